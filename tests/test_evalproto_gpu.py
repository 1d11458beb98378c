"""LFW-protocol evaluation on the device against the golden vectors produced by the reference's
own calculate_accuracy / calculate_val_far / calculate_roc (tests/gen_golden.py)."""
import contextlib
import io
import os

import numpy as np
import pytest

import golden_inputs as gi
from oracle import evalproto as oe

pytestmark = pytest.mark.gpu


def test_accuracy_and_valfar_vs_golden(cuda, golden_dir):
    from deep_insight_face.evaluation import utility
    g = np.load(os.path.join(golden_dir, 'roc.npz'))
    e1, e2, same = gi.roc_inputs()
    for m in (0, 1):
        dist = utility.distance(e1, e2, m)
        acc = np.array([utility.calculate_accuracy(t, dist, same) for t in (0.2, 0.5, 1.0, 1.5)])
        vf = np.array([utility.calculate_val_far(t, dist, same) for t in (0.2, 0.5, 1.0, 1.5)])
        # integer counts: equal unless a distance sits within float32 rounding of a threshold
        np.testing.assert_allclose(acc, g['acc_m%d' % m], atol=1.0 / len(same) + 1e-12)
        np.testing.assert_allclose(vf, g['valfar_m%d' % m], atol=2.0 / len(same) + 1e-12)


@pytest.mark.parametrize('metric,sub', [(0, False), (1, False), (0, True), (1, True)])
def test_roc_vs_golden(cuda, golden_dir, metric, sub):
    from deep_insight_face.evaluation import utility
    g = np.load(os.path.join(golden_dir, 'roc.npz'))
    e1, e2, same = gi.roc_inputs()
    thresholds = np.arange(0, 4, 0.01)
    with contextlib.redirect_stdout(io.StringIO()):
        tpr, fpr, acc, f1 = utility.calculate_roc(thresholds, e1, e2, same, 10, metric, sub)
    k = 'm%d_s%d' % (metric, int(sub))
    tol = 2.0 / (len(same) / 10)            # one pair of one fold may flip at a threshold edge
    np.testing.assert_allclose(tpr, g['tpr_' + k], atol=tol / 10 + 1e-12)
    np.testing.assert_allclose(fpr, g['fpr_' + k], atol=tol / 10 + 1e-12)
    np.testing.assert_allclose(acc, g['acc_' + k], atol=tol)
    np.testing.assert_allclose(f1, g['f1_' + k], atol=2 * tol)


def test_evaluate_runs_and_val_matches_oracle_curve(cuda):
    from deep_insight_face.evaluation import utility
    e1, e2, same = gi.roc_inputs(npairs=400, seed=3)
    emb = np.empty((800, e1.shape[1]), dtype=np.float32)
    emb[0::2], emb[1::2] = e1, e2
    with contextlib.redirect_stdout(io.StringIO()):
        tpr, fpr, acc, f1, val, val_std, far = utility.evaluate(emb, same, nrof_folds=10, distance_metric=0)
    assert tpr.shape == (400,) and acc.shape == (10,)
    assert 0.0 <= val <= 1.0 and 0.0 <= far <= 0.05
    # the train FAR curve of fold 0 equals the oracle's restatement of utility.py:104-107
    thresholds = np.arange(0, 4, 0.001)
    fold_ids = utility._kfold_ids(400, 10)
    dist = utility.distance(e1, e2, 0)
    counts = utility._threshold_counts(dist, same, thresholds, fold_ids, 10)
    train = counts.sum(0) - counts[0]
    tr = fold_ids != 0
    want = oe.far_train_curve(thresholds, dist[tr], same[tr])
    got = train[:, 1] / float((~same[tr]).sum())
    np.testing.assert_allclose(got, want, atol=1.0 / (~same[tr]).sum() + 1e-12)
