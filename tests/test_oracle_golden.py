"""Pins the CPU oracle against the golden vectors produced by the reference's own
importable modules (tests/gen_golden.py).  Runs without a GPU."""
import os

import numpy as np
import pytest

import golden_inputs as gi
from oracle import distance as od
from oracle import evalproto as oe


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_inputs_reproduce(golden_dir):
    g = load(golden_dir, 'distance_pairs.npz')
    assert np.array_equal(g['sha'], gi.digest(*gi.pair_inputs()))
    g = load(golden_dir, 'match_b8_g1000.npz')
    assert np.array_equal(g['sha'], gi.digest(*gi.match_inputs()))
    g = load(golden_dir, 'match_ties.npz')
    assert np.array_equal(g['sha'], gi.digest(*gi.match_tie_inputs()))


def test_distance_bit_exact(golden_dir):
    g = load(golden_dir, 'distance_pairs.npz')
    e1, e2 = gi.pair_inputs()
    for m, key in ((0, 'd0'), (1, 'd1')):
        out = od.distance(e1, e2, m)
        assert out.dtype == np.float32
        assert np.array_equal(out, g[key])
    assert np.array_equal(od.get_emd_distance(e1, e2, 0), g['emd0'])
    assert np.array_equal(od.get_emd_distance(e1, e2, 1), g['emd1'])


def test_distance_undefined_metric():
    e1, e2 = gi.pair_inputs(4)
    with pytest.raises(RuntimeError, match='Undefined distance metric 10'):
        od.distance(e1, e2, 10)      # the CLI default of the reference is the invalid 10


@pytest.mark.parametrize('name,maker', [('match_b8_g1000.npz', gi.match_inputs),
                                        ('match_ties.npz', gi.match_tie_inputs),
                                        ('match_unnormalised.npz', gi.match_unnormalised_inputs)])
def test_match_bit_exact(golden_dir, name, maker):
    g = load(golden_dir, name)
    probes, gallery = maker()
    for m in (0, 1):
        idx, best, full = od.match(probes, gallery, m)
        assert np.array_equal(full, g['full%d' % m], equal_nan=True)
        assert np.array_equal(idx, g['idx%d' % m])
        assert np.array_equal(best, g['full%d' % m][np.arange(len(idx)), idx], equal_nan=True)


def test_match_ties_take_first(golden_dir):
    g = load(golden_dir, 'match_ties.npz')
    # probes near rows 100..139 have exact copies at 300.. and 500..: first index wins
    assert list(g['idx1'][:5]) == [100, 105, 119, 120, 139]
    assert g['idx1'][5] == 101 and g['idx1'][6] == 110


def test_match_blas_agrees_on_separated_data():
    probes, gallery = gi.match_inputs(16, 2000)
    for m in (0, 1):
        i1, d1, _ = od.match(probes, gallery, m)
        i2, d2 = od.match_blas(probes, gallery, m)
        assert np.array_equal(i1, i2)
        np.testing.assert_allclose(d1, d2, atol=2e-5)


def test_scalars(golden_dir):
    g = load(golden_dir, 'scalars.npz')
    a, b = gi.vector_inputs()
    assert np.float64(od.sq_l2(a, b)) == g['sq_l2']
    assert np.array_equal(np.array([od.distance_to_proba(x) for x in g['d']]), g['proba'])
    assert np.array_equal(np.array([od.gaussian_kernel_dist_to_prob(x) for x in g['d']]), g['gauss'])
    assert np.array_equal(np.array([od.gaussian_kernel_dist_to_prob(x, 2.0) for x in g['d']]), g['gauss_t2'])


def test_face_distance_and_compare():
    a, b = gi.vector_inputs()
    assert od.face_distance([], b).shape == (0,)
    d = od.face_distance(a, b)
    assert np.isclose(d, np.sqrt(od.sq_l2(a, b)))
    dist, proba = od.compare_faces([a * 0.01], [b * 0.01])
    assert dist <= 0.6 and np.isclose(proba, np.exp(-dist / 2))
    dist, proba = od.compare_faces([a], [b])
    assert dist > 0.6 and np.isclose(proba, 1 / (1 + dist))


def test_roc_pieces(golden_dir):
    g = load(golden_dir, 'roc.npz')
    e1, e2, same = gi.roc_inputs()
    assert np.array_equal(g['sha'], gi.digest(e1, e2, same.astype(np.float32)))
    thresholds = np.arange(0, 4, 0.01)
    for m in (0, 1):
        dist = od.distance(e1, e2, m)
        acc = np.array([oe.calculate_accuracy(t, dist, same) for t in (0.2, 0.5, 1.0, 1.5)])
        assert np.array_equal(acc, g['acc_m%d' % m])
        vf = np.array([oe.calculate_val_far(t, dist, same) for t in (0.2, 0.5, 1.0, 1.5)])
        assert np.array_equal(vf, g['valfar_m%d' % m])
        for sub in (False, True):
            k = 'm%d_s%d' % (m, int(sub))
            tpr, fpr, accs, f1 = oe.calculate_roc(thresholds, e1, e2, same, 10, m, sub)
            assert np.array_equal(tpr, g['tpr_' + k])
            assert np.array_equal(fpr, g['fpr_' + k])
            assert np.array_equal(accs, g['acc_' + k])
            assert np.array_equal(f1, g['f1_' + k])


def test_near_tie_fixture(golden_dir):
    """The near-tie fixture: inputs reproduce, the oracle's arg-min equals the reference's, and the
    fixture really holds what it is for -- probes with several rows at the minimal float32 distance,
    and probes whose reference distance is NaN (similarity rounded above 1: np.argmin returns the
    first such row)."""
    g = load(golden_dir, 'match_near_ties.npz')
    probes, gallery = gi.match_near_tie_inputs()
    assert np.array_equal(g['sha'], gi.digest(probes, gallery))
    with np.errstate(invalid='ignore'):
        for m in (0, 1):
            idx, best, _ = od.match(probes, gallery, m)
            assert np.array_equal(idx, g['idx%d' % m])
            assert np.array_equal(best, g['dmin%d' % m], equal_nan=True)
    assert (g['nties0'] > 1).sum() >= 10 and (g['nties1'] > 1).sum() >= 10
    assert (g['nnan1'] > 0).sum() >= 5 and g['nnan0'].sum() == 0


def test_numpy_summation_order_restated():
    """oracle.distance.np_pairwise_sum == np.sum(axis=1), bit for bit, for the row lengths the product
    meets (512, 128) and the awkward ones (tails, < 8, splits that are not powers of two)."""
    rng = np.random.default_rng(0)
    for d in (512, 128, 100, 7, 1000, 513, 2048, 64, 96, 129, 255, 8, 9, 1):
        a = rng.standard_normal((6, d)).astype(np.float32)
        want = np.sum(a, axis=1)
        got = np.array([od.np_pairwise_sum(r) for r in a])
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), d
    # the three sums of the cosine distance, as the reference forms them (products rounded first)
    q = rng.standard_normal((1, 512)).astype(np.float32)
    gal = rng.standard_normal((20, 512)).astype(np.float32)
    dot = np.sum(np.multiply(q, gal), axis=1)
    nrm = np.linalg.norm(gal, axis=1)
    for r in range(20):
        assert od.np_pairwise_sum(q[0] * gal[r]) == dot[r]
        assert np.sqrt(od.np_pairwise_sum(gal[r] * gal[r])) == nrm[r]


def test_degenerate_fixture(golden_dir):
    """Zero-norm / non-finite / tiny / huge rows and probes, anti-parallel rows: the
    oracle's arg-min and minimal distance equal the reference's on every case, NaN pattern included."""
    g = load(golden_dir, 'match_degenerate.npz')
    cases = gi.match_degenerate_cases()
    assert len(cases) >= 10
    seen_nan = 0
    with np.errstate(all='ignore'):
        for name, probes, gallery in cases:
            assert gallery.shape[0] >= 4096
            assert np.array_equal(g[name + '_sha'], gi.digest(probes, gallery)), name
            for m in (0, 1):
                idx, best, _ = od.match(probes, gallery, m)
                assert np.array_equal(idx, g['%s_idx%d' % (name, m)]), (name, m)
                assert np.array_equal(best, g['%s_dmin%d' % (name, m)], equal_nan=True), (name, m)
                seen_nan += int(np.isnan(best).sum())
    assert seen_nan > 100
    # what the fixture is for: a zero-norm row is np.argmin's answer for every finite probe under metric 1 ...
    assert set(g['zero_rows_idx1']) == {417} and np.all(np.isnan(g['zero_rows_dmin1']))
    # ... but an ordinary row under metric 0; anti-parallel rows produce NaN at the far end of the ranking
    assert not np.isnan(g['zero_rows_dmin0']).any()
    assert 5 <= np.isnan(g['antiparallel_dmin1']).sum() < len(g['antiparallel_dmin1'])
