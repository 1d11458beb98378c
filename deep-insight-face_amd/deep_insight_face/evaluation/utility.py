"""Distance functions of the evaluation path, computed on the MI355X.

Mirrors deep_insight_face/evaluation/utility.py of the reference: same names, argument
meaning and error behaviour.  NumPy in -> NumPy float32 out (as the reference);
torch tensors in -> torch tensor (on the GPU) out.
"""
import torch

from .. import _native as N


def distance(embeddings1, embeddings2, distance_metric=0):
    """Row-paired distance between two [n, d] batches (either side may be a single
    row, NumPy-broadcast style).  metric 0: squared L2; metric 1: arccos(cos)/pi.
    Reference: evaluation/utility.py:52-66 (RuntimeError on any other metric)."""
    if distance_metric not in (0, 1):
        raise RuntimeError('Undefined distance metric %d' % distance_metric)
    dev = N.require_device()
    a, a_np = N.to_device_f32(embeddings1, dev)
    b, b_np = N.to_device_f32(embeddings2, dev)
    if a.dim() == 1:
        a = a[None, :]
    if b.dim() == 1:
        b = b[None, :]
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[1]:
        raise ValueError('operands could not be broadcast together with shapes %s %s'
                         % (tuple(a.shape), tuple(b.shape)))
    n1, n2 = a.shape[0], b.shape[0]
    if n1 != n2 and 1 not in (n1, n2):
        raise ValueError('operands could not be broadcast together with shapes %s %s'
                         % (tuple(a.shape), tuple(b.shape)))
    n = 0 if 0 in (n1, n2) else max(n1, n2)
    out = torch.empty((n,), dtype=torch.float32, device=dev)
    if n:
        N.check(N.lib.dif_pairwise(N.ptr(a), n1, N.ptr(b), n2, a.shape[1], distance_metric,
                                   N.ptr(out), N.stream_ptr()), RuntimeError)
    return out.cpu().numpy() if (a_np and b_np) else out


def get_emd_distance(embeddings1, embeddings2, distance_metric=0):
    """Twin of `distance` (evaluation/utility.py:174-188).  Its metric 0 reduces over
    axis 0 instead of axis 1 -- a quirk of the reference kept here: the sum over rows of
    (a-b)^2 per column is the row-paired distance of the transposed operands."""
    if distance_metric == 0:
        dev = N.require_device()
        a, a_np = N.to_device_f32(embeddings1, dev)
        b, b_np = N.to_device_f32(embeddings2, dev)
        a, b = torch.broadcast_tensors(a, b)
        if a.dim() == 1:
            r = distance(a[None, :], b[None, :], 0)[0]
            return r.cpu().numpy() if (a_np and b_np) else r
        r = distance(a.t().contiguous(), b.t().contiguous(), 0)
        return r.cpu().numpy() if (a_np and b_np) else r
    if distance_metric == 1:
        return distance(embeddings1, embeddings2, 1)
    raise RuntimeError('Undefined distance metric %d' % distance_metric)


# ------------------------------------------------------------------------------ LFW protocol
# evaluation/utility.py:10-33 (evaluate), :36-49 (calculate_accuracy), :69-77
# (calculate_val_far), :80-119 (calculate_val), :122-171 (calculate_roc).  The reference sweeps
# 400 / 4000 thresholds x 10 folds in Python; here every sweep is one device pass
# (dif_threshold_counts) that returns, per fold and threshold, the number of accepted same /
# different pairs.  All ratios are then formed on the host from those integers exactly as the
# reference forms them.
import numpy as np   # noqa: E402


def _kfold_ids(n, n_splits):
    """fold id of every pair under sklearn KFold(n_splits, shuffle=False) (utility.py:92,135):
    contiguous test ranges, the first n % k folds one longer."""
    sizes = np.full(n_splits, n // n_splits, dtype=np.int64)
    sizes[: n % n_splits] += 1
    return np.repeat(np.arange(n_splits, dtype=np.int32), sizes)


def _threshold_counts(dist, actual_issame, thresholds, fold_ids, n_folds):
    """-> int array [n_folds, T, 2]: accepted same / accepted different pairs per test fold."""
    dev = N.require_device()
    d, _ = N.to_device_f32(dist, dev)
    same = torch.as_tensor(np.asarray(actual_issame).astype(np.uint8)).to(dev)
    fold = torch.as_tensor(np.asarray(fold_ids, dtype=np.int32)).to(dev)
    thr = torch.as_tensor(np.asarray(thresholds, dtype=np.float64).reshape(-1)).to(dev)
    T = int(thr.numel())
    out = torch.zeros((n_folds, T, 2), dtype=torch.int32, device=dev)
    N.check(N.lib.dif_threshold_counts(N.ptr(d), N.ptr(same), N.ptr(fold), d.numel(), N.ptr(thr), T, n_folds,
                                       N.ptr(out), N.stream_ptr()))
    return out.cpu().numpy().astype(np.int64)


def _ratios(tp, fp, n_same, n_diff, size):
    """tpr, fpr, acc, f1 from the accept counts, as calculate_accuracy does (utility.py:38-49)."""
    fn, tn = n_same - tp, n_diff - fp
    tpr = 0 if (tp + fn == 0) else float(tp) / float(tp + fn)
    fpr = 0 if (fp + tn == 0) else float(fp) / float(fp + tn)
    acc = float(tp + tn) / size
    precision = 0 if (tp + fp == 0) else float(tp) / float(tp + fp)
    recall = 0 if (tp + fn == 0) else float(tp) / float(tp + fn)
    f1score = 0 if float(precision + recall) == 0.0 else 2 * (float(precision * recall) / float(precision + recall))
    return tpr, fpr, acc, f1score


def calculate_accuracy(threshold, dist, actual_issame, display_cm=False):
    same = np.asarray(actual_issame).astype(bool)
    dist = np.asarray(dist)
    c = _threshold_counts(dist, same, [threshold], np.zeros(dist.size, np.int32), 1)[0, 0]
    return _ratios(int(c[0]), int(c[1]), int(same.sum()), int((~same).sum()), dist.size)


def calculate_val_far(threshold, dist, actual_issame):
    same = np.asarray(actual_issame).astype(bool)
    dist = np.asarray(dist)
    c = _threshold_counts(dist, same, [threshold], np.zeros(dist.size, np.int32), 1)[0, 0]
    n_same, n_diff = int(same.sum()), int((~same).sum())
    val = 0 if n_same == 0 else float(c[0]) / float(n_same)
    far = 0 if n_diff == 0 else float(c[1]) / float(n_diff)
    return val, far


def _fold_distances(embeddings1, embeddings2, fold_ids, f, distance_metric, subtract_mean):
    if subtract_mean:
        train = fold_ids != f
        mean = np.mean(np.concatenate([embeddings1[train], embeddings2[train]]), axis=0)
    else:
        mean = 0.0
    return distance(embeddings1 - mean, embeddings2 - mean, distance_metric)


def calculate_roc(thresholds, embeddings1, embeddings2, actual_issame, nrof_folds=10, distance_metric=0,
                  subtract_mean=False):
    """utility.py:122-171 -> (tpr[T], fpr[T], accuracy[folds], f1scores[folds])."""
    assert(embeddings1.shape[0] == embeddings2.shape[0])
    assert(embeddings1.shape[1] == embeddings2.shape[1])
    same = np.asarray(actual_issame).astype(bool)
    n = min(len(same), embeddings1.shape[0])
    e1, e2, same = np.asarray(embeddings1)[:n], np.asarray(embeddings2)[:n], same[:n]
    T = len(thresholds)
    fold_ids = _kfold_ids(n, nrof_folds)
    tprs, fprs = np.zeros((nrof_folds, T)), np.zeros((nrof_folds, T))
    accuracy, f1scores = np.zeros(nrof_folds), np.zeros(nrof_folds)
    n_same_f = np.array([int(same[fold_ids == f].sum()) for f in range(nrof_folds)])
    n_all_f = np.array([int((fold_ids == f).sum()) for f in range(nrof_folds)])
    counts = None
    for f in range(nrof_folds):
        if counts is None or subtract_mean:
            dist = _fold_distances(e1, e2, fold_ids, f, distance_metric, subtract_mean)
            counts = _threshold_counts(dist, same, thresholds, fold_ids, nrof_folds)
        test = counts[f]                                    # [T, 2]
        train = counts.sum(axis=0) - test
        ns_te, nd_te = n_same_f[f], n_all_f[f] - n_same_f[f]
        ns_tr, nd_tr = n_same_f.sum() - ns_te, (n_all_f.sum() - n_same_f.sum()) - nd_te
        acc_train = np.array([_ratios(int(train[t, 0]), int(train[t, 1]), ns_tr, nd_tr, ns_tr + nd_tr)[2]
                              for t in range(T)])
        best = int(np.argmax(acc_train))
        for t in range(T):
            tprs[f, t], fprs[f, t], _, _ = _ratios(int(test[t, 0]), int(test[t, 1]), ns_te, nd_te, ns_te + nd_te)
        _, _, accuracy[f], f1scores[f] = _ratios(int(test[best, 0]), int(test[best, 1]), ns_te, nd_te, ns_te + nd_te)
        print("Best Threshold value %02d" % best)
    return np.mean(tprs, 0), np.mean(fprs, 0), accuracy, f1scores


def calculate_val(thresholds, embeddings1, embeddings2, actual_issame, far_target, nrof_folds=10,
                  distance_metric=0, subtract_mean=False):
    """utility.py:80-119 -> (val_mean, val_std, far_mean).  The reference interpolates the train
    FAR curve with scipy interp1d(kind='slinear'), which raises on the duplicate FAR values every
    real curve has under current SciPy; np.interp is used instead --
    same piecewise-linear result wherever SciPy accepts the curve."""
    assert(embeddings1.shape[0] == embeddings2.shape[0])
    assert(embeddings1.shape[1] == embeddings2.shape[1])
    same = np.asarray(actual_issame).astype(bool)
    n = min(len(same), embeddings1.shape[0])
    e1, e2, same = np.asarray(embeddings1)[:n], np.asarray(embeddings2)[:n], same[:n]
    thresholds = np.asarray(thresholds, dtype=np.float64)
    T = len(thresholds)
    fold_ids = _kfold_ids(n, nrof_folds)
    val, far = np.zeros(nrof_folds), np.zeros(nrof_folds)
    n_same_f = np.array([int(same[fold_ids == f].sum()) for f in range(nrof_folds)])
    n_all_f = np.array([int((fold_ids == f).sum()) for f in range(nrof_folds)])
    counts = dist = None
    for f in range(nrof_folds):
        if counts is None or subtract_mean:
            dist = _fold_distances(e1, e2, fold_ids, f, distance_metric, subtract_mean)
            counts = _threshold_counts(dist, same, thresholds, fold_ids, nrof_folds)
        train = counts.sum(axis=0) - counts[f]
        nd_te = n_all_f[f] - n_same_f[f]
        nd_tr = (n_all_f.sum() - n_same_f.sum()) - nd_te
        far_train = np.zeros(T) if nd_tr == 0 else train[:, 1].astype(np.float64) / float(nd_tr)
        if np.max(far_train) >= far_target:
            threshold = float(np.interp(far_target, far_train, thresholds))
        else:
            threshold = 0.0
        te = fold_ids == f
        val[f], far[f] = calculate_val_far(threshold, np.asarray(dist)[te], same[te])
    return np.mean(val), np.std(val), np.mean(far)


def evaluate(embeddings, labels, nrof_folds=10, distance_metric=0, subtract_mean=False,
             thresholds=np.arange(0, 4, 0.01)):
    """utility.py:10-33: embeddings[0::2] vs embeddings[1::2]; ROC over `thresholds`, VAL@FAR=1e-3
    over arange(0, 4, 0.001)."""
    embeddings1 = embeddings[0::2]
    embeddings2 = embeddings[1::2]
    tpr, fpr, accuracy, f1scores = calculate_roc(thresholds, embeddings1, embeddings2, np.asarray(labels),
                                                 nrof_folds=nrof_folds, distance_metric=distance_metric,
                                                 subtract_mean=subtract_mean)
    thresholds = np.arange(0, 4, 0.001)
    far_target = 1e-3
    val, val_std, far = calculate_val(thresholds, embeddings1, embeddings2, np.asarray(labels), far_target,
                                      nrof_folds=nrof_folds, distance_metric=distance_metric,
                                      subtract_mean=subtract_mean)
    return tpr, fpr, accuracy, f1scores, val, val_std, far


from .pairs import add_extension, get_paths, read_pairs  # noqa: E402,F401  (utility.py:222-262)
