"""NumPy restatement of the reference's LFW-protocol evaluation (the direct
caller of the distance path).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows deep_insight_face/evaluation/utility.py:36-49 (calculate_accuracy),
:69-77 (calculate_val_far), :80-119 (calculate_val), :122-171 (calculate_roc).
"""
import numpy as np

from .distance import distance


def kfold_indices(n, n_splits):
    """Contiguous, unshuffled K-fold (what sklearn's KFold(shuffle=False) yields,
    used at evaluation/utility.py:92,135): the first n % k folds get one extra
    sample.  Yields (train_idx, test_idx)."""
    sizes = np.full(n_splits, n // n_splits, dtype=int)
    sizes[: n % n_splits] += 1
    idx = np.arange(n)
    start = 0
    for s in sizes:
        test = idx[start:start + s]
        train = np.concatenate([idx[:start], idx[start + s:]])
        yield train, test
        start += s


def calculate_accuracy(threshold, dist, actual_issame):
    """evaluation/utility.py:36-49 -> (tpr, fpr, acc, f1)."""
    pred = dist < threshold
    same = np.asarray(actual_issame, dtype=bool)
    tp = int(np.sum(pred & same))
    fp = int(np.sum(pred & ~same))
    tn = int(np.sum(~pred & ~same))
    fn = int(np.sum(~pred & same))
    tpr = 0 if tp + fn == 0 else tp / (tp + fn)
    fpr = 0 if fp + tn == 0 else fp / (fp + tn)
    acc = (tp + tn) / dist.size
    prec = 0 if tp + fp == 0 else tp / (tp + fp)
    rec = 0 if tp + fn == 0 else tp / (tp + fn)
    f1 = 0 if (prec + rec) == 0.0 else 2 * (prec * rec / (prec + rec))
    return tpr, fpr, acc, f1


def calculate_val_far(threshold, dist, actual_issame):
    """evaluation/utility.py:69-77 -> (val, far)."""
    pred = dist < threshold
    same = np.asarray(actual_issame, dtype=bool)
    ta = int(np.sum(pred & same))
    fa = int(np.sum(pred & ~same))
    n_same = int(np.sum(same))
    n_diff = int(np.sum(~same))
    val = 0 if n_same == 0 else ta / n_same
    far = 0 if n_diff == 0 else fa / n_diff
    return val, far


def calculate_roc(thresholds, embeddings1, embeddings2, actual_issame,
                  nrof_folds=10, distance_metric=0, subtract_mean=False):
    """evaluation/utility.py:122-171 -> (tpr[T], fpr[T], accuracy[F], f1[F])."""
    actual_issame = np.asarray(actual_issame)
    n = min(len(actual_issame), embeddings1.shape[0])
    T = len(thresholds)
    tprs = np.zeros((nrof_folds, T))
    fprs = np.zeros((nrof_folds, T))
    accuracy = np.zeros(nrof_folds)
    f1s = np.zeros(nrof_folds)
    for f, (train, test) in enumerate(kfold_indices(n, nrof_folds)):
        if subtract_mean:
            mean = np.mean(np.concatenate([embeddings1[train], embeddings2[train]]), axis=0)
        else:
            mean = 0.0
        dist = distance(embeddings1 - mean, embeddings2 - mean, distance_metric)
        acc_train = np.zeros(T)
        for t, thr in enumerate(thresholds):
            acc_train[t] = calculate_accuracy(thr, dist[train], actual_issame[train])[2]
        best = int(np.argmax(acc_train))
        for t, thr in enumerate(thresholds):
            tprs[f, t], fprs[f, t], _, _ = calculate_accuracy(thr, dist[test], actual_issame[test])
        _, _, accuracy[f], f1s[f] = calculate_accuracy(thresholds[best], dist[test], actual_issame[test])
    return np.mean(tprs, 0), np.mean(fprs, 0), accuracy, f1s


def far_train_curve(thresholds, dist, actual_issame):
    """Inner loop of calculate_val (evaluation/utility.py:104-107)."""
    out = np.zeros(len(thresholds))
    for t, thr in enumerate(thresholds):
        out[t] = calculate_val_far(thr, dist, actual_issame)[1]
    return out
