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
