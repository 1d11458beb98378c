"""NumPy restatement of the reference's distance functions and of the 1:N match
that is composed from them.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Every function cites the reference lines it restates (paths relative to
/root/reference).  The arithmetic is kept in the dtype the reference computes in
(float32 in -> float32 out) so that results are bit-comparable with the golden
vectors in tests/golden/, which were produced by the reference itself.
"""
import math

import numpy as np


def distance(embeddings1, embeddings2, distance_metric=0):
    """Row-paired distance.  deep_insight_face/evaluation/utility.py:52-66.

    metric 0: squared L2, sum over axis 1 of (a-b)^2.
    metric 1: arccos(cosine similarity) / pi.
    Anything else raises RuntimeError('Undefined distance metric %d').
    """
    if distance_metric == 0:
        d = embeddings1 - embeddings2
        return (d * d).sum(axis=1)
    if distance_metric == 1:
        num = (embeddings1 * embeddings2).sum(axis=1)
        den = np.linalg.norm(embeddings1, axis=1) * np.linalg.norm(embeddings2, axis=1)
        return np.arccos(num / den) / math.pi
    raise RuntimeError('Undefined distance metric %d' % distance_metric)


def get_emd_distance(embeddings1, embeddings2, distance_metric=0):
    """Twin of ``distance`` whose metric 0 reduces over axis 0 (a quirk of the
    reference).  deep_insight_face/evaluation/utility.py:174-188."""
    if distance_metric == 0:
        d = embeddings1 - embeddings2
        return (d * d).sum(axis=0)
    if distance_metric == 1:
        return distance(embeddings1, embeddings2, 1)
    raise RuntimeError('Undefined distance metric %d' % distance_metric)


def sq_l2(emb1, emb2):
    """Squared L2 over every element.  deep_insight_face/networks/utils.py:4-9."""
    return np.sum(np.square(emb1 - emb2))


def distance_to_proba(d):
    """deep_insight_face/networks/utils.py:12-17."""
    return 1 / (1 + d)


def gaussian_kernel_dist_to_prob(d, tuning_factor=1.0):
    """deep_insight_face/networks/utils.py:20-29."""
    return np.exp(-d / (2 * tuning_factor ** 2))


def face_distance(face_encodings, face_to_compare):
    """L2 norm along axis 0; empty input -> np.empty((0)).
    deep_insight_face/api.py:94-104."""
    if len(face_encodings) == 0:
        return np.empty((0))
    return np.linalg.norm(face_encodings - face_to_compare, axis=0)


def compare_faces(known_face_encodings, face_encoding_to_check, tolerance=0.6):
    """(distance, probability) with the 0.6 tolerance switch between the two
    probability maps.  deep_insight_face/api.py:242-256."""
    d = face_distance(known_face_encodings[0], face_encoding_to_check[0])
    if d <= tolerance:
        p = gaussian_kernel_dist_to_prob(d)
    else:
        p = distance_to_proba(d)
    return d, p


def euclidean_distance(x, y, eps=1e-7):
    """sqrt(max(sum((x-y)^2, axis=1, keepdims), K.epsilon())).
    deep_insight_face/networks/siamese.py:22-24 (K.epsilon() == 1e-7)."""
    s = np.sum(np.square(x - y), axis=1, keepdims=True)
    return np.sqrt(np.maximum(s, np.asarray(eps, dtype=s.dtype)))


def verify_distance(encoding, stored):
    """float(np.linalg.norm(enc - db[id])).
    deep_insight_face/predictions.py:126."""
    return float(np.linalg.norm(encoding - stored))


def match(probes, gallery, distance_metric=1):
    """1:N top-1 search.  The reference has no such entry point (SURVEY.md
    section 3 "1:N gallery search"); its semantics are pinned as the composition
    the reference's own functions allow: for each probe row q,
    ``d = distance(q[None, :], gallery, metric)`` (NumPy broadcast of
    evaluation/utility.py:52-66) followed by ``np.argmin(d)`` (first minimum).

    Returns (idx[B] int64, dist[B] float32, full[B,G] float32)."""
    B = probes.shape[0]
    G = gallery.shape[0]
    idx = np.zeros((B,), dtype=np.int64)
    best = np.zeros((B,), dtype=np.float32)
    full = np.zeros((B, G), dtype=np.float32)
    for b in range(B):
        d = distance(probes[b][None, :], gallery, distance_metric)
        full[b] = d
        idx[b] = int(np.argmin(d))
        best[b] = d[idx[b]]
    return idx, best, full


def match_blas(probes, gallery, distance_metric=1):
    """Same search written the way a CPU user would write it for speed: one
    sgemm of the probe block against the gallery, then a row-wise arg-extremum.
    Used as the 'fair' CPU baseline in bench.py and as a second opinion in tests;
    precedent for 'cosine = matmul of normalised rows' in the reference:
    deep_insight_face/common/losses.py:39-40,137-138."""
    p = np.ascontiguousarray(probes, dtype=np.float32)
    g = np.ascontiguousarray(gallery, dtype=np.float32)
    dots = p @ g.T
    if distance_metric == 1:
        pn = np.linalg.norm(p, axis=1)
        gn = np.linalg.norm(g, axis=1)
        sim = dots / (pn[:, None] * gn[None, :])
        idx = np.argmax(sim, axis=1)
        s = sim[np.arange(p.shape[0]), idx]
        return idx.astype(np.int64), (np.arccos(np.clip(s, -1.0, 1.0)) / math.pi).astype(np.float32)
    if distance_metric == 0:
        d = (p * p).sum(1)[:, None] + (g * g).sum(1)[None, :] - 2.0 * dots
        idx = np.argmin(d, axis=1)
        return idx.astype(np.int64), d[np.arange(p.shape[0]), idx].astype(np.float32)
    raise RuntimeError('Undefined distance metric %d' % distance_metric)


# --------------------------------------------------------------------------- summation order
def np_pairwise_sum(a):
    """What ``np.sum(x, axis=1)`` / ``np.add.reduce`` does to ONE contiguous float32 row, spelled out
    (NumPy 2.2.6, numpy/_core/src/umath/loops_utils.h.src: @TYPE@_pairwise_sum): n < 8 a plain loop;
    n <= 128 eight interleaved accumulators, combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the
    n % 8 leftovers in order; larger n split at n/2 rounded down to a multiple of 8, recursively.
    The reference's distances (evaluation/utility.py:54-62) are these sums of float32 products, so this
    order -- not the mathematical sum -- is what decides its arg-min on near-ties.  csrc/match.hip
    (np_sum) evaluates the same tree on the device; tests/test_oracle_golden.py pins this restatement
    against np.sum itself bit for bit."""
    f32 = np.float32
    a = np.asarray(a, dtype=np.float32)
    n = a.shape[0]
    if n < 8:
        res = f32(0.0)
        for x in a:
            res = f32(res + x)
        return res
    if n <= 128:
        r = [a[j] for j in range(8)]
        body = n - (n % 8)
        for i in range(8, body, 8):
            for j in range(8):
                r[j] = f32(r[j] + a[i + j])
        res = f32(f32(f32(r[0] + r[1]) + f32(r[2] + r[3])) + f32(f32(r[4] + r[5]) + f32(r[6] + r[7])))
        for i in range(body, n):
            res = f32(res + a[i])
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return f32(np_pairwise_sum(a[:n2]) + np_pairwise_sum(a[n2:]))


def similarity(embeddings1, embeddings2):
    """The cosine similarity inside ``distance(.., 1)`` (evaluation/utility.py:58-60), float32."""
    num = (embeddings1 * embeddings2).sum(axis=1)
    den = np.linalg.norm(embeddings1, axis=1) * np.linalg.norm(embeddings2, axis=1)
    return num / den
