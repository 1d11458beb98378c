"""1:N gallery match ("one shot" identification) on the MI355X.

The reference's oneshot.py is an unfinished Omniglot stub with no distance code
(deep_insight_face/oneshot.py:8 "TODO: FIX THIS MODULE"); north_star houses the
pairwise cosine-distance gallery match under this module name.  Semantics: for every
probe row q, ``np.argmin(evaluation.utility.distance(q[None, :], gallery, metric))``
(evaluation/utility.py:52-66 broadcast over the gallery; first minimum wins).
"""
import ctypes
import os

import torch

from . import _native as N


class Gallery:
    """Device-resident gallery of enrolled embeddings, [G, d] float32.

    One instance per process/GPU.  With ``index_base`` the rows are a shard of a larger
    gallery (see deep_insight_face.parallel.ShardedGallery)."""

    def __init__(self, embeddings=None, emd_size=None, index_base=0):
        self._h = ctypes.c_void_p()
        self._dev = N.require_device()
        if embeddings is not None and emd_size is None:
            emd_size = int(embeddings.shape[1])
        if emd_size is None:
            raise ValueError('Gallery needs embeddings or emd_size')
        N.check(N.lib.dif_gallery_create(ctypes.byref(self._h), int(emd_size)), ValueError)
        self.emd_size = int(emd_size)
        # development hook (A/B runs of bench.py and the tools, like DIF_OPTIONS for the networks): "key=value,..." applied to
        # every gallery of the process at creation
        for kv in os.environ.get('DIF_GALLERY_OPTIONS', '').split(','):
            if '=' in kv:
                self.set_option(kv.split('=')[0].strip(), int(kv.split('=')[1]))
        if embeddings is not None:
            self.set(embeddings, index_base)

    def set(self, embeddings, index_base=0):
        g, _ = N.to_device_f32(embeddings, self._dev)
        if g.dim() != 2 or g.shape[1] != self.emd_size:
            raise ValueError('gallery must be [G, %d], got %s' % (self.emd_size, tuple(g.shape)))
        N.check(N.lib.dif_gallery_set(self._h, N.ptr(g), g.shape[0], int(index_base), N.stream_ptr()))
        torch.cuda.current_stream().synchronize()   # g may be a temporary: the copy must have landed

    def update(self, embeddings, first_row=None):
        """Enrol incrementally: overwrite rows [first_row, first_row + k) or, with first_row None (or == len(self)),
        append -- O(k) on the device (`set` is a pass over the whole gallery).  Appending beyond the capacity grows it
        by half (one reallocation + copy)."""
        g, _ = N.to_device_f32(embeddings, self._dev)
        if g.dim() == 1:
            g = g[None, :]
        if g.dim() != 2 or g.shape[1] != self.emd_size:
            raise ValueError('rows must be [k, %d], got %s' % (self.emd_size, tuple(g.shape)))
        n = len(self)
        first_row = n if first_row is None else int(first_row)
        if first_row < 0 or first_row > n:
            raise ValueError('first_row %d outside [0, %d]' % (first_row, n))
        need = first_row + g.shape[0]
        if need > self.capacity:
            self.reserve(max(need, self.capacity + self.capacity // 2))
        N.check(N.lib.dif_gallery_update(self._h, N.ptr(g), g.shape[0], first_row, N.stream_ptr()), ValueError)
        torch.cuda.current_stream().synchronize()   # g may be a temporary: the copy must have landed

    def reserve(self, capacity):
        N.check(N.lib.dif_gallery_reserve(self._h, int(capacity), N.stream_ptr()))

    @property
    def capacity(self):
        return int(N.lib.dif_gallery_capacity(self._h))

    def __len__(self):
        return int(N.lib.dif_gallery_size(self._h))

    def set_option(self, key, value):
        """'filter': 2 (default) runs the MFMA filter stage on operands rounded to bf16 once, 1 on two-term split-bf16
        operands, 0 in float32: the same results (the filter only proposes candidates), different speed and memory.
        'frag': layout of the one-term filter's copy -- 1 (default) MFMA-fragment order from 2^18 rows up (embedding sizes that
        are multiples of 128 up to 512), 2 always, 0 row-major; same results (include/dif.h).
        'clamp_nan': 1 reports distance 0 / 1 where the reference's distance is NaN (a similarity rounded
        beyond +-1); the default 0 reports NaN like the reference.  The arg-min is unaffected."""
        N.check(N.lib.dif_gallery_set_option(self._h, key.encode(), int(value)), ValueError)

    def stat(self, key):
        """'split_copy': 1 when the filter's bf16 copy of the rows exists (+ 50 % device memory for the one-term
        filter, + 100 % for the two-term one; when it cannot be allocated the float32 filter serves);
        'filter_terms': bf16 terms per operand the next match's filter runs on (0: float32 rows);
        'frag_copy': 1 when the one-term copy is held in MFMA-fragment order (option 'frag': match_g1_kernel serves);
        'row_bytes': device bytes per row;
        'exact_probes': probes the last match sent to the exact whole-gallery search (synchronises)."""
        v = ctypes.c_int64(0)
        N.check(N.lib.dif_gallery_get_stat(self._h, key.encode(), ctypes.byref(v), N.stream_ptr()), ValueError)
        return int(v.value)

    def match_into(self, probes, distance_metric, idx, dist, key=None):
        """Allocation-free form for a serving loop: `probes` [B, d] float32 CUDA, results written into the
        caller's CUDA tensors idx [B] int64, dist [B] float32 and (optional) key [B] float32 -- which may be
        slices of one packed buffer (see ShardedGallery)."""
        if distance_metric not in (0, 1):
            raise RuntimeError('Undefined distance metric %d' % distance_metric)
        # the library reads `probes` as dense float32 [B, emd_size] and writes B results: anything else would be
        # read out of bounds or reinterpreted silently, so it is refused here (no conversion: this form allocates nothing)
        if not torch.is_tensor(probes) or probes.dim() != 2 or probes.shape[1] != self.emd_size:
            raise ValueError('probes must be a [B, %d] tensor, got %s' % (
                self.emd_size, tuple(probes.shape) if hasattr(probes, 'shape') else type(probes).__name__))
        B = probes.shape[0]
        for name, t, dt in (('probes', probes, torch.float32), ('idx', idx, torch.int64), ('dist', dist, torch.float32),
                            ('key', key, torch.float32)):
            if t is None and name == 'key':
                continue
            if not torch.is_tensor(t) or t.dtype != dt or t.device != self._dev or not t.is_contiguous():
                raise ValueError('match_into: %s must be a contiguous %s tensor on %s' % (name, dt, self._dev))
            if name != 'probes' and (t.dim() != 1 or t.shape[0] != B):
                raise ValueError('match_into: %s must have shape [%d], got %s' % (name, B, tuple(t.shape)))
        if B:
            if len(self) == 0:
                raise ValueError('attempt to get argmin of an empty sequence')
            N.check(N.lib.dif_match(self._h, N.ptr(probes), B, distance_metric, N.ptr(idx), N.ptr(dist),
                                    N.ptr(key) if key is not None else None, N.stream_ptr()))

    def match(self, probes, distance_metric=1, return_key=False):
        """-> (idx[B] int64, dist[B] float32) [, key[B]]; NumPy in -> NumPy out."""
        if distance_metric not in (0, 1):
            raise RuntimeError('Undefined distance metric %d' % distance_metric)
        p, was_np = N.to_device_f32(probes, self._dev)
        if p.dim() == 1:
            p = p[None, :]
        if p.dim() != 2 or p.shape[1] != self.emd_size:
            raise ValueError('probes must be [B, %d], got %s' % (self.emd_size, tuple(p.shape)))
        B = p.shape[0]
        idx = torch.empty((B,), dtype=torch.int64, device=self._dev)
        dist = torch.empty((B,), dtype=torch.float32, device=self._dev)
        key = torch.empty((B,), dtype=torch.float32, device=self._dev)
        if B:
            if len(self) == 0:
                raise ValueError('attempt to get argmin of an empty sequence')   # what np.argmin raises
            N.check(N.lib.dif_match(self._h, N.ptr(p), B, distance_metric, N.ptr(idx), N.ptr(dist),
                                    N.ptr(key), N.stream_ptr()))
        if was_np:
            idx, dist, key = idx.cpu().numpy(), dist.cpu().numpy(), key.cpu().numpy()
        return (idx, dist, key) if return_key else (idx, dist)

    def close(self):
        if self._h:
            N.lib.dif_gallery_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def match(probes, gallery, distance_metric=1):
    """One-call form: top-1 gallery index and distance for each probe row."""
    g = gallery if isinstance(gallery, Gallery) else Gallery(gallery)
    try:
        return g.match(probes, distance_metric)
    finally:
        if g is not gallery:
            g.close()


def one_shot_clf(probe, gallery, distance_metric=1):
    """Identify one face: index of the nearest enrolled embedding and its distance
    (name kept from the reference stub, deep_insight_face/oneshot.py:110)."""
    idx, dist = match(probe, gallery, distance_metric)
    return int(idx[0]), float(dist[0])


def cosine_similarity_matrix(embeddings1, embeddings2=None):
    """All-pairs cosine similarity ``l2norm(E1) @ l2norm(E2).T`` -> [B, G] float32 (E2 defaults to E1):
    the reference's own "cosine = matmul of normalised rows" (common/losses.py:39-40, :137-138), as one
    MFMA GEMM with both normalisations in its epilogue (csrc/arcmargin.hip with scale 1, no margin).
    NumPy in -> NumPy out."""
    from .networks.arcmargin import ArcMarginHead
    if embeddings2 is None:
        embeddings2 = embeddings1
    head = ArcMarginHead(embeddings2, s=1.0, m=0.0)
    try:
        return head.logits(embeddings1)
    finally:
        head.close()
