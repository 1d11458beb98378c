"""Data-parallel embed + gallery-sharded match across the GPUs of one node.

The reference is single-process (SURVEY.md section 5 "distributed communication backend:
none"); this is new work named by north_star.  One process per GPU, torch.distributed
for the exchange (backend "nccl" is RCCL over xGMI on ROCm; "gloo" on CPU for tests):

  1. every rank embeds its own slice of the batch                    [b, d]
  2. ONE all-gather of the per-rank embeddings                       [R*b, d]   (8 MiB at B=4096)
  3. every rank matches ALL probes against ITS gallery shard         (key, idx, dist)[R*b]
  4. ONE all-gather of the packed per-rank results                   [R, 3, R*b] (tiny)
  5. merge: lowest key, then lowest GLOBAL index == np.argmin over the whole gallery

There is no other collective on the data path.  The local compute (steps 1, 3, 5) is the
HIP library; tests on CPU inject stand-ins for it to exercise steps 2 and 4.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_rows, world_size, rank):
    """Contiguous row range [lo, hi) of `rank`: the first n % R ranks hold one extra row."""
    base, extra = divmod(int(n_rows), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _hip_match(gallery, probes, metric):
    idx, d, key = gallery.match(probes, metric, return_key=True)
    return key, idx, d


def _hip_merge(keys, idx, dists):
    from . import _native as N
    R, B = keys.shape
    oi = torch.empty((B,), dtype=torch.int64, device=keys.device)
    od = torch.empty((B,), dtype=torch.float32, device=keys.device)
    N.check(N.lib.dif_match_merge(N.ptr(keys), N.ptr(idx), N.ptr(dists), R, B, N.ptr(oi), N.ptr(od), N.stream_ptr()))
    return oi, od


class ShardedGallery:
    """Row shard of a gallery plus the two collectives around the local match."""

    def __init__(self, shard_rows, index_base, group=None, match_fn=None, merge_fn=None, gallery=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._match = match_fn or _hip_match
        self._merge = merge_fn or _hip_merge
        if gallery is not None:
            self.gallery = gallery
        elif match_fn is None:
            from .oneshot import Gallery
            self.gallery = Gallery(shard_rows, index_base=index_base)
        else:
            self.gallery = (shard_rows, index_base)     # stand-in compute gets the raw shard

    def all_gather_embeddings(self, local):
        """Step 2.  `local` is [b, d] on every rank (same b); returns [R*b, d]."""
        if self.world == 1:
            return local
        out = torch.empty((self.world * local.shape[0], local.shape[1]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)
        return out

    def match(self, local_embeddings, distance_metric=1):
        """Steps 2-5: returns (idx[R*b] int64 global, dist[R*b] float32), identical on every rank."""
        probes = self.all_gather_embeddings(local_embeddings)
        key, idx, d = self._match(self.gallery, probes, distance_metric)
        if self.world == 1:
            return idx, d
        B = probes.shape[0]
        # (key, dist) travel as one float32 tensor, idx as int64: two all-gathers of a few KiB each
        kd = torch.cat([key.to(torch.float32), d.to(torch.float32)])               # [2B]
        kd_all = torch.empty((self.world * 2 * B,), dtype=kd.dtype, device=kd.device)
        ix_all = torch.empty((self.world * B,), dtype=torch.int64, device=idx.device)
        dist.all_gather_into_tensor(kd_all, kd.contiguous(), group=self.group)
        dist.all_gather_into_tensor(ix_all, idx.contiguous(), group=self.group)
        kd_all = kd_all.view(self.world, 2, B)
        ix_all = ix_all.view(self.world, B)
        return self._merge(kd_all[:, 0].contiguous(), ix_all, kd_all[:, 1].contiguous())
