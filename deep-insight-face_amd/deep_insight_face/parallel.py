"""Data-parallel embed + gallery-sharded match across the GPUs of one node.

The reference is single-process (SURVEY.md section 5 "distributed communication backend:
none"); this is new work named by north_star.  One process per GPU, torch.distributed
for the exchange (backend "nccl" is RCCL over xGMI on ROCm; "gloo" on CPU for tests):

  1. every rank embeds its own slice of the batch                    [b, d]
  2. ONE all-gather of the per-rank embeddings                       [R*b, d]   (8 MiB at B=4096)
  3. every rank matches ALL probes against ITS gallery shard         (key, idx, dist)[R*b]
  4. ONE all-gather of the packed per-rank results                   [R][{key, dist, idx}[R*b]] (tiny)
  5. merge: lowest key (= the reference distance), then lowest GLOBAL index == np.argmin over the whole gallery

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


class _StepBuffers:
    """Device buffers of one (B, d) step shape, allocated once: the gathered probes, this rank's packed
    partial result and the gathered partial results, the merged output."""

    def __init__(self, world, b, d, dev):
        B = world * b
        self.probes = torch.empty((B, d), dtype=torch.float32, device=dev)
        # one rank's record = { float key[B]; float dist[B]; int64 idx[B]; } (include/dif.h: dif_match_merge_packed)
        self.packed = torch.empty((world, 4 * B), dtype=torch.float32, device=dev)
        self.local = torch.empty((4 * B,), dtype=torch.float32, device=dev)      # this rank's record (not aliasing `packed`)
        self.out_idx = torch.empty((B,), dtype=torch.int64, device=dev)
        self.out_dist = torch.empty((B,), dtype=torch.float32, device=dev)

    def record(self, B):
        rec = self.local
        return rec[0:B], rec[B:2 * B], rec[2 * B:4 * B].view(torch.int64)


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

    def __init__(self, shard_rows, index_base, group=None, match_fn=None, merge_fn=None, gallery=None,
                 force_collectives=False, check_batch='always'):
        """`force_collectives`: with a world of ONE rank the two all-gathers and the merge are short-cut (they
        would move a rank's data to itself); True runs them all the same -- the whole N > 1 branch over the real
        backend on a single GPU (tests/test_parallel_gpu.py::test_rccl_single_rank_runs_the_sharded_branch).
        Needs an initialised process group.
        `check_batch`: every rank must bring the SAME number of probes b to a step (all_gather_into_tensor sizes its
        buffers from this rank's b: a ragged last batch on one rank would otherwise hang or gather garbage).
        'always' (default) compares the ranks' b before every step -- one 8-byte all-gather and a host read, raising
        ValueError on EVERY rank when they differ; 'first' only when this rank meets a (b, d) shape for the first
        time (no host read in a steady serving loop: what bench.py times; a rank whose shape is cached does not take
        part, so it protects the first step of a shape only); 'never' trusts the caller."""
        if check_batch not in ('always', 'first', 'never'):
            raise ValueError("check_batch must be 'always', 'first' or 'never'")
        self.check_batch = check_batch
        self.phase_events = None      # bench.py: a list to which every step appends its five CUDA events (see match)
        self.group = group
        self.force = bool(force_collectives)
        if self.force and not dist.is_initialized():
            raise RuntimeError('force_collectives needs torch.distributed.init_process_group first')
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._match = match_fn
        self._merge = merge_fn
        self._bufs = {}
        if gallery is not None:
            self.gallery = gallery
        elif match_fn is None:
            from .oneshot import Gallery
            self.gallery = Gallery(shard_rows, index_base=index_base)
        else:
            self.gallery = (shard_rows, index_base)     # stand-in compute gets the raw shard

    def all_gather_embeddings(self, local, out=None):
        """Step 2.  `local` is [b, d] on every rank (same b); returns [R*b, d]."""
        if self.world == 1 and not self.force:
            return local
        if out is None:
            out = torch.empty((self.world * local.shape[0], local.shape[1]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)
        return out

    def _check_same_batch(self, b, device, first_time):
        """Raises ValueError on every rank when the ranks' probe counts differ (see __init__: check_batch)."""
        if self.world == 1 or self.check_batch == 'never' or (self.check_batch == 'first' and not first_time):
            return
        mine = torch.tensor([int(b)], dtype=torch.int64, device=device)
        sizes = torch.empty((self.world,), dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(sizes, mine, group=self.group)
        sizes = [int(v) for v in sizes.cpu().tolist()]
        if any(v != sizes[0] for v in sizes):
            raise ValueError('ShardedGallery.match: every rank must bring the same number of probes per step, got %s '
                             '(pad the ragged last batch, or drop it, on all ranks alike)' % sizes)

    def match(self, local_embeddings, distance_metric=1, copy=True):
        """Steps 2-5: returns (idx[R*b] int64 global, dist[R*b] float32), identical on every rank.
        `local_embeddings`: [b, emd_size], NumPy or torch on any device / float dtype / strides (converted to a
        dense float32 tensor on this rank's GPU, like Gallery.match; a wrong width raises ValueError).
        On the HIP path the match writes its (key, dist, idx) straight into this rank's record of one packed,
        preallocated buffer, ONE all-gather moves the records, dif_match_merge_packed reduces them.
        copy=True (default) returns fresh tensors; copy=False returns the step buffers themselves, which the
        NEXT call of the same shape overwrites -- the allocation-free form for a serving loop."""
        if self._match is not None:
            return self._match_injected(local_embeddings, distance_metric)
        from . import _native as N
        local_embeddings, _ = N.to_device_f32(local_embeddings, self.gallery._dev)
        if local_embeddings.dim() != 2 or local_embeddings.shape[1] != self.gallery.emd_size:
            raise ValueError('embeddings must be [b, %d], got %s' % (self.gallery.emd_size,
                                                                      tuple(local_embeddings.shape)))
        b, d = local_embeddings.shape
        key = (b, d, local_embeddings.device)
        buf = self._bufs.get(key)
        self._check_same_batch(b, local_embeddings.device, buf is None)
        if buf is None:
            buf = self._bufs[key] = _StepBuffers(self.world, b, d, local_embeddings.device)
        B = self.world * b
        evs = None
        if self.phase_events is not None:                     # [start, embeddings gathered, matched, records gathered, merged]
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
            self.phase_events.append(evs)
            evs[0].record()
        probes = self.all_gather_embeddings(local_embeddings, buf.probes)
        if evs:
            evs[1].record()
        k, dd, ix = buf.record(B)
        if self.world == 1 and not self.force:
            self.gallery.match_into(probes, distance_metric, buf.out_idx, buf.out_dist)
            if evs:
                for e in evs[2:]:
                    e.record()
        else:
            self.gallery.match_into(probes, distance_metric, ix, dd, k)
            if evs:
                evs[2].record()
            dist.all_gather_into_tensor(buf.packed.view(-1), buf.local, group=self.group)
            if evs:
                evs[3].record()
            N.check(N.lib.dif_match_merge_packed(N.ptr(buf.packed), self.world, B, N.ptr(buf.out_idx),
                                                 N.ptr(buf.out_dist), N.stream_ptr()))
            if evs:
                evs[4].record()
        if copy:
            return buf.out_idx.clone(), buf.out_dist.clone()
        return buf.out_idx, buf.out_dist

    def _match_injected(self, local_embeddings, distance_metric):
        """The same exchange with stand-in compute (CPU tests over gloo: the HIP kernels need a GPU)."""
        shape = tuple(local_embeddings.shape)
        self._check_same_batch(shape[0], local_embeddings.device, shape not in self._bufs)
        self._bufs.setdefault(shape, True)
        probes = self.all_gather_embeddings(local_embeddings)
        key, idx, d = self._match(self.gallery, probes, distance_metric)
        if self.world == 1 and not self.force:
            return idx, d
        B = probes.shape[0]
        rec = torch.cat([key.to(torch.float32), d.to(torch.float32), idx.to(torch.int64).view(torch.float32)])   # [4B]
        allrec = torch.empty((self.world * 4 * B,), dtype=torch.float32, device=rec.device)
        dist.all_gather_into_tensor(allrec, rec.contiguous(), group=self.group)
        allrec = allrec.view(self.world, 4 * B)
        keys, dists = allrec[:, 0:B].contiguous(), allrec[:, B:2 * B].contiguous()
        ix = allrec[:, 2 * B:4 * B].contiguous().view(torch.int64)
        return (self._merge or _hip_merge)(keys, ix, dists)
