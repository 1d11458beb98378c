"""Two ranks sharing the one GPU of the test box run the whole sharded pipeline with the
real HIP kernels: per-rank embed -> all-gather -> gallery-sharded dif_match -> all-gather
of partials -> dif_match_merge.  gloo carries the collectives here (RCCL needs one GPU per
rank); the data path and every kernel are the ones bench.py --gpus N uses."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_inputs as gi

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _CpuCollectives:
    """all_gather_into_tensor for CUDA tensors over gloo: stage through the host."""

    def __init__(self):
        self._orig = dist.all_gather_into_tensor

    def __call__(self, out, inp, group=None):
        o = torch.empty(out.shape, dtype=out.dtype)
        self._orig(o, inp.cpu(), group=group)
        out.copy_(o)


def _worker(rank, world, port, G, b, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        dist.all_gather_into_tensor = _CpuCollectives()
        from deep_insight_face.parallel import ShardedGallery, shard_bounds
        gal = gi.gallery(G, seed=3)
        gal[G - 40:G - 10] = gal[5:35]
        probes, _ = gi.probes_from(gal, world * b, seed=4)
        lo, hi = shard_bounds(G, world, rank)
        sg = ShardedGallery(torch.from_numpy(gal[lo:hi]).cuda(), lo)
        res = {}
        for m in (0, 1):
            idx, d = sg.match(torch.from_numpy(probes[rank * b:(rank + 1) * b]).cuda(), m)
            res['idx%d' % m], res['d%d' % m] = idx.cpu().numpy(), d.cpu().numpy()
        np.savez(os.path.join(out_dir, 'r%d.npz' % rank), **res)
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu(cuda, tmp_path):
    from oracle import distance as od
    world, G, b = 2, 3001, 37
    mp.spawn(_worker, args=(world, _free_port(), G, b, str(tmp_path)), nprocs=world, join=True)
    gal = gi.gallery(G, seed=3)
    gal[G - 40:G - 10] = gal[5:35]
    probes, _ = gi.probes_from(gal, world * b, seed=4)
    for m in (0, 1):
        want_idx, want_d, _ = od.match(probes, gal, m)
        for r in range(world):
            z = np.load(os.path.join(str(tmp_path), 'r%d.npz' % r))
            assert np.array_equal(z['idx%d' % m], want_idx)
            if m == 0:
                np.testing.assert_allclose(z['d%d' % m], want_d, atol=1e-5)
            else:
                np.testing.assert_allclose(np.cos(z['d%d' % m].astype(np.float64) * np.pi),
                                           np.cos(want_d.astype(np.float64) * np.pi), atol=1e-5)


def _step_worker(rank, world, port, b, G, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        dist.all_gather_into_tensor = _CpuCollectives()
        from deep_insight_face.networks.triplet import DifEmbedder
        from deep_insight_face.parallel import ShardedGallery, shard_bounds
        model = DifEmbedder('resnet', 'v2', 512, (112, 112, 3), max_batch=world * b).init_synthetic(2024)
        model.set_input_transform(scale=1 / 255.)
        rng = np.random.default_rng(77)
        enrol = rng.integers(0, 256, (world * b, 112, 112, 3), dtype=np.uint8)
        noisy = np.clip(enrol.astype(np.int64) + rng.integers(-5, 6, enrol.shape), 0, 255).astype(np.uint8)
        gal = gi.gallery(G, seed=3)
        pos = np.random.default_rng(78).choice(G, world * b, replace=False)
        gal_t = torch.from_numpy(gal).cuda()
        gal_t[torch.from_numpy(pos).cuda()] = model.embed(torch.from_numpy(enrol).cuda())
        lo, hi = shard_bounds(G, world, rank)
        sg = ShardedGallery(gal_t[lo:hi], lo)
        mine = torch.from_numpy(noisy[rank * b:(rank + 1) * b]).cuda()
        for _ in range(2):                                   # second pass reuses the preallocated step buffers
            idx, d = sg.match(model.embed(mine), 1)
        np.savez(os.path.join(out_dir, 's%d.npz' % rank), idx=idx.cpu().numpy(), d=d.cpu().numpy(), pos=pos)
    finally:
        dist.destroy_process_group()


def test_two_ranks_whole_step(cuda, tmp_path):
    """The whole N > 1 step on two ranks: per-rank embed of its half of the batch -> all-gather of the
    embeddings -> match against the rank's gallery shard -> ONE all-gather of the packed partial records
    -> dif_match_merge_packed.  Every rank must name the planted enrolment of every probe of both ranks."""
    world, b, G = 2, 6, 2001
    mp.spawn(_step_worker, args=(world, _free_port(), b, G, str(tmp_path)), nprocs=world, join=True)
    z0 = np.load(os.path.join(str(tmp_path), 's0.npz'))
    z1 = np.load(os.path.join(str(tmp_path), 's1.npz'))
    assert np.array_equal(z0['idx'], z0['pos']) and np.array_equal(z1['idx'], z0['pos'])
    assert np.array_equal(z0['d'], z1['d']) and float(z0['d'].max()) < 0.2


def test_single_rank_input_forms(cuda):
    """with world == 1 ShardedGallery.match hands the caller's tensor to dif_match.  A
    non-contiguous view, another dtype / device or NumPy input must be converted (as Gallery.match does), a
    wrong width refused; the allocation-free forms refuse what they cannot take as it is; copy=False returns
    the step buffers the next call overwrites, copy=True (default) does not."""
    from deep_insight_face import oneshot
    from deep_insight_face.parallel import ShardedGallery
    gal = gi.gallery(3000, seed=5)
    probes, pick = gi.probes_from(gal, 20, seed=6)
    sg = ShardedGallery(torch.from_numpy(gal).cuda(), 0)
    assert sg.world == 1
    want = torch.from_numpy(pick)
    wide = torch.zeros((20, 1024), device='cuda')
    wide[:, :512] = torch.from_numpy(probes).cuda()           # e.g. the first half of a flipped-concat embedding
    for form in (wide[:, :512], torch.from_numpy(probes).cuda().double(), torch.from_numpy(probes).half(),
                 torch.from_numpy(probes), probes, torch.from_numpy(probes).cuda().t().contiguous().t()):
        idx, d = sg.match(form, 1)
        assert torch.equal(idx.cpu(), want), type(form)
    with pytest.raises(ValueError, match=r'must be \[b, 512\]'):
        sg.match(wide, 1)
    with pytest.raises(ValueError):
        sg.match(torch.zeros(512, device='cuda'), 1)
    # buffer-reuse contract
    p_t = torch.from_numpy(probes).cuda()
    i1, _ = sg.match(p_t, 1)
    i2, _ = sg.match(p_t.flip(0), 1)
    assert torch.equal(i1.cpu(), want) and torch.equal(i2.cpu(), want.flip(0))
    j1, _ = sg.match(p_t, 1, copy=False)
    j2, _ = sg.match(p_t.flip(0), 1, copy=False)
    assert j1.data_ptr() == j2.data_ptr() and torch.equal(j1.cpu(), want.flip(0))
    # Gallery.match_into: no conversion, so everything it cannot read as dense float32 [B, 512] is refused
    g = oneshot.Gallery(torch.from_numpy(gal).cuda())
    oi = torch.empty(20, dtype=torch.int64, device='cuda')
    od_ = torch.empty(20, dtype=torch.float32, device='cuda')
    g.match_into(p_t, 1, oi, od_)
    assert torch.equal(oi.cpu(), want)
    for bad in (wide[:, :512], p_t.double(), p_t.cpu(), probes, p_t[:, :256]):
        with pytest.raises(ValueError):
            g.match_into(bad, 1, oi, od_)
    with pytest.raises(ValueError):
        g.match_into(p_t, 1, oi[:10], od_)
    with pytest.raises(ValueError):
        g.match_into(p_t, 1, oi.int(), od_)
    with pytest.raises(ValueError):
        g.match_into(p_t, 1, oi, od_, key=od_.double())
    g.close()


def test_rccl_single_rank_runs_the_sharded_branch(cuda, golden_dir, tmp_path):
    """RCCL itself executes the N > 1 branch on the one GPU there is.  A FRESH child
    process initialises backend "nccl" (= RCCL) with world_size 1 before any other GPU call and runs
    ShardedGallery(force_collectives=True): all_gather_into_tensor of the embeddings -> dif_match into the
    packed record -> the packed all-gather -> dif_match_merge_packed.  The answers must be the plain
    Gallery.match's and the reference's (tests/golden/match_near_ties.npz)."""
    import subprocess
    import sys
    out = os.path.join(str(tmp_path), 'rccl.npz')
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'rccl_single_rank_child.py')
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    r = subprocess.run([sys.executable, child, str(_free_port()), out], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    z = np.load(out)
    g = np.load(os.path.join(golden_dir, 'match_near_ties.npz'))
    assert str(z['backend']) == 'nccl' and int(z['world']) == 1
    assert bool(z['gathered_equal']) and bool(z['copy_false_same_buffer'])
    for m in (0, 1):
        assert np.array_equal(z['idx%d' % m], g['idx%d' % m])
        assert np.array_equal(z['idx%d' % m], z['plain_idx%d' % m])
        assert np.array_equal(z['d%d' % m], z['plain_d%d' % m], equal_nan=True)
    assert np.array_equal(z['idx_ragged'], g['idx1'][:5])
