"""The N>1 data path (all-gather of per-rank embeddings -> gallery-sharded match ->
all-gather of partials -> lowest-index merge) with world_size 2, 3 and 8 on the gloo
backend, CPU only.  The local compute is stood in by the oracle (the HIP kernels need
a GPU; their sharded-merge parity is covered by tests/test_match_gpu.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_inputs as gi
from oracle import distance as od


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_match(gallery, probes, metric):
    rows, base = gallery
    idx, best, _ = od.match(probes.numpy(), rows, metric)
    # search key with the properties the HIP key has: monotone in the distance, comparable across shards
    return torch.from_numpy(best.copy()), torch.from_numpy(idx + base), torch.from_numpy(best)


def _cpu_merge(keys, idx, dists):
    k, i, d = keys.numpy(), idx.numpy(), dists.numpy()
    R, B = k.shape
    oi = np.zeros(B, dtype=np.int64)
    o_d = np.zeros(B, dtype=np.float32)
    for b in range(B):
        order = sorted(range(R), key=lambda r: (k[r, b], i[r, b]))
        oi[b], o_d[b] = i[order[0], b], d[order[0], b]
    return torch.from_numpy(oi), torch.from_numpy(o_d)


def _worker(rank, world, port, G, b, metric, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from deep_insight_face.parallel import ShardedGallery, shard_bounds
        gal = gi.gallery(G, seed=3)
        gal[G - 40:G - 10] = gal[5:35]              # equal minima living on different shards
        probes, _ = gi.probes_from(gal, world * b, seed=4)
        lo, hi = shard_bounds(G, world, rank)
        sg = ShardedGallery(gal[lo:hi], lo, match_fn=_oracle_match, merge_fn=_cpu_merge)
        idx, d = sg.match(torch.from_numpy(probes[rank * b:(rank + 1) * b]), metric)
        np.savez(os.path.join(out_dir, 'r%d.npz' % rank), idx=idx.numpy(), d=d.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,metric', [(2, 1), (2, 0), (3, 1), (8, 1)])     # 8: the node bench.py's largest line runs on (1001 rows: ragged shards)
def test_sharded_match_equals_whole(tmp_path, world, metric):
    G, b = 1001, 5
    port = _free_port()
    mp.spawn(_worker, args=(world, port, G, b, metric, str(tmp_path)), nprocs=world, join=True)
    gal = gi.gallery(G, seed=3)
    gal[G - 40:G - 10] = gal[5:35]
    probes, _ = gi.probes_from(gal, world * b, seed=4)
    want_idx, want_d, _ = od.match(probes, gal, metric)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), 'r%d.npz' % r))
        assert np.array_equal(z['idx'], want_idx)          # identical on every rank, first minimum wins
        assert np.array_equal(z['d'], want_d)


def test_shard_bounds_cover():
    from deep_insight_face.parallel import shard_bounds
    for G in (0, 1, 7, 8, 100_000, 1_000_003):
        for R in (1, 2, 3, 8):
            edges = [shard_bounds(G, R, r) for r in range(R)]
            assert edges[0][0] == 0 and edges[-1][1] == G
            assert all(edges[i][1] == edges[i + 1][0] for i in range(R - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


def _force_worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from deep_insight_face.parallel import ShardedGallery
        gal = gi.gallery(501, seed=3)
        probes, _ = gi.probes_from(gal, 6, seed=4)
        calls = {'n': 0}
        orig = dist.all_gather_into_tensor

        def counting(out, inp, group=None):
            calls['n'] += 1
            return orig(out, inp, group=group)

        dist.all_gather_into_tensor = counting
        res = {}
        for force in (False, True):
            calls['n'] = 0
            sg = ShardedGallery(gal, 0, match_fn=_oracle_match, merge_fn=_cpu_merge, force_collectives=force)
            idx, d = sg.match(torch.from_numpy(probes), 1)
            res['idx%d' % force], res['d%d' % force], res['calls%d' % force] = idx.numpy(), d.numpy(), calls['n']
        np.savez(os.path.join(out_dir, 'f.npz'), **res)
    finally:
        dist.destroy_process_group()


def test_force_collectives_world_of_one(tmp_path):
    """A world of one rank short-cuts both all-gathers; force_collectives=True runs them (what the GPU suite
    uses to put RCCL under the N > 1 branch on a single GPU).  Same answers either way."""
    from deep_insight_face.parallel import ShardedGallery
    with pytest.raises(RuntimeError, match='init_process_group'):
        ShardedGallery(np.zeros((4, 8), np.float32), 0, match_fn=_oracle_match, force_collectives=True)
    mp.spawn(_force_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    z = np.load(os.path.join(str(tmp_path), 'f.npz'))
    assert int(z['calls0']) == 0 and int(z['calls1']) == 2
    assert np.array_equal(z['idx0'], z['idx1']) and np.array_equal(z['d0'], z['d1'])


def _ragged_worker(rank, world, port, mode, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from deep_insight_face.parallel import ShardedGallery, shard_bounds
        G = 301
        gal = gi.gallery(G, seed=3)
        probes, _ = gi.probes_from(gal, 16, seed=4)
        lo, hi = shard_bounds(G, world, rank)
        sg = ShardedGallery(gal[lo:hi], lo, match_fn=_oracle_match, merge_fn=_cpu_merge, check_batch=mode)
        log = []
        if mode == 'always':
            # a full step first (shape cached on both ranks), THEN the ragged last batch: rank 0 still brings 5 probes
            idx, _ = sg.match(torch.from_numpy(probes[rank * 5:(rank + 1) * 5]), 1)
            log.append('ok%d' % len(idx))
        b = 5 if rank == 0 else 4
        try:
            sg.match(torch.from_numpy(probes[:b]), 1)
            log.append('no error')
        except ValueError as e:
            log.append('ValueError: %s' % e)
        # the group is still usable afterwards (both ranks left the step at the same collective)
        idx, _ = sg.match(torch.from_numpy(probes[rank * 3:(rank + 1) * 3]), 1)
        log.append('ok%d' % len(idx))
        with open(os.path.join(out_dir, 'r%d.txt' % rank), 'w') as fh:
            fh.write('\n'.join(log))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('mode', ['always', 'first'])
def test_ragged_batch_across_ranks_raises(tmp_path, mode):
    """ShardedGallery.match sized its gather buffers from THIS rank's b; ranks with b = 5 and 4
    (a ragged last batch) gave a size-mismatched all_gather_into_tensor -- hang or garbage.  Now every rank raises
    ValueError naming the sizes, before any buffer is touched, and the next well-formed step works."""
    mp.spawn(_ragged_worker, args=(2, _free_port(), mode, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        log = open(os.path.join(str(tmp_path), 'r%d.txt' % r)).read().split('\n')
        if mode == 'always':
            assert log[0] == 'ok10'
            log = log[1:]
        assert log[0].startswith('ValueError') and '[5, 4]' in log[0], log
        assert log[1] == 'ok6'
    from deep_insight_face.parallel import ShardedGallery
    with pytest.raises(ValueError, match='check_batch'):
        ShardedGallery(np.zeros((4, 8), np.float32), 0, match_fn=_oracle_match, check_batch='sometimes')
