"""Parity of the HIP distance / 1:N match path (through the C ABI) against the golden
vectors of the reference, against the CPU oracle on seeded inputs, and -- at BASELINE
sizes -- through size-independent properties.  Tolerance: distances within 1e-5
(float32), arg-min indices bit-identical (north_star)."""
import os

import numpy as np
import pytest
import torch

import golden_inputs as gi
from oracle import distance as od

pytestmark = pytest.mark.gpu
ATOL = 1e-5


def _cos_ok(full_row_sim):
    # arccos amplifies dot-product rounding near s -> 1 (SURVEY.md section 7 "arccos conditioning")
    return full_row_sim <= 0.999


def test_library_is_the_hip_one(cuda):
    from deep_insight_face import _native
    assert _native.lib.dif_device_count() >= 1
    assert os.path.basename(_native.LIB_PATH) == 'libdif.so'


def test_distance_vs_golden(cuda, golden_dir):
    from deep_insight_face.evaluation import utility
    g = np.load(os.path.join(golden_dir, 'distance_pairs.npz'))
    e1, e2 = gi.pair_inputs()
    d0 = utility.distance(e1, e2, 0)
    d1 = utility.distance(e1, e2, 1)
    assert isinstance(d0, np.ndarray) and d0.dtype == np.float32
    np.testing.assert_allclose(d0, g['d0'], rtol=2e-6, atol=ATOL)
    np.testing.assert_allclose(d1, g['d1'], atol=ATOL)
    np.testing.assert_allclose(utility.get_emd_distance(e1, e2, 1), g['emd1'], atol=ATOL)
    np.testing.assert_allclose(utility.get_emd_distance(e1, e2, 0), g['emd0'], rtol=2e-6, atol=ATOL)


def test_distance_broadcast_and_errors(cuda):
    from deep_insight_face.evaluation import utility
    e1, e2 = gi.pair_inputs(16)
    for m in (0, 1):
        np.testing.assert_allclose(utility.distance(e1[3][None, :], e2, m), od.distance(e1[3][None, :], e2, m),
                                   rtol=2e-6, atol=ATOL)
        np.testing.assert_allclose(utility.distance(e1, e2[5], m), od.distance(e1, e2[5][None, :], m),
                                   rtol=2e-6, atol=ATOL)
    with pytest.raises(RuntimeError, match='Undefined distance metric 10'):
        utility.distance(e1, e2, 10)
    with pytest.raises(ValueError):
        utility.distance(e1[:3], e2[:4], 0)
    assert utility.distance(e1[:0], e2[:0], 0).shape == (0,)
    t = utility.distance(torch.from_numpy(e1).cuda(), torch.from_numpy(e2).cuda(), 1)
    assert torch.is_tensor(t) and t.is_cuda
    # odd embedding sizes (the reference default is 128-d; any d works for row pairs)
    a = np.random.default_rng(0).standard_normal((5, 77)).astype(np.float32)
    b = np.random.default_rng(1).standard_normal((5, 77)).astype(np.float32)
    np.testing.assert_allclose(utility.distance(a, b, 0), od.distance(a, b, 0), rtol=2e-6)


@pytest.mark.parametrize('name,maker', [('match_b8_g1000.npz', gi.match_inputs),
                                        ('match_ties.npz', gi.match_tie_inputs),
                                        ('match_unnormalised.npz', gi.match_unnormalised_inputs)])
def test_match_vs_golden(cuda, golden_dir, name, maker):
    from deep_insight_face import oneshot
    g = np.load(os.path.join(golden_dir, name))
    probes, gallery = maker()
    gal = oneshot.Gallery(gallery)
    assert len(gal) == gallery.shape[0]
    for m in (0, 1):
        idx, dist = gal.match(probes, m)
        assert idx.dtype == np.int64 and dist.dtype == np.float32
        assert np.array_equal(idx, g['idx%d' % m]), (m, idx, g['idx%d' % m])
        want = g['full%d' % m][np.arange(len(idx)), idx]
        if m == 0:
            np.testing.assert_allclose(dist, want, rtol=2e-6, atol=ATOL)
        else:
            sim = np.cos(want.astype(np.float64) * np.pi)
            ok = _cos_ok(sim)
            np.testing.assert_allclose(dist[ok], want[ok], atol=ATOL)
            # similarity itself always within 1e-5
            np.testing.assert_allclose(np.cos(dist.astype(np.float64) * np.pi), sim, atol=ATOL)
    gal.close()


@pytest.mark.parametrize('B,G,D', [(1, 1, 32), (3, 129, 64), (64, 1000, 512), (65, 257, 512), (200, 5000, 128),
                                   (256, 4096, 512), (300, 777, 512)])
def test_match_vs_oracle_shapes(cuda, B, G, D):
    """Ragged sizes: tile tails in both the gallery and the probe dimension."""
    from deep_insight_face import oneshot
    gal_np = gi.gallery(G, seed=100 + G, d=D)
    p_np, pick = gi.probes_from(gal_np, min(B, G), seed=B)
    if B > G:
        p_np = np.concatenate([p_np] * (B // G + 1))[:B]
    gal = oneshot.Gallery(gal_np)
    for m in (0, 1):
        idx, dist = gal.match(p_np, m)
        oi, od_, full = od.match(p_np, gal_np, m)
        assert np.array_equal(idx, oi)
        if m == 0:
            np.testing.assert_allclose(dist, od_, rtol=2e-6, atol=ATOL)
        else:
            np.testing.assert_allclose(np.cos(dist.astype(np.float64) * np.pi),
                                       np.cos(od_.astype(np.float64) * np.pi), atol=ATOL)
    gal.close()


def test_match_empty_and_errors(cuda):
    from deep_insight_face import oneshot
    gal_np = gi.gallery(10)
    gal = oneshot.Gallery(gal_np)
    idx, dist = gal.match(np.zeros((0, 512), dtype=np.float32), 1)
    assert idx.shape == (0,) and dist.shape == (0,)
    with pytest.raises(RuntimeError, match='Undefined distance metric 2'):
        gal.match(gal_np[:2], 2)
    with pytest.raises(ValueError):
        gal.match(np.zeros((2, 128), dtype=np.float32), 1)
    empty = oneshot.Gallery(emd_size=512)
    with pytest.raises(ValueError):
        empty.match(gal_np[:2], 1)
    with pytest.raises(ValueError):
        oneshot.Gallery(emd_size=100)      # embedding size must be a multiple of 32
    i, d = oneshot.one_shot_clf(gal_np[4], gal_np, 1)
    assert i == 4 and d < 1e-3


def test_match_full_size_properties(cuda):
    """BASELINE config 2 size (B=256, G=100k, d=512), checked without the oracle:
    planted probes are found, every reported distance equals the row-paired distance
    to the reported row, no other row is closer than the reported one (spot-checked on
    random rows), and the match is invariant under probe scaling (cosine)."""
    from deep_insight_face import oneshot
    from deep_insight_face.evaluation import utility
    G, B = 100_000, 256
    gal_np = gi.gallery(G, seed=7)
    p_np, pick = gi.probes_from(gal_np, B, seed=11)
    gal_t = torch.from_numpy(gal_np).cuda()
    p_t = torch.from_numpy(p_np).cuda()
    gal = oneshot.Gallery(gal_t)
    for m in (0, 1):
        idx, dist = gal.match(p_t, m)
        assert torch.is_tensor(idx) and idx.is_cuda
        assert np.array_equal(idx.cpu().numpy(), pick)
        paired = utility.distance(p_t, gal_t[idx], m)
        np.testing.assert_allclose(dist.cpu().numpy(), paired.cpu().numpy(), atol=1e-6)
        rnd = torch.from_numpy(np.random.default_rng(m).integers(0, G, (B,))).cuda()
        other = utility.distance(p_t, gal_t[rnd], m)
        assert bool(torch.all(other >= dist - 1e-6))
    i2, _ = gal.match(p_t * 3.5, 1)
    assert torch.equal(i2.cpu(), torch.from_numpy(pick))
    # first-minimum rule at scale: duplicate the winners at the end of the gallery
    gal2 = torch.cat([gal_t, gal_t[torch.from_numpy(pick).cuda()]])
    g2 = oneshot.Gallery(gal2)
    i3, _ = g2.match(p_t, 1)
    assert np.array_equal(i3.cpu().numpy(), pick)
    gal.close()
    g2.close()


def test_sharded_merge_equals_whole(cuda):
    """Row-sharded gallery + dif_match_merge == one gallery (the 8-GPU data path, run
    here on one device): lowest key then lowest global index."""
    from deep_insight_face import oneshot, _native as N
    G, B, R = 4000, 100, 4
    gal_np = gi.gallery(G, seed=3)
    gal_np[3000:3050] = gal_np[500:550]      # ties across shards
    p_np, _ = gi.probes_from(gal_np, B, seed=4)
    whole = oneshot.Gallery(gal_np)
    for m in (0, 1):
        wi, wd = whole.match(p_np, m)
        keys, idxs, dists = [], [], []
        for r in range(R):
            lo, hi = r * G // R, (r + 1) * G // R
            sh = oneshot.Gallery(gal_np[lo:hi], index_base=lo)
            i, d, k = sh.match(torch.from_numpy(p_np).cuda(), m, return_key=True)
            keys.append(k), idxs.append(i), dists.append(d)
            sh.close()
        keys, idxs, dists = torch.stack(keys), torch.stack(idxs), torch.stack(dists)
        oi = torch.empty(B, dtype=torch.int64, device='cuda')
        odist = torch.empty(B, dtype=torch.float32, device='cuda')
        N.check(N.lib.dif_match_merge(N.ptr(keys), N.ptr(idxs), N.ptr(dists), R, B, N.ptr(oi), N.ptr(odist),
                                      N.stream_ptr()))
        assert np.array_equal(oi.cpu().numpy(), wi)
        np.testing.assert_allclose(odist.cpu().numpy(), wd, atol=1e-6)
    whole.close()


def test_match_million_row_gallery(cuda):
    """BASELINE config 4 scale on one GPU: 1M x 512 gallery (2 GB), 512 probes.  Checked through
    size-independent properties: planted probes are found at their planted rows, the reported
    distance equals the row-paired distance to the reported row, a sharded search over 8 row
    shards + merge gives the same answer, and duplicating the winners at the END of the gallery
    does not move the arg-min (first minimum wins)."""
    from deep_insight_face import oneshot, _native as N
    from deep_insight_face.evaluation import utility
    from deep_insight_face.parallel import shard_bounds
    G, B, R = 1_000_000, 512, 8
    g = torch.Generator(device='cuda').manual_seed(7)
    gal = torch.nn.functional.normalize(torch.randn((G, 512), generator=g, device='cuda'), dim=1)
    pick = torch.randperm(G, generator=g, device='cuda')[:B]
    probes = torch.nn.functional.normalize(gal[pick] + 0.03 * torch.randn((B, 512), generator=g, device='cuda'), dim=1)
    whole = oneshot.Gallery(gal)
    for m in (1, 0):
        idx, dist = whole.match(probes, m)
        assert torch.equal(idx, pick)
        paired = utility.distance(probes, gal[idx], m)
        assert torch.allclose(dist, paired, atol=1e-6)
    keys, idxs, dists = [], [], []
    for r in range(R):
        lo, hi = shard_bounds(G, R, r)
        sh = oneshot.Gallery(gal[lo:hi], index_base=lo)
        i, d, k = sh.match(probes, 1, return_key=True)
        keys.append(k), idxs.append(i), dists.append(d)
        sh.close()
    keys, idxs, dists = torch.stack(keys), torch.stack(idxs), torch.stack(dists)
    oi = torch.empty(B, dtype=torch.int64, device='cuda')
    od_ = torch.empty(B, dtype=torch.float32, device='cuda')
    N.check(N.lib.dif_match_merge(N.ptr(keys), N.ptr(idxs), N.ptr(dists), R, B, N.ptr(oi), N.ptr(od_), N.stream_ptr()))
    assert torch.equal(oi, pick)
    whole.close()
    g2 = oneshot.Gallery(torch.cat([gal, gal[pick]]))
    i2, _ = g2.match(probes, 1)
    assert torch.equal(i2, pick)
    g2.close()
