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
            assert np.array_equal(dist, want)               # bit-identical: same float32 operations, same order
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


# ------------------------------------------------------------------------------------------------
# Near-ties: the search key only filters; the winner is chosen on the reference's own float32
# arithmetic (csrc/match.hip stages 2 and 3)
def test_reference_arithmetic_bit_exact(cuda, golden_dir):
    """dif_pairwise evaluates utility.py:54-62 in NumPy's summation order: metric 0 is bit-identical to
    the reference's output, and so is the cosine similarity inside metric 1 (DIF_METRIC_SIMILARITY); the
    distance of metric 1 differs from the reference's only through arccos (NumPy: SVML, <= 2 ulp)."""
    from deep_insight_face import _native as N
    from deep_insight_face.evaluation import utility
    g = np.load(os.path.join(golden_dir, 'distance_pairs.npz'))
    e1, e2 = gi.pair_inputs()
    d0 = utility.distance(e1, e2, 0)
    assert np.array_equal(d0.view(np.uint32), g['d0'].view(np.uint32))
    rng = np.random.default_rng(5)
    for d in (512, 128, 96, 77, 8, 7, 1000, 129):
        a = (rng.standard_normal((33, d)) * rng.uniform(0.1, 30, (33, 1))).astype(np.float32)
        b = (a + 0.3 * rng.standard_normal((33, d))).astype(np.float32)
        b[0] = a[0]
        ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        for metric, want in ((0, od.distance(a, b, 0)), (2, od.similarity(a, b))):
            out = torch.empty(33, dtype=torch.float32, device='cuda')
            N.check(N.lib.dif_pairwise(N.ptr(ta), 33, N.ptr(tb), 33, d, metric, N.ptr(out), N.stream_ptr()))
            assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32)), (d, metric)
        with np.errstate(invalid='ignore'):
            want1 = od.distance(a, b, 1)
        got1 = utility.distance(a, b, 1)
        ok = ~np.isnan(want1)
        assert np.abs(got1[ok].view(np.int32).astype(np.int64) - want1[ok].view(np.int32)).max() <= 4   # ulps
        assert np.all(np.isnan(got1[~ok]))                 # reference NaN (s > 1) <-> NaN


@pytest.mark.parametrize('metric', [0, 1])
def test_match_near_ties_vs_golden(cuda, golden_dir, metric):
    """Gallery rows 1 ulp / 1e-7 / 1e-4 apart from each other and from the probes, exact duplicates,
    unnormalised rows with large |g|^2: the HIP arg-min equals the reference's on every probe, including
    the probes whose reference distance is NaN (np.argmin: first NaN)."""
    from deep_insight_face import oneshot
    g = np.load(os.path.join(golden_dir, 'match_near_ties.npz'))
    probes, gallery = gi.match_near_tie_inputs()
    assert np.array_equal(g['sha'], gi.digest(probes, gallery))
    gal = oneshot.Gallery(gallery)
    idx, dist, key = gal.match(probes, metric, return_key=True)
    assert np.array_equal(idx, g['idx%d' % metric]), np.nonzero(idx != g['idx%d' % metric])
    want = g['dmin%d' % metric]
    if metric == 0:
        assert np.array_equal(dist.view(np.uint32), want.view(np.uint32))
        assert np.array_equal(key, dist)
    else:
        nan = np.isnan(want)
        assert nan.sum() >= 5
        assert np.all(np.isnan(dist[nan])) and np.all(np.isneginf(key[nan]))     # NaN like the reference
        assert np.abs(dist[~nan].view(np.int32).astype(np.int64) - want[~nan].view(np.int32)).max() <= 4
        # option "clamp_nan": same winners, distance 0 in place of the NaN of a similarity rounded above 1
        gal.set_option('clamp_nan', 1)
        i3, d3 = gal.match(probes, metric)
        assert np.array_equal(i3, idx) and np.all(d3[nan] == 0.0) and np.array_equal(d3[~nan], dist[~nan])
        gal.set_option('clamp_nan', 0)
    # one probe at a time and in ragged groups (other tiles, other block counts): same winners
    for lo, hi in ((0, 1), (1, 34), (34, 48)):
        i2, _ = gal.match(probes[lo:hi], metric)
        assert np.array_equal(i2, g['idx%d' % metric][lo:hi])
    # the same probes three times over and then some (48 -> 160 > 64 probes, ragged against the 128-probe tile), on the
    # one-term bf16 filter (match_b1_kernel: the default, also what served the calls above), the two-term filter's B-direct
    # kernel (match_bd_kernel) and, 'bd' = 0, its match_tile_kernel.  Every copy of a probe must get the fixture's answer.
    rep = np.concatenate([probes, probes, probes, probes[:16]])
    want_rep = np.concatenate([g['idx%d' % metric]] * 3 + [g['idx%d' % metric][:16]])
    # ... and the one-term filter on its row-major copy ('frag' = 0: match_b1_kernel) as well as on the fragment-order one
    # (match_g1_kernel, the default where the embedding size allows it: it served every call above)
    for flt, bd, frag in ((2, 1, 2), (2, 1, 0), (1, 1, 1), (1, 0, 1)):
        gal.set_option('filter', flt)
        gal.set_option('bd', bd)
        gal.set_option('frag', frag)
        i5, d5 = gal.match(rep, metric)
        assert np.array_equal(i5, want_rep), (flt, bd, np.nonzero(i5 != want_rep)[0])
        assert np.array_equal(d5[:48], dist, equal_nan=True)
        i4, d4, k4 = gal.match(probes, metric, return_key=True)
        assert np.array_equal(i4, idx) and np.array_equal(d4, dist, equal_nan=True) and np.array_equal(k4, key)
    gal.set_option('bd', 1)
    gal.set_option('frag', 1)
    # the filter stage on the f32 MFMA instead of bf16 operands: the same answers, bit for bit (the filter only
    # proposes candidates; the reference arithmetic decides)
    gal.set_option('filter', 0)
    i4, d4, k4 = gal.match(probes, metric, return_key=True)
    assert np.array_equal(i4, idx) and np.array_equal(d4, dist, equal_nan=True) and np.array_equal(k4, key)
    for lo, hi in ((0, 1), (1, 34), (34, 48)):
        i2, _ = gal.match(probes[lo:hi], metric)
        assert np.array_equal(i2, g['idx%d' % metric][lo:hi])
    gal.close()


@pytest.mark.parametrize('metric', [0, 1])
def test_match_near_ties_sharded(cuda, golden_dir, metric):
    """The same fixture row-sharded 3 and 7 ways + the packed merge (parallel.py's N > 1 record layout)."""
    from deep_insight_face import oneshot, _native as N
    from deep_insight_face.parallel import shard_bounds
    g = np.load(os.path.join(golden_dir, 'match_near_ties.npz'))
    probes, gallery = gi.match_near_tie_inputs()
    B = probes.shape[0]
    p_t = torch.from_numpy(probes).cuda()
    for R in (3, 7):
        packed = torch.empty((R, 4 * B), dtype=torch.float32, device='cuda')
        for r in range(R):
            lo, hi = shard_bounds(gallery.shape[0], R, r)
            sh = oneshot.Gallery(gallery[lo:hi], index_base=lo)
            rec = packed[r]
            sh.match_into(p_t, metric, rec[2 * B:].view(torch.int64), rec[B:2 * B], rec[:B])
            torch.cuda.synchronize()
            sh.close()
        oi = torch.empty(B, dtype=torch.int64, device='cuda')
        odist = torch.empty(B, dtype=torch.float32, device='cuda')
        N.check(N.lib.dif_match_merge_packed(N.ptr(packed), R, B, N.ptr(oi), N.ptr(odist), N.stream_ptr()))
        assert np.array_equal(oi.cpu().numpy(), g['idx%d' % metric]), R


def _degenerate_cases():
    return [c[0] for c in gi.match_degenerate_cases()]


@pytest.mark.parametrize('name', _degenerate_cases())
def test_match_degenerate_vs_golden(cuda, golden_dir, name):
    """zero-norm / tiny / huge / non-finite gallery rows and probes, rows anti-parallel
    to a probe, an all-zero gallery.  evaluation/utility.py:58-62 guards none of it, so the reference's
    answer is IEEE arithmetic + np.argmin's first-NaN rule; the fixture holds what the reference returned.
    Index bit-identical; distance NaN exactly where the reference's is NaN, bit-identical for metric 0,
    within 4 ulp (arccos) for metric 1.  Also sharded 3 ways through the packed merge."""
    from deep_insight_face import oneshot, _native as N
    from deep_insight_face.parallel import shard_bounds
    g = np.load(os.path.join(golden_dir, 'match_degenerate.npz'))
    probes, gallery = [(p, gl) for n, p, gl in gi.match_degenerate_cases() if n == name][0]
    assert np.array_equal(g[name + '_sha'], gi.digest(probes, gallery))
    B = probes.shape[0]
    gal = oneshot.Gallery(gallery)
    p_t = torch.from_numpy(probes).cuda()
    for metric in (0, 1):
        want_i, want_d = g['%s_idx%d' % (name, metric)], g['%s_dmin%d' % (name, metric)]
        idx, dist, key = gal.match(probes, metric, return_key=True)
        assert np.array_equal(idx, want_i), (metric, np.nonzero(idx != want_i)[0], idx, want_i)
        nan = np.isnan(want_d)
        assert np.array_equal(np.isnan(dist), nan), metric
        assert np.all(np.isneginf(key[nan]))
        if metric == 0:
            assert np.array_equal(dist[~nan].view(np.uint32), want_d[~nan].view(np.uint32))
        elif (~nan).any():
            assert np.abs(dist[~nan].view(np.int32).astype(np.int64) - want_d[~nan].view(np.int32)).max() <= 4
        # more than 64 probes (the fixture's, repeated): match_b1_kernel / match_bd_kernel and their wave-local epilogue
        reps = (80 + B - 1) // B + 1
        for flt, frag in ((2, 2), (2, 0), (1, 1)):               # frag 2 / 0: match_g1_kernel / match_b1_kernel
            gal.set_option('filter', flt)
            gal.set_option('frag', frag)
            i6, d6 = gal.match(np.concatenate([probes] * reps), metric)
            assert np.array_equal(i6, np.concatenate([want_i] * reps)), (metric, 'filter', flt, frag)
            assert np.array_equal(np.isnan(d6), np.concatenate([nan] * reps))
        # ragged groups of probes (other tile shapes), on every filter (f32, two-term, one-term)
        for flt in (0, 1, 2):
            gal.set_option('filter', flt)
            for lo, hi in ((0, 1), (1, 12), (12, B)):
                i2, _ = gal.match(probes[lo:hi], metric)
                assert np.array_equal(i2, want_i[lo:hi]), (metric, lo, hi, flt)
        R = 3
        packed = torch.empty((R, 4 * B), dtype=torch.float32, device='cuda')
        for r in range(R):
            lo, hi = shard_bounds(gallery.shape[0], R, r)
            sh = oneshot.Gallery(gallery[lo:hi], index_base=lo)
            rec = packed[r]
            sh.match_into(p_t, metric, rec[2 * B:].view(torch.int64), rec[B:2 * B], rec[:B])
            torch.cuda.synchronize()
            sh.close()
        oi = torch.empty(B, dtype=torch.int64, device='cuda')
        odist = torch.empty(B, dtype=torch.float32, device='cuda')
        N.check(N.lib.dif_match_merge_packed(N.ptr(packed), R, B, N.ptr(oi), N.ptr(odist), N.stream_ptr()))
        assert np.array_equal(oi.cpu().numpy(), want_i), metric
        assert np.array_equal(np.isnan(odist.cpu().numpy()), nan)
    gal.close()


def test_match_zero_probes_take_no_exact_search(cuda):
    """padded (all-zero) probe rows under the cosine metric used to overflow every block's
    candidate list and cost a whole-gallery exact scan each.  They are now answered directly (row 0, NaN --
    what np.argmin over an all-NaN row returns), and a gallery of zero rows (an unfilled shard) is answered
    from its first row: neither may send a probe to the exact search."""
    from deep_insight_face import oneshot
    G, B = 200_000, 256
    gt = torch.nn.functional.normalize(torch.randn((G, 512), device='cuda', generator=torch.Generator('cuda').manual_seed(3)), dim=1)
    gal = oneshot.Gallery(gt)
    good = torch.nn.functional.normalize(gt[:B] + 0.01 * torch.randn((B, 512), device='cuda'), dim=1)
    zeros = torch.zeros((B, 512), device='cuda')
    # the mechanism itself, not a wall-clock gate: how many probes the call sent to the exact search
    i_good, _ = gal.match(good, 1)
    assert torch.equal(i_good.cpu(), torch.arange(B)) and gal.stat('exact_probes') == 0
    i_zero, d_zero = gal.match(zeros, 1)
    assert torch.all(i_zero == 0) and torch.all(torch.isnan(d_zero))
    assert gal.stat('exact_probes') == 0
    inf_probe = good.clone()
    inf_probe[3, 7] = float('inf')                # an out-of-range probe that is NOT all-NaN does take it
    gal.match(inf_probe, 0)
    assert gal.stat('exact_probes') == 1
    empty_shard = oneshot.Gallery(torch.zeros((G, 512), device='cuda'), index_base=1000)
    i_es, d_es = empty_shard.match(good, 1)
    assert torch.all(i_es == 1000) and torch.all(torch.isnan(d_es))
    assert empty_shard.stat('exact_probes') == 0
    i0, d0 = empty_shard.match(good, 0)          # metric 0: zero rows are ordinary rows, all tied -> the first
    assert torch.all(i0 == 1000) and torch.allclose(d0, torch.ones(B, device='cuda'), atol=1e-5)
    gal.close()
    empty_shard.close()


def test_match_exact_search_on_overflow(cuda):
    """More near-minimal rows in one block than a candidate list holds (KCAND = 8): the probe is re-searched
    exactly over the whole gallery.  300 copies of the best row, the genuinely lowest index hidden among
    later ones, plus a row one ulp better at a high index for metric 0."""
    from deep_insight_face import oneshot
    G = 3000
    gal_np = gi.gallery(G, seed=55)
    rng = np.random.default_rng(56)
    probes, _ = gi.probes_from(gal_np, 6, seed=57)
    for b in range(6):
        base = gi._unit((probes[b] + 0.01 * rng.standard_normal(512))[None].astype(np.float32))[0]
        rows = np.sort(rng.choice(G, 300, replace=False))
        gal_np[rows] = base
    gal = oneshot.Gallery(gal_np)
    for m in (0, 1):
        with np.errstate(invalid='ignore'):
            oi, odist, _ = od.match(probes, gal_np, m)
        idx, dist = gal.match(probes, m)
        assert np.array_equal(idx, oi)
        if m == 0:
            assert np.array_equal(dist, odist)
    # a 4-probe batch next to them that does NOT overflow is unaffected
    clean, pick = gi.probes_from(gi.gallery(G, seed=55), 4, seed=58)
    i2, _ = gal.match(np.concatenate([probes[:2], clean]), 1)
    with np.errstate(invalid='ignore'):
        oi2, _, _ = od.match(np.concatenate([probes[:2], clean]), gal_np, 1)
    assert np.array_equal(i2, oi2)
    gal.close()


@pytest.mark.parametrize('B', [6, 130])
def test_one_term_filter_wide_net(cuda, B):
    """The one-term bf16 filter's proven bound is ~0.008 |q| (two-term: 1.6e-4 |q|).  A cluster of rows whose cosine to the
    probe differs by 1e-4 .. 3e-3 -- many images of one identity -- lies INSIDE that net and far outside the two-term one:
    the one-term filter must still return the reference's row (re-ranked among the cluster; the cluster's best row is not
    the one its bf16 dot ranks first for most probes), with 10 rows per cluster from its lists, with 40 -- in ONE wave row
    of one tile, more than its list of 13 holds -- from the exact search, and say so in 'exact_probes'; the two-term filter
    (two or three rows of the cluster inside ITS bound) needs neither for the clusters of 10."""
    from deep_insight_face import oneshot
    G = 20_000
    rng = np.random.default_rng(600 + B)
    gal_np = gi.gallery(G, seed=61)
    probes, _ = gi.probes_from(gal_np, B, seed=62)
    for size in (10, 40):
        g = gal_np.copy()
        for b in range(B):
            rows = 128 * b + 5 + np.arange(size)                                         # one wave row of one tile: one candidate list
            noise = rng.standard_normal((size, 512)).astype(np.float32)
            noise -= (noise @ probes[b])[:, None] * probes[b][None]                      # orthogonal to the probe
            noise /= np.linalg.norm(noise, axis=1, keepdims=True)
            t = np.sqrt(2 * rng.uniform(1e-4, 3e-3, size)).astype(np.float32)            # cos = 1 / sqrt(1 + t^2): 1e-4 .. 3e-3 below 1, evenly
            g[rows] = gi._unit(probes[b][None] + t[:, None] * noise)
        with np.errstate(invalid='ignore'):
            want = {m: od.match(probes, g, m)[0] for m in (0, 1)}
        gal = oneshot.Gallery(g)
        for m in (0, 1):
            gal.set_option('filter', 2)
            i2, d2 = gal.match(probes, m)
            flagged2 = gal.stat('exact_probes')
            gal.set_option('filter', 1)
            i1, d1 = gal.match(probes, m)
            flagged1 = gal.stat('exact_probes')
            assert np.array_equal(i2, want[m]) and np.array_equal(i1, want[m]), (size, m)
            assert np.array_equal(d2, d1)
            assert size == 40 or flagged1 == 0          # (40 rows 7e-5 apart can chain past its lists of 8 too: a running minimum that creeps down never restarts one)
            assert (flagged2 == 0) if size == 10 else (flagged2 > 0), (size, m, flagged2)
        gal.close()


def test_cosine_similarity_matrix(cuda):
    """the all-pairs cosine matrix of common/losses.py:39-40 as an entry point."""
    from deep_insight_face import oneshot
    rng = np.random.default_rng(8)
    a = (rng.standard_normal((70, 512)) * rng.uniform(0.2, 5, (70, 1))).astype(np.float32)
    b = (rng.standard_normal((333, 512)) * rng.uniform(0.2, 5, (333, 1))).astype(np.float32)
    an = a / np.linalg.norm(a, axis=1, keepdims=True)
    bn = b / np.linalg.norm(b, axis=1, keepdims=True)
    got = oneshot.cosine_similarity_matrix(a, b)
    assert got.shape == (70, 333) and got.dtype == np.float32
    np.testing.assert_allclose(got, an.astype(np.float64) @ bn.astype(np.float64).T, atol=2e-6)
    self_sim = oneshot.cosine_similarity_matrix(torch.from_numpy(a).cuda())
    assert torch.is_tensor(self_sim) and self_sim.shape == (70, 70)
    np.testing.assert_allclose(np.diag(self_sim.cpu().numpy()), 1.0, atol=2e-6)


def test_filter_option_in_any_order(cuda):
    """the filter's bf16 copy adds to the gallery's device memory (one-term default: + 50 %;
    two-term "filter" = 1: + 100 %).  Switching "filter" to 0 -- before OR after the rows are set -- does without it /
    frees it; switching it back on rebuilds it at the next match; the answers never change."""
    from deep_insight_face import oneshot
    gal_np = gi.gallery(5000, seed=91)
    probes, pick = gi.probes_from(gal_np, 70, seed=92)
    g = oneshot.Gallery(gal_np)
    assert g.stat('split_copy') == 1 and g.stat('row_bytes') == 2048 + 1024 + 8
    i0, d0 = g.match(probes, 1)
    g.set_option('filter', 0)                       # after the rows: the copy is given back
    assert g.stat('split_copy') == 0 and g.stat('row_bytes') == 2048 + 8
    i1, d1 = g.match(probes, 1)
    g.set(gal_np[::-1].copy())                      # a new set under filter 0 builds no copy
    assert g.stat('split_copy') == 0
    g.set(gal_np)
    g.set_option('filter', 1)                       # back on: built by the next match, not silently left on f32
    assert g.stat('split_copy') == 0
    i2, d2 = g.match(probes, 1)
    assert g.stat('split_copy') == 1 and g.stat('row_bytes') == 2 * 2048 + 8
    g.set_option('filter', 2)                       # the one-term copy replaces the two-term one
    assert g.stat('split_copy') == 0 and g.stat('row_bytes') == 2048 + 8
    i4, d4 = g.match(probes, 1)
    assert g.stat('split_copy') == 1 and g.stat('row_bytes') == 2048 + 1024 + 8
    assert np.array_equal(i4, pick) and np.array_equal(d4, d0)
    g.set(gal_np[:3000])                            # smaller set inside the capacity: the copy is refreshed
    i3, _ = g.match(probes[pick < 3000], 1)
    for i in (i0, i1, i2):
        assert np.array_equal(i, pick)
    assert np.array_equal(d0, d1) and np.array_equal(d0, d2)
    assert np.array_equal(i3, pick[pick < 3000])
    g.close()
    h = oneshot.Gallery(emd_size=512)
    h.set_option('filter', 0)                       # before the rows
    h.set(gal_np)
    assert h.stat('split_copy') == 0
    assert np.array_equal(h.match(probes, 1)[0], pick)
    with pytest.raises(ValueError):
        h.stat('nonsense')
    with pytest.raises(ValueError):
        h.set_option('filter', 3)
    h.close()


@pytest.mark.parametrize('G,B', [(300_003, 200), (70_000, 65), (1_000, 513)])
def test_match_bd_kernel_equals_tile_kernel(cuda, G, B):
    """The split-bf16 filter runs on match_bd_kernel from 65 probes up (256 x 128 tiles, probes in fragment order from L2, gallery
    by LDS-DMA, candidate lists per wave row) and on match_tile_kernel below / with option 'bd' = 0: same candidates in the
    end, so the same index, distance and key, bit for bit, both metrics -- gallery sizes that leave a ragged last tile of
    256 rows, probe counts that leave a ragged 128-probe block and a ragged 32-probe fragment, a gallery smaller than one
    tile per block, near-duplicate rows, and against the oracle on a sample."""
    from deep_insight_face import oneshot
    gen = torch.Generator(device='cuda').manual_seed(G + B)
    gal_t = torch.nn.functional.normalize(torch.randn((G, 512), device='cuda', generator=gen), dim=1)
    gal_t[G // 2:G // 2 + 40] = gal_t[7:47] * (1 + 1e-7)                      # near-duplicates far apart
    pick = torch.randperm(G, device='cuda', generator=gen)[:B]
    probes = torch.nn.functional.normalize(gal_t[pick] + 0.05 * torch.randn((B, 512), device='cuda', generator=gen), dim=1)
    probes[B // 3] = -gal_t[5]                                                  # an anti-parallel pair
    g = oneshot.Gallery(gal_t)
    for metric in (0, 1):
        g.set_option('filter', 2)                   # the default: one-term bf16 filter, match_g1_kernel on the fragment-order copy
        i2, d2, k2 = g.match(probes, metric, return_key=True)
        g.set_option('frag', 2)                     # ... match_g1_kernel on the fragment-order copy whatever the gallery's size (rebuilt by this call)
        i3, d3, k3 = g.match(probes, metric, return_key=True)
        g.set_option('frag', 0)                     # ... and match_b1_kernel on the row-major one
        i4, d4, k4 = g.match(probes, metric, return_key=True)
        assert torch.equal(i2, i4) and torch.equal(d2.view(torch.int32), d4.view(torch.int32)) and torch.equal(k2.view(torch.int32), k4.view(torch.int32))
        g.set_option('frag', 1)
        assert torch.equal(i2, i3) and torch.equal(d2.view(torch.int32), d3.view(torch.int32)) and torch.equal(k2.view(torch.int32), k3.view(torch.int32))
        g.set_option('filter', 1)
        g.set_option('bd', 1)
        i1, d1, k1 = g.match(probes, metric, return_key=True)
        g.set_option('bd', 0)
        i0, d0, k0 = g.match(probes, metric, return_key=True)
        assert torch.equal(i1, i0) and torch.equal(d1.view(torch.int32), d0.view(torch.int32)) and torch.equal(k1.view(torch.int32), k0.view(torch.int32))
        assert torch.equal(i2, i0) and torch.equal(d2.view(torch.int32), d0.view(torch.int32)) and torch.equal(k2.view(torch.int32), k0.view(torch.int32))
        rows = [0, 1, B // 3, B - 1]
        oi, _, _ = od.match(probes[rows].cpu().numpy(), gal_t.cpu().numpy(), metric)
        assert np.array_equal(i1[rows].cpu().numpy(), oi)
    g.close()


@pytest.mark.parametrize('flt', [2, 1, 0])
def test_gallery_update_equals_set(cuda, golden_dir, flt):
    """dif_gallery_update (round 5: enrolling k identities costs O(k)): overwriting rows in place,
    appending within and beyond the capacity, replacing a degenerate (zero / NaN / huge) row by an ordinary one and the
    other way round -- after every step the answers (index, distance bits, key bits; both metrics) are those of a fresh
    dif_gallery_set with the same rows, and on the degenerate fixture those of the reference."""
    from deep_insight_face import oneshot
    rng = np.random.default_rng(500 + flt)
    base = gi.gallery(6000, seed=93)
    probes, pick = gi.probes_from(base, 96, seed=94)

    def same_as_fresh(g, rows, p=probes):
        fresh = oneshot.Gallery(emd_size=512)
        fresh.set_option('filter', flt)
        fresh.set(rows)
        for metric in (0, 1):
            a = g.match(p, metric, return_key=True)
            b = fresh.match(p, metric, return_key=True)
            assert np.array_equal(a[0], b[0]), metric
            assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), metric
            assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32)), metric
        fresh.close()

    g = oneshot.Gallery(emd_size=512)
    g.set_option('filter', flt)
    g.set(base[:5000])
    assert g.capacity == 5000 and len(g) == 5000
    cur = base[:5000].copy()
    # overwrite in place: 8 rows become noisy copies of probes (the bench's "plant" step)
    new = probes[:8] + np.float32(0.001) * rng.standard_normal((8, 512)).astype(np.float32)
    g.update(new, 1234)
    cur[1234:1242] = new
    same_as_fresh(g, cur)
    i, _ = g.match(probes[:8], 1)
    assert np.array_equal(i, np.arange(1234, 1242))
    # append beyond the capacity (grows), then inside it
    g.update(base[5000:5600])
    cur = np.concatenate([cur, base[5000:5600]])
    assert len(g) == 5600 and g.capacity >= 5600
    same_as_fresh(g, cur)
    g.reserve(9000)
    assert g.capacity == 9000 and len(g) == 5600
    g.update(base[5600:6000], 5600)
    cur = np.concatenate([cur, base[5600:6000]])
    same_as_fresh(g, cur)
    # degenerate rows come ...
    bad = np.zeros((4, 512), np.float32)
    bad[1, 3] = np.nan
    bad[2] = base[77] * np.float32(1e19)
    bad[3] = base[78] * np.float32(1e-25)
    g.update(bad, 40)
    cur[40:44] = bad
    same_as_fresh(g, cur)
    # ... and go (the special lists must forget them: a stale first-NaN row would hide a later one)
    g.update(base[40:42], 40)
    cur[40:42] = base[40:42]
    same_as_fresh(g, cur)
    g.update(base[42:44], 42)
    cur[42:44] = base[42:44]
    same_as_fresh(g, cur)
    with pytest.raises(ValueError):
        g.update(base[:3], len(g) + 1)              # a gap
    with pytest.raises(ValueError):
        g.update(base[:3, :100])
    g.close()
    # the reference's own answers: every degenerate fixture rebuilt row group by row group through update
    gold = np.load(os.path.join(golden_dir, 'match_degenerate.npz'))
    for name, p, gl in gi.match_degenerate_cases():
        h = oneshot.Gallery(emd_size=gl.shape[1])
        h.set_option('filter', flt)
        h.set(gl[::-1].copy())                      # other rows first (special rows at other places)
        step = max(1, gl.shape[0] // 3 + 1)
        for lo in range(0, gl.shape[0], step):
            h.update(gl[lo:lo + step], lo)
        for metric in (0, 1):
            idx, dist = h.match(p, metric)
            assert np.array_equal(idx, gold['%s_idx%d' % (name, metric)]), (name, metric)
            assert np.array_equal(np.isnan(dist), np.isnan(gold['%s_dmin%d' % (name, metric)])), (name, metric)
        h.close()


def test_probe_workspace_survives_a_non_monotone_batch_sequence(cuda):
    """the probes' fragment-order copy was grown under the per-probe capacity but sized B + 32, so a
    call that only grew the partial-result workspace could reallocate it SMALLER than an earlier, larger batch needed and
    the next such batch wrote past its end (two-term filter: 4 B per value).  Batches 513 -> 482 -> 513 on a gallery
    large enough for the part count to differ; answers equal the one-term filter's and the planted rows."""
    from deep_insight_face import oneshot
    G = 1_000_000
    gen = torch.Generator(device='cuda').manual_seed(11)
    gal_t = torch.nn.functional.normalize(torch.randn((G, 512), device='cuda', generator=gen), dim=1)
    pick = torch.randperm(G, device='cuda', generator=gen)[:513]
    probes = torch.nn.functional.normalize(gal_t[pick] + 0.02 * torch.randn((513, 512), device='cuda', generator=gen), dim=1)
    g = oneshot.Gallery(gal_t)
    g.set_option('filter', 1)
    guard = torch.full((1 << 20,), 7.0, device='cuda')          # a neighbour for a stray write to land in
    for B in (513, 482, 513, 100, 513):
        i, _ = g.match(probes[:B], 1)
        assert torch.equal(i, pick[:B]), B
    torch.cuda.synchronize()
    assert bool((guard == 7.0).all())
    g.close()


@pytest.mark.parametrize('D', [128, 256, 384, 512, 192])
def test_fragment_order_copy_follows_updates_and_reserve(cuda, D):
    """The one-term filter's copy in MFMA-fragment order (gallery option 'frag', match_g1_kernel; embedding sizes that are
    multiples of 128 up to 512 -- 192 stays row-major on match_b1_kernel): enrolled whole (staged 8 rows at a time), updated in
    ragged runs that straddle 8-, 32- and 64-row boundaries (piece by piece), appended to past a reserve (the copy is moved),
    switched to the row-major layout and back (rebuilt by the next match) -- after every step the answers equal the f32
    filter's, bit for bit, and 'frag' = 0 gives the same."""
    from deep_insight_face import oneshot
    gen = torch.Generator(device='cuda').manual_seed(D)
    G, B = 5000, 150
    rows = torch.nn.functional.normalize(torch.randn((G + 700, D), device='cuda', generator=gen), dim=1)
    g = oneshot.Gallery(emd_size=D)
    g.set_option('frag', 2)                                   # (1, the default, keeps galleries below 2^18 rows row-major)
    g.set(rows[:G])

    def check(n, tag):
        pick = torch.randperm(n, device='cuda', generator=gen)[:B]
        probes = torch.nn.functional.normalize(rows_now[pick] + 0.05 * torch.randn((B, D), device='cuda', generator=gen), dim=1)
        out = {}
        for name, flt, frag in (('g1', 2, 2), ('b1', 2, 0), ('f32', 0, 1)):
            g.set_option('filter', flt)
            g.set_option('frag', frag)
            out[name] = [g.match(probes, m, return_key=True) for m in (0, 1)]
        g.set_option('filter', 2)
        g.set_option('frag', 2)
        for name in ('g1', 'b1'):
            for (i, d, k), (i0, d0, k0) in zip(out[name], out['f32']):
                assert torch.equal(i, i0) and torch.equal(d.view(torch.int32), d0.view(torch.int32)), (tag, name)
        assert torch.equal(out['g1'][1][0], pick), tag

    rows_now = rows[:G].clone()
    check(G, 'set')
    for first, count in ((3, 1), (29, 7), (60, 9), (1000, 333), (G - 5, 5)):
        new = torch.nn.functional.normalize(torch.randn((count, D), device='cuda', generator=gen), dim=1)
        g.update(new, first)
        rows_now[first:first + count] = new
    check(G, 'updated')
    g.reserve(G + 700)
    g.update(rows[G:G + 700], G)
    rows_now = torch.cat([rows_now, rows[G:G + 700]])
    check(G + 700, 'appended')
    g.close()


def test_fragment_layout_follows_the_row_count(cuda):
    """Gallery option 'frag' = 1 (the default): the one-term copy is kept in fragment order -- and the filter runs on
    match_g1_kernel -- from 2^18 rows up; a gallery that grows past the threshold by updates has its copy rewritten by the
    next match.  Same answers either side of it."""
    from deep_insight_face import oneshot
    gen = torch.Generator(device='cuda').manual_seed(18)
    D, G0, G1 = 128, (1 << 18) - 100, (1 << 18) + 150
    rows = torch.nn.functional.normalize(torch.randn((G1, D), device='cuda', generator=gen), dim=1)
    pick = torch.randperm(G0, device='cuda', generator=gen)[:70]
    probes = torch.nn.functional.normalize(rows[pick] + 0.05 * torch.randn((70, D), device='cuda', generator=gen), dim=1)
    g = oneshot.Gallery(emd_size=D)
    g.reserve(G1)
    g.update(rows[:G0], 0)
    i0, d0 = g.match(probes, 1)
    assert g.stat('frag_copy') == 0 and g.stat('filter_terms') == 1
    assert torch.equal(i0, pick)
    g.update(rows[G0:G1])                                        # across the threshold: the copy is stale in its layout ...
    i1, d1 = g.match(probes, 1)                                  # ... and rewritten here
    assert g.stat('frag_copy') == 1 and g.stat('filter_terms') == 1
    assert torch.equal(i1, i0) and torch.equal(d1.view(torch.int32), d0.view(torch.int32))
    big = oneshot.Gallery(rows)                                  # enrolled whole: fragment order from the start
    assert big.stat('frag_copy') == 1
    i2, d2 = big.match(probes, 1)
    assert torch.equal(i2, i0) and torch.equal(d2.view(torch.int32), d0.view(torch.int32))
    big.set_option('frag', 0)
    i3, d3 = big.match(probes, 1)
    assert big.stat('frag_copy') == 0 and torch.equal(i3, i0) and torch.equal(d3.view(torch.int32), d0.view(torch.int32))
    g.close()
    big.close()
