"""Host-side pieces of bench.py and of the oracle that need no GPU."""
import importlib.util
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gallery_shards_do_not_depend_on_the_world_size():
    """Every rank generates only its shard, from per-block seeds: a row's value depends on its global index alone, so the
    shards of any world size concatenate to the same gallery (VERDICT r03 #1 b)."""
    from deep_insight_face.parallel import shard_bounds
    b = _bench()
    b.GALLERY_BLOCK = 1000                       # small blocks so that shards cut through and across them
    G = 4321
    whole = b.synthetic_gallery(0, G, 16, 7, 'cpu')
    assert whole.shape == (G, 16)
    assert torch.allclose(whole.norm(dim=1), torch.ones(G), atol=1e-6)
    for world in (2, 3, 8):
        parts = [b.synthetic_gallery(*shard_bounds(G, world, r), 16, 7, 'cpu') for r in range(world)]
        assert torch.equal(torch.cat(parts), whole), world
    assert b.synthetic_gallery(5, 5, 16, 7, 'cpu').shape == (0, 16)       # an empty shard
    assert not torch.equal(b.synthetic_gallery(0, 10, 16, 8, 'cpu'), whole[:10])


def test_layer_rooflines_and_kernel_shares():
    b = _bench()
    prof = [('a', 'k1', 1e9, 0.5), ('b', 'k2', 1e6, 0.25), ('c', 'k1', 2e9, 0.25)]
    traffic = [(1e6, 4e6), (8e9, 0.0), (2e6, 0.0)]
    t_ms, by, hbm = b.layer_rooflines(prof, traffic, 2)
    t_a = max(2 * 1e9 * 2 / (b.PEAK_F32_MFMA_TFLOPS * 1e12), (1e6 * 2 + 4e6) / (b.HBM_COPY_TBS * 1e12))
    t_b = max(2 * 1e6 * 2 / (b.PEAK_F32_MFMA_TFLOPS * 1e12), 16e9 / (b.HBM_COPY_TBS * 1e12))
    t_c = max(2 * 2e9 * 2 / (b.PEAK_F32_MFMA_TFLOPS * 1e12), 4e6 / (b.HBM_COPY_TBS * 1e12))
    assert abs(t_ms - (t_a + t_b + t_c) * 1e3) < 1e-9 and hbm == ['b'] and by == 6e6 + 16e9 + 4e6
    shares = b.kernel_shares(prof)
    assert shares[0] == ('k1', 0.75, 2) and shares[1] == ('k2', 0.25, 1)


def test_area_resize_restatement_properties():
    """oracle/imageops.area_resize restates cv2's uint8 INTER_AREA paths (unpinned: cv2 is not installed).  What can be
    checked without cv2: a constant image stays constant on every path; the 2 x 2 path is (a + b + c + d + 2) >> 2; other
    integer ratios are the block mean rounded half to even; fractional shrinking stays within one level of the float64
    coverage mean; enlarging reproduces the source at the pixels whose footprint lies inside one source pixel; (width,
    height) order."""
    from oracle import imageops as oi
    rng = np.random.default_rng(5)
    for shape in ((224, 224), (336, 336), (160, 160), (96, 96), (300, 180), (64, 200)):
        c = np.full(shape + (3,), 77, np.uint8)
        assert np.all(oi.area_resize(c, 112) == 77), shape
    img = rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)
    blocks = img.reshape(112, 2, 112, 2, 3).astype(np.int32).sum(axis=(1, 3))
    assert np.array_equal(oi.area_resize(img, 112), ((blocks + 2) >> 2).astype(np.uint8))
    img = rng.integers(0, 256, (336, 336, 3), dtype=np.uint8)
    mean = img.reshape(112, 3, 112, 3, 3).astype(np.float64).mean(axis=(1, 3))
    got = oi.area_resize(img, 112).astype(np.float64)
    assert np.abs(got - mean).max() <= 0.5 + 1e-4
    img = rng.integers(0, 256, (250, 250, 3), dtype=np.uint8)
    s = 250 / 112
    cov = np.zeros((112, 250))
    for o in range(112):
        for i in range(int(np.floor(o * s)), min(int(np.ceil((o + 1) * s)), 250)):
            cov[o, i] = max(min((o + 1) * s, i + 1) - max(o * s, i), 0.0)
        cov[o] /= cov[o].sum()
    want = np.einsum('yi,ijc,xj->yxc', cov, img.astype(np.float64), cov)
    assert np.abs(oi.area_resize(img, 112).astype(np.float64) - want).max() <= 0.5 + 1e-3
    img = rng.integers(0, 256, (56, 56, 3), dtype=np.uint8)
    up = oi.area_resize(img, 112)                    # exactly 2x: every destination pixel lies inside one source pixel
    assert np.array_equal(up, np.repeat(np.repeat(img, 2, axis=0), 2, axis=1))
    assert oi.area_resize(rng.integers(0, 256, (200, 100, 3), dtype=np.uint8), (50, 100)).shape == (100, 50, 3)
