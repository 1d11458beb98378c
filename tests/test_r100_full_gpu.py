"""The benchmarked configurations, checked at their real sizes (VERDICT r01 "test what you benchmark").

bench.py's default line is IResNet-100 at 512 faces per GPU against a 1M-row gallery, and the
north_star fraction is quoted on the batch-256 forward: both run the DEFAULT executor policy (two
lanes, full-chip stream-K grids, the software-pipelined kernel on the 64-channel stage), none of which
a 2-image batch reaches.  These tests run exactly those paths (DIF_STREAMS is NOT set) and check
size-independent properties plus spot rows against the oracle (oracle/nets.py: parity unpinned, see
DESIGN.md section 3).  Tolerance: cosine 1e-5 (north_star), arg-min identities exact.
"""
import numpy as np
import pytest
import torch

from oracle import distance as od
from oracle import nets

pytestmark = pytest.mark.gpu
TOL = 1e-5


def crops_u8(n, hw=112, seed=1234):
    return np.random.default_rng(seed).integers(0, 256, (n, hw, hw, 3), dtype=np.uint8)


def scaled(u8):
    return u8.astype(np.float32) / np.float32(255.0)


def cosine_gap(a, b):
    a = a.reshape(a.shape[0], -1).astype(np.float64)
    b = b.reshape(b.shape[0], -1).astype(np.float64)
    return 1.0 - (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))


@pytest.fixture(scope='module')
def r100(cuda):
    """One IResNet-100 (max_batch 512, default lane policy) shared by the tests of this module."""
    import os
    assert 'DIF_STREAMS' not in os.environ and 'DIF_PIPE' not in os.environ
    from deep_insight_face.networks.triplet import DifEmbedder
    model = DifEmbedder('iresnet100', 'v2', 512, (112, 112, 3), max_batch=512).init_synthetic(2024)
    model.set_input_transform(scale=1 / 255.)
    yield model, model.get_weights()
    model.close()


@pytest.mark.parametrize('batch', [256, 512])
def test_iresnet100_full_batch(r100, batch):
    """Batch 256 (the north_star roofline configuration) and 512 (bench.py's per-GPU batch): unit
    norm, finite, rows equal the same crops in a batch of 8 (one lane, no split), permutation
    equivariance, run-to-run determinism, and 8 spot rows -- both lanes, both ends -- vs the oracle."""
    model, p = r100
    u8 = crops_u8(batch, seed=4200 + batch)
    dev = torch.from_numpy(u8).cuda()
    full = model.embed(dev)
    again = model.embed(dev)
    assert torch.equal(full, again)                               # stream-K split is a pure function of the size
    full = full.cpu().numpy()
    assert full.shape == (batch, 512) and np.all(np.isfinite(full))
    np.testing.assert_allclose(np.linalg.norm(full, axis=1), 1.0, atol=1e-5)
    half = batch // 2
    for lo in (0, half - 4, batch - 8):                           # lane 0 head, the lane seam, lane 1 tail
        small = model.predict_on_batch(u8[lo:lo + 8])
        assert cosine_gap(small, full[lo:lo + 8]).max() < 1e-6, lo
    perm = np.random.default_rng(0).permutation(batch)
    permuted = model.predict_on_batch(u8[perm])
    assert cosine_gap(permuted, full[perm]).max() < 1e-6
    rows = [0, 1, half - 1, half, half + 1, batch - 65, batch - 2, batch - 1]
    want = nets.embed(scaled(u8[rows]), p, 'iresnet100', 512, 'v2')
    assert cosine_gap(full[rows], want).max() < TOL
    # every pairwise cosine distance among the spot rows within 1e-5 of the oracle's
    for i in range(len(rows)):
        a = od.distance(np.repeat(full[rows][i][None], len(rows), 0), full[rows], 1)
        b = od.distance(np.repeat(want[i][None], len(rows), 0), want, 1)
        mask = np.arange(len(rows)) != i
        np.testing.assert_allclose(a[mask], b[mask], atol=TOL)


def test_iresnet100_one_lane_vs_default(r100, monkeypatch):
    """The default policy gives IResNet-100 two lanes; DIF_STREAMS=1 forces the single-lane executor.
    Same weights, same 256 crops: embeddings agree to float32 rounding (different stream-K split points)."""
    from deep_insight_face.networks.triplet import DifEmbedder
    model, p = r100
    u8 = crops_u8(256, seed=77)
    two = model.predict_on_batch(u8)
    monkeypatch.setenv('DIF_STREAMS', '1')
    one = DifEmbedder('iresnet100', 'v2', 512, (112, 112, 3), max_batch=256)
    one.set_weights(p)
    one.set_input_transform(scale=1 / 255.)
    one._finalize()
    monkeypatch.delenv('DIF_STREAMS')
    single = one.predict_on_batch(u8)
    one.close()
    assert cosine_gap(two, single).max() < 1e-6
    np.testing.assert_allclose(two, single, atol=5e-6)


def test_config2_embed_then_arcmargin(r100):
    """BASELINE configs[2] end to end at its real size: 512 crops -> IResNet-100 -> ArcMargin logits over
    85 742 classes (HIP), against the oracle's logits on all rows x sampled class columns (the label
    columns, where the margin is applied, included)."""
    from deep_insight_face.networks.arcmargin import ArcMarginHead
    model, p = r100
    C = 85_742
    u8 = crops_u8(512, seed=314)
    emb = model.embed(torch.from_numpy(u8).cuda())
    g = torch.Generator(device='cpu').manual_seed(99)
    w = torch.randn((C, 512), generator=g)
    labels = torch.randint(0, C, (512,), generator=g)
    head = ArcMarginHead(w.cuda())
    logits = head.logits(emb, labels.cuda()).cpu().numpy()
    assert logits.shape == (512, C) and np.all(np.isfinite(logits))
    cols = np.unique(np.concatenate([labels.numpy(), np.random.default_rng(1).integers(0, C, 256), [0, C - 1]]))
    remap = {int(c): i for i, c in enumerate(cols)}
    sub_labels = np.array([remap[int(c)] for c in labels.numpy()])
    want = nets.arcmargin_logits(emb.cpu().numpy(), w.numpy()[cols], sub_labels)
    np.testing.assert_allclose(logits[:, cols], want, atol=64 * 2e-6 + 1e-5, rtol=1e-5)    # s = 64: cos within ~2e-6
    # the embeddings behind them are the oracle's (spot rows)
    rows = [0, 255, 256, 511]
    assert cosine_gap(emb[rows].cpu().numpy(), nets.embed(scaled(u8[rows]), p, 'iresnet100', 512, 'v2')).max() < TOL


def test_config3_per_gpu_shape_embed_then_1m_match(r100):
    """BASELINE configs[3], one GPU's share: 512 faces -> IResNet-100 -> top-1 over a 1M-row gallery.
    Every probe's enrolment (its embedding + noise, re-normalised) is planted at a random row: the
    match must return exactly those rows; 4 probes are also checked against the reference formula
    over the whole gallery (oracle/distance.py == evaluation/utility.py:52-66 + np.argmin)."""
    from deep_insight_face import oneshot
    model, _ = r100
    G = 1_000_000
    u8 = crops_u8(512, seed=2718)
    emb = model.embed(torch.from_numpy(u8).cuda())
    g = torch.Generator(device='cuda').manual_seed(7)
    gal = torch.nn.functional.normalize(torch.randn((G, 512), generator=g, device='cuda'), dim=1)
    pos = torch.from_numpy(np.random.default_rng(5).choice(G, 512, replace=False)).cuda()
    planted = torch.nn.functional.normalize(emb + 0.02 * torch.randn(emb.shape, generator=g, device='cuda'), dim=1)
    gal[pos] = planted
    gallery = oneshot.Gallery(gal)
    idx, dist = gallery.match(emb, 1)
    assert torch.equal(idx, pos)
    assert float(dist.max()) < 0.2 and float(dist.min()) > 0.0
    gal_np = gal.cpu().numpy()
    e_np = emb.cpu().numpy()
    for r in (0, 17, 300, 511):
        oi, odist, _ = od.match(e_np[r:r + 1], gal_np, 1)
        assert int(oi[0]) == int(idx[r])
        assert abs(np.cos(float(odist[0]) * np.pi) - np.cos(float(dist[r]) * np.pi)) < TOL
    gallery.close()


@pytest.mark.parametrize('arch,head,emd,hw,n', [
    ('resnet', 'v2', 512, 112, 37), ('resnet', 'v2', 512, 112, 257), ('resnet', 'v2', 128, 96, 130),
    ('resnet', 'v1', 64, 128, 70), ('iresnet50', 'v2', 512, 112, 65), ('iresnet50', 'v2', 256, 96, 33),
    ('iresnet100', 'v2', 512, 112, 129), ('mobilenet', 'v2', 512, 112, 129), ('vgg16', 'sv2', 64, 96, 31),
    ('nn4', 'v2', 128, 96, 200), ('yolov3', 'v3', 1, 320, 5)])
def test_odd_sizes_pipelined_vs_plain_vs_oracle(cuda, monkeypatch, arch, head, emd, hw, n):
    """Odd batch and input sizes (ragged tiles, ragged lane splits, partial XCD chunks): the
    software-pipelined kernel against the plain one, and the first / last row against the oracle."""
    from deep_insight_face.networks.triplet import DifEmbedder
    rng = np.random.default_rng(n * 7 + hw)
    x = torch.from_numpy(rng.integers(0, 256, (n, hw, hw, 3), dtype=np.uint8)).cuda()
    m = DifEmbedder(arch, head, emd, (hw, hw, 3), max_batch=n).init_synthetic(3)
    m.set_input_transform(scale=1 / 255.)
    m.set_option('pipe', 1)
    a = m.embed(x)
    m.set_option('pipe', 0)
    b = m.embed(x)
    m.set_option('pipe', 1)
    al, bl = (a if isinstance(a, list) else [a]), (b if isinstance(b, list) else [b])
    for ta, tb in zip(al, bl):
        assert bool(torch.isfinite(ta).all())
        assert float((ta - tb).abs().max()) <= 5e-6 * max(float(tb.abs().max()), 1.0)
    if arch not in ('yolov3', 'nn4'):
        rows = [0, n - 1]
        want = nets.embed(scaled(x[rows].cpu().numpy()), m.get_weights(), arch, emd, head)
        assert cosine_gap(a[rows].cpu().numpy(), want).max() < TOL
    m.close()


@pytest.mark.parametrize('batch', [256, 512])
@pytest.mark.parametrize('compute', ['bf16x3', 'bf16x2'])
def test_iresnet100_full_batch_bf16x3(r100, batch, compute):
    """The split-bf16 throughput modes ("bf16x3": six products; "bf16x2": hi + mid, three products) (conv.hip: gemm_mainloop_patch_bf3, the B-direct halo-patch kernel on 128 x 128 tiles
    with the stream-K grid cut at slice boundaries) at the benchmarked sizes and under the default executor: against the
    float32 path of the same weights on the whole batch (cosine gap < 1e-6), against the oracle on spot rows of both lanes
    (gap < 1e-5, pairwise cosine distances within 1e-5), finite, unit norm, deterministic, and on a gallery built from the
    float32 embeddings every split-bf16 probe names its own row."""
    from deep_insight_face import oneshot
    from deep_insight_face.networks.triplet import DifEmbedder
    model, p = r100
    b3 = DifEmbedder('iresnet100', 'v2', 512, (112, 112, 3), max_batch=512, compute=compute)
    b3.set_weights(p)
    b3.set_input_transform(scale=1 / 255.)
    u8 = crops_u8(batch, seed=5200 + batch)
    dev = torch.from_numpy(u8).cuda()
    ref = model.embed(dev)
    got = b3.embed(dev)
    assert torch.equal(got, b3.embed(dev))
    assert not torch.equal(got, ref)                                # really another arithmetic
    g, r = got.cpu().numpy(), ref.cpu().numpy()
    assert np.all(np.isfinite(g))
    np.testing.assert_allclose(np.linalg.norm(g, axis=1), 1.0, atol=1e-5)
    assert cosine_gap(g, r).max() < 1e-6
    half = batch // 2
    rows = [0, half - 1, half, batch - 1]
    want = nets.embed(scaled(u8[rows]), p, 'iresnet100', 512, 'v2')
    assert cosine_gap(g[rows], want).max() < TOL
    for i in range(len(rows)):
        a = od.distance(np.repeat(g[rows][i][None], len(rows), 0), g[rows], 1)
        b = od.distance(np.repeat(want[i][None], len(rows), 0), want, 1)
        mask = np.arange(len(rows)) != i
        np.testing.assert_allclose(a[mask], b[mask], atol=TOL)
    gal = oneshot.Gallery(ref)
    idx, _ = gal.match(got, 1)
    assert torch.equal(idx.cpu(), torch.arange(batch))
    gal.close()
    b3.close()
