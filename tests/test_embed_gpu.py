"""Parity of the HIP embedding forward (through the C ABI) against the CPU oracle on
the same seeded crops and weights.  Tolerance (north_star): cosine distances within 1e-5
in float32, arg-min identities bit-identical."""
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


def build(arch, head, emd, max_batch=8):
    from deep_insight_face.networks.triplet import bottleneck_network
    if head == 'sv2':                      # the siamese builder's v2 head (networks/siamese.py:107-128)
        from deep_insight_face.networks import siamese
        model = siamese.bottleneck_network(arch, emd, (112, 112, 3), max_batch=max_batch)('v2')
    else:
        model = bottleneck_network(arch, emd, (112, 112, 3), max_batch=max_batch)(head)
    model.init_synthetic(2024)
    return model, model.get_weights()


@pytest.mark.parametrize('arch,head,emd,n', [('resnet', 'v2', 512, 8), ('resnet', 'v1', 128, 5),
                                             ('resnet', 'v3', 512, 3), ('iresnet50', 'v2', 512, 3),
                                             ('iresnet100', 'v2', 512, 2), ('vgg16', 'v2', 512, 3),
                                             ('mobilenet', 'v2', 512, 5), ('mobilenet', 'v3', 512, 2),
                                             ('mobilenet', 'v1', 128, 2), ('resnet', 'sv2', 128, 3),
                                             ('vgg16', 'sv2', 128, 2),
                                             # the reference's own call shapes: ONE image per call
                                             # (predictions.py:152-156) and batch 12 (scripts/insight_face.py:112)
                                             ('resnet', 'v2', 512, 1), ('resnet', 'v2', 512, 12),
                                             ('iresnet100', 'v2', 512, 1), ('iresnet100', 'v2', 512, 12)])
def test_embed_vs_oracle(cuda, arch, head, emd, n):
    model, p = build(arch, head, emd, max_batch=max(8, n))
    x = scaled(crops_u8(n))
    got = model.predict_on_batch(x)
    want = nets.embed(x, p, arch, emd, head)
    assert isinstance(got, np.ndarray) and got.dtype == np.float32 and got.shape == want.shape
    assert np.all(np.isfinite(got))
    gap = cosine_gap(got, want)
    assert gap.max() < TOL, gap
    scale = np.abs(want).max()
    np.testing.assert_allclose(got, want, atol=2e-4 * scale, rtol=2e-3)
    if head != 'v3':
        # every pairwise cosine distance between embeddings agrees within 1e-5
        for i in range(n):
            a = od.distance(np.repeat(got[i][None], n, 0), got, 1)
            b = od.distance(np.repeat(want[i][None], n, 0), want, 1)
            mask = np.arange(n) != i
            np.testing.assert_allclose(a[mask], b[mask], atol=TOL)
    model.close()


def test_config1_embed_then_match(cuda):
    """BASELINE config 1 (ResNet-50 512-d, batch 8, 1k gallery): identities found through
    the HIP embed + HIP match equal those of the oracle embed + reference-formula match."""
    from deep_insight_face import oneshot
    model, p = build('resnet', 'v2', 512, max_batch=64)
    enrolled = scaled(crops_u8(40, seed=5))
    probes_u8 = crops_u8(40, seed=5)[[3, 17, 0, 39, 21, 8, 30, 11]]
    noise = np.random.default_rng(9).integers(-6, 7, probes_u8.shape)
    probes = scaled(np.clip(probes_u8.astype(np.int64) + noise, 0, 255).astype(np.uint8))
    filler = np.random.default_rng(10).standard_normal((960, 512)).astype(np.float32)
    filler /= np.linalg.norm(filler, axis=1, keepdims=True)

    g_emb = model.predict_on_batch(enrolled)
    q_emb = model.predict_on_batch(probes)
    o_g = nets.embed(enrolled, p, 'resnet', 512, 'v2')
    o_q = nets.embed(probes, p, 'resnet', 512, 'v2')
    gal = oneshot.Gallery(np.concatenate([g_emb, filler]))
    for m in (0, 1):
        idx, dist = gal.match(q_emb, m)
        oi, odist, _ = od.match(o_q, np.concatenate([o_g, filler]), m)
        assert np.array_equal(idx, oi)
        assert np.array_equal(idx, [3, 17, 0, 39, 21, 8, 30, 11])
        if m == 0:
            np.testing.assert_allclose(dist, odist, atol=TOL)
        else:
            np.testing.assert_allclose(np.cos(dist.astype(np.float64) * np.pi),
                                       np.cos(odist.astype(np.float64) * np.pi), atol=TOL)
    gal.close()
    model.close()


def test_input_forms_agree(cuda):
    """NHWC float (the reference's form), NCHW float, uint8 + fused 1/255 scaling, torch
    CUDA tensors and batches larger than max_batch all give the same embeddings."""
    model, _ = build('resnet', 'v2', 512, max_batch=4)
    u8 = crops_u8(6)
    x = scaled(u8)
    base = model.predict_on_batch(x)
    nchw = model.predict_on_batch(np.ascontiguousarray(x.transpose(0, 3, 1, 2)))
    assert np.array_equal(base, nchw)
    t = model.predict_on_batch(torch.from_numpy(x).cuda())
    assert torch.is_tensor(t) and t.is_cuda and np.array_equal(t.cpu().numpy(), base)
    model.set_input_transform(scale=1 / 255.)
    fused = model.predict_on_batch(u8)
    fused_nchw = model.predict_on_batch(np.ascontiguousarray(u8.transpose(0, 3, 1, 2)))
    model.set_input_transform()
    assert cosine_gap(fused, base).max() < 1e-6
    assert np.array_equal(fused, fused_nchw)
    # a row's embedding does not depend on its batch beyond float32 rounding (the stream-K
    # split points of a convolution move with the batch size, which reorders a few sums) ...
    one = model.predict_on_batch(x[2:3])
    assert cosine_gap(one, base[2:3]).max() < 1e-6
    np.testing.assert_allclose(one[0], base[2], atol=2e-6)
    # ... and the same batch gives bit-identical results run to run
    assert np.array_equal(model.predict_on_batch(x), base)
    with pytest.raises(ValueError):
        model.predict_on_batch(np.zeros((2, 96, 96, 3), dtype=np.float32))
    model.close()


def test_bgr_mean_transform(cuda):
    """The siamese path's keras vgg16 preprocess_input (predictions.py:95): BGR swap and
    mean subtraction, fused into the input kernel."""
    model, p = build('resnet', 'v2', 512, max_batch=4)
    x = scaled(crops_u8(2, seed=3))
    mean = np.array([103.939, 116.779, 123.68], dtype=np.float32)
    want = nets.embed(x[..., ::-1] - mean, p, 'resnet', 512, 'v2')
    model.set_input_transform(scale=1.0, bias=tuple(-mean), bgr=True)
    got = model.predict_on_batch(x)
    assert cosine_gap(got, want).max() < TOL
    model.close()


def test_flipped_concat(cuda):
    """use_flipped_images (scripts/insight_face.py:117-118): embeddings of the image and of its
    mirror, concatenated; the mirror is fused into the input kernel for every input form."""
    import torch
    model, p = build('resnet', 'v2', 512, max_batch=4)
    u8 = crops_u8(3, seed=8)
    x = scaled(u8)
    want = np.concatenate([nets.embed(x, p, 'resnet', 512, 'v2'),
                           nets.embed(x[:, :, ::-1, :].copy(), p, 'resnet', 512, 'v2')], axis=1)
    got = model.embed_flipped_concat(x).cpu().numpy()
    assert got.shape == (3, 1024)
    assert cosine_gap(got[:, :512], want[:, :512]).max() < TOL
    assert cosine_gap(got[:, 512:], want[:, 512:]).max() < TOL
    # bitwise: fused mirror == mirroring the array first; NCHW uint8 input too; transform restored
    assert np.array_equal(got[:, 512:], model.predict_on_batch(x[:, :, ::-1, :].copy()))
    model.set_input_transform(scale=1 / 255.)
    nchw = torch.from_numpy(u8).permute(0, 3, 1, 2).contiguous()
    got2 = model.embed_flipped_concat(nchw).cpu().numpy()
    assert np.array_equal(got2[:, 512:], model.predict_on_batch(u8[:, :, ::-1, :].copy()))
    assert np.array_equal(got2[:, :512], model.predict_on_batch(u8))
    model.close()


def test_weights_roundtrip(cuda, tmp_path):
    model, p = build('resnet', 'v1', 128, max_batch=2)
    x = scaled(crops_u8(2))
    a = model.predict_on_batch(x)
    path = str(tmp_path / 'w.npz')
    model.save_weights(path)
    from deep_insight_face.networks.triplet import DifEmbedder
    other = DifEmbedder('resnet', 'v1', 128, (112, 112, 3), max_batch=2)
    with pytest.raises(Exception):
        other.predict_on_batch(x)           # weights never set
    other.load_weights(path)
    assert np.array_equal(other.predict_on_batch(x), a)
    with pytest.raises(ValueError):
        other.set_weights({k: v for k, v in p.items() if 'post_bn' not in k})
    # the reference's own container: model.save_weights('x.h5') / load_weights('x.h5') (inceptionv3.py:79-88), written
    # and read without h5py on this image (networks/h5lite.py)
    h5 = str(tmp_path / 'w.h5')
    model.save_weights(h5)
    third = DifEmbedder('resnet', 'v1', 128, (112, 112, 3), max_batch=2)
    third.load_weights(h5)
    assert np.array_equal(third.predict_on_batch(x), a)
    third.close()
    model.close()
    other.close()


def test_arcmargin_vs_oracle(cuda):
    from deep_insight_face.networks.arcmargin import ArcMarginHead
    rng = np.random.default_rng(0)
    for (B, C) in ((5, 100), (130, 1000), (64, 4097)):
        e = rng.standard_normal((B, 512)).astype(np.float32)
        w = rng.standard_normal((C, 512)).astype(np.float32)
        lab = rng.integers(0, C, (B,))
        head = ArcMarginHead(w)
        got = head.logits(e)
        want = nets.arcmargin_logits(e, w)
        np.testing.assert_allclose(got, want, atol=64 * TOL)
        gotm = head.logits(e, lab)
        wantm = nets.arcmargin_logits(e, w, lab)
        np.testing.assert_allclose(gotm, wantm, atol=64 * 2e-5)
        assert np.array_equal(np.argmax(got, 1), np.argmax(want, 1))
        head.close()


def test_arcmargin_full_size_properties(cuda):
    """BASELINE configs[2] shape (512 embeddings x 85 742 classes) through size-independent
    properties: without labels the logits are s * cosine (|logit| <= s, scale invariance in both
    operands, the row of a class centre fed back as an embedding peaks at s on its own class);
    with labels only the label column changes, to s*cos(theta + m) or the easy-margin guard; and a
    sample of columns equals the oracle."""
    import torch
    from deep_insight_face.networks.arcmargin import ArcMarginHead
    B, C = 512, 85_742
    g = torch.Generator().manual_seed(5)
    w = torch.randn((C, 512), generator=g)
    e = torch.randn((B, 512), generator=g)
    e[:16] = w[1000:1016] * 3.0                          # embeddings that ARE class centres (scaled)
    lab = torch.randint(0, C, (B,), generator=g)
    lab[:16] = torch.arange(1000, 1016)
    head = ArcMarginHead(w.cuda())
    plain = head.logits(e.cuda())
    assert plain.shape == (B, C) and bool(torch.isfinite(plain).all())
    assert float(plain.abs().max()) <= 64.0 + 1e-3
    assert torch.equal(plain[:16].argmax(1).cpu(), torch.arange(1000, 1016))
    assert float((plain[:16].max(1).values - 64.0).abs().max()) < 1e-3
    scaled_e = head.logits((e * 7.5).cuda())
    assert float((scaled_e - plain).abs().max()) < 2e-3   # cosine does not see the embedding's norm
    with_m = head.logits(e.cuda(), lab.cuda())
    diff = (with_m - plain).ne(0)
    assert int(diff.sum()) <= B and bool(diff[torch.arange(B), lab].sum() >= B - 2)
    others = diff.clone()
    others[torch.arange(B), lab] = False
    assert not bool(others.any())                         # only the label column moves
    cos = (plain[torch.arange(B), lab] / 64.0).double().clamp(-1, 1).cpu().numpy()
    th = np.cos(np.pi - 0.5)
    want = np.where(cos > th, np.cos(np.arccos(cos) + 0.5), cos - np.sin(np.pi - 0.5) * 0.5) * 64.0
    np.testing.assert_allclose(with_m[torch.arange(B), lab].cpu().numpy(), want, atol=5e-3)
    cols = np.random.default_rng(1).choice(C, 256, replace=False)
    ref = nets.arcmargin_logits(e.numpy(), w.numpy()[cols])
    np.testing.assert_allclose(plain[:, torch.from_numpy(cols).cuda()].cpu().numpy(), ref, atol=64 * TOL)
    head.close()


def test_pipelined_kernel_equals_plain_kernel(cuda, monkeypatch):
    """Layers with a short K loop and several tiles per resident block run on the software-pipelined
    kernel (conv_pipe_kernel: persistent blocks, the previous tile's epilogue retired inside the next
    tile's K-steps, stores straight from the MFMA layout).  Same arithmetic per output element as
    conv_igemm_kernel, so the two must agree to float32 rounding of the activation's zero sign:
    ResNet50V2 at batch 192 (pre-activation, shortcuts, ragged last tile), IResNet-50 (PReLU, 3x3
    layers with 64 input channels) and YOLOv3-face (LeakyReLU slopes, shortcut adds, 18-channel heads,
    concat views) are run both ways in one process."""
    import torch
    from deep_insight_face.networks.triplet import DifEmbedder
    rng = np.random.default_rng(21)
    cases = [('resnet', 'v2', 512, (112, 112, 3), 192), ('iresnet50', 'v2', 512, (112, 112, 3), 96),
             ('yolov3', 'v3', 1, (416, 416, 3), 2)]
    for arch, head, emd, shape, n in cases:
        x = torch.from_numpy(rng.integers(0, 256, (n,) + shape, dtype=np.uint8)).cuda()
        m = DifEmbedder(arch, head, emd, shape, max_batch=n).init_synthetic(5)
        m.set_input_transform(scale=1 / 255.)
        m.set_option('sk2', 0)                 # compare the two families themselves (round 5's split-K path would take the small layers)
        m.set_option('pipe', 1)
        a = m.embed(x)
        a2 = m.embed(x)
        m.set_option('pipe', 0)
        b = m.embed(x)
        m.set_option('pipe', 1)
        a, a2, b = [t if isinstance(t, list) else [t] for t in (a, a2, b)]
        for ta, ta2, tb in zip(a, a2, b):
            assert torch.equal(ta, ta2)                                   # deterministic
            scale = float(tb.abs().max())
            assert float((ta - tb).abs().max()) <= 2e-6 * max(scale, 1.0), arch
        m.close()


def test_deferred_epilogue_kernel_equals_plain_kernel(cuda, monkeypatch):
    """3x3 / stride 1 layers with several tiles per resident block run on conv_bdp_kernel: B fragments straight from L2,
    a finished tile's accumulators parked in LDS and retired two registers per K-step inside the next tile's K loop,
    always on the persistent stream-K grid.  Option 'bdp' = 2 forces it wherever its restrictions allow (also onto tiny
    grids, tiles split between blocks, 8x8-tile and linear patches, ragged last tiles), 'bdp' = 0 keeps
    conv_igemm_kernel: same products, same order per output element, so the embeddings agree to float32 rounding of the
    ReLU's zero sign.  Run once more with the stream-K owner forced to recompute its partners' ranges
    (DIF_SK_SPIN_LIMIT=-1: the rare branch of the hand-over)."""
    import torch
    from deep_insight_face.networks.triplet import DifEmbedder
    rng = np.random.default_rng(44)
    for arch, n, spin in (('iresnet50', 37, None), ('resnet', 70, None), ('vgg16', 5, None), ('iresnet50', 9, '-1')):
        if spin is not None:
            monkeypatch.setenv('DIF_SK_SPIN_LIMIT', spin)
        x = torch.from_numpy(rng.integers(0, 256, (n, 112, 112, 3), dtype=np.uint8)).cuda()
        m = DifEmbedder(arch, 'v2', 512, (112, 112, 3), max_batch=n).init_synthetic(11)
        m.set_input_transform(scale=1 / 255.)
        m.set_option('sk2', 0)                 # (the small-batch split-K path would take these layers at these batches)
        m.set_option('bdp', 2)
        a = m.embed(x)
        a2 = m.embed(x)
        m.set_option('bdp', 0)
        b = m.embed(x)
        assert torch.equal(a, a2)                                         # deterministic
        assert float((a - b).abs().max()) <= 2e-6, arch
        m.close()
        if spin is not None:
            monkeypatch.delenv('DIF_SK_SPIN_LIMIT')


def test_lean_epilogue_equals_general_epilogue(cuda):
    """conv_igemm_kernel finishes a tile through conv_epilogue_fast when the output is plain (whole tensor, Cout % 4 == 0,
    unit-stride shortcut, 32-bit offsets) and through the general conv_epilogue otherwise; option 'dbg' bit 1024 keeps every
    layer on the general one.  Same operations per element in the same order: bit-identical embeddings, on networks that
    cover ReLU / PReLU / ReLU6 / no activation, shortcuts prefetched in the mainloop's tail and fetched in the epilogue,
    8x8-tile and linear patches, stream-K partial tiles.  (The split-bf16 kernel has the lean epilogue only: its dispatch
    admits nothing else.)"""
    import torch
    from deep_insight_face.networks.triplet import DifEmbedder
    rng = np.random.default_rng(45)
    for arch, n, compute in (('iresnet50', 37, 'f32'), ('resnet', 70, 'f32'), ('mobilenet', 20, 'f32'), ('vgg16', 5, 'f32')):
        x = torch.from_numpy(rng.integers(0, 256, (n, 112, 112, 3), dtype=np.uint8)).cuda()
        m = DifEmbedder(arch, 'v2', 512, (112, 112, 3), max_batch=n, compute=compute).init_synthetic(12)
        m.set_input_transform(scale=1 / 255.)
        m.set_option('sk2', 0)
        m.set_option('bdp', 0)                 # keep the 3x3 layers on conv_igemm_kernel, whose epilogue this is
        a = m.embed(x)
        m.set_option('dbg', 1024)
        b = m.embed(x)
        m.set_option('dbg', 0)
        assert torch.equal(a, b), (arch, compute, float((a - b).abs().max()))
        m.close()


def test_two_subtile_kernel_equals_plain_kernel(cuda):
    """The short-K 3x3 layers on maps whose sides are multiples of 8 (IResNet's 64-channel 112 x 112 and 56 x 56 layers, VGG16's
    first stages, the detector's) run on conv_t2_kernel (option 't2' = 0 keeps them on the 64 x 64 kernel): a 128-pixel x 64-channel tile of two 8x8 sub-tiles per block, two
    row fragments per wave, one whole tile per block.  Same products in the same order per output element as the 64 x 64
    kernel: the embeddings agree bit for bit up to the sign of an activation's zero.  Odd batches leave an odd number of
    8x8 tiles (a block whose second sub-tile does not exist); IResNet's last 64-channel layer writes its first output at
    even pixels only (y_sub)."""
    import torch
    from deep_insight_face.networks.triplet import DifEmbedder
    rng = np.random.default_rng(47)
    for arch, n in (('iresnet50', 9), ('iresnet50', 64), ('iresnet100', 3), ('vgg16', 5)):
        x = torch.from_numpy(rng.integers(0, 256, (n, 112, 112, 3), dtype=np.uint8)).cuda()
        m = DifEmbedder(arch, 'v2', 512, (112, 112, 3), max_batch=n).init_synthetic(14)
        m.set_input_transform(scale=1 / 255.)
        m.set_option('sk2', 0)                 # compare the two families themselves (round 5's split-K path would take the small layers)
        a = m.embed(x)
        a2 = m.embed(x)
        kernels = {k for _, k, _ in m.op_table()}
        m.set_option('t2', 0)
        b = m.embed(x)
        assert not any(k.startswith('conv_t2_kernel') for _, k, _ in m.op_table())
        assert torch.equal(a, a2)
        assert any(k.startswith('conv_t2_kernel') for k in kernels), kernels
        assert float((a - b).abs().max()) <= 2e-6, (arch, n, float((a - b).abs().max()))
        m.close()


def test_wide_tile_kernel_equals_deferred_epilogue_kernel(cuda):
    """The linear-patch 3x3 / stride 1 layers with whole 128-channel column blocks and several tiles per resident block run on
    conv_tn_kernel (64 pixels x 128 channels per block, a wave = 32 pixels x 64 channels, one whole tile per block); option
    'tn' = 0 sends them back to conv_bdp_kernel (persistent stream-K, deferred epilogue).  Same products; conv_bdp_kernel
    splits a tile's K range between blocks and adds the partial sums, so the two agree to float32 rounding through the
    network's ~50 / ~100 layers, not bit for bit (conv_tn_kernel itself adds in conv_igemm_kernel's order).  IResNet-50 at batch 192
    (the default dispatch, two lanes), and with 'dbg' bit 512 -- the kernel also where conv_bdp_kernel would not run --
    IResNet-100 / -50 at batches that leave ragged last tiles, tiles spanning two images and few tiles; ResNet-50V2 and VGG16 for the shortcut / no-shortcut / ReLU epilogues; the
    sub-sampled first output of a stage's last block."""
    import torch
    from deep_insight_face.networks.triplet import DifEmbedder
    rng = np.random.default_rng(48)
    for arch, n, dbg in (('iresnet50', 192, 0), ('iresnet100', 37, 512), ('iresnet50', 9, 512), ('resnet', 70, 512), ('vgg16', 5, 512)):
        x = torch.from_numpy(rng.integers(0, 256, (n, 112, 112, 3), dtype=np.uint8)).cuda()
        m = DifEmbedder(arch, 'v2', 512, (112, 112, 3), max_batch=n).init_synthetic(15)
        m.set_input_transform(scale=1 / 255.)
        m.set_option('sk2', 0)                 # compare the two families themselves (round 5's split-K path would take the small layers)
        m.set_option('dbg', dbg)
        a = m.embed(x)
        a2 = m.embed(x)
        kernels = {k for _, k, _ in m.op_table()}
        m.set_option('tn', 0)
        m.set_option('dbg', 0)
        b = m.embed(x)
        assert not any(k.startswith('conv_tn_kernel') for _, k, _ in m.op_table())
        assert torch.equal(a, a2)
        assert any(k.startswith('conv_tn_kernel') for k in kernels), (arch, kernels)
        assert float((a - b).abs().max()) <= 5e-6, (arch, n, float((a - b).abs().max()))
        m.set_option('bdp', 0)                     # ... and against conv_igemm_kernel's patch form (stream-K as well at these sizes)
        c = m.embed(x)
        assert float((a - c).abs().max()) <= 5e-6, (arch, n, float((a - c).abs().max()))
        m.close()


def test_stem_kernels_equal_general_kernel(cuda):
    """3-channel first layers run on their own kernels: IResNet's 3x3 and ResNet50V2's 7x7 / stride 2 (64 filters) on
    the MFMA with the input patch in LDS and the true K (stem.hip), YOLOv3-face's 3x3 (32 filters) as a direct
    convolution.  Option 'stem' = 0 sends them through the general implicit-GEMM kernel instead: same products, another
    summation order, so the outputs agree to float32 rounding.  ResNet50V2's 56x56 output leaves ragged 16x16 tiles on
    the bottom and right edges (masked pixels); the batches are odd so the last persistent round is partial."""
    import torch
    from deep_insight_face.networks.triplet import DifEmbedder
    rng = np.random.default_rng(33)
    cases = [('resnet', 'v2', 512, (112, 112, 3), 37), ('iresnet50', 'v2', 512, (112, 112, 3), 19),
             ('yolov3', 'v3', 1, (416, 416, 3), 2)]
    for arch, head, emd, shape, n in cases:
        x = torch.from_numpy(rng.integers(0, 256, (n,) + shape, dtype=np.uint8)).cuda()
        m = DifEmbedder(arch, head, emd, shape, max_batch=n).init_synthetic(7)
        m.set_input_transform(scale=1 / 255.)
        a = m.embed(x)
        names = [k for _, k, _ in m.op_table()]
        assert any(k.startswith('stem') for k in names), names
        a2 = m.embed(x)
        m.set_option('stem', 0)
        assert not any(k.startswith('stem') for _, k, _ in m.op_table())
        b = m.embed(x)
        m.set_option('stem', 1)
        a, a2, b = [t if isinstance(t, list) else [t] for t in (a, a2, b)]
        for ta, ta2, tb in zip(a, a2, b):
            assert torch.equal(ta, ta2)
            scale = float(tb.abs().max())
            assert float((ta - tb).abs().max()) <= 2e-5 * max(scale, 1.0), arch                # YOLOv3: 75 layers deep
        m.close()


def test_streamk_fallback_branch(cuda, monkeypatch):
    """The stream-K owner normally adds its partners' partial slabs; if a partner is not
    co-resident it recomputes the missing K range itself.  That branch is rare and
    scheduling-dependent, so force it (DIF_SK_SPIN_LIMIT=-1) and check parity again
    (cdna_hip_programming.md rule 26: a rare branch needs its own test)."""
    monkeypatch.setenv('DIF_SK_SPIN_LIMIT', '-1')
    monkeypatch.setenv('DIF_OPTIONS', 'sk2=0')       # batch 3 would otherwise run on the split-K path: no stream-K at all
    model, p = build('iresnet50', 'v2', 512, max_batch=3)
    x = scaled(crops_u8(3, seed=77))
    got = model.predict_on_batch(x)
    monkeypatch.delenv('DIF_SK_SPIN_LIMIT')
    monkeypatch.delenv('DIF_OPTIONS')
    want = nets.embed(x, p, 'iresnet50', 512, 'v2')
    assert cosine_gap(got, want).max() < TOL
    ref_model, _ = build('iresnet50', 'v2', 512, max_batch=3)
    normal = ref_model.predict_on_batch(x)
    assert cosine_gap(got, normal).max() < 1e-6
    model.close()
    ref_model.close()


def test_embed_full_batch_properties(cuda):
    """BASELINE config-2 batch (256 crops, ResNet50V2+GDC 512-d) without the oracle: outputs are
    unit-norm and finite, rows equal the same crops embedded in a small batch (up to float32
    rounding), a permuted batch gives permuted embeddings, and 8 spot rows match the oracle."""
    model, p = build('resnet', 'v2', 512, max_batch=256)
    u8 = crops_u8(256, seed=42)
    model.set_input_transform(scale=1 / 255.)
    full = model.predict_on_batch(u8)
    assert full.shape == (256, 512) and np.all(np.isfinite(full))
    np.testing.assert_allclose(np.linalg.norm(full, axis=1), 1.0, atol=1e-5)
    small = model.predict_on_batch(u8[100:108])
    assert cosine_gap(small, full[100:108]).max() < 1e-6
    perm = np.random.default_rng(0).permutation(256)
    permuted = model.predict_on_batch(u8[perm])
    assert cosine_gap(permuted, full[perm]).max() < 1e-6
    model.set_input_transform()
    rows = [0, 31, 64, 127, 128, 200, 254, 255]
    want = nets.embed(scaled(u8[rows]), p, 'resnet', 512, 'v2')
    assert cosine_gap(full[rows], want).max() < TOL
    model.close()


@pytest.mark.parametrize('arch,head,emd,n', [('iresnet50', 'v2', 512, 96), ('iresnet50', 'v2', 512, 37), ('iresnet50', 'v2', 512, 130),
                                             ('iresnet100', 'v2', 512, 64),
                                             ('resnet', 'v2', 512, 512), ('resnet', 'v1', 128, 512),
                                             ('vgg16', 'v2', 512, 16)])
@pytest.mark.parametrize('compute', ['bf16x3', 'bf16x2'])
def test_embed_bf16x3_mode_vs_oracle(cuda, arch, head, emd, n, compute):
    """The split-bf16 throughput modes -- "bf16x3": three bf16 terms per f32 operand, six MFMA products; "bf16x2" (round 4):
    two terms, three products -- f32 accumulation, against the float32 path on the whole batch and against the oracle on
    spot rows, at the same gates: cosine gap < 1e-5 (oracle) / 1e-6 (f32 path), pairwise cosine distances within 1e-5.
    The batches are large enough for the mode to engage (conv.hip: bf3_pays -- short or small layers stay on
    the f32 kernels), including the halo-patch 3x3 path; `launches` below checks that it did."""
    from deep_insight_face.networks.triplet import DifEmbedder
    u8 = crops_u8(n, seed=31)
    f32 = DifEmbedder(arch, head, emd, (112, 112, 3), max_batch=n).init_synthetic(2024)
    b3 = DifEmbedder(arch, head, emd, (112, 112, 3), max_batch=n, compute=compute)
    p = f32.get_weights()
    b3.set_weights(p)
    for m in (f32, b3):
        m.set_input_transform(scale=1 / 255.)
    got = b3.predict_on_batch(u8)
    ref = f32.predict_on_batch(u8)
    assert np.all(np.isfinite(got))
    assert cosine_gap(got, ref).max() < 1e-6
    rows = [0, 1, n // 2, n - 1]
    want = nets.embed(scaled(u8[rows]), p, arch, emd, head)
    assert cosine_gap(got[rows], want).max() < TOL
    np.testing.assert_allclose(got[rows], want, atol=2e-4 * np.abs(want).max(), rtol=2e-3)
    for i in range(len(rows)):
        a = od.distance(np.repeat(got[rows][i][None], len(rows), 0), got[rows], 1)
        b = od.distance(np.repeat(want[i][None], len(rows), 0), want, 1)
        mask = np.arange(len(rows)) != i
        np.testing.assert_allclose(a[mask], b[mask], atol=TOL)
    assert np.array_equal(b3.predict_on_batch(u8), got)           # deterministic
    assert not np.array_equal(got, ref)                            # ... and really another arithmetic
    f32.close()
    b3.close()


def test_mfma_clock_probe(cuda):
    """dif_probe_mfma_clock (measurement aid behind bench.py's `held_mfma_clock_ghz`): the clock held under back-to-back
    f32 MFMAs lies between the 1.0 GHz floor seen under power caps and the 2.4 GHz peak, and the loop's own rate is
    that clock's share of the 157.3 TFLOP/s peak (it issues nothing but MFMAs: within 10 %)."""
    import ctypes
    import torch
    from deep_insight_face import _native as N
    ghz, tf = ctypes.c_double(0.0), ctypes.c_double(0.0)
    N.check(N.lib.dif_probe_mfma_clock(ctypes.byref(ghz), ctypes.byref(tf), torch.cuda.current_stream().cuda_stream))
    assert 1.0 < ghz.value <= 2.45, ghz.value
    assert 0.9 < tf.value / (157.3 * ghz.value / 2.4) < 1.05, (tf.value, ghz.value)
    # ... and the clock inside the convolution kernels of a forward (every block records cycles and 100 MHz ticks)
    model, _ = build('iresnet50', 'v2', 512, max_batch=64)
    x = torch.from_numpy(crops_u8(64, seed=3)).cuda()
    ref = model.embed(x)
    conv_ghz = model.held_clock_ghz(x)
    assert 1.0 < conv_ghz <= 2.45, conv_ghz
    assert torch.equal(model.embed(x), ref)                    # the measuring forward leaves no state behind
    model.close()


@pytest.mark.parametrize('arch,head,emd,shape,n', [('resnet', 'v2', 512, (112, 112, 3), 1), ('resnet', 'v2', 512, (112, 112, 3), 12),
                                                   ('resnet', 'v1', 128, (112, 112, 3), 3), ('iresnet50', 'v2', 512, (112, 112, 3), 1),
                                                   ('iresnet100', 'v2', 512, (112, 112, 3), 8), ('iresnet50', 'v2', 512, (112, 112, 3), 33),
                                                   ('iresnet50', 'v2', 512, (112, 112, 3), 16), ('iresnet50', 'v2', 512, (112, 112, 3), 20),
                                                   ('mobilenet', 'v2', 512, (112, 112, 3), 5), ('vgg16', 'v2', 512, (112, 112, 3), 2),
                                                   ('yolov3', 'v3', 1, (416, 416, 3), 1)])
def test_splitk_path_equals_streamk_path(cuda, arch, head, emd, shape, n):
    """Round 5: at small batches the few-tile / long-K layers run as conv_sk_kernel (split-K partials, no hand-over inside
    the launch) + conv_sk_reduce_kernel (fixed-order sum + the layer's epilogue from the MFMA layout); option 'sk2' = 0
    keeps them on round 4's kernels.  Same products, another (fixed) summation order: embeddings agree to float32 rounding,
    run to run bit-identical.  Networks cover pre-activation (ResNet50V2), PReLU + two outputs + sub-sampled first outputs +
    strided shortcuts (IResNet), ReLU6 / depthwise neighbours (MobileNetV2), the flattening fc, concat views and 18-channel
    heads (YOLOv3-face); batches of 16 and 20 are sk2_plan's second look (392 tiles split three ways: a grid of more blocks
    than the CUs hold at once; 248 lone tiles split in two)."""
    import torch
    from deep_insight_face.networks.triplet import DifEmbedder
    rng = np.random.default_rng(n * 7 + len(arch))
    x = torch.from_numpy(rng.integers(0, 256, (n,) + shape, dtype=np.uint8)).cuda()
    m = DifEmbedder(arch, head, emd, shape, max_batch=n).init_synthetic(13)
    m.set_input_transform(scale=1 / 255.)
    m.set_option('mt', 0)                      # (the one-image kernel would take most layers at n = 1: its own test below)
    a = m.embed(x)
    a2 = m.embed(x)
    kernels = [k for _, k, _, _ in m.profile(x)]
    assert any(k.startswith('conv_sk') for k in kernels), kernels
    m.set_option('dbg', 16384)                 # the gather form of the split-K block where the patch form ran
    g = m.embed(x)
    m.set_option('dbg', 0)
    m.set_option('sk2', 0)
    b = m.embed(x)
    assert not any(k.startswith('conv_sk') for _, k, _, _ in m.profile(x))
    g = g if isinstance(g, list) else [g]
    for tg, tb in zip(g, b if isinstance(b, list) else [b]):
        assert float((tg - tb).abs().max()) <= (2e-5 if arch == 'yolov3' else 4e-6) * max(float(tb.abs().max()), 1.0), arch
    a, a2, b = [t if isinstance(t, list) else [t] for t in (a, a2, b)]
    for ta, ta2, tb in zip(a, a2, b):
        assert torch.equal(ta, ta2)                                       # deterministic
        scale = max(float(tb.abs().max()), 1.0)
        assert float((ta - tb).abs().max()) <= (2e-5 if arch == 'yolov3' else 4e-6) * scale, arch
    m.close()


@pytest.mark.parametrize('arch,head,emd,shape,n', [('resnet', 'v2', 512, (112, 112, 3), 1), ('resnet', 'v1', 128, (112, 112, 3), 2),
                                                   ('iresnet50', 'v2', 512, (112, 112, 3), 1), ('iresnet100', 'v2', 512, (112, 112, 3), 1),
                                                   ('iresnet50', 'v2', 512, (112, 112, 3), 3), ('mobilenet', 'v2', 512, (112, 112, 3), 1),
                                                   ('vgg16', 'v2', 512, (112, 112, 3), 1), ('resnet', 'v2', 512, (96, 96, 3), 1),
                                                   ('yolov3', 'v3', 1, (416, 416, 3), 1)])
def test_one_image_kernel_equals_other_paths(cuda, arch, head, emd, shape, n):
    """Round 5: at ONE image per call (predictions.py:152-156, the reference's own call shape) a layer runs in one launch on
    16 x 16 tiles (conv_mt_kernel: operands straight from L2 into v_mfma_f32_16x16x4_f32, K split over the block's waves,
    their sums added in wave order); option 'mt' = 0 sends the same layers to the split-K pair / the large-batch kernels.
    Same products, another fixed summation order: embeddings agree to float32 rounding and are bit-identical run to run.
    Covers pre-activation on both loaders (ResNet50V2), PReLU + two outputs + sub-sampled first outputs + strided
    shortcuts + stride-2 3x3 layers (IResNet), 1x1 expansions and ReLU6 (MobileNetV2), ragged 16-pixel tiles (a 96 x 96
    input: 6 x 6, 3 x 3 maps), a batch of 2-3 where only the small layers qualify, concat views / 18-channel heads (YOLOv3)."""
    import torch
    from deep_insight_face.networks.triplet import DifEmbedder
    rng = np.random.default_rng(n * 11 + len(arch))
    x = torch.from_numpy(rng.integers(0, 256, (n,) + shape, dtype=np.uint8)).cuda()
    m = DifEmbedder(arch, head, emd, shape, max_batch=n).init_synthetic(17)
    m.set_input_transform(scale=1 / 255.)
    a = m.embed(x)
    a2 = m.embed(x)
    kernels = [k for _, k, _, _ in m.profile(x)]
    assert any(k.startswith('conv_mt_kernel') for k in kernels), kernels
    m.set_option('mt', 0)
    b = m.embed(x)
    assert not any(k.startswith('conv_mt_kernel') for _, k, _, _ in m.profile(x))
    a, a2, b = [t if isinstance(t, list) else [t] for t in (a, a2, b)]
    for ta, ta2, tb in zip(a, a2, b):
        assert torch.equal(ta, ta2)                                       # deterministic
        scale = max(float(tb.abs().max()), 1.0)
        assert float((ta - tb).abs().max()) <= (2e-5 if arch == 'yolov3' else 4e-6) * scale, arch
    m.close()


def test_small_batches_on_a_large_max_batch_model(cuda):
    """bench.py's `latency` block (and any serving process) embeds 1 / 8 / 32 images on a model finalized for hundreds: the
    small-batch kernels are chosen from the batch at hand, not from max_batch, and the result does not depend on max_batch --
    bit-identical to a model finalized for exactly that batch, and within 1e-5 of the oracle."""
    u8 = crops_u8(12, seed=9)
    big, p = build('iresnet50', 'v2', 512, max_batch=192)
    big.set_input_transform(scale=1 / 255.)
    want = nets.embed(scaled(u8), p, 'iresnet50', 512, 'v2')
    for n in (1, 8, 12):
        small, _ = build('iresnet50', 'v2', 512, max_batch=n)
        small.set_input_transform(scale=1 / 255.)
        a = big.predict_on_batch(u8[:n])
        b = small.predict_on_batch(u8[:n])
        assert np.array_equal(a, b), n
        assert cosine_gap(a, want[:n]).max() < TOL
        kinds = {k.split('<')[0] for _, k, _, _ in big.profile(torch.from_numpy(u8[:n]).cuda())}
        assert ('conv_mt_kernel' in kinds) if n == 1 else (kinds & {'conv_sk_kernel', 'conv_skp_kernel'}), (n, kinds)
        small.close()
    full = big.predict_on_batch(crops_u8(192, seed=10))                    # the large-batch kernels still run on it afterwards
    assert np.all(np.isfinite(full)) and full.shape == (192, 512)
    big.close()


def test_gdc_tail_one_and_two_images(cuda):
    """The GDC tail (networks/triplet.py:129-138: depthwise over the whole map -> BN -> 1x1 conv -> Dense -> l2_normalize) at
    one and two images runs as two launches of emd / 32 blocks (gdc_tail_a_kernel / gdc_tail_b_kernel: the second's last
    block normalises); three and more on the one-launch kernel.  Same products, another fixed summation order: within 1e-5
    of the oracle either way, rows equal to float32 rounding across the two forms, and repeated calls bit-identical (the
    ticket the blocks draw is back at zero after every call)."""
    model, p = build('resnet', 'v2', 512, max_batch=4)
    x = scaled(crops_u8(3, seed=21))
    want = nets.embed(x, p, 'resnet', 512, 'v2')
    three = model.predict_on_batch(x)
    assert cosine_gap(three, want).max() < TOL
    for n in (1, 2):
        got = model.predict_on_batch(x[:n])
        kernels = [k for _, k, _, _ in model.profile(torch.from_numpy(x[:n]).cuda())]
        assert 'gdc_tail_a_kernel+gdc_tail_b_kernel' in kernels, kernels
        assert cosine_gap(got, want[:n]).max() < TOL
        np.testing.assert_allclose(got, three[:n], atol=2e-6)
        for _ in range(3):
            assert np.array_equal(model.predict_on_batch(x[:n]), got)
    assert 'gdc_tail_kernel' in [k for _, k, _, _ in model.profile(torch.from_numpy(x).cuda())]
    model.close()
