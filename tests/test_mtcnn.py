"""MTCNN (BASELINE configs[4] as worded; not in the reference -- PARITY UNPINNED, oracle/mtcnn.py says why): the NumPy
restatement against an independently written torch-CPU implementation and the library's parameter table (no GPU), then
the HIP networks and the whole static-shape cascade against the restatement on the same seeded frames and weights."""
import numpy as np
import pytest
import torch

from oracle import mtcnn as om
from oracle import torch_nets as torch_ref


def _synth(seed=2024):
    from deep_insight_face.networks.weights import synth_params
    params = {k: synth_params(om.spec(k), seed + i) for i, k in enumerate(('pnet', 'rnet', 'onet'))}
    for k in params:
        params[k]['head/bias'][0], params[k]['head/bias'][1] = -2.0, 2.0
        params[k]['head/kernel'][..., 2:] *= np.float32(0.02)
        params[k]['head/bias'][2:] *= np.float32(0.02)
    params['pnet']['conv1/kernel'][..., 10:] = 0
    params['pnet']['conv1/bias'][10:] = 0
    params['pnet']['head/kernel'][..., 6:] = 0
    params['pnet']['head/bias'][6:] = 0
    return params


def _frames(n, h, w, seed):
    rng = np.random.default_rng(seed)
    # smooth blobs + noise: different cells score differently, so the suppression has something to order
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    out = np.empty((n, h, w, 3), np.uint8)
    for i in range(n):
        img = np.zeros((h, w, 3), np.float32)
        for _ in range(6):
            cy, cx, r = rng.uniform(0, h), rng.uniform(0, w), rng.uniform(8, 40)
            img += rng.uniform(40, 160, 3) * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * r * r))[..., None]
        out[i] = np.clip(img + rng.normal(0, 12, img.shape), 0, 255).astype(np.uint8)
    return out


def test_oracle_networks_against_torch():
    p = _synth()
    rng = np.random.default_rng(1)
    for stage, shape, out in (('pnet', (2, 37, 52, 3), (2, 14, 21, 8)), ('pnet', (1, 12, 12, 3), (1, 1, 1, 8)),
                              ('rnet', (3, 24, 24, 3), (3, 8)), ('onet', (3, 48, 48, 3), (3, 16))):
        x = om.normalise(rng.integers(0, 256, shape, dtype=np.uint8))
        a = getattr(om, stage)(x, p[stage])
        b = torch_ref.mtcnn(x, p[stage], stage)
        assert a.shape == out and b.shape == out, (stage, a.shape, b.shape)
        np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-5 * np.abs(b).max())
    # public layer tables: parameter counts of the three networks (P-Net 10 -> 12 first-layer filters, heads merged + padded)
    n = lambda spec: sum(int(np.prod(s)) for _, s in spec)   # noqa: E731
    assert n(om.spec('rnet')) == 100_178 + 128 * 2 + 2          # published 100 178 + the head's two zero filters
    assert om.pyramid_scales(480, 640) == pytest.approx([0.6 * 0.709 ** i for i in range(10)])


def test_library_parameter_table_matches_oracle():
    from deep_insight_face.networks.triplet import DifEmbedder
    for stage, hw in (('pnet', (60, 80)), ('rnet', (24, 24)), ('onet', (48, 48))):
        m = DifEmbedder('mtcnn_' + stage, 'v3', 1, hw + (3,))
        assert dict(m.param_spec()) == dict(om.spec(stage)), stage
        m.close()
    with pytest.raises(ValueError):
        DifEmbedder('mtcnn_rnet', 'v3', 1, (32, 32, 3))
    with pytest.raises(ValueError):
        DifEmbedder('mtcnn_pnet', 'v3', 1, (8, 40, 3))


@pytest.mark.gpu
@pytest.mark.parametrize('stage,shape', [('pnet', (3, 37, 52, 3)), ('pnet', (2, 288, 384, 3)), ('pnet', (5, 12, 17, 3)),
                                         ('rnet', (70, 24, 24, 3)), ('onet', (33, 48, 48, 3))])
def test_networks_vs_oracle(cuda, stage, shape):
    from deep_insight_face.networks.triplet import DifEmbedder
    p = _synth()[stage]
    u8 = np.random.default_rng(shape[1]).integers(0, 256, shape, dtype=np.uint8)
    m = DifEmbedder('mtcnn_' + stage, 'v3', 1, shape[1:], max_batch=shape[0])
    m.set_weights(p)
    m.set_input_transform(scale=1 / 128., bias=(-127.5 / 128.,) * 3)
    got = m.predict_on_batch(u8)
    want = getattr(om, stage)(om.normalise(u8), p)
    got = got.reshape(want.shape)
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-5 * np.abs(want).max())
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize('hw,n', [((96, 128), 3), ((120, 90), 2)])
def test_cascade_vs_oracle(cuda, hw, n):
    """The whole cascade, stage by stage: same slots (boxes bit for bit, scores to float32 rounding of exp) after P-Net +
    two suppressions + calibration, after R-Net, after O-Net.  A slot whose probability sits within 1e-6 of a threshold,
    or two slots whose scores tie within 1e-6, could legitimately come out differently; the seeds avoid that (asserted)."""
    from deep_insight_face.detector.mtcnn import MtcnnDetector
    p = _synth()
    frames = _frames(n, hw[0], hw[1], seed=hw[0])
    cap = (24, 12, 8)
    det = MtcnnDetector(hw, max_batch=n, cap=cap)
    det.set_weights(p)
    assert det.scales == om.pyramid_scales(*hw)
    b3, s3, st = det.detect(frames, return_stages=True)
    ob, os_, dbg = om.detect(frames, p, cap=cap)
    for name in ('stage1', 'stage2'):
        gb, gs = st[name + '_boxes'].cpu().numpy(), st[name + '_scores'].cpu().numpy()
        wb, ws = np.stack(dbg[name + '_boxes']), np.stack(dbg[name + '_scores'])
        assert (ws >= 0).sum() >= n, name                      # the synthetic weights give every stage work
        assert np.array_equal(gs >= 0, ws >= 0), name
        np.testing.assert_allclose(gs, ws, atol=2e-6, err_msg=name)
        assert np.array_equal(gb, wb), (name, np.abs(gb - wb).max())
    gb, gs = b3.cpu().numpy(), s3.cpu().numpy()
    assert np.array_equal(gs >= 0, os_ >= 0) and (os_ >= 0).sum() >= n
    np.testing.assert_allclose(gs, os_, atol=2e-6)
    np.testing.assert_allclose(gb, ob, rtol=1e-5, atol=1e-3)     # the last regression is not truncated
    det.close()


@pytest.mark.gpu
def test_pyramid_streams_give_the_same_detections(cuda):
    """The pyramid's scales run round-robin on `streams` HIP streams (default 4) and meet before the merge: same boxes and
    scores, bit for bit, as one scale after the other on the caller's stream, call after call."""
    import torch
    from deep_insight_face.detector.mtcnn import MtcnnDetector
    hw, n = (120, 160), 4
    frames = _frames(n, hw[0], hw[1], seed=11)
    one = MtcnnDetector(hw, max_batch=n, cap=(24, 12, 8), streams=1).init_synthetic(7)
    ref_b, ref_s = one.detect(frames)
    for k in (2, 3, 5):
        det = MtcnnDetector(hw, max_batch=n, cap=(24, 12, 8), streams=k)
        det.set_weights(one.get_weights())
        for _ in range(3):
            b, s = det.detect(frames)
            assert torch.equal(b, ref_b) and torch.equal(s, ref_s), k
        det.close()
    one.close()


@pytest.mark.gpu
def test_detection_wrapper_and_frame_pipeline(cuda):
    """detector/run.py:120-173's calling convention (image -> crops, boxes; ValueError when nothing is found), and the
    batched device pipeline frames -> best face -> crop -> embedding -> match, whose crops equal the per-image path's."""
    from deep_insight_face import oneshot
    from deep_insight_face.detector.mtcnn import MtcnnDetection, MtcnnDetector, MtcnnFramePipeline
    from deep_insight_face.networks.triplet import DifEmbedder
    hw, n = (96, 128), 5
    frames = _frames(n, hw[0], hw[1], seed=5)
    det = MtcnnDetector(hw, max_batch=2, cap=(24, 12, 8)).init_synthetic(7)
    crops, boxes = MtcnnDetection(model=det, margin=8)(frames[0])
    assert len(crops) == 1 and crops[0].ndim == 3 and len(boxes[0]) == 4
    many, _ = MtcnnDetection(model=det, margin=8, detect_multiple_faces=True)(frames[0])
    assert len(many) >= 1
    quiet = MtcnnDetector(hw, max_batch=1, cap=(24, 12, 8), thresholds=(0.6, 0.7, 1.1))
    quiet.set_weights(det.get_weights())
    with pytest.raises(ValueError, match='Bounding box not found'):
        MtcnnDetection(model=quiet)(frames[0])
    emb = DifEmbedder('resnet', 'v2', 512, (112, 112, 3), max_batch=8).init_synthetic(3)
    emb.set_input_transform(scale=1 / 255.)
    pipe = MtcnnFramePipeline(det, emb, None, margin=8)
    bx, sc, e = pipe(frames)                                   # five frames through a detector of max_batch 2
    assert bx.shape == (n, 4) and e.shape == (n, 512) and bool((sc > 0).all())
    gal = oneshot.Gallery(e)
    _, _, _, idx, dist = MtcnnFramePipeline(det, emb, gal, margin=8)(frames)
    assert idx.tolist() == list(range(n))
    b0, s0 = det.detect(frames[:1])
    assert torch.equal(b0[0, 0], bx[0]) and float(s0[0, 0]) == float(sc[0])
    for m in (det, quiet, emb, gal):
        m.close()
