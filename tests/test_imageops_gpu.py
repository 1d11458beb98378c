"""GPU letterbox / crop+resize kernels vs the oracle (oracle/imageops.py) and the device-resident
frames -> detect -> crop -> embed -> match pipeline (BASELINE configs[4])."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _frames(n, h, w, seed):
    rng = np.random.default_rng(seed)
    # smooth-ish content plus noise so that resampling is exercised on gradients and on edges
    yy, xx = np.mgrid[0:h, 0:w]
    base = (127 + 100 * np.sin(xx / 17.0)[None] * np.cos(yy / 11.0)[None])[..., None]
    noise = rng.integers(-40, 40, size=(n, h, w, 3))
    return np.clip(base + noise, 0, 255).astype(np.uint8)


@pytest.mark.parametrize('h,w,size', [(480, 640, 416), (300, 200, 416), (96, 128, 64), (416, 416, 416)])
def test_letterbox_matches_pil(h, w, size):
    from deep_insight_face.detector.yolov3 import letterbox_batch
    from oracle import imageops
    frames = _frames(3, h, w, 5)
    got = letterbox_batch(frames, size).cpu().numpy()
    want = np.stack([imageops.letterbox(f, size) for f in frames])
    diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
    # PIL works in 22-bit fixed point; float weights can flip a rounding by one grey level
    assert diff.max() <= 1
    assert (diff == 0).mean() > 0.98


def test_crop_resize_matches_oracle():
    from deep_insight_face.detector.run import crop_faces
    from oracle import imageops
    frames = _frames(6, 240, 320, 7)
    boxes = np.array([[40.3, 30.9, 200.2, 180.7],      # shrink
                      [100.0, 100.0, 150.0, 160.0],    # enlarge
                      [-20.0, -10.0, 400.0, 300.0],    # clamped to the frame
                      [10.0, 20.0, 122.0, 132.0],      # 112 + margin: integer scale
                      [50.0, 50.0, 50.0, 90.0],        # zero width after margin 0 -> see below
                      [np.nan] * 4], dtype=np.float32)
    for margin in (8, 0):
        got = crop_faces(frames, boxes, margin, 112).cpu().numpy()
        for i in range(6):
            if np.isnan(boxes[i, 0]):
                assert not got[i].any()
                continue
            want = imageops.crop_resize(frames[i], boxes[i], margin, 112)
            # round 4: oracle and kernel both follow cv2's uint8 INTER_AREA paths operation by operation -> equal
            assert np.array_equal(got[i], want), (i, margin, int(np.abs(got[i].astype(np.int32) - want).max()))


@pytest.mark.parametrize('nframes,h,w', [(4, 240, 320), (96, 480, 640), (256, 480, 640)])
def test_frame_pipeline_end_to_end(nframes, h, w):
    """Device pipeline == the same stages run one by one through the host-visible API; at 4 small frames and
    at 96 and at 256 frames (= BASELINE configs[4]'s batch) of its 640x480 (the detector then runs in chunks of
    its max_batch = 64, the embedder splits its batch over two lanes...)."""
    from deep_insight_face import oneshot
    from deep_insight_face.detector import run as drun, yolov3 as yolo
    from deep_insight_face.networks.triplet import bottleneck_network
    from deep_insight_face.networks.weights import synth_params
    det = drun.yolo_v3_face(max_batch=min(nframes, 64))
    p = synth_params(det.param_spec(), seed=11)
    for k in p:                                   # keep exp() in the box decode finite
        if k in ('conv_58/kernel', 'conv_66/kernel', 'conv_74/kernel'):
            p[k] = p[k] * 1e-5
        if k in ('conv_58/bias', 'conv_66/bias', 'conv_74/bias'):
            p[k] = np.zeros_like(p[k])
            p[k][4::6] = 2.0
            p[k][5::6] = 2.0
    det.set_weights(p)
    det.set_input_transform(scale=1 / 255.)
    emb = bottleneck_network('resnet', emd_size=512, input_shape=(112, 112, 3), max_batch=nframes)('v2')
    emb.init_synthetic(3)
    emb.set_input_transform(scale=1 / 255.)
    rng = np.random.default_rng(0)
    gal = rng.standard_normal((1000, 512)).astype(np.float32)
    pipe = drun.FramePipeline(det, emb, oneshot.Gallery(gal), margin=8, score=0.4)
    frames = _frames(nframes, h, w, 9)
    boxes, scores, e, idx, dist = pipe(frames)
    boxes, scores = boxes.cpu().numpy(), scores.cpu().numpy()
    assert np.isfinite(boxes).all() and (scores >= 0.4).all()
    # stage by stage
    lb = yolo.letterbox_batch(frames, 416)
    maps = det.embed(lb)
    for i in range(0, nframes, max(1, nframes // 8)):
        b, s, c = yolo.get_yolo_output([m[i:i + 1] for m in maps], drun.ANCHORS.reshape(-1, 2), 1, (h, w),
                                       max_boxes=1, score_threshold=0.4)
        top, left, bottom, right = b[0]
        np.testing.assert_allclose(boxes[i], [left, top, right, bottom], rtol=0, atol=0)
        np.testing.assert_allclose(scores[i], s[0], rtol=0, atol=0)
    crops = drun.crop_faces(frames, boxes, 8, 112)
    e2 = emb.embed(crops)
    assert torch.equal(e, e2)
    i2, d2 = oneshot.Gallery(gal).match(e2, 1)
    assert torch.equal(idx, i2) and torch.equal(dist, d2)
