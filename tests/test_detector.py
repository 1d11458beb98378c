"""YOLOv3-face post-processing (detector/yolov3.py:36-172): the device decode + suppression
against the oracle's restatement on seeded feature maps."""
import numpy as np
import pytest

from oracle import detector as odet


def feature_maps(n, num_classes=1, seed=0, hot=12):
    """Random 13/26/52 maps with a few confident, overlapping detections planted."""
    rng = np.random.default_rng(seed)
    outs = []
    for g in (13, 26, 52):
        f = rng.standard_normal((n, g, g, 3 * (5 + num_classes))).astype(np.float32)
        f[..., 4::(5 + num_classes)] -= 6.0                      # low objectness everywhere ...
        outs.append(f)
    for i in range(n):
        for _ in range(hot):                                     # ... except a few cells
            l = int(rng.integers(0, 3))
            g = outs[l].shape[1]
            y, x, a = int(rng.integers(0, g)), int(rng.integers(0, g)), int(rng.integers(0, 3))
            base = a * (5 + num_classes)
            outs[l][i, y, x, base + 4] = rng.uniform(2.0, 6.0)
            outs[l][i, y, x, base + 5:base + 5 + num_classes] = rng.uniform(1.0, 5.0, num_classes)
            if x + 1 < g:                                        # a neighbour that overlaps it
                outs[l][i, y, x + 1, base:base + 5 + num_classes] = outs[l][i, y, x, base:base + 5 + num_classes]
                outs[l][i, y, x + 1, base + 4] -= 0.5
    return outs


def test_oracle_nms_basics():
    boxes = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10]], dtype=np.float32)
    scores = np.array([0.9, 0.8, 0.7, 0.9], dtype=np.float32)
    assert list(odet.non_max_suppression(boxes, scores, 10, 0.5)) == [0, 2]        # 3 ties with 0: lower index wins
    assert list(odet.non_max_suppression(boxes, scores, 1, 0.5)) == [0]
    assert list(odet.non_max_suppression(boxes[:, [2, 3, 0, 1]], scores, 10, 0.5)) == [0, 2]   # corner order free


@pytest.mark.gpu
@pytest.mark.parametrize('num_classes,image_shape', [(1, (480, 640)), (1, (416, 416)), (2, (300, 900))])
def test_decode_and_nms_vs_oracle(cuda, num_classes, image_shape):
    from deep_insight_face.detector import yolov3
    outs = feature_maps(1, num_classes, seed=num_classes)
    boxes, scores = yolov3.boxes_and_scores_all(outs, odet.ANCHORS, num_classes, image_shape)
    want_b, want_s = [], []
    for l, mask in enumerate(([6, 7, 8], [3, 4, 5], [0, 1, 2])):
        b, s = odet.boxes_and_scores(outs[l], odet.ANCHORS[mask], num_classes, (416, 416), image_shape)
        want_b.append(b)
        want_s.append(s)
    want_b, want_s = np.concatenate(want_b), np.concatenate(want_s)
    assert boxes.shape == (1, 10647, 4)
    np.testing.assert_allclose(boxes[0], want_b, rtol=2e-5, atol=2e-3)
    np.testing.assert_allclose(scores[0], want_s, rtol=2e-5, atol=1e-6)
    for thr in (0.6, 0.3):
        gb, gs, gc = yolov3.get_yolo_output(outs, odet.ANCHORS, num_classes, image_shape, 20, thr, 0.5)
        ob, os_, oc = odet.get_yolo_output(outs, odet.ANCHORS, num_classes, image_shape, 20, thr, 0.5)
        assert len(gs) == len(os_) and len(gs) > 0
        assert np.array_equal(gc, oc)
        np.testing.assert_allclose(gs, os_, rtol=2e-5)
        np.testing.assert_allclose(gb, ob, rtol=2e-5, atol=2e-3)


@pytest.mark.gpu
def test_batched_detection_and_plain_nms(cuda):
    from deep_insight_face.detector import yolov3
    outs = feature_maps(3, 1, seed=9)
    shapes = [(480, 640), (720, 1280), (416, 416)]
    res = yolov3.get_yolo_output(outs, odet.ANCHORS, 1, shapes, 20, 0.5, 0.45)
    assert len(res) == 3
    for i in range(3):
        one = [o[i:i + 1] for o in outs]
        ob, os_, oc = odet.get_yolo_output(one, odet.ANCHORS, 1, shapes[i], 20, 0.5, 0.45)
        np.testing.assert_allclose(res[i][1], os_, rtol=2e-5)
        np.testing.assert_allclose(res[i][0], ob, rtol=2e-5, atol=2e-3)
    rng = np.random.default_rng(1)
    b = rng.uniform(0, 100, (300, 4)).astype(np.float32)
    s = rng.uniform(0, 1, 300).astype(np.float32)
    s[10] = s[200]                                              # an exact tie
    assert list(yolov3.non_max_suppression(b, s, 50, 0.3)) == list(odet.non_max_suppression(b, s, 50, 0.3))


# ------------------------------------------------------------------------------ the detector network
def _yolo_params(num_classes=1):
    from deep_insight_face.networks.weights import synth_params
    return synth_params(odet.yolov3_spec(num_classes))


def _frames(n, hw, seed=0):
    return np.random.default_rng(seed).integers(0, 256, (n, hw, hw, 3), dtype=np.uint8).astype(np.float32) / np.float32(255)


def test_yolov3_structure_matches_the_cfg():
    """75 convolutions (72 with BN + leaky, 3 linear heads with bias); the cfg itself is pinned layer by
    layer by tests/test_structure.py against tests/golden/structure.json."""
    spec = odet.yolov3_spec(1)
    kernels = [s for n, s in spec if n.endswith('/kernel')]
    assert len(kernels) == 75 and len([n for n, _ in spec if n.endswith('/bias')]) == 3
    assert kernels[-1][-1] == 18 and kernels[0] == (3, 3, 3, 32)
    from deep_insight_face.detector.run import yolo_v3_face
    net = yolo_v3_face(1, 416)
    assert dict(net.param_spec()) == dict(spec)
    assert net.output_shapes == [(13, 13, 18), (26, 26, 18), (52, 52, 18)]
    assert abs(net.flops_per_image / 1e9 - 65.3) < 0.2


def test_yolov3_oracle_vs_torch():
    from oracle import torch_nets as torch_ref
    p = _yolo_params()
    x = _frames(1, 96)
    a = odet.yolov3_forward(x, p)
    b = torch_ref.yolov3(x, p)
    assert [t.shape for t in a] == [(1, 3, 3, 18), (1, 6, 6, 18), (1, 12, 12, 18)]
    for u, v in zip(a, b):
        np.testing.assert_allclose(u, v, rtol=2e-3, atol=2e-4 * np.abs(v).max())


@pytest.mark.gpu
@pytest.mark.parametrize('n,hw', [(2, 160), (1, 416)])
def test_yolov3_gpu_vs_oracle(cuda, n, hw):
    from deep_insight_face.detector.run import yolo_v3_face
    net = yolo_v3_face(1, hw, max_batch=2)
    net.init_synthetic(2024)
    p = net.get_weights()
    x = _frames(n, hw, seed=3)
    got = net.predict_on_batch(x)
    want = odet.yolov3_forward(x, p)
    assert len(got) == 3
    for g, w in zip(got, want):
        assert g.shape == w.shape
        np.testing.assert_allclose(g, w, rtol=2e-3, atol=2e-4 * np.abs(w).max())


@pytest.mark.gpu
def test_yolov3_two_lanes_equal_one_lane(cuda, monkeypatch):
    """The three-output detector takes the two-lane executor as well (round 4: a lane writes its images' part of each of
    the three maps; the split threshold counts pixels -- 5 frames at 416 x 416, 49 at 128 x 128): the same maps as the
    single-lane executor up to the kernels' batch-dependent summation splits, for an odd batch too, and a batch below
    the threshold stays on one lane."""
    from deep_insight_face.detector.run import yolo_v3_face
    monkeypatch.delenv('DIF_STREAMS', raising=False)
    two = yolo_v3_face(1, 128, max_batch=101)
    two.init_synthetic(2024)
    monkeypatch.setenv('DIF_STREAMS', '1')
    one = yolo_v3_face(1, 128, max_batch=101)
    one.set_weights(two.get_weights())
    monkeypatch.delenv('DIF_STREAMS')
    x = _frames(101, 128, seed=9)
    for n in (101, 64, 3):
        a, b = two.predict_on_batch(x[:n]), one.predict_on_batch(x[:n])
        assert [t.shape for t in a] == [(n, 4, 4, 18), (n, 8, 8, 18), (n, 16, 16, 18)]
        for u, v in zip(a, b):
            np.testing.assert_allclose(u, v, rtol=1e-4, atol=1e-5 * np.abs(v).max())
    want = odet.yolov3_forward(x[99:101], two.get_weights())     # the second lane's last images, against the oracle
    for g, w in zip(two.predict_on_batch(x), want):
        np.testing.assert_allclose(g[99:101], w, rtol=2e-3, atol=2e-4 * np.abs(w).max())


@pytest.mark.gpu
def test_yolov3_wide_patch_kernel_equals_gather_kernel(cuda):
    """Maps 29 .. 59 wide (the detector's 52 x 52 stage at 416 x 416) take conv_tn_kernel with a 256-entry halo patch where
    the launch has several tiles per resident block (batch 64 in the bench); 'dbg' bit 512 sends them there at any batch,
    bit 8192 keeps them on the per-K-step gather (stream-K there at these tile counts: split sums).  Same products: the three
    maps agree to float32 rounding carried through 75 layers (2e-5 of the map's largest value; an indexing slip shows as O(1));
    several frames so that tiles span the image boundary; 44- and 36-wide maps (352, 288) beside the 52-wide one."""
    from deep_insight_face.detector.run import yolo_v3_face
    for hw, n in ((416, 2), (352, 3), (288, 2)):
        net = yolo_v3_face(1, hw, max_batch=n)
        net.init_synthetic(7)
        x = _frames(n, hw, seed=hw)
        net.set_option('sk2', 0)                    # (round 5's small-batch paths would take these layers at these batches)
        net.set_option('mt', 0)
        net.set_option('dbg', 512)
        a = net.predict_on_batch(x)
        kernels = {k for _, k, _ in net.op_table()}
        assert any(k.startswith('conv_tn_kernel<64x128,patch256') for k in kernels), kernels
        net.set_option('dbg', 512 + 8192)
        b = net.predict_on_batch(x)
        assert not any(k.startswith('conv_tn_kernel<64x128,patch256') for _, k, _ in net.op_table())
        for u, v in zip(a, b):
            np.testing.assert_allclose(u, v, rtol=1e-3, atol=2e-5 * np.abs(v).max())
        net.close()


@pytest.mark.gpu
def test_detection_pipeline_end_to_end(cuda):
    """letterbox -> network -> decode -> NMS through the reference-named entry points, compared with
    the oracle fed with the same (GPU-produced) maps, and with the oracle's own network forward."""
    from PIL import Image
    from deep_insight_face.detector import run, yolov3
    net = run.yolo_v3_face(1, 416, max_batch=1)
    p = _yolo_params()
    for head in (58, 66, 74):                            # tame the random heads: exp(t_w) must stay finite
        p['conv_%d/kernel' % head] = p['conv_%d/kernel' % head] * np.float32(1e-5)
    p['conv_58/bias'] = p['conv_58/bias'].copy()      # make the coarse head fire: objectness bias up
    p['conv_58/bias'][4::6] += 2.0
    p['conv_58/bias'][5::6] += 2.0
    for head in (66, 74):
        p['conv_%d/bias' % head] = p['conv_%d/bias' % head].copy()
        p['conv_%d/bias' % head][4::6] -= 6.0
    net.set_weights(p)
    img = Image.fromarray(np.random.default_rng(5).integers(0, 256, (300, 500, 3), dtype=np.uint8))
    boxes, boxed = run.get_bounding_box(net, img, run.ANCHORS, 1, (416, 416), score_threshold=0.3)
    assert boxed.size == (416, 416)
    x = np.expand_dims(np.array(boxed, dtype='float32') / 255., 0)
    maps = net.predict(x)
    ob, os_, oc = odet.get_yolo_output(maps, odet.ANCHORS, 1, (300, 500), 20, 0.3, 0.5)
    assert len(boxes) == len(ob) > 0
    for (left, top, right, bottom), o in zip(boxes, ob):
        np.testing.assert_allclose([top, left, bottom, right], o, rtol=2e-5, atol=2e-3)
    det = run.YoloDetection(model=net, score=0.3)
    crops, bbs = det(np.array(img))
    assert len(crops) == len(boxes) and all(c.ndim == 3 for c in crops)
    with pytest.raises(AssertionError, match='Invalid image format'):
        det('nope')
