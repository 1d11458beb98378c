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
