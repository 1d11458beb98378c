"""NumPy restatement of the reference's YOLOv3-face post-processing.  TEST INFRASTRUCTURE
ONLY (see oracle/__init__.py).  PARITY UNPINNED: the reference code is Keras-backend /
TensorFlow (``K.sigmoid``, ``tf.image.non_max_suppression``), which cannot run here; the
arithmetic below follows it line by line and restates the TF primitive from its public
definition.

Follows deep_insight_face/detector/yolov3.py:36-66 (yolo_head), :69-93 (correct_boxes),
:96-106 (boxes_and_scores), :122-172 (get_yolo_output).
"""
import numpy as np

ANCHORS = np.array([[10, 13], [16, 30], [33, 23], [30, 61], [62, 45], [59, 119], [116, 90], [156, 198], [373, 326]],
                   dtype=np.float32)   # detector/yolo_cfg/yolo_anchors.txt


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def yolo_head(feats, anchors, num_classes, input_shape):
    """yolov3.py:36-66.  feats [N,gh,gw,na*(5+C)] -> box_xy, box_wh (fractions of the input),
    box_confidence, box_class_probs, each [N,gh,gw,na,*]."""
    na = len(anchors)
    n, gh, gw = feats.shape[:3]
    f = feats.reshape(n, gh, gw, na, num_classes + 5).astype(np.float32)
    gy, gx = np.meshgrid(np.arange(gh, dtype=np.float32), np.arange(gw, dtype=np.float32), indexing='ij')
    grid = np.stack([gx, gy], axis=-1)[None, :, :, None, :]
    box_xy = (_sigmoid(f[..., :2]) + grid) / np.array([gw, gh], dtype=np.float32)
    box_wh = np.exp(f[..., 2:4]) * anchors.reshape(1, 1, 1, na, 2).astype(np.float32) / \
        np.array(input_shape[::-1], dtype=np.float32)
    return box_xy, box_wh, _sigmoid(f[..., 4:5]), _sigmoid(f[..., 5:])


def correct_boxes(box_xy, box_wh, input_shape, image_shape):
    """yolov3.py:69-93: undo the letterbox, return [y_min, x_min, y_max, x_max] in image pixels."""
    box_yx = box_xy[..., ::-1]
    box_hw = box_wh[..., ::-1]
    input_shape = np.asarray(input_shape, dtype=np.float32)
    image_shape = np.asarray(image_shape, dtype=np.float32)
    new_shape = np.round(image_shape * np.min(input_shape / image_shape))
    offset = (input_shape - new_shape) / 2. / input_shape
    scale = input_shape / new_shape
    box_yx = (box_yx - offset) * scale
    box_hw = box_hw * scale
    mins = box_yx - box_hw / 2.
    maxes = box_yx + box_hw / 2.
    boxes = np.concatenate([mins[..., 0:1], mins[..., 1:2], maxes[..., 0:1], maxes[..., 1:2]], axis=-1)
    return (boxes * np.concatenate([image_shape, image_shape])).astype(np.float32)


def boxes_and_scores(feats, anchors, num_classes, input_shape, image_shape):
    """yolov3.py:96-106 for ONE image (the reference runs batch 1)."""
    xy, wh, conf, probs = yolo_head(feats, anchors, num_classes, input_shape)
    boxes = correct_boxes(xy, wh, input_shape, image_shape).reshape(-1, 4)
    return boxes, (conf * probs).reshape(-1, num_classes).astype(np.float32)


def _iou(a, b):
    """IoU as tf.image.non_max_suppression computes it (corner order normalised)."""
    ay0, ay1 = min(a[0], a[2]), max(a[0], a[2])
    ax0, ax1 = min(a[1], a[3]), max(a[1], a[3])
    by0, by1 = min(b[0], b[2]), max(b[0], b[2])
    bx0, bx1 = min(b[1], b[3]), max(b[1], b[3])
    area_a, area_b = (ay1 - ay0) * (ax1 - ax0), (by1 - by0) * (bx1 - bx0)
    if area_a <= 0 or area_b <= 0:
        return 0.0
    ih = max(min(ay1, by1) - max(ay0, by0), 0.0)
    iw = max(min(ax1, bx1) - max(ax0, bx0), 0.0)
    inter = ih * iw
    return inter / (area_a + area_b - inter)


def non_max_suppression(boxes, scores, max_output_size, iou_threshold):
    """Greedy NMS with tf.image.non_max_suppression's ordering: highest score first, equal
    scores by lower index; a candidate is dropped when its IoU with an already selected box is
    > iou_threshold."""
    order = sorted(range(len(scores)), key=lambda i: (-float(scores[i]), i))
    keep = []
    for i in order:
        if len(keep) >= max_output_size:
            break
        if all(_iou(boxes[i], boxes[j]) <= iou_threshold for j in keep):
            keep.append(i)
    return np.array(keep, dtype=np.int64)


def get_yolo_output(outputs, anchors, num_classes, image_shape, max_boxes=20, score_threshold=.6, iou_threshold=.5):
    """yolov3.py:122-172 for one image: outputs = list of [1,gh,gw,na*(5+C)] maps, coarse first.
    -> (boxes_[K,4], scores_[K], classes_[K])."""
    num_layers = len(outputs)
    anchor_mask = [[6, 7, 8], [3, 4, 5], [0, 1, 2]] if num_layers == 3 else [[3, 4, 5], [1, 2, 3]]
    input_shape = (outputs[0].shape[1] * 32, outputs[0].shape[2] * 32)
    boxes, box_scores = [], []
    for l in range(num_layers):
        b, s = boxes_and_scores(outputs[l], anchors[anchor_mask[l]], num_classes, input_shape, image_shape)
        boxes.append(b)
        box_scores.append(s)
    boxes = np.concatenate(boxes, axis=0)
    box_scores = np.concatenate(box_scores, axis=0)
    mask = box_scores >= score_threshold
    boxes_, scores_, classes_ = [], [], []
    for c in range(num_classes):
        cb = boxes[mask[:, c]]
        cs = box_scores[:, c][mask[:, c]]
        keep = non_max_suppression(cb, cs, max_boxes, iou_threshold)
        boxes_.append(cb[keep])
        scores_.append(cs[keep])
        classes_.append(np.full(len(keep), c, dtype=np.int32))
    return np.concatenate(boxes_, axis=0), np.concatenate(scores_, axis=0), np.concatenate(classes_, axis=0)


# --------------------------------------------------------------------------- YOLOv3-face network
# The detector network itself is not defined in Python in the reference: it is the Darknet cfg
# detector/yolo_cfg/yolov3-face.cfg turned into a Keras model by scripts/yolo_convert_tf.py:60-215.
# Restated here from that converter's layer semantics (Conv2D without bias -> BatchNormalization
# (Keras default epsilon 1e-3) -> LeakyReLU(0.1); stride 2 = ZeroPadding2D(((1,0),(1,0))) + VALID;
# shortcut = Add; route = concat; upsample = nearest x2; heads = linear Conv2D with bias).
def _leaky(x):
    return np.where(x >= 0, x, x * np.asarray(0.1, x.dtype))


def yolov3_spec(num_classes=1):
    spec, ci = [], [0]

    def conv(cin, cout, k, bn=True):
        i = ci[0]
        ci[0] += 1
        s = [('conv_%d/kernel' % i, (k, k, cin, cout))]
        if bn:
            s += [('bn_%d/%s' % (i, n), (cout,)) for n in ('gamma', 'beta', 'moving_mean', 'moving_variance')]
        else:
            s += [('conv_%d/bias' % i, (cout,))]
        spec.extend(s)
        return cout

    c = conv(3, 32, 3)
    for cout, blocks in ((64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)):
        c = conv(c, cout, 3)
        for _ in range(blocks):
            conv(cout, cout // 2, 1)
            conv(cout // 2, cout, 3)
    n_out = 3 * (5 + num_classes)

    def head(cin, w):
        conv(cin, w, 1), conv(w, 2 * w, 3), conv(2 * w, w, 1), conv(w, 2 * w, 3), conv(2 * w, w, 1)
        conv(w, 2 * w, 3), conv(2 * w, n_out, 1, bn=False)

    head(1024, 512)
    conv(512, 256, 1)
    head(256 + 512, 256)
    conv(256, 128, 1)
    head(128 + 256, 128)
    return spec


def yolov3_forward(x, p, num_classes=1):
    """x [N,H,W,3] (letterboxed, /255) -> [y13, y26, y52] maps of 3*(5+classes) channels."""
    from .nets import batchnorm, conv2d
    ci = [0]

    def cbl(x, k, stride):
        i = ci[0]
        ci[0] += 1
        if stride == 2:
            y = conv2d(x, p['conv_%d/kernel' % i], stride=2, pad=(1, 0, 1, 0))
        else:
            pd = 1 if k == 3 else 0
            y = conv2d(x, p['conv_%d/kernel' % i], pad=(pd, pd, pd, pd))
        return _leaky(batchnorm(y, p, 'bn_%d' % i, 1e-3))

    def linear(x):
        i = ci[0]
        ci[0] += 1
        return conv2d(x, p['conv_%d/kernel' % i], p['conv_%d/bias' % i])

    y = cbl(x, 3, 1)
    routes = {}
    for cout, blocks in ((64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)):
        y = cbl(y, 3, 2)
        for _ in range(blocks):
            y = y + cbl(cbl(y, 1, 1), 3, 1)
        routes[cout] = y

    def head(y):
        for k in (1, 3, 1, 3, 1):
            y = cbl(y, k, 1)
        return y, linear(cbl(y, 3, 1))

    def up(y):
        return np.repeat(np.repeat(y, 2, axis=1), 2, axis=2)

    b, y13 = head(y)
    b, y26 = head(np.concatenate([up(cbl(b, 1, 1)), routes[512]], axis=3))
    b, y52 = head(np.concatenate([up(cbl(b, 1, 1)), routes[256]], axis=3))
    return [y13, y26, y52]
