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
