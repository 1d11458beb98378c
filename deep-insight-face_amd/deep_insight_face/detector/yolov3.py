"""YOLOv3-face post-processing with the reference's signatures, computed on the MI355X.

Mirrors deep_insight_face/detector/yolov3.py:36-172: ``boxes_and_scores`` (box decode, letterbox
correction, confidence x class probability) and ``get_yolo_output`` (score filter + per-class greedy
non-max suppression, which the reference delegates to ``tf.image.non_max_suppression``).  The
feature maps are the three output tensors of the detector network, coarse grid first,
``[N, gh, gw, 3*(5+classes)]``.  Extension over the reference (which runs batch 1): a batch of N
images with per-image ``image_shape``.
"""
import ctypes
import typing

import numpy as np
import torch
from PIL import Image

from .. import _native as N

ANCHOR_MASK_3 = [[6, 7, 8], [3, 4, 5], [0, 1, 2]]      # yolov3.py:133
ANCHOR_MASK_2 = [[3, 4, 5], [1, 2, 3]]


def _decode(outputs, anchors, num_classes, image_shape):
    dev = N.require_device()
    feats = [N.to_device_f32(o, dev)[0] for o in outputs]
    n = feats[0].shape[0]
    num_layers = len(feats)
    mask = ANCHOR_MASK_3 if num_layers == 3 else ANCHOR_MASK_2
    anchors = np.asarray(anchors, dtype=np.float32).reshape(-1, 2)
    input_h, input_w = int(feats[0].shape[1]) * 32, int(feats[0].shape[2]) * 32
    grid = (ctypes.c_int32 * (2 * num_layers))(*[int(v) for f in feats for v in f.shape[1:3]])
    anc = (ctypes.c_float * (6 * num_layers))(*[float(v) for l in range(num_layers) for v in anchors[mask[l]].reshape(-1)])
    ptrs = (ctypes.c_void_p * num_layers)(*[f.data_ptr() for f in feats])
    ishape = np.asarray(image_shape, dtype=np.float32).reshape(-1, 2)
    if ishape.shape[0] == 1 and n > 1:
        ishape = np.repeat(ishape, n, axis=0)
    ishape_t = torch.from_numpy(np.ascontiguousarray(ishape)).to(dev)
    ntot = sum(int(f.shape[1]) * int(f.shape[2]) * 3 for f in feats)
    boxes = torch.empty((n, ntot, 4), dtype=torch.float32, device=dev)
    scores = torch.empty((n, ntot, num_classes), dtype=torch.float32, device=dev)
    N.check(N.lib.dif_yolo_decode(ptrs, grid, anc, num_layers, n, num_classes, input_h, input_w, N.ptr(ishape_t),
                                  N.ptr(boxes), N.ptr(scores), N.stream_ptr()))
    return boxes, scores


def boxes_and_scores_all(outputs, anchors, num_classes, image_shape):
    """All layers at once: (boxes [N, n_boxes, 4] as y_min, x_min, y_max, x_max in image pixels,
    scores [N, n_boxes, classes]) -- yolov3.py:96-106 applied to every output layer and
    concatenated in the order get_yolo_output concatenates them (:139-147)."""
    boxes, scores = _decode(outputs, anchors, num_classes, image_shape)
    return boxes.cpu().numpy(), scores.cpu().numpy()


def non_max_suppression(boxes, scores, max_output_size, iou_threshold=0.5, score_threshold=float('-inf')):
    """tf.image.non_max_suppression for one list of boxes [K,4] / scores [K] -> kept indices."""
    dev = N.require_device()
    b, _ = N.to_device_f32(np.asarray(boxes, dtype=np.float32).reshape(1, -1, 4), dev)
    s, _ = N.to_device_f32(np.asarray(scores, dtype=np.float32).reshape(1, -1, 1), dev)
    k = b.shape[1]
    keep = torch.empty((1, 1, max_output_size), dtype=torch.int32, device=dev)
    cnt = torch.empty((1, 1), dtype=torch.int32, device=dev)
    ws = torch.empty((max(k, 1),), dtype=torch.uint8, device=dev)
    N.check(N.lib.dif_nms(N.ptr(b), N.ptr(s), 1, k, 1, int(max_output_size), float(score_threshold),
                          float(iou_threshold), N.ptr(ws), N.ptr(keep), N.ptr(cnt), N.stream_ptr()))
    return keep[0, 0, :int(cnt[0, 0])].cpu().numpy().astype(np.int64)


def get_yolo_output(outputs: typing.List, anchors: np.ndarray, num_classes: int, image_shape: typing.Tuple,
                    max_boxes: int = 20, score_threshold: float = .6, iou_threshold: float = .5):
    """yolov3.py:122-172.  For a batch of one (the reference's case) returns (boxes_, scores_,
    classes_) as NumPy arrays; for N > 1 a list of such triples."""
    boxes, scores = _decode(outputs, anchors, num_classes, image_shape)
    n, ntot = boxes.shape[0], boxes.shape[1]
    dev = boxes.device
    keep = torch.empty((n, num_classes, max_boxes), dtype=torch.int32, device=dev)
    cnt = torch.empty((n, num_classes), dtype=torch.int32, device=dev)
    ws = torch.empty((n * num_classes * ntot,), dtype=torch.uint8, device=dev)
    N.check(N.lib.dif_nms(N.ptr(boxes), N.ptr(scores), n, ntot, num_classes, int(max_boxes), float(score_threshold),
                          float(iou_threshold), N.ptr(ws), N.ptr(keep), N.ptr(cnt), N.stream_ptr()))
    boxes_h, scores_h, keep_h, cnt_h = boxes.cpu().numpy(), scores.cpu().numpy(), keep.cpu().numpy(), cnt.cpu().numpy()
    results = []
    for i in range(n):
        b_, s_, c_ = [], [], []
        for c in range(num_classes):
            idx = keep_h[i, c, :cnt_h[i, c]]
            b_.append(boxes_h[i, idx])
            s_.append(scores_h[i, idx, c])
            c_.append(np.full(len(idx), c, dtype=np.int32))
        results.append((np.concatenate(b_, axis=0), np.concatenate(s_, axis=0), np.concatenate(c_, axis=0)))
    return results[0] if n == 1 else results


def letterbox_image(image: Image.Image, size: typing.Tuple[int, int]) -> Image.Image:
    """Resize with unchanged aspect ratio and grey padding (yolov3.py:108-119)."""
    iw, ih = image.size
    w, h = size
    scale = min(w / iw, h / ih)
    nw, nh = int(iw * scale), int(ih * scale)
    canvas = Image.new('RGB', size, (128, 128, 128))
    canvas.paste(image.resize((nw, nh), Image.BICUBIC), ((w - nw) // 2, (h - nh) // 2))
    return canvas


def letterbox_batch(frames, size: int = 416) -> torch.Tensor:
    """``letterbox_image`` for a whole batch on the device: uint8 [N,H,W,3] (ndarray or tensor)
    -> uint8 CUDA tensor [N,size,size,3] (dif_letterbox; PIL BICUBIC semantics, rounding may
    differ from PIL's fixed-point passes by one grey level)."""
    dev = N.require_device()
    t = torch.from_numpy(np.ascontiguousarray(frames)) if not torch.is_tensor(frames) else frames
    if t.dim() != 4 or t.shape[3] != 3 or t.dtype != torch.uint8:
        raise ValueError('expected uint8 frames [N,H,W,3], got %s %s' % (t.dtype, tuple(t.shape)))
    t = t.to(dev).contiguous()
    out = torch.empty((t.shape[0], size, size, 3), dtype=torch.uint8, device=dev)
    N.check(N.lib.dif_letterbox(N.ptr(t), t.shape[0], t.shape[1], t.shape[2], N.ptr(out), size, N.stream_ptr()))
    return out
