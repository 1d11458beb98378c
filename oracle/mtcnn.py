"""NumPy restatement of the MTCNN cascade (BASELINE configs[4] as worded).  TEST INFRASTRUCTURE ONLY
(see oracle/__init__.py).  **PARITY UNPINNED**: MTCNN is NOT in the reference -- `config.py:37` and
`detector/run.py:124` mention it in comments only; the detector the reference ships is YOLOv3-face
(oracle/detector.py).  The three networks restate the public definition (Zhang, Zhang, Li, Qiao 2016, "Joint Face
Detection and Alignment using Multi-task Cascaded Convolutional Networks"; every convolution VALID with bias and
PReLU, Caffe max-pooling = ceil mode); the cascade restates the public algorithm (image pyramid by 0.709 from
12 / min_face, per-scale proposals from the P-Net map with stride 2 and cell 12, NMS, box regression, squaring,
24 x 24 crops through R-Net, 48 x 48 crops through O-Net) WITH THE CHOICES THE DEVICE PATH MAKES, each of which is
a deviation a port of someone else's weights must know about:
  * static shapes -- a fixed number of slots per frame and stage (`cap`), empty slots carry score -1: the whole
    cascade runs for a batch of frames without a host round trip; the published code keeps variable-length lists;
  * IoU as the library's `dif_nms` computes it (continuous boxes, union in the denominator) for every suppression;
    the published code adds one pixel to widths and heights and uses min(area) in the last stage;
  * crops are clamped to the frame and resampled by area coverage (`dif_crop_resize`'s arithmetic, oracle/imageops.py)
    -- the published code zero-pads outside the frame and resamples with cv2.resize;
  * the sibling heads of a network are one layer (filters concatenated: [logits 2 | box 4 | landmarks 10 | zeros]),
    P-Net's first layer holds 12 filters of which two are zero; face probability = softmax over the two logits.
The calling convention kept from the reference is `detector/run.py:120-173` (a detector object called with an image
returns crops and boxes).

Frames are uint8 [N, H, W, 3]; networks see (x - 127.5) / 128.
"""
import math

import numpy as np

from . import imageops
from . import nets

STRIDE, CELL = 2, 12


# --------------------------------------------------------------------------- networks
def _ceil_pool(x, k, stride):
    """Caffe max-pooling: the last window may hang over the bottom / right edge."""
    n, h, w, c = x.shape
    ho = -(-(h - k) // stride) + 1
    wo = -(-(w - k) // stride) + 1
    ph = (ho - 1) * stride + k - h
    pw = (wo - 1) * stride + k - w
    return nets.maxpool(x, k, stride, pad=(0, ph, 0, pw), pad_value=-np.inf)


def _cp(x, p, name, prelu=True):
    y = nets.conv2d(x, p[name + '/kernel'], p[name + '/bias'])
    return nets.prelu(y, p[name + '_prelu/alpha']) if prelu else y


def pnet(x, p):
    """[N, H, W, 3] (already normalised) -> head map [N, H', W', 8] = [logits 2 | box 4 | 0 0]."""
    x = _cp(x, p, 'conv1')
    x = _ceil_pool(x, 2, 2)
    x = _cp(x, p, 'conv2')
    x = _cp(x, p, 'conv3')
    return _cp(x, p, 'head', prelu=False)


def rnet(x, p):
    """[N, 24, 24, 3] -> [N, 8]."""
    x = _cp(x, p, 'conv1')
    x = _ceil_pool(x, 3, 2)
    x = _cp(x, p, 'conv2')
    x = _ceil_pool(x, 3, 2)
    x = _cp(x, p, 'conv3')
    x = _cp(x, p, 'fc1')
    return _cp(x, p, 'head', prelu=False).reshape(x.shape[0], -1)


def onet(x, p):
    """[N, 48, 48, 3] -> [N, 16] = [logits 2 | box 4 | landmarks 10]."""
    x = _cp(x, p, 'conv1')
    x = _ceil_pool(x, 3, 2)
    x = _cp(x, p, 'conv2')
    x = _ceil_pool(x, 3, 2)
    x = _cp(x, p, 'conv3')
    x = _ceil_pool(x, 2, 2)
    x = _cp(x, p, 'conv4')
    x = _cp(x, p, 'fc1')
    return _cp(x, p, 'head', prelu=False).reshape(x.shape[0], -1)


def spec(stage):
    """(name, shape) table of the library's parameters for 'pnet' / 'rnet' / 'onet'."""
    def cp(name, k, cin, cout, prelu=True):
        rows = [(name + '/kernel', (k, k, cin, cout)), (name + '/bias', (cout,))]
        return ([(name + '_prelu/alpha', (cout,))] + rows) if prelu else rows
    if stage == 'pnet':
        return cp('conv1', 3, 3, 12) + cp('conv2', 3, 12, 16) + cp('conv3', 3, 16, 32) + cp('head', 1, 32, 8, False)
    if stage == 'rnet':
        return (cp('conv1', 3, 3, 28) + cp('conv2', 3, 28, 48) + cp('conv3', 2, 48, 64) + cp('fc1', 3, 64, 128) +
                cp('head', 1, 128, 8, False))
    return (cp('conv1', 3, 3, 32) + cp('conv2', 3, 32, 64) + cp('conv3', 3, 64, 64) + cp('conv4', 2, 64, 128) +
            cp('fc1', 3, 128, 256) + cp('head', 1, 256, 16, False))


def normalise(u8):
    return (u8.astype(np.float32) - np.float32(127.5)) * np.float32(1.0 / 128.0)


# --------------------------------------------------------------------------- cascade
def pyramid_scales(h, w, min_face=20, factor=0.709):
    m = 12.0 / min_face
    side = min(h, w) * m
    out = []
    while side >= 12:
        out.append(m)
        m *= factor
        side *= factor
    return out


def scaled_size(h, w, scale):
    return int(math.ceil(h * scale)), int(math.ceil(w * scale))


def face_prob(logits2):
    """softmax over [not-face, face] -> P(face), float32, as the device evaluates it: 1 / (1 + exp(l0 - l1))."""
    d = (logits2[..., 0] - logits2[..., 1]).astype(np.float32)
    return (np.float32(1.0) / (np.float32(1.0) + np.exp(d, dtype=np.float32))).astype(np.float32)


def propose(head_map, scale, threshold):
    """P-Net head map [gh, gw, 8] of one frame at one scale -> dense proposals, one per cell:
    boxes [gh*gw, 4] (x1, y1, x2, y2 in frame pixels), scores [gh*gw] (-1 below the threshold), reg [gh*gw, 4]."""
    gh, gw, _ = head_map.shape
    prob = face_prob(head_map[..., 0:2])
    ys, xs = np.meshgrid(np.arange(gh, dtype=np.float32), np.arange(gw, dtype=np.float32), indexing='ij')
    inv = np.float32(1.0) / np.float32(scale)
    x1 = np.trunc((np.float32(STRIDE) * xs + np.float32(1.0)) * inv)
    y1 = np.trunc((np.float32(STRIDE) * ys + np.float32(1.0)) * inv)
    x2 = np.trunc((np.float32(STRIDE) * xs + np.float32(CELL)) * inv)
    y2 = np.trunc((np.float32(STRIDE) * ys + np.float32(CELL)) * inv)
    boxes = np.stack([x1, y1, x2, y2], -1).reshape(-1, 4).astype(np.float32)
    scores = np.where(prob >= np.float32(threshold), prob, np.float32(-1.0)).reshape(-1).astype(np.float32)
    return boxes, scores, head_map[..., 2:6].reshape(-1, 4).astype(np.float32)


def _iou32(a, b):
    """IoU in float32, operation by operation as dif_nms evaluates it (csrc/detector.hip: box_iou) -- the cells' boxes have
    integer corners, so IoUs hit a threshold EXACTLY in exact arithmetic and the rounding of each step decides."""
    f = np.float32
    ax0, ax1, ay0, ay1 = min(a[0], a[2]), max(a[0], a[2]), min(a[1], a[3]), max(a[1], a[3])
    bx0, bx1, by0, by1 = min(b[0], b[2]), max(b[0], b[2]), min(b[1], b[3]), max(b[1], b[3])
    area_a = f(f(ax1 - ax0) * f(ay1 - ay0))
    area_b = f(f(bx1 - bx0) * f(by1 - by0))
    if area_a <= 0 or area_b <= 0:
        return f(0)
    iw = max(f(min(ax1, bx1) - max(ax0, bx0)), f(0))
    ih = max(f(min(ay1, by1) - max(ay0, by0)), f(0))
    inter = f(iw * ih)
    return f(inter / f(f(area_a + area_b) - inter))


def nms_slots(boxes, scores, cap, iou):
    """Greedy suppression over the slots with score >= 0 -> indices of the kept slots, best first, -1 padded to cap.
    dif_nms: highest score first, ties by lower index; a slot is dropped when its IoU with a kept one is > iou."""
    order = sorted((i for i in range(len(scores)) if scores[i] >= 0), key=lambda i: (-float(scores[i]), i))
    boxes = boxes.astype(np.float32)
    thr = np.float32(iou)
    keep = []
    for i in order:
        if len(keep) >= cap:
            break
        if all(_iou32(boxes[i], boxes[j]) <= thr for j in keep):
            keep.append(i)
    out = np.full(cap, -1, dtype=np.int32)
    out[:len(keep)] = keep
    return out


def gather_slots(keep, *arrays):
    """Rows `keep` of every array; -1 -> an empty slot (zeros, score -1 for 1-D arrays)."""
    outs = []
    for a in arrays:
        o = np.zeros((len(keep),) + a.shape[1:], dtype=a.dtype)
        if a.ndim == 1:
            o[:] = -1
        ok = keep >= 0
        o[ok] = a[keep[ok]]
        outs.append(o)
    return outs


def calibrate(boxes, reg):
    """Box regression, then squaring around the centre, then truncation -- float32 throughout, as the device."""
    w = boxes[:, 2] - boxes[:, 0] + np.float32(1.0)
    h = boxes[:, 3] - boxes[:, 1] + np.float32(1.0)
    x1 = boxes[:, 0] + reg[:, 0] * w
    y1 = boxes[:, 1] + reg[:, 1] * h
    x2 = boxes[:, 2] + reg[:, 2] * w
    y2 = boxes[:, 3] + reg[:, 3] * h
    w = x2 - x1
    h = y2 - y1
    side = np.maximum(w, h)
    x1 = x1 + w * np.float32(0.5) - side * np.float32(0.5)
    y1 = y1 + h * np.float32(0.5) - side * np.float32(0.5)
    out = np.stack([np.trunc(x1), np.trunc(y1), np.trunc(x1 + side), np.trunc(y1 + side)], -1)
    return out.astype(np.float32)


def crops_of(frame, boxes, scores, size):
    """One size x size crop per slot (an empty slot or a box without area gives a black crop)."""
    out = np.zeros((len(boxes), size, size, 3), dtype=np.uint8)
    for i, (b, s) in enumerate(zip(boxes, scores)):
        if s >= 0:
            out[i] = imageops.crop_resize(frame, b, 0.0, size)
    return out


def detect(frames, params, min_face=20, thresholds=(0.6, 0.7, 0.7), cap=(64, 32, 16), factor=0.709):
    """frames uint8 [N, H, W, 3]; params = {'pnet': {...}, 'rnet': {...}, 'onet': {...}}.
    -> boxes [N, cap[2], 4] (x1, y1, x2, y2), scores [N, cap[2]] (-1 = empty slot), best first;
       plus the intermediate stage outputs for the tests (dict)."""
    n, h, w, _ = frames.shape
    scales = pyramid_scales(h, w, min_face, factor)
    dbg = {'scales': scales}
    all_boxes, all_scores = [], []
    for f in range(n):
        fb, fs, fr = [], [], []
        for sc in scales:
            hs, ws = scaled_size(h, w, sc)
            img = imageops.area_resize(frames[f], (ws, hs))      # (width, height), as cv2 takes it
            m = pnet(normalise(img)[None], params['pnet'])[0]
            b, s, r = propose(m, sc, thresholds[0])
            keep = nms_slots(b, s, cap[0], 0.5)
            kb, ks, kr = gather_slots(keep, b, s, r)
            fb.append(kb)
            fs.append(ks)
            fr.append(kr)
        b, s, r = np.concatenate(fb), np.concatenate(fs), np.concatenate(fr)
        keep = nms_slots(b, s, cap[1], 0.7)
        b, s, r = gather_slots(keep, b, s, r)
        b = calibrate(b, r)
        dbg.setdefault('stage1_boxes', []).append(b)
        dbg.setdefault('stage1_scores', []).append(s)
        # R-Net
        o = rnet(normalise(crops_of(frames[f], b, s, 24)), params['rnet'])
        p = face_prob(o[:, 0:2])
        s = np.where((s >= 0) & (p >= np.float32(thresholds[1])), p, np.float32(-1.0)).astype(np.float32)
        keep = nms_slots(b, s, cap[2], 0.7)
        b, s, r = gather_slots(keep, b, s, o[:, 2:6].astype(np.float32))
        b = calibrate(b, r)
        dbg.setdefault('stage2_boxes', []).append(b)
        dbg.setdefault('stage2_scores', []).append(s)
        # O-Net
        o = onet(normalise(crops_of(frames[f], b, s, 48)), params['onet'])
        p = face_prob(o[:, 0:2])
        s = np.where((s >= 0) & (p >= np.float32(thresholds[2])), p, np.float32(-1.0)).astype(np.float32)
        b = calibrate_plain(b, o[:, 2:6].astype(np.float32))
        keep = nms_slots(b, s, cap[2], 0.7)
        b, s = gather_slots(keep, b, s)
        all_boxes.append(b)
        all_scores.append(s)
    return np.stack(all_boxes), np.stack(all_scores), dbg


def calibrate_plain(boxes, reg):
    """Last stage: regression only (no squaring), as the public cascade ends."""
    w = boxes[:, 2] - boxes[:, 0] + np.float32(1.0)
    h = boxes[:, 3] - boxes[:, 1] + np.float32(1.0)
    return np.stack([boxes[:, 0] + reg[:, 0] * w, boxes[:, 1] + reg[:, 1] * h, boxes[:, 2] + reg[:, 2] * w,
                     boxes[:, 3] + reg[:, 3] * h], -1).astype(np.float32)
