"""CPU restatement of the image resampling around the detector.  TEST INFRASTRUCTURE ONLY (see
oracle/__init__.py).

* ``letterbox``: deep_insight_face/detector/yolov3.py:108-119, executed with Pillow -- the very
  library call the reference makes (``image.resize((nw, nh), Image.BICUBIC)`` + paste), so this
  half is pinned by construction.
* ``crop_margin`` / ``area_resize``: detector/run.py:63-87 (filter_bounding_box) and the resize at
  predictions.py:93,154, ``cv2.resize(image, size, interpolation=Image.BICUBIC)``: PIL's BICUBIC
  constant is 3, which cv2 interprets as INTER_AREA.  OpenCV (opencv-python, unpinned in the
  reference's setup.py) is not installed here, so cv2's uint8 INTER_AREA paths are restated from
  OpenCV's published source, operation by operation (see `area_resize`).  PARITY UNPINNED for this half.
"""
import numpy as np
from PIL import Image


def letterbox(frame: np.ndarray, size: int) -> np.ndarray:
    image = Image.fromarray(frame)
    iw, ih = image.size
    scale = min(size / iw, size / ih)
    nw, nh = int(iw * scale), int(ih * scale)
    canvas = Image.new('RGB', (size, size), (128, 128, 128))
    canvas.paste(image.resize((nw, nh), Image.BICUBIC), ((size - nw) // 2, (size - nh) // 2))
    return np.array(canvas)


def crop_margin(frame: np.ndarray, box_ltrb, margin: float) -> np.ndarray:
    h, w = frame.shape[:2]
    bb = np.zeros(4, dtype=np.int32)
    bb[0] = np.maximum(box_ltrb[0] - margin / 2, 0)
    bb[1] = np.maximum(box_ltrb[1] - margin / 2, 0)
    bb[2] = np.minimum(box_ltrb[2] + margin / 2, w)
    bb[3] = np.minimum(box_ltrb[3] + margin / 2, h)
    return frame[bb[1]:bb[3], bb[0]:bb[2], :]


# ---------------------------------------------------------------------------------------------------------------
# cv2.resize(..., interpolation=INTER_AREA) for uint8 images, restated from OpenCV's published source
# (modules/imgproc/src/resize.cpp, 4.x: `resize`, `computeResizeAreaTab`, `ResizeArea_Invoker`, `resizeAreaFast_Invoker`,
# `ResizeAreaFastVec_SIMD_8u`, and -- when an axis is ENLARGED -- the linear path with `area_mode` coefficients in
# 11-bit fixed point: `HResizeLinear`, `VResizeLinear<uchar, int, short, FixedPtCast<int, uchar, 22>>`).
# PARITY UNPINNED: cv2 is not installed in this image; what is pinned is that the device kernel (csrc/imageops.hip)
# performs exactly these operations in exactly this order (tests/test_wrappers_gpu.py: bit-equal).
#
#   both axes shrink or stay (scale >= 1):
#     integer ratio 2 x 2:  (a + b + c + d + 2) >> 2                       (the SIMD body; OpenCV's scalar tail of a row
#                                                                           uses the next formula -- where it starts
#                                                                           depends on the build's vector width)
#     other integer ratios: cvRound(float(sum of the block) * (1.f / area))
#     fractional ratios:    per axis a table of (source index, float32 weight) from double arithmetic; per destination
#                           row, buf[dx] = sum_k S[sx_k] * alpha_k and sum[dx] = sum_j beta_j * buf_j[dx], every product
#                           and every addition rounded to float32, in table order; cvRound (half to even) at the end
#   any axis enlarges:      both axes take the linear path: sx = floor(dx * scale), fx = (dx + 1) - (sx + 1) / scale
#                           (clamped to [0, 1)), coefficients round(2048 (1 - fx)), round(2048 fx) as int16; rows
#                           are combined as ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2


def _area_tab(ssize: int, dsize: int, scale: float):
    """computeResizeAreaTab: per destination index the (source index, float32 weight) pairs, in order."""
    tab = [[] for _ in range(dsize)]
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = int(np.ceil(fsx1)), int(np.floor(fsx2))
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab[dx].append((sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab[dx].append((sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab[dx].append((sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def _linear_tab(ssize: int, dsize: int):
    """The linear path's area_mode coefficients: per destination index (s0, s1, a0, a1) with int16 weights of 2048."""
    inv = dsize / ssize
    scale = 1.0 / inv
    out = []
    for dx in range(dsize):
        sx = int(np.floor(dx * scale))
        fx = np.float32((dx + 1) - (sx + 1) * inv)
        fx = np.float32(0) if fx <= 0 else np.float32(fx - np.floor(fx))
        if sx < 0:
            fx, sx = np.float32(0), 0
        if sx >= ssize - 1:
            fx, sx = np.float32(0), ssize - 1
        a0 = int(np.rint(np.float32(np.float32(1) - fx) * np.float32(2048)))
        a1 = int(np.rint(fx * np.float32(2048)))
        out.append((sx, min(sx + 1, ssize - 1), a0, a1))
    return out


def area_resize(img: np.ndarray, size) -> np.ndarray:
    """img [H, W, C] uint8 -> [dh, dw, C] uint8; `size` = side of a square or (width, height) as cv2 takes it."""
    dw, dh = (size, size) if np.isscalar(size) else (int(size[0]), int(size[1]))
    sh, sw, C = img.shape
    scale_x, scale_y = 1.0 / (dw / sw), 1.0 / (dh / sh)
    if scale_x >= 1 and scale_y >= 1:
        ix, iy = int(scale_x), int(scale_y)
        fast = abs(scale_x - ix) < np.finfo(np.float64).eps and abs(scale_y - iy) < np.finfo(np.float64).eps
        if fast:
            blocks = img[:dh * iy, :dw * ix].reshape(dh, iy, dw, ix, C).astype(np.int32).sum(axis=(1, 3))
            if ix == 2 and iy == 2:
                return ((blocks + 2) >> 2).astype(np.uint8)
            v = blocks.astype(np.float32) * np.float32(np.float32(1) / np.float32(ix * iy))
            return np.clip(np.rint(v), 0, 255).astype(np.uint8)
        xtab, ytab = _area_tab(sw, dw, scale_x), _area_tab(sh, dh, scale_y)
        ntap = max(len(t) for t in xtab)
        out = np.zeros((dh, dw, C), dtype=np.uint8)
        src = img.astype(np.float32)
        rowbuf = {}

        def hbuf(sy):                                     # buf[dx] = sum_k S[sx_k] * alpha_k, sequential float32
            if sy not in rowbuf:
                buf = np.zeros((dw, C), dtype=np.float32)
                for t in range(ntap):
                    sel = np.array([dx for dx in range(dw) if len(xtab[dx]) > t], dtype=np.int64)
                    if sel.size == 0:
                        break
                    si = np.array([xtab[dx][t][0] for dx in sel], dtype=np.int64)
                    al = np.array([xtab[dx][t][1] for dx in sel], dtype=np.float32)[:, None]
                    buf[sel] = buf[sel] + src[sy, si] * al
                rowbuf[sy] = buf
            return rowbuf[sy]

        for dy in range(dh):
            acc = None
            for sy, beta in ytab[dy]:
                term = beta * hbuf(sy)
                acc = term if acc is None else acc + term
            out[dy] = np.clip(np.rint(acc), 0, 255).astype(np.uint8)
            for sy in [k for k in rowbuf if k < ytab[dy][-1][0]]:
                del rowbuf[sy]
        return out
    # an axis is enlarged: both take the linear path with area_mode coefficients, 11-bit fixed point
    xt, yt = _linear_tab(sw, dw), _linear_tab(sh, dh)
    s0 = np.array([t[0] for t in xt]); s1 = np.array([t[1] for t in xt])
    a0 = np.array([t[2] for t in xt], dtype=np.int32)[:, None]; a1 = np.array([t[3] for t in xt], dtype=np.int32)[:, None]
    src = img.astype(np.int32)
    hrow = src[:, s0] * a0[None] + src[:, s1] * a1[None]                       # [sh, dw, C], values * 2048
    out = np.zeros((dh, dw, C), dtype=np.uint8)
    for dy, (r0, r1, b0, b1) in enumerate(yt):
        v = (((b0 * (hrow[r0] >> 4)) >> 16) + ((b1 * (hrow[r1] >> 4)) >> 16) + 2) >> 2
        out[dy] = np.clip(v, 0, 255).astype(np.uint8)
    return out


def crop_resize(frame: np.ndarray, box_ltrb, margin: float, size: int) -> np.ndarray:
    crop = crop_margin(frame, box_ltrb, margin)
    if crop.shape[0] == 0 or crop.shape[1] == 0:
        return np.zeros((size, size, 3), dtype=np.uint8)
    return area_resize(crop, size)
