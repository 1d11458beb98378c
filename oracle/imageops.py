"""CPU restatement of the image resampling around the detector.  TEST INFRASTRUCTURE ONLY (see
oracle/__init__.py).

* ``letterbox``: deep_insight_face/detector/yolov3.py:108-119, executed with Pillow -- the very
  library call the reference makes (``image.resize((nw, nh), Image.BICUBIC)`` + paste), so this
  half is pinned by construction.
* ``crop_margin`` / ``area_resize``: detector/run.py:63-87 (filter_bounding_box) and the resize at
  predictions.py:93,154, ``cv2.resize(image, size, interpolation=Image.BICUBIC)``: PIL's BICUBIC
  constant is 3, which cv2 interprets as INTER_AREA.  OpenCV (opencv-python, unpinned in the
  reference's setup.py) is not installed here, so the area-coverage resampling is restated from
  its published definition (each destination pixel = mean of the source over its footprint, source
  pixels weighted by covered area; uint8 results rounded to nearest).  PARITY UNPINNED for this half.
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


def _coverage(n_in: int, n_out: int) -> np.ndarray:
    """[n_out, n_in] matrix of covered fractions, rows normalised."""
    s = n_in / n_out
    m = np.zeros((n_out, n_in), dtype=np.float64)
    for o in range(n_out):
        a, b = o * s, (o + 1) * s
        for i in range(int(np.floor(a)), min(int(np.ceil(b)), n_in)):
            m[o, i] = max(min(b, i + 1) - max(a, i), 0.0)
        m[o] /= m[o].sum()
    return m


def area_resize(img: np.ndarray, size: int) -> np.ndarray:
    my, mx = _coverage(img.shape[0], size), _coverage(img.shape[1], size)
    out = np.einsum('yi,ijc,xj->yxc', my, img.astype(np.float64), mx)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def crop_resize(frame: np.ndarray, box_ltrb, margin: float, size: int) -> np.ndarray:
    crop = crop_margin(frame, box_ltrb, margin)
    if crop.shape[0] == 0 or crop.shape[1] == 0:
        return np.zeros((size, size, 3), dtype=np.uint8)
    return area_resize(crop, size)
