"""Face detection front end with the reference's entry points, network + post-processing on the
MI355X.

Mirrors deep_insight_face/detector/run.py:63-173: ``filter_bounding_box`` (margin + clamp + crop),
``get_bounding_box`` (letterbox -> network -> get_yolo_output -> (left, top, right, bottom)) and the
``YoloDetection`` callable.  The network is YOLOv3-face (Darknet-53, three heads, one class) built
inside libdif.so (csrc/net.hip: build_yolov3); the reference loads the same graph as a converted
Keras model (run.py:140).
"""
import typing

import numpy as np
import torch
from PIL import Image

from . import yolov3 as yolo
from .. import _native as N
from ..networks.triplet import DifEmbedder

ANCHORS = np.array([10, 13, 16, 30, 33, 23, 30, 61, 62, 45, 59, 119, 116, 90, 156, 198, 373, 326],
                   dtype=np.float32)      # detector/yolo_cfg/yolo_anchors.txt (run.py:24-31)


def _to_rgb(img: np.ndarray):
    """GRAYSCALE -> RGB (run.py:34-40)."""
    return np.repeat(np.asarray(img, dtype=np.uint8)[:, :, None], 3, axis=2)


def yolo_v3_face(num_classes: int = 1, image_size: int = 416, max_batch: int = 16) -> DifEmbedder:
    """The detector network; ``predict`` / ``predict_on_batch`` return its three output maps."""
    return DifEmbedder('yolov3', 'v3', num_classes, (image_size, image_size, 3), max_batch=max_batch,
                       name='yolov3-face')


def filter_bounding_box(img: Image.Image, bounding_boxes, margin: int = 8, detect_multiple_faces: bool = False):
    """Crop every box with a margin, clamped to the image (run.py:63-87)."""
    assert isinstance(img, Image.Image), "Invalid image type"
    arr = np.array(img)
    h, w = arr.shape[0], arr.shape[1]
    cropped_images, boxes = [], []
    for det in bounding_boxes:
        det = np.squeeze(np.asarray(det))
        bb = np.zeros(4, dtype=np.int32)
        bb[0] = np.maximum(det[0] - margin / 2, 0)
        bb[1] = np.maximum(det[1] - margin / 2, 0)
        bb[2] = np.minimum(det[2] + margin / 2, w)
        bb[3] = np.minimum(det[3] + margin / 2, h)
        cropped_images.append(np.asarray(arr[bb[1]:bb[3], bb[0]:bb[2], :]))
        boxes.append(bb)
    return cropped_images, boxes


def get_bounding_box(infer_model, image: Image.Image, anchors, num_classes: int,
                     target_size: typing.Tuple = (416, 416), score_threshold: float = .6, iou_threshold: float = .5):
    """-> ([(left, top, right, bottom), ...], letterboxed image) (run.py:90-117)."""
    boxed_image = yolo.letterbox_image(image, target_size)
    image_data = np.expand_dims(np.array(boxed_image, dtype='float32') / 255., 0)
    yolo_output = infer_model.predict(image_data)
    image_shape = (image.size[1], image.size[0])
    boxes, scores, classes = yolo.get_yolo_output(yolo_output, np.asarray(anchors, dtype=np.float32).reshape(-1, 2),
                                                  num_classes, image_shape, score_threshold=score_threshold,
                                                  iou_threshold=iou_threshold)
    return [(left, top, right, bottom) for top, left, bottom, right in boxes], boxed_image


class YoloDetection:
    """Detect faces and return (cropped images, boxes) (run.py:120-173).  ``model`` replaces the
    reference's ``model_path`` (a Keras file): any object whose ``predict`` returns the three maps."""

    def __init__(self, margin: int = 8, detect_multiple_faces: bool = False, image_size: int = 416, **kwargs) -> None:
        self.margin = margin
        self.detect_multiple_faces = detect_multiple_faces
        self.image_size = image_size
        self.score = kwargs.pop("score", 0.4)
        self.iou = kwargs.pop("iou", 0.5)
        self.anchors = np.asarray(kwargs.pop("anchors", ANCHORS), dtype=np.float32)
        self.classes = kwargs.pop("classes", ['face'])
        self.model = kwargs.pop("model", None)
        if self.model is None:
            raise ValueError("YoloDetection needs model=<detector network> (see yolo_v3_face())")

    def __call__(self, img: np.ndarray):
        assert isinstance(img, np.ndarray), "Invalid image format"
        if img.ndim < 2:
            raise ValueError(f'Unable to align {img.shape}')
        if img.ndim == 2:
            img = _to_rgb(img)
        img = img[:, :, 0:3]
        image = Image.fromarray(img)
        bounding_boxes, _ = get_bounding_box(self.model, image, self.anchors, len(self.classes),
                                             (self.image_size, self.image_size), self.score, self.iou)
        if len(bounding_boxes) > 0:
            # the reference crops from the letterboxed image although the boxes are in original-image
            # pixels (run.py:163-165, marked TODO there); crop from the original image instead
            return filter_bounding_box(image, bounding_boxes, self.margin, self.detect_multiple_faces)
        raise ValueError("Bounding box not found")


def crop_faces(frames, boxes_ltrb, margin: int = 8, size: int = 112) -> torch.Tensor:
    """``filter_bounding_box`` + the resize to the embedder's input (predictions.py:94,154) for one
    box per frame, on the device: uint8 frames [N,H,W,3], boxes [N,4] (left, top, right, bottom)
    -> uint8 CUDA tensor [N,size,size,3].  A NaN / empty box gives a black crop."""
    dev = N.require_device()
    t = torch.from_numpy(np.ascontiguousarray(frames)) if not torch.is_tensor(frames) else frames
    if t.dim() != 4 or t.shape[3] != 3 or t.dtype != torch.uint8:
        raise ValueError('expected uint8 frames [N,H,W,3], got %s %s' % (t.dtype, tuple(t.shape)))
    t = t.to(dev).contiguous()
    b, _ = N.to_device_f32(boxes_ltrb, dev)
    if tuple(b.shape) != (t.shape[0], 4):
        raise ValueError('expected one (left, top, right, bottom) box per frame')
    out = torch.empty((t.shape[0], size, size, 3), dtype=torch.uint8, device=dev)
    N.check(N.lib.dif_crop_resize(N.ptr(t), t.shape[0], t.shape[1], t.shape[2], N.ptr(b), float(margin), N.ptr(out),
                                  size, N.stream_ptr()))
    return out


class FramePipeline:
    """Raw frames -> best face per frame -> embedding -> top-1 gallery match, everything resident
    on the device (BASELINE configs[4]; the reference chains the same stages per image on the
    host: run.py:149-173 -> predictions.py:150-158 -> oneshot).

    ``detector``: yolo_v3_face(); ``embedder``: a DifEmbedder whose input transform expects uint8
    crops; ``gallery``: oneshot.Gallery or None."""

    def __init__(self, detector, embedder, gallery=None, margin: int = 8, score: float = 0.4, anchors=ANCHORS,
                 num_classes: int = 1, distance_metric: int = 1):
        self.detector, self.embedder, self.gallery = detector, embedder, gallery
        self.margin, self.score, self.num_classes, self.metric = margin, score, num_classes, distance_metric
        self.anchors = np.asarray(anchors, dtype=np.float32).reshape(-1, 2)
        self.det_size = detector.input_shape[0]
        self.crop_size = embedder.input_shape[0]

    def detect(self, frames: torch.Tensor):
        """-> (boxes [N,4] left, top, right, bottom; NaN where nothing passed the threshold,
        scores [N])."""
        n, h, w = frames.shape[0], frames.shape[1], frames.shape[2]
        maps = self.detector.embed(yolo.letterbox_batch(frames, self.det_size))
        boxes, scores = yolo._decode(maps, self.anchors, self.num_classes, (h, w))
        ntot = boxes.shape[1]
        dev = boxes.device
        keep = torch.empty((n, self.num_classes, 1), dtype=torch.int32, device=dev)
        cnt = torch.empty((n, self.num_classes), dtype=torch.int32, device=dev)
        ws = torch.empty((n * self.num_classes * ntot,), dtype=torch.uint8, device=dev)
        # max_boxes = 1: the first greedy pick = highest score, lowest index on ties
        N.check(N.lib.dif_nms(N.ptr(boxes), N.ptr(scores), n, ntot, self.num_classes, 1, float(self.score), 0.5,
                              N.ptr(ws), N.ptr(keep), N.ptr(cnt), N.stream_ptr()))
        found = cnt[:, 0] > 0
        idx = torch.where(found, keep[:, 0, 0], torch.zeros_like(keep[:, 0, 0])).long()
        best = boxes[torch.arange(n, device=dev), idx][:, [1, 0, 3, 2]]
        best = torch.where(found[:, None], best, torch.full_like(best, float('nan')))
        sc = torch.where(found, scores[torch.arange(n, device=dev), idx, 0], torch.zeros((), device=dev))
        return best, sc

    def __call__(self, frames):
        dev = N.require_device()
        t = torch.from_numpy(np.ascontiguousarray(frames)) if not torch.is_tensor(frames) else frames
        t = t.to(dev).contiguous()
        boxes, scores = self.detect(t)
        crops = crop_faces(t, boxes, self.margin, self.crop_size)
        emb = self.embedder.embed(crops)
        if self.gallery is None:
            return boxes, scores, emb
        idx, dist = self.gallery.match(emb, self.metric)
        return boxes, scores, emb, idx, dist
