"""MTCNN cascade on the device (BASELINE configs[4] as worded: "MTCNN P/R/O-Net forward (conv-only) fused ahead of embed").

NOT IN THE REFERENCE: ``config.py:37`` and ``detector/run.py:124`` name MTCNN in comments only; the detector the
reference ships is YOLOv3-face (``detector/run.py``, ``detector/yolov3.py`` here).  What is kept from the reference is
its detector calling convention (``detector/run.py:120-173``): an object called with an image returns
``(cropped_images, boxes)``; ``MtcnnFramePipeline`` is the batched, device-resident form (frames -> best face -> crop
-> embedding -> gallery match), the twin of ``run.FramePipeline``.

The networks restate the public MTCNN definition (Zhang et al. 2016) on the library's convolution kernels (archs
``mtcnn_pnet`` / ``mtcnn_rnet`` / ``mtcnn_onet``, include/dif.h); the cascade runs on STATIC shapes -- ``cap`` slots
per frame and stage, an empty slot has score -1 -- so that a batch of frames needs no host round trip between the
stages.  Deviations from the published code a port of someone else's weights must know: the library's IoU for every
suppression (continuous boxes, union), crops clamped to the frame and resampled by area coverage, sibling heads as one
layer ``head`` = [logits 2 | box 4 | landmarks 10 | zero filters], P-Net's ``conv1`` held as 12 filters (two zero).
"""
import ctypes
import math
import os
import typing

import numpy as np
import torch
from PIL import Image

from .. import _native as N
from ..networks.triplet import DifEmbedder
from ..networks import weights as W
from .run import _to_rgb, filter_bounding_box

STAGES = ('pnet', 'rnet', 'onet')


def pyramid_scales(h: int, w: int, min_face: int = 20, factor: float = 0.709):
    """12 / min_face, then x factor while the shorter side stays >= 12 pixels."""
    m = 12.0 / min_face
    side = min(h, w) * m
    out = []
    while side >= 12:
        out.append(m)
        m *= factor
        side *= factor
    return out


class MtcnnDetector:
    """P-Net over an image pyramid -> R-Net -> O-Net for batches of equally sized uint8 frames.

    ``detect(frames) -> (boxes [n, cap[2], 4] (x1, y1, x2, y2), scores [n, cap[2]])``: CUDA tensors, best slot first,
    score -1 = empty slot."""

    def __init__(self, frame_hw=(480, 640), max_batch: int = 16, min_face: int = 20, thresholds=(0.6, 0.7, 0.7),
                 cap=(64, 32, 16), factor: float = 0.709, streams: int = 4):
        self.h, self.w = int(frame_hw[0]), int(frame_hw[1])
        # the pyramid's scales are independent until their candidates are merged: they run round-robin on this many HIP
        # streams (1 = one after the other on the caller's stream); the small scales' launches do not fill the chip
        self.n_streams = max(1, int(os.environ.get('DIF_MTCNN_STREAMS', streams)))      # (env: A/B runs of bench.py)
        self._side = None
        self.max_batch = int(max_batch)
        self.thresholds = tuple(float(t) for t in thresholds)
        self.cap = tuple(int(c) for c in cap)
        self.scales = pyramid_scales(self.h, self.w, min_face, factor)
        if not self.scales:
            raise ValueError('frames of %dx%d are smaller than one 12-pixel P-Net cell at min_face %d' % (self.h, self.w, min_face))
        self.sizes = [(int(math.ceil(self.h * s)), int(math.ceil(self.w * s))) for s in self.scales]
        norm = dict(scale=1.0 / 128.0, bias=(-127.5 / 128.0,) * 3)
        self.pnets = []
        for hs, ws in self.sizes:
            net = DifEmbedder('mtcnn_pnet', 'v3', 1, (hs, ws, 3), max_batch=self.max_batch, name='mtcnn-pnet-%dx%d' % (hs, ws))
            net.set_input_transform(**norm)
            self.pnets.append(net)
        self.rnet = DifEmbedder('mtcnn_rnet', 'v3', 1, (24, 24, 3), max_batch=self.max_batch * self.cap[1], name='mtcnn-rnet')
        self.onet = DifEmbedder('mtcnn_onet', 'v3', 1, (48, 48, 3), max_batch=self.max_batch * self.cap[2], name='mtcnn-onet')
        self.rnet.set_input_transform(**norm)
        self.onet.set_input_transform(**norm)

    # ---- weights: {'pnet': {...}, 'rnet': {...}, 'onet': {...}}, names and shapes of param_spec() -------------------
    def param_spec(self):
        return {'pnet': self.pnets[0].param_spec(), 'rnet': self.rnet.param_spec(), 'onet': self.onet.param_spec()}

    def set_weights(self, params: typing.Mapping[str, typing.Mapping[str, np.ndarray]]):
        for net in self.pnets:
            net.set_weights(params['pnet'])
        self.rnet.set_weights(params['rnet'])
        self.onet.set_weights(params['onet'])

    def get_weights(self):
        return {'pnet': self.pnets[0].get_weights(), 'rnet': self.rnet.get_weights(), 'onet': self.onet.get_weights()}

    def init_synthetic(self, seed: int = 2024, logit_scale: float = 1.0):
        """Seeded random weights (no MTCNN weights are obtainable offline); the heads' logit biases are shifted so that a
        share of the cells passes every threshold and the later stages have work, their regression filters are scaled
        down so that the boxes stay on the frame.  ``logit_scale`` << 1 (bench.py) flattens the face / not-face logits, so that
        EVERY cell and slot passes and every frame yields a detection whatever its pixels are -- the arithmetic per frame does
        not depend on the values."""
        spec = self.param_spec()
        params = {k: W.synth_params(spec[k], seed + i) for i, k in enumerate(STAGES)}
        for k in STAGES:
            b = params[k]['head/bias']
            params[k]['head/kernel'][..., 0:2] *= np.float32(logit_scale)
            b[0], b[1] = -2.0, 2.0
            params[k]['head/kernel'][..., 2:] *= np.float32(0.02)     # box regression (and landmarks) of a few percent of a
            b[2:] *= np.float32(0.02)                                  # side: random filters would throw the boxes off the frame
        params['pnet']['conv1/kernel'][..., 10:] = 0          # the two padding filters
        params['pnet']['conv1/bias'][10:] = 0
        params['pnet']['head/kernel'][..., 6:] = 0
        params['pnet']['head/bias'][6:] = 0
        self.set_weights(params)
        return self

    def load_weights(self, path):
        z = W.load_npz(path)
        self.set_weights({k: {n[len(k) + 1:]: a for n, a in z.items() if n.startswith(k + '/')} for k in STAGES})

    def save_weights(self, path):
        W.save_npz(path, {k + '/' + n: a for k, p in self.get_weights().items() for n, a in p.items()})

    @property
    def flops_per_image(self):
        f = sum(net.flops_per_image for net in self.pnets)
        return f + self.cap[1] * self.rnet.flops_per_image + self.cap[2] * self.onet.flops_per_image

    def op_table(self):
        """(name, kernel, MACs per frame) of every launch of the three networks for one chunk of frames."""
        rows = []
        for tag, net, mult in [('pnet@%dx%d' % s, n, 1) for s, n in zip(self.sizes, self.pnets)] + \
                [('rnet', self.rnet, self.cap[1]), ('onet', self.onet, self.cap[2])]:
            rows += [('%s/%s' % (tag, name), kern, macs * mult) for name, kern, macs in net.op_table()]
        return rows

    def close(self):
        for net in self.pnets + [self.rnet, self.onet]:
            net.close()

    # ---- the cascade -----------------------------------------------------------------------------------------------
    @staticmethod
    def _nms(boxes, scores, cap, iou):
        n, nb = scores.shape
        dev = scores.device
        keep = torch.empty((n, 1, cap), dtype=torch.int32, device=dev)
        cnt = torch.empty((n, 1), dtype=torch.int32, device=dev)
        ws = torch.empty((n * nb,), dtype=torch.uint8, device=dev)
        N.check(N.lib.dif_nms(N.ptr(boxes), N.ptr(scores), n, nb, 1, cap, 0.0, float(iou), N.ptr(ws), N.ptr(keep), N.ptr(cnt),
                              N.stream_ptr()))
        return keep

    @staticmethod
    def _gather(keep, cap, sboxes, sscores, sreg_ptr, sreg_ld, nsrc, dboxes, dscores, dreg, ndst, off, calibrate):
        n = sscores.shape[0]
        N.check(N.lib.dif_mtcnn_gather(N.ptr(keep), n, cap, N.ptr(sboxes), N.ptr(sscores), sreg_ptr, sreg_ld, nsrc, N.ptr(dboxes),
                                       N.ptr(dscores), N.ptr(dreg) if dreg is not None else None, ndst, off, int(calibrate),
                                       N.stream_ptr()))

    def detect(self, frames, return_stages: bool = False):
        dev = N.require_device()
        t = torch.from_numpy(np.ascontiguousarray(frames)) if not torch.is_tensor(frames) else frames
        if t.dim() != 4 or tuple(t.shape[1:]) != (self.h, self.w, 3) or t.dtype != torch.uint8:
            raise ValueError('expected uint8 frames [n,%d,%d,3], got %s %s' % (self.h, self.w, t.dtype, tuple(t.shape)))
        n = t.shape[0]
        if n > self.max_batch:
            raise ValueError('at most max_batch = %d frames per call' % self.max_batch)
        t = t.to(dev).contiguous()
        c0, c1, c2 = self.cap
        f32 = dict(dtype=torch.float32, device=dev)
        S = len(self.scales)
        mb = torch.empty((n, S * c0, 4), **f32)
        ms = torch.empty((n, S * c0), **f32)
        mr = torch.empty((n, S * c0, 4), **f32)
        main = torch.cuda.current_stream()
        if self.n_streams > 1 and self._side is None:
            self._side = [torch.cuda.Stream(device=dev) for _ in range(self.n_streams - 1)]
        lanes = [main] + (self._side or [])
        ready = torch.cuda.Event()
        ready.record(main)                                         # frames and the merge buffers exist from here on
        for si, (sc, (hs, ws), net) in enumerate(zip(self.scales, self.sizes, self.pnets)):
            lane = lanes[si % len(lanes)]
            with torch.cuda.stream(lane):                          # (every launch below reads the current stream)
                if lane is not main:
                    lane.wait_event(ready)
                st = N.stream_ptr()
                img = torch.empty((n, hs, ws, 3), dtype=torch.uint8, device=dev)
                N.check(N.lib.dif_area_resize(N.ptr(t), n, self.h, self.w, N.ptr(img), hs, ws, st))
                head = net.embed(img)                              # [n, gh, gw, 8]
                gh, gw, ld = head.shape[1], head.shape[2], head.shape[3]
                pb = torch.empty((n, gh * gw, 4), **f32)
                ps = torch.empty((n, gh * gw), **f32)
                N.check(N.lib.dif_mtcnn_propose(N.ptr(head), n, gh, gw, ld, float(sc), self.thresholds[0], N.ptr(pb), N.ptr(ps), st))
                keep = self._nms(pb, ps, c0, 0.5)
                reg_ptr = ctypes.c_void_p(head.data_ptr() + 2 * 4)  # the map's box channels, rows `ld` floats apart
                self._gather(keep, c0, pb, ps, reg_ptr, ld, gh * gw, mb, ms, mr, S * c0, si * c0, 0)
        for lane in lanes[1:]:                                     # the merge waits for every scale
            done = torch.cuda.Event()
            done.record(lane)
            main.wait_event(done)
        st = N.stream_ptr()
        b1 = torch.empty((n, c1, 4), **f32)
        s1 = torch.empty((n, c1), **f32)
        keep = self._nms(mb, ms, c1, 0.7)
        self._gather(keep, c1, mb, ms, N.ptr(mr), 4, S * c0, b1, s1, None, c1, 0, 1)
        stages = {'stage1_boxes': b1.clone(), 'stage1_scores': s1.clone()} if return_stages else None

        def refine(net, size, boxes, scores, k, thr, plain):
            crops = torch.empty((n * k, size, size, 3), dtype=torch.uint8, device=dev)
            N.check(N.lib.dif_crop_resize_multi(N.ptr(t), n, self.h, self.w, N.ptr(boxes), N.ptr(scores), k, 0.0, N.ptr(crops), size, st))
            out = net.embed(crops).reshape(n * k, -1)
            reg = torch.empty((n, k, 4), **f32)
            N.check(N.lib.dif_mtcnn_rescore(N.ptr(out), n * k, out.shape[1], float(thr), N.ptr(scores), N.ptr(reg),
                                            N.ptr(boxes) if plain else None, int(plain), st))
            return reg

        r1 = refine(self.rnet, 24, b1, s1, c1, self.thresholds[1], False)
        b2 = torch.empty((n, c2, 4), **f32)
        s2 = torch.empty((n, c2), **f32)
        keep = self._nms(b1, s1, c2, 0.7)
        self._gather(keep, c2, b1, s1, N.ptr(r1), 4, c1, b2, s2, None, c2, 0, 1)
        if return_stages:
            stages.update(stage2_boxes=b2.clone(), stage2_scores=s2.clone())
        refine(self.onet, 48, b2, s2, c2, self.thresholds[2], True)
        b3 = torch.empty((n, c2, 4), **f32)
        s3 = torch.empty((n, c2), **f32)
        keep = self._nms(b2, s2, c2, 0.7)
        self._gather(keep, c2, b2, s2, None, 4, c2, b3, s3, None, c2, 0, 0)
        return (b3, s3, stages) if return_stages else (b3, s3)


class MtcnnDetection:
    """The reference's detector call (``detector/run.py:120-173``): ``detect(img) -> (cropped_images, boxes)``, raising
    ``ValueError("Bounding box not found")`` when nothing passes.  ``model``: an MtcnnDetector built for the image's size
    (one is created per new size otherwise, with the given weights)."""

    def __init__(self, margin: int = 8, detect_multiple_faces: bool = False, **kwargs) -> None:
        self.margin = margin
        self.detect_multiple_faces = detect_multiple_faces
        self.weights = kwargs.pop('weights', None)
        self.model = kwargs.pop('model', None)
        self.kwargs = kwargs
        if self.model is None and self.weights is None:
            raise ValueError('MtcnnDetection needs model=<MtcnnDetector> or weights=<{pnet, rnet, onet}>')

    def _model_for(self, h, w):
        if self.model is None or (self.model.h, self.model.w) != (h, w):
            weights = self.weights if self.weights is not None else self.model.get_weights()
            self.model = MtcnnDetector((h, w), max_batch=1, **self.kwargs)
            self.model.set_weights(weights)
        return self.model

    def __call__(self, img: np.ndarray):
        assert isinstance(img, np.ndarray), "Invalid image format"
        if img.ndim < 2:
            raise ValueError(f'Unable to align {img.shape}')
        if img.ndim == 2:
            img = _to_rgb(img)
        img = np.ascontiguousarray(img[:, :, 0:3], dtype=np.uint8)
        boxes, scores = self._model_for(img.shape[0], img.shape[1]).detect(img[None])
        boxes, scores = boxes[0].cpu().numpy(), scores[0].cpu().numpy()
        found = [tuple(b) for b, s in zip(boxes, scores) if s >= 0]
        if not found:
            raise ValueError("Bounding box not found")
        if not self.detect_multiple_faces:
            found = found[:1]
        return filter_bounding_box(Image.fromarray(img), found, self.margin, self.detect_multiple_faces)


class MtcnnFramePipeline:
    """Raw frames -> MTCNN -> best face per frame -> crop -> embedding -> top-1 gallery match on the device (BASELINE
    configs[4] as worded; the YOLOv3-face twin is run.FramePipeline)."""

    def __init__(self, detector: MtcnnDetector, embedder, gallery=None, margin: int = 8, distance_metric: int = 1):
        self.detector, self.embedder, self.gallery = detector, embedder, gallery
        self.margin, self.metric = margin, distance_metric
        self.crop_size = embedder.input_shape[0]

    def detect(self, frames: torch.Tensor):
        """-> (boxes [N, 4] left, top, right, bottom of the best face; NaN where nothing passed, scores [N])."""
        mb = self.detector.max_batch
        parts = [self.detector.detect(frames[lo:lo + mb]) for lo in range(0, frames.shape[0], mb)]
        b, s = torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
        best, sc = b[:, 0, :], s[:, 0]
        found = sc >= 0
        return torch.where(found[:, None], best, torch.full_like(best, float('nan'))), torch.where(found, sc, torch.zeros_like(sc))

    def __call__(self, frames):
        from .run import crop_faces
        dev = N.require_device()
        t = torch.from_numpy(np.ascontiguousarray(frames)) if not torch.is_tensor(frames) else frames
        t = t.to(dev).contiguous()
        boxes, scores = self.detect(t)
        emb = self.embedder.embed(crop_faces(t, boxes, self.margin, self.crop_size))
        if self.gallery is None:
            return boxes, scores, emb
        idx, dist = self.gallery.match(emb, self.metric)
        return boxes, scores, emb, idx, dist
