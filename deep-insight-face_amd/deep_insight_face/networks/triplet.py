"""Embedding-model builder with the reference's constructor, backed by libdif.so.

Reference: deep_insight_face/networks/triplet.py:60-146 (``bottleneck_network``): pick a
backbone by name, then ``__call__(default_model_ver)`` returns a model object whose
``predict_on_batch(x[N,H,W,3]) -> [N,emd]`` the rest of the code base calls
(predictions.py:96,156; evaluation/evals.py:56).  Here the returned object is a
``DifEmbedder``: same duck type, forward pass = hand-written HIP kernels on the MI355X.
"""
import ctypes
import typing

import numpy as np
import torch

from .. import _native as N
from . import weights as W


class DifEmbedder:
    """Keras-model stand-in: ``predict_on_batch``, ``__call__``, ``predict``,
    ``load_weights``, ``save_weights`` (networks/inceptionv3.py:73-91, api.py:87)."""

    def __init__(self, arch='resnet', head='v2', emd_size=128, input_shape=(112, 112, 3), max_batch=256, name=None,
                 compute='f32'):
        if len(input_shape) != 3 or input_shape[2] != 3:
            raise ValueError('input_shape must be (H, W, 3), got %r' % (input_shape,))
        self.arch, self.head, self.emd_size = arch, head, int(emd_size)
        self.input_shape = tuple(int(s) for s in input_shape)
        self.max_batch = int(max_batch)
        self.name = name or arch
        self._h = ctypes.c_void_p()
        N.check(N.lib.dif_net_create(ctypes.byref(self._h), arch.encode(), head.encode(), self.emd_size,
                                     self.input_shape[0], self.input_shape[1]), ValueError)
        self._ready = False
        if compute not in ('f32', 'bf16x3', 'bf16x2'):
            raise ValueError("compute must be 'f32' (the reference's arithmetic), 'bf16x3' (split-bf16 throughput mode: three "
                             "bf16 terms per operand, six products) or 'bf16x2' (two terms, three products)")
        self.compute = compute
        if compute in ('bf16x3', 'bf16x2'):
            N.check(N.lib.dif_net_set_option(self._h, b'bf16x3', 1), ValueError)
            N.check(N.lib.dif_net_set_option(self._h, b'bf_terms', 3 if compute == 'bf16x3' else 2), ValueError)
        shp = (ctypes.c_int64 * 3)()
        N.check(N.lib.dif_net_output_dim(self._h, shp))
        c, h, w = int(shp[0]), int(shp[1]), int(shp[2])
        self.output_shape = (c,) if (h == 1 and w == 1) else (h, w, c)
        # networks with several outputs (the detector): one (h, w, c) per output
        self.output_shapes = []
        for i in range(N.lib.dif_net_output_count(self._h)):
            N.check(N.lib.dif_net_output_info(self._h, i, shp))
            self.output_shapes.append((int(shp[1]), int(shp[2]), int(shp[0])))

    # ---- parameters -----------------------------------------------------------------
    def param_spec(self):
        out = []
        name, ndim, shape = ctypes.c_char_p(), ctypes.c_int(), (ctypes.c_int64 * 4)()
        for i in range(N.lib.dif_net_param_count(self._h)):
            N.check(N.lib.dif_net_param_info(self._h, i, ctypes.byref(name), ctypes.byref(ndim), shape))
            out.append((name.value.decode(), tuple(int(shape[k]) for k in range(ndim.value))))
        return out

    def count_params(self):
        return int(sum(int(np.prod(s)) for _, s in self.param_spec()))

    def set_weights(self, params: typing.Mapping[str, np.ndarray]):
        spec = dict(self.param_spec())
        missing = sorted(set(spec) - set(params))
        if missing:
            raise ValueError('missing weights: %s%s' % (missing[:5], ' ...' if len(missing) > 5 else ''))
        for name, shape in spec.items():
            a = np.ascontiguousarray(params[name], dtype=np.float32)
            if tuple(a.shape) != shape:
                raise ValueError('weight %s has shape %s, expected %s' % (name, a.shape, shape))
            N.check(N.lib.dif_net_set_param(self._h, name.encode(), a.ctypes.data_as(ctypes.c_void_p), a.size),
                    ValueError)
        self._ready = False

    def get_weights(self):
        out = {}
        for name, shape in self.param_spec():
            a = np.empty(shape, dtype=np.float32)
            N.check(N.lib.dif_net_get_param(self._h, name.encode(), a.ctypes.data_as(ctypes.c_void_p), a.size),
                    ValueError)
            out[name] = a
        return out

    def init_synthetic(self, seed=2024):
        """Seeded random weights (no pretrained weights are obtainable offline)."""
        self.set_weights(W.synth_params(self.param_spec(), seed))
        return self

    def load_weights(self, path):
        """``model.load_weights(path)`` (api.py:87, inceptionv3.py:79-82): a Keras HDF5 weight file (read with h5py
        where it is installed, otherwise by networks/h5lite.py) or the native ``.npz`` (one entry per weight name)."""
        if str(path).endswith(('.h5', '.hdf5')):
            from . import h5lite
            spec = dict(self.param_spec())
            params = {}
            for name, a in h5lite.read_keras_weights(path).items():
                if name in spec and tuple(a.shape) != tuple(spec[name]):
                    # only singleton axes may differ (PReLU alpha saved as (1, 1, C) with shared spatial axes); a
                    # kernel in another layout (OIHW vs HWIO, a transposed Dense) is refused, as Keras refuses it
                    if tuple(d for d in a.shape if d != 1) != tuple(d for d in spec[name] if d != 1):
                        raise ValueError('weight %s has shape %s in %s, expected %s'
                                         % (name, tuple(a.shape), path, tuple(spec[name])))
                    a = a.reshape(spec[name])
                params[name] = a
            self.set_weights(params)
            return
        self.set_weights(W.load_npz(path))

    def save_weights(self, path):
        if str(path).endswith(('.h5', '.hdf5')):
            W.save_keras_h5(path, self.get_weights())           # Keras save_weights layout (h5py, or networks/h5lite.py)
            return
        W.save_npz(path, self.get_weights())

    def set_input_transform(self, scale=1.0, bias=(0.0, 0.0, 0.0), bgr=False, hflip=False):
        """y[c] = x[2-c if bgr else c] * scale + bias[c] (and a left-right mirror with ``hflip``),
        fused into the first kernel."""
        b = (ctypes.c_float * 3)(*[float(v) for v in bias])
        self._transform = (float(scale), tuple(float(v) for v in bias), bool(bgr), bool(hflip))
        N.check(N.lib.dif_net_set_input_transform(self._h, float(scale), b, int(bool(bgr)) | (2 if hflip else 0)))

    def set_option(self, key, value):
        """Execution option of the library (include/dif.h: dif_net_set_option), e.g. ``('pipe', 0)``."""
        N.check(N.lib.dif_net_set_option(self._h, key.encode(), int(value)), ValueError)

    def embed_flipped_concat(self, x, layout=None):
        """[embed(x), embed(mirror(x))] concatenated along the feature axis -> [N, 2*emd]: the
        ``use_flipped_images`` option of the evaluation entry point (scripts/insight_face.py:117-118,
        'Concatenates embeddings for the image and its horizontally flipped counterpart').  The
        mirror happens while the first kernel reads the input; the images are not copied."""
        scale, bias, bgr, hflip = getattr(self, '_transform', (1.0, (0.0, 0.0, 0.0), False, False))
        a = self.embed(x, layout)
        self.set_input_transform(scale, bias, bgr, not hflip)
        try:
            b = self.embed(x, layout)
        finally:
            self.set_input_transform(scale, bias, bgr, hflip)
        return torch.cat([a, b], dim=1)

    def _finalize(self):
        if not self._ready:
            N.require_device()
            N.check(N.lib.dif_net_finalize(self._h, self.max_batch))
            self._ready = True

    @property
    def flops_per_image(self):
        return float(N.lib.dif_net_flops_per_image(self._h))

    # ---- forward ----------------------------------------------------------------------
    def embed(self, x, layout=None):
        """x: torch tensor or ndarray, [N,H,W,3] (NHWC, the reference's layout) or
        [N,3,H,W] (NCHW), float or uint8.  Returns a float32 CUDA tensor [N, *output_shape]."""
        dev = N.require_device()
        self._finalize()
        t = torch.from_numpy(np.ascontiguousarray(x)) if not torch.is_tensor(x) else x
        if t.dim() != 4:
            raise ValueError('expected a 4-D batch, got shape %s' % (tuple(t.shape),))
        H, Wd, _ = self.input_shape
        if layout is None:
            if tuple(t.shape[1:]) == (H, Wd, 3):
                layout = N.LAYOUT_NHWC
            elif tuple(t.shape[1:]) == (3, H, Wd):
                layout = N.LAYOUT_NCHW
            else:
                raise ValueError('input %s matches neither [N,%d,%d,3] nor [N,3,%d,%d]'
                                 % (tuple(t.shape), H, Wd, H, Wd))
        if t.dtype == torch.uint8:
            dtype = N.DTYPE_U8
        else:
            dtype = N.DTYPE_F32
            t = t.to(torch.float32)
        t = t.to(dev).contiguous()
        n = t.shape[0]
        if len(self.output_shapes) > 1:
            return self._embed_multi(t, n, layout, dtype, dev)
        out = torch.empty((n,) + self.output_shape, dtype=torch.float32, device=dev)
        per = int(np.prod(self.output_shape))
        flat = out.view(n, per)
        for s in range(0, n, self.max_batch):
            e = min(n, s + self.max_batch)
            N.check(N.lib.dif_net_embed(self._h, N.ptr(t[s:e]), e - s, layout, dtype, N.ptr(flat[s:e]),
                                        N.stream_ptr()))
        return out

    def embed_into(self, x, out, layout=None):
        """Allocation-free form for a serving loop: `x` a CUDA tensor [n <= max_batch, H, W, 3] or [n, 3, H, W],
        uint8 or float32, contiguous; the embeddings are written into the caller's `out`, a contiguous float32
        CUDA tensor [n, emd].  Nothing is converted or copied -- whatever the library could not read as it is
        raises ValueError."""
        dev = N.require_device()
        self._finalize()
        H, Wd, _ = self.input_shape
        if not torch.is_tensor(x) or x.dim() != 4 or x.device != dev or not x.is_contiguous() \
                or x.dtype not in (torch.uint8, torch.float32):
            raise ValueError('embed_into: x must be a contiguous uint8 / float32 4-D tensor on %s' % dev)
        if layout is None:
            if tuple(x.shape[1:]) == (H, Wd, 3):
                layout = N.LAYOUT_NHWC
            elif tuple(x.shape[1:]) == (3, H, Wd):
                layout = N.LAYOUT_NCHW
            else:
                raise ValueError('input %s matches neither [N,%d,%d,3] nor [N,3,%d,%d]'
                                 % (tuple(x.shape), H, Wd, H, Wd))
        n = x.shape[0]
        per = int(np.prod(self.output_shape))
        if len(self.output_shapes) > 1 or n > self.max_batch:
            raise ValueError('embed_into: one output and at most max_batch = %d images per call' % self.max_batch)
        if not torch.is_tensor(out) or out.dtype != torch.float32 or out.device != dev or not out.is_contiguous() \
                or out.numel() != n * per or out.shape[0] != n:
            raise ValueError('embed_into: out must be a contiguous float32 tensor [%d, %d] on %s' % (n, per, dev))
        if n:
            N.check(N.lib.dif_net_embed(self._h, N.ptr(x), n, layout,
                                        N.DTYPE_U8 if x.dtype == torch.uint8 else N.DTYPE_F32, N.ptr(out),
                                        N.stream_ptr()))
        return out

    def _embed_multi(self, t, n, layout, dtype, dev):
        """Several outputs: the library writes output 0 for the whole chunk, then output 1, ..."""
        sizes = [int(np.prod(s)) for s in self.output_shapes]
        outs = [torch.empty((n,) + s, dtype=torch.float32, device=dev) for s in self.output_shapes]
        for s0 in range(0, n, self.max_batch):
            e = min(n, s0 + self.max_batch)
            c = e - s0
            flat = torch.empty((c * sum(sizes),), dtype=torch.float32, device=dev)
            N.check(N.lib.dif_net_embed(self._h, N.ptr(t[s0:e]), c, layout, dtype, N.ptr(flat), N.stream_ptr()))
            off = 0
            for o, sz, shp in zip(outs, sizes, self.output_shapes):
                o[s0:e] = flat[off:off + c * sz].view((c,) + shp)
                off += c * sz
        return outs

    def op_table(self):
        """The static launch list of one forward: (op name, kernel name, MACs per image) per launch."""
        rows = []
        name, kern, macs = ctypes.c_char_p(), ctypes.c_char_p(), ctypes.c_double()
        for i in range(N.lib.dif_net_launch_count(self._h)):
            N.check(N.lib.dif_net_op_info(self._h, i, ctypes.byref(name), ctypes.byref(kern), ctypes.byref(macs)))
            rows.append((name.value.decode(), kern.value.decode(), macs.value))
        return rows

    def op_traffic(self):
        """Compulsory HBM bytes of every launch: (activation bytes per image, parameter bytes) -- the mixed roofline's
        memory side (include/dif.h: dif_net_op_traffic)."""
        rows = []
        a, p = ctypes.c_double(), ctypes.c_double()
        for i in range(N.lib.dif_net_launch_count(self._h)):
            N.check(N.lib.dif_net_op_traffic(self._h, i, ctypes.byref(a), ctypes.byref(p)))
            rows.append((a.value, p.value))
        return rows

    def profile(self, x):
        """Per-launch milliseconds of one forward of a CUDA uint8/float batch (diagnostic):
        list of (op name, kernel name, MACs per image, ms)."""
        dev = N.require_device()
        self._finalize()
        t = x.to(dev).contiguous()
        n = t.shape[0]
        if n > self.max_batch:
            raise ValueError('profile one batch of at most max_batch images')
        layout = N.LAYOUT_NHWC if tuple(t.shape[1:]) == self.input_shape else N.LAYOUT_NCHW
        dtype = N.DTYPE_U8 if t.dtype == torch.uint8 else N.DTYPE_F32
        out = torch.empty((n * sum(int(np.prod(s)) for s in self.output_shapes),), dtype=torch.float32, device=dev)
        k = N.lib.dif_net_launch_count(self._h)
        ms = (ctypes.c_float * k)()
        N.check(N.lib.dif_net_embed_profile(self._h, N.ptr(t), n, layout, dtype, N.ptr(out), N.stream_ptr(), ms))
        return [(n, kn, m, float(ms[i])) for i, (n, kn, m) in enumerate(self.op_table())]

    def held_clock_ghz(self, x):
        """Shader clock (GHz) held inside the convolution kernels over one single-lane forward of the CUDA batch x
        (diagnostic; include/dif.h: dif_net_embed_clock)."""
        dev = N.require_device()
        self._finalize()
        t = x.to(dev).contiguous()
        n = t.shape[0]
        layout = N.LAYOUT_NHWC if tuple(t.shape[1:]) == self.input_shape else N.LAYOUT_NCHW
        dtype = N.DTYPE_U8 if t.dtype == torch.uint8 else N.DTYPE_F32
        out = torch.empty((n * sum(int(np.prod(s)) for s in self.output_shapes),), dtype=torch.float32, device=dev)
        ghz = ctypes.c_double(0.0)
        N.check(N.lib.dif_net_embed_clock(self._h, N.ptr(t), n, layout, dtype, N.ptr(out), ctypes.byref(ghz), N.stream_ptr()))
        return ghz.value

    def predict_on_batch(self, x):
        """NumPy in -> NumPy float32 out (Keras semantics); torch in -> CUDA tensor out."""
        out = self.embed(x)
        if isinstance(out, list):
            return out if torch.is_tensor(x) else [o.cpu().numpy() for o in out]
        return out if torch.is_tensor(x) else out.cpu().numpy()

    __call__ = predict_on_batch

    def predict(self, x, batch_size=None, **_):
        return self.predict_on_batch(x)

    def close(self):
        if self._h:
            N.lib.dif_net_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class bottleneck_network:
    """Same constructor and call as the reference (networks/triplet.py:73-85).  ``net`` is
    'resnet' (ResNet50V2, the hot path), 'mobilenet' (MobileNetV2) or 'vgg16' as in the reference
    (triplet.py:87-93), or -- extensions named by north_star -- 'iresnet50' / 'iresnet100'."""

    def __init__(self, net: str = "resnet", emd_size: int = 128, input_shape: typing.Tuple = (96, 96, 3), **kwargs):
        # 'inception' is handled by the reference's picker (triplet.py:94-99) although its assert omits it
        assert net in ('mobilenet', 'resnet', 'vgg16', 'inception', 'iresnet50', 'iresnet100'), \
            "Invalid bottleneck network"
        self.net = net
        self.emd_size = emd_size
        self.input_shape = input_shape
        self.kwargs = kwargs

    def __call__(self, default_model_ver='v1', dropout=.2) -> DifEmbedder:
        # dropout is identity at inference (triplet.py:133-134)
        return getattr(self, 'build_models_' + default_model_ver)(dropout=dropout)

    def _build(self, head):
        arch = 'nn4' if self.net == 'inception' else self.net
        return DifEmbedder(arch, head, self.emd_size, self.input_shape,
                           max_batch=self.kwargs.get('max_batch', 256))

    def build_models_v1(self, dropout=0.3):
        return self._build('v1')

    def build_models_v2(self, dropout=0.3):
        return self._build('v2')

    def build_models_v3(self, dropout=0.3):
        return self._build('v3')
