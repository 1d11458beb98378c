"""ctypes binding of libdif.so (include/dif.h) plus the torch plumbing around it.

PyTorch-ROCm is used for device memory, streams and torch.distributed only; every
computation of the hot path happens inside libdif.so.  There is no CPU fallback:
importing this module without the built library raises, and calling a compute entry
point without a HIP device raises.
"""
import ctypes
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), 'lib', os.environ.get('DIF_LIB', 'libdif.so'))   # DIF_LIB: same-box A/B of two builds

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libdif.so not found at %s -- build it with `python deep-insight-face_amd/build.py` "
        "(there is no CPU fallback for the embedding/match hot path)" % LIB_PATH)

lib = ctypes.CDLL(LIB_PATH)

c_void_p, c_int, c_int64, c_float, c_double, c_char_p = (
    ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double, ctypes.c_char_p)
P = ctypes.POINTER

# name -> (restype, argtypes); mirrors include/dif.h one to one
SIGNATURES = {
    'dif_version': (c_int, []),
    'dif_last_error': (c_char_p, []),
    'dif_device_count': (c_int, []),
    'dif_probe_mfma_clock': (c_int, [P(c_double), P(c_double), c_void_p]),
    'dif_net_embed_clock': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, P(c_double), c_void_p]),
    'dif_pairwise': (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    'dif_threshold_counts': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'dif_yolo_decode': (c_int, [P(c_void_p), P(ctypes.c_int32), P(c_float), c_int, c_int, c_int, c_int, c_int,
                                c_void_p, c_void_p, c_void_p, c_void_p]),
    'dif_nms': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p,
                        c_void_p, c_void_p]),
    'dif_letterbox': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    'dif_crop_resize': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_float, c_void_p, c_int, c_void_p]),
    'dif_area_resize': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    'dif_crop_resize_multi': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_float, c_void_p, c_int, c_void_p]),
    'dif_mtcnn_propose': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    'dif_mtcnn_gather': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                 c_int, c_int, c_int, c_void_p]),
    'dif_mtcnn_rescore': (c_int, [c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'dif_gallery_create': (c_int, [P(c_void_p), c_int]),
    'dif_gallery_destroy': (c_int, [c_void_p]),
    'dif_gallery_set': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    'dif_gallery_size': (c_int64, [c_void_p]),
    'dif_gallery_update': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    'dif_gallery_reserve': (c_int, [c_void_p, c_int64, c_void_p]),
    'dif_gallery_capacity': (c_int64, [c_void_p]),
    'dif_gallery_set_option': (c_int, [c_void_p, c_char_p, c_int]),
    'dif_gallery_get_stat': (c_int, [c_void_p, c_char_p, P(c_int64), c_void_p]),
    'dif_match': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'dif_match_merge': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'dif_match_merge_packed': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'dif_net_create': (c_int, [P(c_void_p), c_char_p, c_char_p, c_int, c_int, c_int]),
    'dif_net_destroy': (c_int, [c_void_p]),
    'dif_net_param_count': (c_int, [c_void_p]),
    'dif_net_param_info': (c_int, [c_void_p, c_int, P(c_char_p), P(c_int), P(c_int64)]),
    'dif_net_set_param': (c_int, [c_void_p, c_char_p, c_void_p, c_int64]),
    'dif_net_get_param': (c_int, [c_void_p, c_char_p, c_void_p, c_int64]),
    'dif_net_set_input_transform': (c_int, [c_void_p, c_float, P(c_float), c_int]),
    'dif_net_finalize': (c_int, [c_void_p, c_int]),
    'dif_net_set_option': (c_int, [c_void_p, c_char_p, c_int]),
    'dif_net_option_name': (c_char_p, [c_int]),
    'dif_gallery_option_name': (c_char_p, [c_int]),
    'dif_net_output_dim': (c_int, [c_void_p, P(c_int64)]),
    'dif_net_output_count': (c_int, [c_void_p]),
    'dif_net_output_info': (c_int, [c_void_p, c_int, P(c_int64)]),
    'dif_net_embed': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'dif_net_flops_per_image': (c_double, [c_void_p]),
    'dif_net_launch_count': (c_int, [c_void_p]),
    'dif_net_op_info': (c_int, [c_void_p, c_int, P(c_char_p), P(c_char_p), P(c_double)]),
    'dif_net_op_traffic': (c_int, [c_void_p, c_int, P(c_double), P(c_double)]),
    'dif_net_embed_profile': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, P(c_float)]),
    'dif_arcmargin_create': (c_int, [P(c_void_p), c_int, c_int64, c_float, c_float]),
    'dif_arcmargin_destroy': (c_int, [c_void_p]),
    'dif_arcmargin_set_weight': (c_int, [c_void_p, c_void_p, c_void_p]),
    'dif_arcmargin_logits': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here = header and library out of step
    _fn.restype = _res
    _fn.argtypes = _args

METRIC_SQL2, METRIC_COSINE, METRIC_SIMILARITY = 0, 1, 2
LAYOUT_NHWC, LAYOUT_NCHW = 0, 1
DTYPE_F32, DTYPE_U8 = 0, 1


class DifError(RuntimeError):
    pass


def last_error():
    return lib.dif_last_error().decode('utf-8', 'replace')


def check(rc, exc=DifError):
    if rc != 0:
        raise exc(last_error())


def require_device():
    """The hot path is GPU-only: fail loudly instead of computing on the host."""
    if lib.dif_device_count() < 1 or not torch.cuda.is_available():
        raise DifError("no HIP device visible: the embedding/match hot path has no CPU fallback")
    return torch.device('cuda', torch.cuda.current_device())


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def to_device_f32(x, device):
    """numpy / torch (any device) -> contiguous float32 tensor on `device`; reports
    whether the caller handed NumPy (so results go back as NumPy, like the reference)."""
    was_numpy = not torch.is_tensor(x)
    if was_numpy:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32)))
    else:
        t = x
    t = t.to(device=device, dtype=torch.float32).contiguous()
    return t, was_numpy
