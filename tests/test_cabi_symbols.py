"""The C-ABI library loads and exports every symbol include/dif.h declares (no
compute calls: this runs without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'dif.h')
LIB = os.path.join(ROOT, 'deep-insight-face_amd', 'lib', 'libdif.so')


def declared():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(dif_[a-z0-9_]+)\s*\(', src)))


def test_header_declares_the_path():
    names = declared()
    for must in ('dif_pairwise', 'dif_gallery_set', 'dif_match', 'dif_match_merge', 'dif_net_create',
                 'dif_net_embed', 'dif_arcmargin_logits', 'dif_last_error'):
        assert must in names


def test_library_exports_every_declared_symbol():
    if not os.path.exists(LIB):
        import subprocess
        import sys
        subprocess.check_call([sys.executable, os.path.join(ROOT, 'deep-insight-face_amd', 'build.py')])
    lib = ctypes.CDLL(LIB)
    missing = [n for n in declared() if not hasattr(lib, n)]
    assert not missing, missing
    lib.dif_version.restype = ctypes.c_int
    assert lib.dif_version() == 110
    assert re.search(r'#define DIF_VERSION 110\b', open(HEADER).read())


def _option_block(fn):
    """The comment in front of `int <fn>(` in the header."""
    src = open(HEADER).read()
    end = src.index('int %s(' % fn)
    start = src.rindex('/*', 0, end)
    return src[start:end]


@pytest.mark.parametrize('setter,lister', [('dif_net_set_option', 'dif_net_option_name'),
                                           ('dif_gallery_set_option', 'dif_gallery_option_name')])
def test_header_documents_every_option_the_library_accepts(setter, lister):
    """the header is the boundary, and it listed 4 of ~17 keys.  The library reads out its key table
    (dif_*_option_name); every key must be named, in quotes, with a default, in the comment of its setter."""
    lib = ctypes.CDLL(LIB)
    fn = getattr(lib, lister)
    fn.restype = ctypes.c_char_p
    fn.argtypes = [ctypes.c_int]
    keys = []
    while fn(len(keys)) is not None:
        keys.append(fn(len(keys)).decode())
        assert len(keys) < 100
    assert fn(-1) is None and len(keys) >= 4 and len(set(keys)) == len(keys)
    doc = _option_block(setter)
    missing = [k for k in keys if not re.search(r'"%s"[^"]{0,400}?default' % re.escape(k), doc, flags=re.S)]
    assert not missing, 'include/dif.h does not document: %s' % missing


def test_binding_table_matches_header():
    from deep_insight_face import _native
    assert sorted(_native.SIGNATURES) == declared()


def test_no_cpu_fallback():
    """Without a HIP device the product path must fail loudly, not compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    import numpy as np
    from deep_insight_face import _native
    from deep_insight_face.evaluation import utility
    from deep_insight_face import oneshot
    x = np.zeros((2, 32), dtype=np.float32)
    with pytest.raises(_native.DifError):
        utility.distance(x, x, 0)
    with pytest.raises(_native.DifError):
        oneshot.Gallery(x)
    # argument errors keep the reference's exception type even without a device
    with pytest.raises(RuntimeError, match='Undefined distance metric 7'):
        utility.distance(x, x, 7)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'deep-insight-face_amd')
    bad = []
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.cpp', '.h')):
                txt = open(os.path.join(d, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M) or 'oracle/' in txt:
                    bad.append(os.path.join(d, f))
    assert not bad, bad
