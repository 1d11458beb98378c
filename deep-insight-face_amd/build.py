"""Builds libdif.so (gfx950) in-tree with hipcc.  No GPU needed: hipcc cross-compiles.

    python deep-insight-face_amd/build.py [--force]

Objects go to deep-insight-face_amd/build/, the library to deep-insight-face_amd/lib/libdif.so
(both git-ignored; the .so travels to the GPU box with the gpurun snapshot).
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'build')
LIB = os.path.join(HERE, 'lib', 'libdif.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = os.environ.get("DIF_EXTRA_FLAGS", "").split() + ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
         '-ffp-contract=off' if os.environ.get('DIF_NO_FMA') else '-ffp-contract=fast']


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def _digest(path):
    h = hashlib.sha1()
    for f in sorted(os.listdir(CSRC)) + [os.path.join('..', '..', 'include', 'dif.h')]:
        if f.endswith(('.hpp', '.h')):
            with open(os.path.join(CSRC, f), 'rb') as fh:
                h.update(fh.read())
    with open(path, 'rb') as fh:
        h.update(fh.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def _file_flags(path):
    """A source may pin compiler flags of its own on a `// hipcc-flags: ...` line (match.hip turns FMA
    contraction off: its re-rank restates the reference's float32 arithmetic operation by operation)."""
    with open(path) as fh:
        for line in fh:
            if line.startswith('// hipcc-flags:'):
                return line.split(':', 1)[1].split()
    return []


def _compile(src):
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ, src + '.o')
    stamp = obj + '.sha1'
    dig = _digest(path)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig and '--force' not in sys.argv:
        return obj, False
    cmd = [HIPCC] + FLAGS + _file_flags(path) + ['-c', path, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    with open(stamp, 'w') as fh:
        fh.write(dig)
    return obj, True


def build(verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    with ThreadPoolExecutor(max_workers=4) as ex:
        res = list(ex.map(_compile, _sources()))
    objs = [o for o, _ in res]
    if any(c for _, c in res) or not os.path.exists(LIB):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
        if verbose:
            print('built', LIB)
    elif verbose:
        print('up to date', LIB)
    return LIB


if __name__ == '__main__':
    build()
