#!/usr/bin/env python3
"""Register / spill / scratch table of every kernel of a .hip source (hipcc -Rpass-analysis=kernel-resource-usage,
device-only compile for gfx950; no GPU needed).

    python tools/resource_usage.py deep-insight-face_amd/csrc/conv.hip [more.hip ...] > profiles/r03_resource_usage.txt
"""
import os
import re
import subprocess
import sys
import tempfile

HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def demangle(names):
    for tool in ('c++filt', '/opt/rocm/lib/llvm/bin/llvm-cxxfilt', '/opt/rocm/llvm/bin/llvm-cxxfilt'):
        try:
            out = subprocess.run([tool], input='\n'.join(names), capture_output=True, text=True, check=True).stdout
            return out.strip().split('\n')
        except (OSError, subprocess.CalledProcessError):
            continue
    return names


def table(src):
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=fast']
    with open(src) as fh:
        for line in fh:
            if line.startswith('// hipcc-flags:'):
                flags += line.split(':', 1)[1].split()
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run([HIPCC] + flags + ['--cuda-device-only', '-Rpass-analysis=kernel-resource-usage', '-c', src,
                                              '-o', os.path.join(tmp, 'dev.o')], capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit(r.stderr)
    blocks = re.split(r'remark: [^\n]*Function Name: ', r.stderr)[1:]
    rows = []
    for b in blocks:
        def g(k):
            m = re.search(k + r': (\d+)', b)
            return int(m.group(1)) if m else -1
        rows.append([b.split('\n')[0].strip(), g('VGPRs'), g('AGPRs'), g('VGPRs Spill'), g('SGPRs Spill'),
                     g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')])
    names = demangle([r_[0] for r_ in rows])
    for r_, n in zip(rows, names):
        r_[0] = re.sub(r'\bdif::', '', n)
    return rows


def main():
    print('%-110s %5s %5s %7s %7s %8s %4s' % ('kernel', 'VGPR', 'AGPR', 'VGPRsp', 'SGPRsp', 'scratchB', 'occ'))
    for src in sys.argv[1:]:
        print('# ' + src)
        for r in table(src):
            print('%-110s %5d %5d %7d %7d %8d %4d' % tuple(r[:7]))


if __name__ == '__main__':
    main()
