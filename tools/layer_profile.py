"""Per-launch table of one embedding forward (HIP events around every launch)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-insight-face_amd'))
from deep_insight_face.networks.triplet import DifEmbedder  # noqa: E402


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else 'resnet'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    compute = sys.argv[3] if len(sys.argv) > 3 else 'f32'
    if arch == 'yolov3':
        m = DifEmbedder('yolov3', 'v3', 1, (416, 416, 3), max_batch=B).init_synthetic()
        x = torch.randint(0, 256, (B, 416, 416, 3), dtype=torch.uint8, device='cuda')
    else:
        m = DifEmbedder(arch, 'v2', 512, (112, 112, 3), max_batch=B, compute=compute).init_synthetic()
        x = torch.randint(0, 256, (B, 112, 112, 3), dtype=torch.uint8, device='cuda')
    for _ in range(2):
        m.embed(x)
    rows = m.profile(x)
    rows2 = m.profile(x)
    tot_ms = tot_fl = 0.0
    agg = {}
    print('%-28s %-44s %9s %8s %8s' % ('op', 'kernel', 'GFLOP', 'ms', 'TFLOP/s'))
    for (name, kern, macs, ms), (_, _, _, ms2) in zip(rows, rows2):
        ms = min(ms, ms2)
        fl = 2 * macs * B
        tot_ms += ms
        tot_fl += fl
        a = agg.setdefault(kern, [0.0, 0.0, 0])
        a[0] += ms
        a[1] += fl
        a[2] += 1
        print('%-28s %-44s %9.2f %8.3f %8.1f' % (name, kern, fl / 1e9, ms, fl / ms / 1e9 if ms > 0 else 0))
    print('TOTAL %.2f ms  %.1f TFLOP/s' % (tot_ms, tot_fl / tot_ms / 1e9))
    for k, (ms, fl, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print('  %-44s launches %3d  %8.3f ms (%4.1f%%)  %6.1f TFLOP/s' % (k, n, ms, 100 * ms / tot_ms, fl / ms / 1e9))


if __name__ == '__main__':
    main()
