"""Development check of the split-bf16 mode: cosine gap to the f32 path at full batch, forward time of both."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-insight-face_amd'))
from deep_insight_face.networks.triplet import DifEmbedder  # noqa: E402


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else 'iresnet100'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    f32 = DifEmbedder(arch, 'v2', 512, (112, 112, 3), max_batch=B).init_synthetic(2024)
    b3 = DifEmbedder(arch, 'v2', 512, (112, 112, 3), max_batch=B, compute='bf16x3')
    b3.set_weights(f32.get_weights())
    x = torch.randint(0, 256, (B, 112, 112, 3), dtype=torch.uint8, device='cuda')
    for m in (f32, b3):
        m.set_input_transform(scale=1 / 255.)
    a = f32.embed(x).double()
    b = b3.embed(x).double()
    gap = 1 - (a * b).sum(1) / (a.norm(dim=1) * b.norm(dim=1))
    print('cosine gap bf16x3 vs f32: max %.3e  mean %.3e  finite %s' % (gap.max(), gap.mean(), bool(torch.isfinite(b).all())))
    for name, m in (('f32', f32), ('bf16x3', b3)):
        for _ in range(3):
            m.embed(x)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(5):
            m.embed(x)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / 5
        print('%-7s forward %.2f ms  %.1f TFLOP/s algorithmic  %.0f faces/s' % (name, ms, m.flops_per_image * B / ms / 1e9, B / ms * 1e3))


if __name__ == '__main__':
    main()
