"""Quick timing of the embedding forward (development aid; bench.py is the contract)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-insight-face_amd'))
from deep_insight_face.networks.triplet import DifEmbedder  # noqa: E402


def main():
    cfgs = [('resnet', 'v2', 256), ('iresnet100', 'v2', 256)]
    if len(sys.argv) > 1:
        cfgs = [(sys.argv[1], 'v2', int(sys.argv[2]))]
    for arch, head, B in cfgs:
        m = DifEmbedder(arch, head, 512, (112, 112, 3), max_batch=B).init_synthetic()
        x = torch.randint(0, 256, (B, 112, 112, 3), dtype=torch.uint8, device='cuda')
        m.set_input_transform(1 / 255.)
        for _ in range(2):
            m.embed(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n):
            m.embed(x)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        tf = m.flops_per_image * B / ms / 1e9
        print('%s B=%d  %.2f ms  %.0f faces/s  %.1f TFLOP/s (%.1f%% of 157.3)' % (arch, B, ms, B / ms * 1e3, tf,
                                                                              tf / 157.3 * 100), flush=True)
        m.close()


if __name__ == '__main__':
    main()
