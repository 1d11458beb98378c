"""Quick timing of the match path (development aid; bench.py is the contract)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-insight-face_amd'))
from deep_insight_face import oneshot  # noqa: E402


def main():
    torch.manual_seed(0)
    for (B, G) in ((8, 100_000), (64, 100_000), (256, 100_000), (512, 125_000), (4096, 125_000), (256, 1_000_000)):
        gal = torch.nn.functional.normalize(torch.randn(G, 512, device='cuda'), dim=1)
        p = torch.nn.functional.normalize(torch.randn(B, 512, device='cuda'), dim=1)
        g = oneshot.Gallery(gal)
        for m in (1,):
            for _ in range(3):
                g.match(p, m)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10
            e0.record()
            for _ in range(n):
                g.match(p, m)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            fl = 2.0 * B * G * 512
            by = (G + B) * 512 * 4
            print('B=%5d G=%8d metric=%d  %.3f ms  %.1f TFLOP/s  %.2f TB/s  %.0f probes/s'
                  % (B, G, m, ms, fl / ms / 1e9, by / ms / 1e9, B / ms * 1e3), flush=True)
        g.close()
        del gal, p


if __name__ == '__main__':
    main()
