"""Small-batch latency of the embedding forward (development aid)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-insight-face_amd'))
from deep_insight_face.networks.triplet import DifEmbedder  # noqa: E402

for arch in ('resnet', 'iresnet100'):
    for B in (1, 8, 12, 16, 32):
        m = DifEmbedder(arch, 'v2', 512, (112, 112, 3), max_batch=B).init_synthetic()
        x = torch.randint(0, 256, (B, 112, 112, 3), dtype=torch.uint8, device='cuda')
        for _ in range(5):
            m.embed(x)
        torch.cuda.synchronize()
        n = 50
        t0 = time.perf_counter()
        for _ in range(n):
            m.embed(x)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / n * 1e3
        rows = m.profile(x)
        gpu = sum(r[3] for r in rows)
        print('%-10s B=%2d  wall %.3f ms/forward  (sum of kernel times %.3f ms, %d launches)  %.0f faces/s'
              % (arch, B, wall, gpu, len(rows), B / wall * 1e3), flush=True)
        m.close()
