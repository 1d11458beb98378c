"""One network at one small batch: wall per forward over `reps` back-to-back forwards (run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel durations of exactly these launches).
    python tools/small_batch.py iresnet100 1 [reps] [key=value ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-insight-face_amd'))
from deep_insight_face.networks.triplet import DifEmbedder  # noqa: E402

arch = sys.argv[1]
B = int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
m = DifEmbedder(arch, 'v2', 512, (112, 112, 3), max_batch=B).init_synthetic()
for kv in sys.argv[4:]:
    k, v = kv.split('=')
    m.set_option(k, int(v))
x = torch.randint(0, 256, (B, 112, 112, 3), dtype=torch.uint8, device='cuda')
out = torch.empty((B, 512), dtype=torch.float32, device='cuda')
for _ in range(5):
    m.embed_into(x, out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    m.embed_into(x, out)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / reps * 1e3
print('%s B=%d: %.3f ms/forward wall (%d forwards incl. 5 warm-up), %.0f faces/s' % (arch, B, wall, reps + 5, B / wall * 1e3))
