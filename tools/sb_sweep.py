"""Small-batch sweep: wall per forward at several batches (development aid; DIF_LIB picks the build).
    python tools/sb_sweep.py iresnet100 8,12,16,32 [key=value ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-insight-face_amd'))
from deep_insight_face.networks.triplet import DifEmbedder  # noqa: E402

arch = sys.argv[1]
out_line = []
for B in [int(b) for b in sys.argv[2].split(',')]:
    m = DifEmbedder(arch, 'v2', 512, (112, 112, 3), max_batch=B).init_synthetic()
    for kv in sys.argv[3:]:
        k, v = kv.split('=')
        m.set_option(k, int(v))
    x = torch.randint(0, 256, (B, 112, 112, 3), dtype=torch.uint8, device='cuda')
    out = torch.empty((B, 512), dtype=torch.float32, device='cuda')
    for _ in range(5):
        m.embed_into(x, out)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(40):
            m.embed_into(x, out)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 40 * 1e3)
    out_line.append('B=%d %.3f' % (B, best))
    m.close()
print(os.environ.get('DIF_LIB', 'libdif.so'), arch, ' '.join(sys.argv[3:]), '|', '  '.join(out_line), 'ms', flush=True)
