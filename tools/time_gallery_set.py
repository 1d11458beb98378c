"""Enrolment cost: dif_gallery_set of G x 512 rows and dif_gallery_update of k rows (HIP events; development aid).
    python tools/time_gallery_set.py [G]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-insight-face_amd'))
from deep_insight_face import oneshot, _native as N  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rows = torch.nn.functional.normalize(torch.randn((G, 512), device='cuda'), dim=1)
for flt in (2, 1, 0):
    g = oneshot.Gallery(emd_size=512)
    g.set_option('filter', flt)
    g.set(rows)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    reps = 10
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        N.check(N.lib.dif_gallery_set(g._h, N.ptr(rows), G, 0, N.stream_ptr()))
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / reps
    byts = G * 512 * (4 + 4 + {2: 2, 1: 4, 0: 0}[flt])        # copy in + read + filter copy out
    print('filter %d: dif_gallery_set %d x 512: %.3f ms  (d2d copy + one pass: %.2f TB/s over %.2f GB moved)'
          % (flt, G, ms, byts / ms / 1e9, byts / 1e9))
    for k in (1, 8, 1024):
        ev[0].record()
        for i in range(reps):
            N.check(N.lib.dif_gallery_update(g._h, N.ptr(rows), k, 1000 + i, N.stream_ptr()))
        ev[1].record()
        torch.cuda.synchronize()
        print('          dif_gallery_update of %4d rows: %.4f ms' % (k, ev[0].elapsed_time(ev[1]) / reps))
    g.close()
