"""dif_match at the reference's own shapes -- one probe or a batch of 8 / 12 against a database of a few thousand encodings
(predictions.py:91-96, BASELINE configs[0]) -- next to configs[1]'s: ms per call by HIP events and by the host clock."""
import os
import sys
import time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-insight-face_amd'))
from deep_insight_face import oneshot  # noqa: E402
for G, B in ((1000, 1), (1000, 8), (6000, 12), (10000, 1), (100000, 1), (100000, 256)):
    g = torch.Generator(device='cuda').manual_seed(7)
    gal = torch.nn.functional.normalize(torch.randn((G, 512), generator=g, device='cuda'), dim=1)
    probes = torch.nn.functional.normalize(torch.randn((B, 512), generator=g, device='cuda'), dim=1)
    G_ = oneshot.Gallery(gal)
    idx = torch.empty(B, dtype=torch.int64, device='cuda')
    dist = torch.empty(B, dtype=torch.float32, device='cuda')
    for _ in range(5):
        G_.match_into(probes, 1, idx, dist)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    t0 = time.perf_counter()
    ev[0].record()
    for _ in range(200):
        G_.match_into(probes, 1, idx, dist)
    ev[1].record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 200 * 1e3
    print('%s  G=%d B=%d  %.4f ms per dif_match (HIP events), %.4f ms host clock; exact_probes %d' %
          (os.environ.get('DIF_LIB', 'libdif.so'), G, B, ev[0].elapsed_time(ev[1]) / 200, wall, G_.stat('exact_probes')), flush=True)
    G_.close()
