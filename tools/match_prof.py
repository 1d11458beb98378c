"""dif_match at one bench shape, repeated, for rocprofv3 --kernel-trace --stats (development aid).
    python tools/match_prof.py [G] [B] [reps]"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-insight-face_amd'))
from deep_insight_face import oneshot  # noqa: E402
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
g = torch.Generator(device='cuda').manual_seed(7)
gal = torch.nn.functional.normalize(torch.randn((G, 512), generator=g, device='cuda'), dim=1)
pick = torch.randperm(G, generator=g, device='cuda')[:B]
probes = torch.nn.functional.normalize(gal[pick] + 0.03 * torch.randn((B, 512), generator=g, device='cuda'), dim=1)
G_ = oneshot.Gallery(gal)
for kv in os.environ.get('MATCH_OPTS', '').split(','):
    if '=' in kv:
        G_.set_option(kv.split('=')[0], int(kv.split('=')[1]))
idx = torch.empty(B, dtype=torch.int64, device='cuda')
dist = torch.empty(B, dtype=torch.float32, device='cuda')
for _ in range(3):
    G_.match_into(probes, 1, idx, dist)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(reps):
    G_.match_into(probes, 1, idx, dist)
ev[1].record()
torch.cuda.synchronize()
print('G=%d B=%d %s %.3f ms per dif_match  ok=%s' % (G, B, os.environ.get('MATCH_OPTS', ''), ev[0].elapsed_time(ev[1]) / reps, bool(torch.equal(idx, pick))))
