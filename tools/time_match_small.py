import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-insight-face_amd'))
from deep_insight_face import oneshot
G = 1_000_000
gal = torch.nn.functional.normalize(torch.randn(G, 512, device='cuda'), dim=1)
g = oneshot.Gallery(gal)
for B in (1, 8, 16, 32, 64):
    p = torch.nn.functional.normalize(torch.randn(B, 512, device='cuda'), dim=1)
    for _ in range(3):
        g.match(p, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.match(p, 1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print('B=%3d G=1M: %.3f ms  %.2f TB/s gallery stream  (%.0f%% of 6.3 TB/s)' % (B, ms, G * 2048 / ms / 1e9, G * 2048 / ms / 1e9 / 6.3 * 100))
