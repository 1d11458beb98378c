"""Small-batch forward three ways: default (null) stream, a side stream, a captured hipGraph replay (development aid).
    python tools/small_batch_modes.py iresnet100 1 [reps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-insight-face_amd'))
from deep_insight_face.networks.triplet import DifEmbedder  # noqa: E402

arch = sys.argv[1]
B = int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
m = DifEmbedder(arch, 'v2', 512, (112, 112, 3), max_batch=B).init_synthetic()
x = torch.randint(0, 256, (B, 112, 112, 3), dtype=torch.uint8, device='cuda')
out = torch.empty((B, 512), dtype=torch.float32, device='cuda')


def timed(fn, label):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print('%-10s B=%d %-28s %.3f ms/forward' % (arch, B, label, (time.perf_counter() - t0) / reps * 1e3), flush=True)


timed(lambda: m.embed_into(x, out), 'null stream')
ref = out.clone()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    timed(lambda: m.embed_into(x, out), 'side stream')
    torch.cuda.synchronize()
    assert torch.equal(ref, out)
    g = torch.cuda.CUDAGraph()
    out.zero_()
    with torch.cuda.graph(g, stream=side):
        m.embed_into(x, out)
    timed(g.replay, 'hipGraph replay')
    torch.cuda.synchronize()
    assert torch.equal(ref, out), float((ref - out).abs().max())
