"""Experiment: one forward at batch 256 vs two half-batches on two HIP streams."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-insight-face_amd'))
from deep_insight_face.networks.triplet import DifEmbedder  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else 'resnet'
B = 256
whole = DifEmbedder(arch, 'v2', 512, (112, 112, 3), max_batch=B).init_synthetic()
halves = [DifEmbedder(arch, 'v2', 512, (112, 112, 3), max_batch=B // 2).init_synthetic() for _ in range(2)]
x = torch.randint(0, 256, (B, 112, 112, 3), dtype=torch.uint8, device='cuda')
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run_whole():
    whole.embed(x)


def run_split():
    cur = torch.cuda.current_stream()
    for s in streams:
        s.wait_stream(cur)
    for i, s in enumerate(streams):
        with torch.cuda.stream(s):
            halves[i].embed(x[i * B // 2:(i + 1) * B // 2])
    for s in streams:
        cur.wait_stream(s)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for r in range(3):
    a, b = timeit(run_whole), timeit(run_split)
    print('%s B=%d: one stream %.3f ms | two streams x %d %.3f ms' % (arch, B, a, B // 2, b), flush=True)
