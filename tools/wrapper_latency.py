"""End-to-end latency of the reference's per-image call (predictions.py:152-156: TripletPrediction._embedding on ONE
uint8 crop in host memory -> float32 embedding in host memory), next to the bare device forward (development aid).
    python tools/wrapper_latency.py [arch] [reps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-insight-face_amd'))
from deep_insight_face.networks.triplet import DifEmbedder  # noqa: E402
from deep_insight_face.predictions import TripletPrediction  # noqa: E402

reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
for arch in ([sys.argv[1]] if len(sys.argv) > 1 else ['resnet', 'iresnet100']):
    m = DifEmbedder(arch, 'v2', 512, (112, 112, 3), max_batch=8).init_synthetic()
    wrap = TripletPrediction(m, img_size=(112, 112))
    rng = np.random.default_rng(0)
    imgs = [rng.integers(0, 256, (112, 112, 3), dtype=np.uint8) for _ in range(8)]
    big = rng.integers(0, 256, (250, 250, 3), dtype=np.uint8)         # an LFW-sized image: resized on the device
    for _ in range(10):
        wrap._embedding(imgs[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        e = wrap._embedding(imgs[i & 7])
    dt = (time.perf_counter() - t0) / reps * 1e3
    t0 = time.perf_counter()
    for i in range(reps):
        e = wrap._embedding(big)
    dt_big = (time.perf_counter() - t0) / reps * 1e3
    x = torch.from_numpy(imgs[0][None]).cuda()
    out = torch.empty((1, 512), device='cuda')
    m.set_input_transform(scale=1 / 255.)
    for _ in range(5):
        m.embed_into(x, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        m.embed_into(x, out)
    torch.cuda.synchronize()
    fw = (time.perf_counter() - t0) / reps * 1e3
    print('%-10s _embedding(112x112 crop) %.3f ms per call, _embedding(250x250 image) %.3f ms, bare device forward %.3f ms'
          % (arch, dt, dt_big, fw), flush=True)
    m.close()
