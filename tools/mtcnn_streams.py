"""MTCNN cascade: ms per detect() of 64 frames of 640 x 480 with the pyramid's scales on 1 .. 4 streams (development aid)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-insight-face_amd'))
from deep_insight_face.detector.mtcnn import MtcnnDetector  # noqa: E402

frames = torch.randint(0, 256, (64, 480, 640, 3), dtype=torch.uint8, device='cuda')
for k in (1, 2, 3, 4, 1, 3):
    det = MtcnnDetector((480, 640), max_batch=64, streams=k).init_synthetic(2025, logit_scale=1e-3)
    for _ in range(3):
        det.detect(frames)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        det.detect(frames)
    torch.cuda.synchronize()
    print('streams %d: %.2f ms per 64 frames' % (k, (time.perf_counter() - t0) / 10 * 1e3), flush=True)
    det.close()
