"""One-off stress of the convolution paths (development aid): odd batch sizes and input sizes, the
pipelined kernel against the plain one (DIF_PIPE=0/1) and, for two rows, against the oracle."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-insight-face_amd'))
sys.path.insert(0, ROOT)
from deep_insight_face.networks.triplet import DifEmbedder  # noqa: E402
from oracle import nets  # noqa: E402


def run(arch, head, emd, hw, n, oracle_rows=2):
    rng = np.random.default_rng(n * 7 + hw)
    x = torch.from_numpy(rng.integers(0, 256, (n, hw, hw, 3), dtype=np.uint8)).cuda()
    m = DifEmbedder(arch, head, emd, (hw, hw, 3), max_batch=n).init_synthetic(3)
    m.set_input_transform(scale=1 / 255.)
    os.environ['DIF_PIPE'] = '1'
    a = m.embed(x)
    os.environ['DIF_PIPE'] = '0'
    b = m.embed(x)
    del os.environ['DIF_PIPE']
    al, bl = (a if isinstance(a, list) else [a]), (b if isinstance(b, list) else [b])
    worst = 0.0
    for ta, tb in zip(al, bl):
        worst = max(worst, float((ta - tb).abs().max()) / max(float(tb.abs().max()), 1.0))
    msg = '%-10s %-3s hw=%3d n=%3d  pipe-vs-plain rel %.1e' % (arch, head, hw, n, worst)
    ok = worst < 5e-6 and all(bool(torch.isfinite(t).all()) for t in al)
    if arch != 'yolov3' and oracle_rows:
        rows = [0, n - 1][:oracle_rows]
        xs = x[rows].cpu().numpy().astype(np.float32) / np.float32(255)
        want = nets.embed(xs, m.get_weights(), arch, emd, head)
        got = a[rows].cpu().numpy()
        g = got.reshape(len(rows), -1).astype(np.float64)
        w = want.reshape(len(rows), -1).astype(np.float64)
        gap = 1 - (g * w).sum(1) / (np.linalg.norm(g, axis=1) * np.linalg.norm(w, axis=1))
        msg += '  oracle cosine gap %.1e' % gap.max()
        ok = ok and gap.max() < 1e-5
    print(msg, 'OK' if ok else 'FAIL', flush=True)
    m.close()
    return ok


def main():
    ok = True
    for n in (37, 100, 255, 257):
        ok &= run('resnet', 'v2', 512, 112, n)
    ok &= run('resnet', 'v2', 128, 96, 130)
    ok &= run('resnet', 'v1', 64, 128, 70)
    ok &= run('iresnet50', 'v2', 512, 112, 65)
    ok &= run('iresnet50', 'v2', 256, 96, 33)
    ok &= run('mobilenet', 'v2', 512, 112, 129)
    ok &= run('mobilenet', 'v2', 128, 97 + 15, 64)
    ok &= run('vgg16', 'v2', 512, 112, 48)
    ok &= run('vgg16', 'sv2', 64, 96, 31)
    ok &= run('nn4', 'v2', 128, 96, 200, oracle_rows=0)
    ok &= run('yolov3', 'v3', 1, 416, 3)
    ok &= run('yolov3', 'v3', 1, 320, 5)
    print('ALL OK' if ok else 'SOME FAILED')
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
