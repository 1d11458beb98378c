"""Development: time of a few 3x3 layers of IResNet-100 in bf16x3 mode (per-op HIP events)."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-insight-face_amd'))
from deep_insight_face.networks.triplet import DifEmbedder  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
compute = sys.argv[2] if len(sys.argv) > 2 else 'bf16x3'
m = DifEmbedder('iresnet100', 'v2', 512, (112, 112, 3), max_batch=B, compute=compute).init_synthetic()
x = torch.randint(0, 256, (B, 112, 112, 3), dtype=torch.uint8, device='cuda')
for _ in range(2):
    m.embed(x)
rows = [m.profile(x) for _ in range(3)]
want = ('layer1_1_conv1', 'layer2_5_conv1', 'layer2_5_conv2', 'layer3_10_conv1', 'layer3_10_conv2', 'layer4_1_conv1', 'layer4_1_conv2')
out = []
for i, (name, kern, macs, ms) in enumerate(rows[0]):
    if name in want:
        best = min(r[i][3] for r in rows)
        out.append('%s %.3fms %.0fTF' % (name.replace('layer', 'L').replace('_conv', 'c'), best, 2 * macs * B / best / 1e9))
tot = min(sum(r_[3] for r_ in r) for r in rows)
print(os.environ.get('DIF_OPTIONS', '-'), '| total %.2f ms |' % tot, ' | '.join(out))
