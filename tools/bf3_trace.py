"""Development: per-layer block traces of one IResNet-100 forward (DIF_OPTIONS=dbg=256)."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-insight-face_amd'))
os.environ['DIF_OPTIONS'] = os.environ.get('DIF_OPTIONS', 'dbg=256')
from deep_insight_face.networks.triplet import DifEmbedder  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
compute = sys.argv[2] if len(sys.argv) > 2 else 'bf16x3'
arch = sys.argv[3] if len(sys.argv) > 3 else 'iresnet100'
m = DifEmbedder(arch, 'v2', 512, (112, 112, 3), max_batch=B, compute=compute).init_synthetic()
x = torch.randint(0, 256, (B, 112, 112, 3), dtype=torch.uint8, device='cuda')
for _ in range(3):
    m.embed(x)
print('clock', m.held_clock_ghz(x))
