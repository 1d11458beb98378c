"""Development A/B: IResNet-100 forward time, default executor and one lane, under the DIF_OPTIONS of the environment."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-insight-face_amd'))
from deep_insight_face.networks.triplet import DifEmbedder  # noqa: E402
arch = sys.argv[1] if len(sys.argv) > 1 else 'iresnet100'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
m = DifEmbedder(arch, 'v2', 512, (112, 112, 3), max_batch=B).init_synthetic()
m.set_input_transform(scale=1 / 255.)
x = torch.randint(0, 256, (B, 112, 112, 3), dtype=torch.uint8, device='cuda')
ref = m.embed(x)
for _ in range(3):
    m.embed(x)
best = 1e9
for rep in range(3):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5):
        out = m.embed(x)
    ev[1].record()
    torch.cuda.synchronize()
    best = min(best, ev[0].elapsed_time(ev[1]) / 5)
print('%-24s %s B=%d lanes=%s forward %.3f ms  %.1f TF  %.4f of peak  finite=%s' % (
    os.environ.get('DIF_OPTIONS', '-'), arch, B, os.environ.get('DIF_STREAMS', 'default'), best,
    m.flops_per_image * B / best / 1e9, m.flops_per_image * B / best / 1e9 / 157.3, bool(torch.isfinite(out).all())))
