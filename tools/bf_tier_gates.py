"""the two-term split-bf16 tier against the float32 gates, IResNet-100, default executor.

For compute in (bf16x3, bf16x2): forward time at batch 256 / 512, max cosine gap to the float32 embeddings, max |pairwise
arccos distance - float32's| over all pairs of the batch, and -- on a 1 M-row gallery in which 64 of the probes are
enrolled with noise and the rest are impostors -- whether every probe names the row the float32 embeddings name.
    python tools/bf_tier_gates.py [batch ...]
"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-insight-face_amd'))
from deep_insight_face import oneshot  # noqa: E402
from deep_insight_face.networks.triplet import DifEmbedder  # noqa: E402


def timed(m, x, n=6):
    for _ in range(2):
        m.embed(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        m.embed(x)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def pair_dist(e):
    e = e.double()
    e = e / e.norm(dim=1, keepdim=True)
    s = (e @ e.t()).clamp(-1, 1)
    return torch.arccos(s) / math.pi


def main():
    batches = [int(v) for v in sys.argv[1:]] or [256, 512]
    B = max(batches)
    g = torch.Generator(device='cuda').manual_seed(1234)
    x = torch.randint(0, 256, (B, 112, 112, 3), dtype=torch.uint8, device='cuda', generator=g)
    ref = DifEmbedder('iresnet100', 'v2', 512, (112, 112, 3), max_batch=B).init_synthetic(2024)
    ref.set_input_transform(scale=1 / 255.)
    w = ref.get_weights()
    e32 = {b: ref.embed(x[:b]).clone() for b in batches}
    t32 = {b: timed(ref, x[:b]) for b in batches}
    # gallery: 1 M unit rows; 64 probes of the largest batch enrolled (their float32 embedding + noise), the rest impostors
    G = 1_000_000
    gal = torch.nn.functional.normalize(torch.randn((G, 512), device='cuda', generator=g), dim=1)
    rows = torch.randperm(G, device='cuda', generator=g)[:64]
    eb = e32[B]
    gal[rows] = torch.nn.functional.normalize(eb[:64] + 0.02 * torch.randn((64, 512), device='cuda', generator=g), dim=1)
    gallery = oneshot.Gallery(gal)
    i32, d32 = gallery.match(eb, 1)
    print('float32: ' + '  '.join('b%d %.2f ms' % (b, t32[b]) for b in batches) +
          '   enrolled probes found: %d / 64' % int((i32[:64] == rows).sum()), flush=True)
    for compute in ('bf16x3', 'bf16x2'):
        m = DifEmbedder('iresnet100', 'v2', 512, (112, 112, 3), max_batch=B, compute=compute)
        m.set_weights(w)
        m.set_input_transform(scale=1 / 255.)
        for b in batches:
            e = m.embed(x[:b])
            ms = timed(m, x[:b])
            ef, ed = e32[b].double(), e.double()
            gap = float((1 - (ef * ed).sum(1) / (ef.norm(dim=1) * ed.norm(dim=1))).max())
            dd = float((pair_dist(e) - pair_dist(e32[b])).abs().max())
            line = '%s b%d: %.2f ms (%.3fx the float32 forward)  max cosine gap %.2e  max |pairwise distance diff| %.2e' % (
                compute, b, ms, t32[b] / ms, gap, dd)
            if b == B:
                idx, dist = gallery.match(e, 1)
                same = int((idx == i32).sum())
                line += '  1M gallery: %d / %d probes name the float32 row, max |top-1 distance diff| %.2e' % (
                    same, b, float((dist - d32).abs().max()))
            print(line, flush=True)
        m.close()


if __name__ == '__main__':
    main()
