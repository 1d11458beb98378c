"""Development A/B: dif_match time at bench shapes, split-bf16 filter vs f32 filter."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-insight-face_amd'))
from deep_insight_face import oneshot  # noqa: E402
for G, B in ((1_000_000, 512), (100_000, 256), (125_000, 4096), (1_000_000, 32)):
    g = torch.Generator(device='cuda').manual_seed(7)
    gal = torch.nn.functional.normalize(torch.randn((G, 512), generator=g, device='cuda'), dim=1)
    pick = torch.randperm(G, generator=g, device='cuda')[:B]
    probes = torch.nn.functional.normalize(gal[pick] + 0.03 * torch.randn((B, 512), generator=g, device='cuda'), dim=1)
    rnd = torch.nn.functional.normalize(torch.randn((B, 512), generator=g, device='cuda'), dim=1)     # no enrolment: impostor tails
    G_ = oneshot.Gallery(gal)
    idx = torch.empty(B, dtype=torch.int64, device='cuda')
    dist = torch.empty(B, dtype=torch.float32, device='cuda')
    res = {}
    flagged = {}
    for flt in (0, 1, 2, 3, 4):                         # f32 filter | bf16x2 on match_tile_kernel | bf16x2 on match_bd_kernel | one-term bf16 on match_b1_kernel | ... on match_g1_kernel
        G_.set_option('filter', (0, 1, 1, 2, 2)[flt])
        G_.set_option('bd', 0 if flt == 1 else 1)
        G_.set_option('frag', 1 if flt == 4 else 0)
        for name, p in (('planted', probes), ('random', rnd)):
            for _ in range(3):
                G_.match_into(p, 1, idx, dist)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(10):
                G_.match_into(p, 1, idx, dist)
            ev[1].record()
            torch.cuda.synchronize()
            res[(flt, name)] = (ev[0].elapsed_time(ev[1]) / 10, idx.clone())
            flagged[(flt, name)] = G_.stat('exact_probes')
    ok = all(torch.equal(res[(0, n)][1], res[(f, n)][1]) for f in (1, 2, 3, 4) for n in ('planted', 'random')) \
        and torch.equal(res[(2, 'planted')][1], pick)
    print('G=%d B=%d  f32 filter %.3f / %.3f ms   bf16x2 tile kernel %.3f / %.3f ms   bf16x2 B-direct kernel %.3f / %.3f ms   '
          'one-term bf16 kernel %.3f / %.3f ms   fragment-order one-term kernel %.3f / %.3f ms (planted / random probes; probes sent to the exact search: %d / %d, %d / %d)  same answers: %s'
          % (G, B, res[(0, 'planted')][0], res[(0, 'random')][0], res[(1, 'planted')][0], res[(1, 'random')][0],
             res[(2, 'planted')][0], res[(2, 'random')][0], res[(3, 'planted')][0], res[(3, 'random')][0],
             res[(4, 'planted')][0], res[(4, 'random')][0],
             flagged[(3, 'planted')], flagged[(3, 'random')], flagged[(4, 'planted')], flagged[(4, 'random')], ok), flush=True)
    G_.close()
    del gal
