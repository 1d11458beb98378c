"""Does an initialised RCCL process group slow the embedding forward down?  (development aid)

One process, IResNet-100 at batch 256 (two-lane executor) and ResNet-50V2: forward time before
init_process_group('nccl', world 1), after it, after the first collective, and after destroy.
    python tools/rccl_effect.py [lanes]
"""
import os
import socket
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'deep-insight-face_amd'))


def timed(m, x, n=8):
    for _ in range(2):
        m.embed(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        m.embed(x)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    if len(sys.argv) > 1:
        os.environ['DIF_STREAMS'] = sys.argv[1]
    from deep_insight_face.networks.triplet import DifEmbedder
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    nets = []
    for arch, B in (('iresnet100', 256), ('resnet', 256)):
        m = DifEmbedder(arch, 'v2', 512, (112, 112, 3), max_batch=B).init_synthetic()
        m.set_input_transform(1 / 255.)
        x = torch.randint(0, 256, (B, 112, 112, 3), dtype=torch.uint8, device=dev)
        nets.append((arch, m, x))

    def row(tag):
        print('%-34s' % tag + '  '.join('%s %.3f ms' % (a, timed(m, x)) for a, m, x in nets), flush=True)

    row('before init_process_group')
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(s.getsockname()[1])
    s.close()
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    row('after init (eager, device_id)')
    t = torch.ones(1024, device=dev)
    o = torch.empty(1024, device=dev)
    dist.all_gather_into_tensor(o, t)
    torch.cuda.synchronize()
    row('after the first all_gather')
    row('again')
    dist.destroy_process_group()
    row('after destroy_process_group')


if __name__ == '__main__':
    main()
