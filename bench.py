#!/usr/bin/env python3
"""Benchmark of the embedding + match hot path (BASELINE.json metric: faces/sec).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload r100_1m|r50|r100|r100_arc|frames|frames_mtcnn]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic crops already resident
in HBM: uint8 NHWC crops -> embedding forward (HIP, f32 MFMA) -> [N>1: RCCL all-gather of
the per-rank embeddings] -> top-1 cosine match against the (row-sharded) gallery ->
[N>1: all-gather of the partial results + lowest-index merge].

Default workload = the configuration BASELINE.json's metric is quoted on ("512-d, 1M gallery",
configs[3]'s per-GPU shape): IResNet-100, 512 faces per GPU, a 1M-row gallery (row-sharded for
N > 1).  `--workload r50` is configs[1] (ResNet-50V2+GDC, batch 256, 100k gallery).  Weak
scaling: the per-GPU batch is fixed, the gallery is fixed in total and row-sharded, so both the
embed and the match work per GPU stay constant as N grows.  The same process also times the
IResNet-100 forward at batch 256 (north_star states its >= 70 % MFMA target there) and reports
it as `roofline.b256`.

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (the
convolution kernel family against the f32 MFMA peak) and, at N=1, `cpu_baseline` (the CPU
oracle = a port of the reference's algorithm, timed on this host's cores on a bounded
sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'deep-insight-face_amd'))

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md "Peak FP32 (matrix)"
WORKLOADS = {
    # name: (arch, head, per-GPU batch, gallery rows in total, description)
    'r50': ('resnet', 'v2', 256, 100_000,
            'configs[1]: ResNet-50V2+GDC 512-d embed, batch=256/GPU, 100k gallery cosine match'),
    'r100': ('iresnet100', 'v2', 256, 100_000,
             'north_star target: IResNet-100 512-d embed, batch=256/GPU, 100k gallery cosine match'),
    'r100_arc': ('iresnet100', 'v2', 512, 100_000,
                 'configs[2]: IResNet-100 512-d embed + ArcMargin logits (85742 classes), batch=512/GPU'),
    'r100_1m': ('iresnet100', 'v2', 512, 1_000_000,
                'configs[3]: IResNet-100 embed, batch=512/GPU (4096 on 8 GPUs), 1M gallery row-sharded'),
    'r100_1m_bf16x3': ('iresnet100', 'v2', 512, 1_000_000,
                       'configs[3] shape in the split-bf16 THROUGHPUT mode (not the headline: the headline is float32, the '
                       'reference\'s arithmetic): every f32 operand as three bf16 terms, six bf16 MFMA products, f32 '
                       'accumulation; same 1e-5 cosine parity gate'),
    'r100_1m_bf16x2': ('iresnet100', 'v2', 512, 1_000_000,
                       'configs[3] shape in the two-term split-bf16 THROUGHPUT mode (not the headline): every f32 operand as hi + mid '
                       'in bf16, three bf16 MFMA products, f32 accumulation; passes the float32 gates (cosine gap < 1e-5, pairwise '
                       'distances within 1e-5, same gallery rows)'),
    'frames': ('resnet', 'v2', 256, 100_000,
               'configs[4]: 256 raw 640x480 frames/GPU -> letterbox -> YOLOv3-face -> best box -> crop 112 -> '
               'ResNet-50V2 embed -> 100k gallery match (the reference ships YOLOv3-face, not MTCNN)'),
    'frames_mtcnn': ('resnet', 'v2', 256, 100_000,
                     'configs[4] AS WORDED: 256 raw 640x480 frames/GPU -> MTCNN (P-Net over a 10-scale pyramid, R-Net on 32 and O-Net '
                     'on 16 candidate slots per frame; static shapes, no host round trip) -> best face -> crop 112 -> ResNet-50V2 '
                     'embed -> 100k gallery match.  MTCNN is not in the reference: public layer tables, synthetic weights'),
}
ARC_CLASSES = 85_742     # MS1MV2 identities


GALLERY_BLOCK = 65_536


def synthetic_gallery(lo, hi, d, seed, device):
    """Rows [lo, hi) of the synthetic gallery, generated ON THE DEVICE in blocks of 65 536 rows, block k from
    its own generator seeded `seed + k`: a row's value depends on its global index only, so the gallery is the
    same for every world size and every rank builds its shard alone -- nothing of size G x d exists on the host
    (VERDICT r03 weak #7d: 8 ranks each built the whole 1 M x 512 matrix on the CPU)."""
    out = torch.empty((hi - lo, d), dtype=torch.float32, device=device)
    for k in range(lo // GALLERY_BLOCK, (max(hi, lo + 1) - 1) // GALLERY_BLOCK + 1):
        b_lo, b_hi = k * GALLERY_BLOCK, (k + 1) * GALLERY_BLOCK
        a, b = max(lo, b_lo), min(hi, b_hi)
        if a >= b:
            continue
        g = torch.Generator(device=device).manual_seed(seed + k)
        blk = torch.randn((GALLERY_BLOCK, d), generator=g, dtype=torch.float32, device=device)
        out[a - lo:b - lo] = torch.nn.functional.normalize(blk[a - b_lo:b - b_lo], dim=1)
    return out


def synth_yolo_params(det):
    """He-normal detector weights with the three linear heads tamed (kernels * 1e-5, objectness and
    class logits biased to +2) so that exp() in the box decode stays finite and every frame yields
    a detection -- the arithmetic per frame does not depend on the values."""
    from deep_insight_face.networks.weights import synth_params
    p = synth_params(det.param_spec(), seed=2025)
    heads = [n[:-len('/bias')] for n in p if n.endswith('/bias')]
    for h in heads:
        p[h + '/kernel'] = p[h + '/kernel'] * np.float32(1e-5)
        b = np.zeros_like(p[h + '/bias'])
        b[4::6] = 2.0
        b[5::6] = 2.0
        p[h + '/bias'] = b
    return p


HBM_COPY_TBS = 6.29              # measured copy rate (spec 8.0)
PEAK_BF16_MFMA_TFLOPS = 2500.0   # dense bf16 MFMA peak (MI355X_MICROARCH.md)


def measured_traffic(workload, batch):
    """(HBM bytes per forward of the conv kernels, the committed file they come from) -- rocprofv3 PMC passes
    (FETCH_SIZE doubled + WRITE_SIZE, tools/pmc_traffic.py).  Counters cannot be read from inside the
    process, so this is the committed profile of the same command -- NOT a measurement of this run -- or None."""
    for rnd in ('r05', 'r04', 'r03', 'r02', 'r01'):
        name = '%s_%s_b%d_hbm_traffic.json' % (rnd, workload, batch)
        path = os.path.join(ROOT, 'profiles', name)
        if os.path.exists(path):
            with open(path) as fh:
                return json.load(fh)['total']['conv_hbm_bytes_per_forward'], 'profiles/' + name
    return None, None


def layer_rooflines(prof, traffic, batch):
    """Per launch: algorithmic flops, compulsory HBM bytes (the op's activations once + its parameters once) and
    t_roof = max(flops / f32-MFMA peak, bytes / measured HBM copy rate).  Returns (sum of t_roof in ms, compulsory
    bytes per forward, the launches bound by HBM under that model)."""
    t_sum, b_sum, hbm_bound = 0.0, 0.0, []
    for (name, _, macs, _), (act, par) in zip(prof, traffic):
        fl, by = 2.0 * macs * batch, act * batch + par
        t_m, t_h = fl / (PEAK_F32_MFMA_TFLOPS * 1e12), by / (HBM_COPY_TBS * 1e12)
        t_sum += max(t_m, t_h)
        b_sum += by
        if t_h > t_m and macs > 0:
            hbm_bound.append(name)
    return t_sum * 1e3, b_sum, hbm_bound


def kernel_shares(prof):
    """GPU time of one profiled forward by kernel instantiation: [(kernel, share of the forward, launches)], largest first."""
    tot = sum(ms for _, _, _, ms in prof) or 1.0
    agg = {}
    for _, k, _, ms in prof:
        a = agg.setdefault(k, [0.0, 0])
        a[0] += ms
        a[1] += 1
    return sorted(((k, v[0] / tot, v[1]) for k, v in agg.items()), key=lambda r: -r[1])


def cpu_model_string():
    try:
        with open('/proc/cpuinfo') as fh:
            for line in fh:
                if line.lower().startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def usable_cores():
    """Host threads this process can really run: the affinity mask, cut by the cgroup CPU quota
    (a GPU box hands a job a share of the host, e.g. 16 of 256 hardware threads), then confirmed by
    timing one convolution at the candidate counts -- an oversubscribed torch/OpenMP pool is 10-100x
    slower than a fitting one, which would make the baseline a straw man."""
    import torch.nn.functional as F
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            quota = int(q) / int(p)
    except (OSError, ValueError):
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    x = torch.randn(8, 256, 14, 14)
    w = torch.randn(256, 256, 3, 3)
    best, best_t = n, None
    c = n
    while c >= 4:
        torch.set_num_threads(c)
        F.conv2d(x, w, padding=1)
        t0 = time.perf_counter()
        for _ in range(3):
            F.conv2d(x, w, padding=1)
        t = time.perf_counter() - t0
        if best_t is None or t < 0.9 * best_t:
            best, best_t = c, t
        c //= 2
    return best, quota


def cpu_baseline(arch, head, gallery_rows, budget_s=12.0):
    """CPU stand-in for the reference's TF2/Keras CPU path (which cannot run here: SURVEY 8(c)), as
    SURVEY 8(d) / BASELINE.md section 3 specify it:
      * embedding: the same network on torch-CPU functional ops (oracle/torch_nets.py), identical
        seeded weights, float32, torch.set_num_threads(all host cores);
      * match: the reference's formula probe by probe (evaluation/utility.py:52-66 restated in
        oracle/distance.py -- what a user of the reference runs today) and a BLAS sgemm variant.
    Bounded sample: the batch is sized from a 2-face calibration run to about `budget_s` seconds;
    the NumPy/im2col oracle is timed on a few faces as a second figure."""
    sys.path.insert(0, ROOT)
    from oracle import distance as od
    from oracle import nets, torch_nets
    from deep_insight_face.networks.weights import synth_params
    cores, quota = usable_cores()
    torch.set_num_threads(cores)
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(cores)                              # the NumPy/BLAS legs run on the same threads
    except Exception:
        pass
    p = synth_params(nets.model_spec(arch, 512, 112, head))
    rng = np.random.default_rng(1234)

    def crops(n):
        return rng.integers(0, 256, (n, 112, 112, 3), dtype=np.uint8).astype(np.float32) / np.float32(255)

    torch_nets.embed(crops(2), p, arch, head)                 # warm-up (thread pool, oneDNN primitives)
    t0 = time.perf_counter()
    torch_nets.embed(crops(8), p, arch, head)
    per_face = (time.perf_counter() - t0) / 8
    sample = int(min(256, max(16, budget_s / per_face))) // 8 * 8
    x = crops(sample)
    t0 = time.perf_counter()
    e = torch_nets.embed(x, p, arch, head)
    t_embed = time.perf_counter() - t0

    x_np = x[:4]
    nets.embed(x_np[:1], p, arch, 512, head)
    t0 = time.perf_counter()
    nets.embed(x_np, p, arch, 512, head)
    t_numpy = time.perf_counter() - t0

    gal = rng.standard_normal((gallery_rows, 512)).astype(np.float32)
    gal /= np.linalg.norm(gal, axis=1, keepdims=True)
    probes = int(min(sample, max(4, 8 * 1_000_000 // max(gallery_rows, 1))))
    t0 = time.perf_counter()
    od.match(e[:probes], gal, 1)
    t_match = time.perf_counter() - t0
    t0 = time.perf_counter()
    od.match_blas(e, gal, 1)                                  # the "fair" CPU variant: one sgemm + argmax
    t_blas = time.perf_counter() - t0
    embed_rate, match_rate, blas_rate = sample / t_embed, probes / t_match, sample / t_blas
    return {
        'value': 1.0 / (1.0 / embed_rate + 1.0 / match_rate), 'unit': 'faces/s', 'cores': cores, 'kind': 'port',
        'cpu_model': cpu_model_string(), 'host_hw_threads': os.cpu_count(), 'cgroup_cpu_quota': quota,
        'sample': '%d faces through torch-CPU ops on %d threads (%.2fs) + reference-formula match of %d probes vs '
                  '%d rows (%.2fs); value = 1/(1/embed_rate + 1/match_rate)' % (sample, cores, t_embed, probes,
                                                                               gallery_rows, t_match),
        'embed_torch_cpu_faces_per_s': embed_rate,
        'embed_numpy_oracle_faces_per_s': 4 / t_numpy,
        'match_reference_formula_probes_per_s': match_rate,
        'match_blas_sgemm_probes_per_s': blas_rate,
        'value_with_blas_match': 1.0 / (1.0 / embed_rate + 1.0 / blas_rate),
    }


def cpu_baseline_frames_mtcnn(gallery_rows, det_params, n_frames=2):
    """configs[4] as worded on the host cores: the NumPy restatement of the cascade (oracle/mtcnn.py: area-resampled pyramid,
    P/R/O-Net as im2col matmuls, float32 NMS) -> area-resampled crop -> ResNet-50V2 on torch-CPU ops -> reference-formula
    match.  Two frames: the restatement's suppression runs in Python."""
    sys.path.insert(0, ROOT)
    from oracle import distance as od
    from oracle import imageops as oi
    from oracle import mtcnn as om
    from oracle import nets, torch_nets
    from deep_insight_face.networks.weights import synth_params
    cores, quota = usable_cores()
    torch.set_num_threads(cores)
    p = synth_params(nets.model_spec('resnet', 512, 112, 'v2'))
    rng = np.random.default_rng(1234)
    frames = rng.integers(0, 256, (n_frames, 480, 640, 3), dtype=np.uint8)
    gal = rng.standard_normal((gallery_rows, 512)).astype(np.float32)
    gal /= np.linalg.norm(gal, axis=1, keepdims=True)

    def run(fr):
        boxes, scores, _ = om.detect(fr, det_params)
        crops = [oi.crop_resize(f, b[0] if s[0] >= 0 else [0, 0, 640, 480], 8, 112) for f, b, s in zip(fr, boxes, scores)]
        e = torch_nets.embed(np.stack(crops).astype(np.float32) / np.float32(255), p, 'resnet', 'v2')
        od.match(e, gal, 1)

    t0 = time.perf_counter()
    run(frames)
    dt = time.perf_counter() - t0
    return {'value': n_frames / dt, 'unit': 'frames/s', 'cores': cores, 'kind': 'port', 'cpu_model': cpu_model_string(),
            'host_hw_threads': os.cpu_count(), 'cgroup_cpu_quota': quota,
            'sample': '%d frames 640x480: NumPy restatement of the MTCNN cascade + ResNet-50V2 on torch-CPU ops (%d threads) + '
                      'reference-formula match vs %d rows, %.2fs' % (n_frames, cores, gallery_rows, dt)}


def cpu_baseline_frames(gallery_rows, det_params, n_frames=4):
    """configs[4] on the host cores: PIL letterbox (the reference's own call, detector/yolov3.py:108-119) ->
    YOLOv3-face on torch-CPU ops -> box decode + NMS (oracle/detector.py) -> area-resampled crop (oracle/
    imageops.py) -> ResNet-50V2 on torch-CPU ops -> reference-formula match.  A handful of frames: the detector
    alone is 65 GFLOP per frame."""
    sys.path.insert(0, ROOT)
    from oracle import detector as odet
    from oracle import distance as od
    from oracle import imageops as oi
    from oracle import nets, torch_nets
    from deep_insight_face.networks.weights import synth_params
    cores, quota = usable_cores()
    torch.set_num_threads(cores)
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(cores)
    except Exception:
        pass
    p = synth_params(nets.model_spec('resnet', 512, 112, 'v2'))
    rng = np.random.default_rng(1234)
    frames = rng.integers(0, 256, (n_frames, 480, 640, 3), dtype=np.uint8)
    gal = rng.standard_normal((gallery_rows, 512)).astype(np.float32)
    gal /= np.linalg.norm(gal, axis=1, keepdims=True)
    anchors = np.array([10, 13, 16, 30, 33, 23, 30, 61, 62, 45, 59, 119, 116, 90, 156, 198, 373, 326], dtype=np.float32).reshape(-1, 2)

    def run(fr):
        lb = np.stack([oi.letterbox(f, 416) for f in fr]).astype(np.float32) / np.float32(255)
        maps = torch_nets.yolov3(lb, det_params)
        crops = []
        for i, f in enumerate(fr):
            b, sc, _ = odet.get_yolo_output([m[i:i + 1] for m in maps], anchors, 1, (480, 640), 1, 0.4, 0.5)
            box = [b[0][1], b[0][0], b[0][3], b[0][2]] if len(b) else [0, 0, 640, 480]
            crops.append(oi.crop_resize(f, box, 8, 112))
        e = torch_nets.embed(np.stack(crops).astype(np.float32) / np.float32(255), p, 'resnet', 'v2')
        od.match(e, gal, 1)

    run(frames[:1])
    t0 = time.perf_counter()
    run(frames)
    dt = time.perf_counter() - t0
    return {'value': n_frames / dt, 'unit': 'frames/s', 'cores': cores, 'kind': 'port', 'cpu_model': cpu_model_string(),
            'host_hw_threads': os.cpu_count(), 'cgroup_cpu_quota': quota,
            'sample': '%d frames 640x480: PIL letterbox + YOLOv3-face and ResNet-50V2 on torch-CPU ops (%d threads) + oracle '
                      'decode/NMS/crop + reference-formula match vs %d rows, %.2fs' % (n_frames, cores, gallery_rows, dt)}


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: this process becomes the parent of one
    torch.distributed.run job (one child per GPU) and only relays its output and exit code.  It has made
    no HIP call (importing torch does not initialise the GPU), and nothing that has is ever re-exec'd: the
    ranks are fresh child processes."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC: what RCCL needs on this driver
    env.setdefault('OMP_NUM_THREADS', '4')
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='r100_1m', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=0, help='per-GPU batch override')
    ap.add_argument('--gallery', type=int, default=0, help='total gallery rows override')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-throughput-mode', action='store_true', help='skip the extra split-bf16 forward timing')
    ap.add_argument('--no-latency', action='store_true', help='skip the small-batch (1 / 8 / 12 / 32) forward latency block')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='gloo = rehearsal of the N>1 path on a box with fewer GPUs than ranks (collectives staged '
                         'through the host, ranks share devices); never used for reported numbers')
    ap.add_argument('--force-collectives', action='store_true',
                    help='N=1 only: initialise the process group (RCCL, one rank) and run both all-gathers and the packed '
                         'merge although a world of one needs none -- a rehearsal of the N>1 step on one GPU, not the headline')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))        # before any GPU call in this process
    if world != args.gpus:
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit('bench.py needs a HIP device: the hot path has no CPU fallback')
    if args.backend == 'gloo':
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    grouped = world > 1 or args.force_collectives
    json_fd = None
    if grouped:
        # librccl prints a version banner on STDOUT when the communicator is created; the contract is ONE JSON line there.
        # Everything this process writes to fd 1 goes to stderr from here on, the JSON line to the real stdout at the end.
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if world == 1:
            import socket
            s_ = socket.socket()
            s_.bind(('127.0.0.1', 0))
            os.environ.setdefault('MASTER_PORT', str(s_.getsockname()[1]))
            s_.close()
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world)
            _orig_ag, _orig_ar = dist.all_gather_into_tensor, dist.all_reduce

            def _ag(out, inp, group=None):
                o = torch.empty(out.shape, dtype=out.dtype)
                _orig_ag(o, inp.cpu(), group=group)
                out.copy_(o)

            def _ar(t, op=dist.ReduceOp.SUM, group=None):
                c = t.cpu()
                _orig_ar(c, op=op, group=group)
                t.copy_(c)

            dist.all_gather_into_tensor, dist.all_reduce = _ag, _ar

    from deep_insight_face.networks.triplet import DifEmbedder
    from deep_insight_face.parallel import ShardedGallery, shard_bounds

    arch, head, batch, gallery_rows, desc = WORKLOADS[args.workload]
    compute = 'bf16x3' if args.workload.endswith('_bf16x3') else ('bf16x2' if args.workload.endswith('_bf16x2') else 'f32')
    batch = args.batch or batch
    gallery_rows = args.gallery or gallery_rows

    model = DifEmbedder(arch, head, 512, (112, 112, 3), max_batch=batch, compute=compute).init_synthetic(2024)
    model.set_input_transform(scale=1 / 255.)                 # predictions.py:154 `* rescale`, fused
    # every embedding forward this process runs, by batch size (tools/check_profiles.py divides profiler totals by it)
    forwards = {}
    _embed, _embed_into = model.embed, model.embed_into

    def _counted_embed(x, *a, **k):
        forwards[int(x.shape[0])] = forwards.get(int(x.shape[0]), 0) + 1
        return _embed(x, *a, **k)

    def _counted_embed_into(x, *a, **k):
        forwards[int(x.shape[0])] = forwards.get(int(x.shape[0]), 0) + 1
        return _embed_into(x, *a, **k)
    model.embed, model.embed_into = _counted_embed, _counted_embed_into
    lo, hi = shard_bounds(gallery_rows, world, rank)
    shard_rows = synthetic_gallery(lo, hi, 512, 7, dev)       # kept: the self-check after the timed region plants rows
    # check_batch='first': the ranks' probe counts are compared when a step shape is first met (the warm-up), not inside
    # the timed loop -- the per-step check reads a value back to the host (parallel.py)
    shard = ShardedGallery(shard_rows, lo, force_collectives=args.force_collectives, check_batch='first')
    emb_buf = torch.empty((batch, 512), dtype=torch.float32, device=dev)   # the serving loop allocates nothing per step
    arc = None
    if args.workload == 'r100_arc':
        from deep_insight_face.networks.arcmargin import ArcMarginHead
        gw = torch.Generator(device='cpu').manual_seed(99)
        arc = ArcMarginHead(torch.randn((ARC_CLASSES, 512), generator=gw).to(dev))
        arc_labels = torch.randint(0, ARC_CLASSES, (batch,), generator=gw).to(dev)
    g = torch.Generator(device='cpu').manual_seed(1234 + rank)
    crops = torch.randint(0, 256, (batch, 112, 112, 3), generator=g, dtype=torch.uint8).to(dev)

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    pipe = None
    mtcnn = args.workload == 'frames_mtcnn'
    if mtcnn:
        from deep_insight_face.detector import run as drun
        from deep_insight_face.detector import mtcnn as dm
        det = dm.MtcnnDetector((480, 640), max_batch=64).init_synthetic(2025, logit_scale=1e-3)   # every frame yields a detection
        dp = None
        pipe = dm.MtcnnFramePipeline(det, model, None, margin=8)
        frames = torch.randint(0, 256, (batch, 480, 640, 3), generator=g, dtype=torch.uint8).to(dev)
        det_ms = []
    if args.workload == 'frames':
        from deep_insight_face.detector import run as drun
        det = drun.yolo_v3_face(1, 416, max_batch=64)
        dp = synth_yolo_params(det)
        det.set_weights(dp)
        det.set_input_transform(scale=1 / 255.)               # run.py:98 `/ 255.`, fused
        pipe = drun.FramePipeline(det, model, None, margin=8, score=0.4)
        frames = torch.randint(0, 256, (batch, 480, 640, 3), generator=g, dtype=torch.uint8).to(dev)
        det_ms = []

    def step_frames(i=None):
        if i is not None:
            ev[i][0].record()
        boxes, _ = pipe.detect(frames)
        faces = drun.crop_faces(frames, boxes, 8, 112)
        if i is not None:
            e_det = torch.cuda.Event(enable_timing=True)
            e_det.record()
            det_ms.append(e_det)
        emb = model.embed_into(faces, emb_buf)
        if i is not None:
            ev[i][1].record()
        idx, d = shard.match(emb, 1, copy=False)
        if i is not None:
            ev[i][2].record()
        return idx, d

    def step(i=None):
        if pipe is not None:
            return step_frames(i)
        if i is not None:
            ev[i][0].record()
        emb = model.embed_into(crops, emb_buf)
        if i is not None:
            ev[i][1].record()
        if arc is not None:
            arc.logits(emb, arc_labels)
        idx, d = shard.match(emb, 1, copy=False)
        if i is not None:
            ev[i][2].record()
        return idx, d

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    shard.phase_events = []                                   # five HIP events per timed step around the collectives (parallel.py)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        idx, d = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    step_events, shard.phase_events = shard.phase_events, None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Self-check, after the timed region and never inside it: enrol 8 of this step's probes at known rows
    # spread over the whole (sharded) gallery, run one more step through the same code path and require every
    # rank to name exactly those rows.  A throughput line whose step computed something else fails here.
    B_all = world * batch
    emb_all = shard.all_gather_embeddings(emb_buf).clone()
    n_plant = min(8, B_all, gallery_rows)                     # distinct probes at distinct rows
    probe_ids = sorted({int(v) for v in np.linspace(0, B_all - 1, n_plant).round()})
    plant_rows = sorted({int(v) for v in np.linspace(0, gallery_rows - 1, len(probe_ids)).round()})
    probe_ids = probe_ids[:len(plant_rows)]
    for pid, row in zip(probe_ids, plant_rows):
        if lo <= row < hi:
            shard.gallery.update(emb_all[pid:pid + 1], row - lo)   # O(1) per enrolment (dif_gallery_update), not a whole-gallery pass
    idx_v, d_v = step()
    torch.cuda.synchronize()
    got = [int(idx_v[pid]) for pid in probe_ids]
    dv = d_v[probe_ids].cpu().numpy()
    verified = got == plant_rows and bool(np.all(np.isnan(dv) | (dv < 2e-3)))   # NaN: similarity rounded above 1, as in the reference
    if not verified:
        raise SystemExit('bench.py self-check FAILED on rank %d: planted rows %s, matched %s (dist %s)'
                         % (rank, plant_rows, got, dv.tolist()))

    embed_ms = float(np.mean([ev[i][0].elapsed_time(ev[i][1]) for i in range(args.steps)]))
    match_ms = float(np.mean([ev[i][1].elapsed_time(ev[i][2]) for i in range(args.steps)]))
    # the match phase taken apart (rank 0's view): [start, embeddings gathered, local match done, records gathered, merged]
    sub = {}
    if step_events:
        for name, a_, b_ in (('allgather_embed', 0, 1), ('match_local', 1, 2), ('allgather_packed', 2, 3), ('merge', 3, 4)):
            sub[name] = float(np.mean([e[a_].elapsed_time(e[b_]) for e in step_events]))
    # every rank's embed / match phase, so that a non-linear point of the scaling curve can be placed (slowest rank, or the exchange)
    per_rank = None
    if world > 1:
        mine = torch.tensor([embed_ms, match_ms] + [sub.get(k, 0.0) for k in ('allgather_embed', 'match_local', 'allgather_packed', 'merge')],
                            dtype=torch.float64, device=dev)
        allr = torch.empty((world * mine.numel(),), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allr, mine)
        allr = allr.view(world, -1).cpu().numpy()
        per_rank = {k: {'min': float(allr[:, i].min()), 'max': float(allr[:, i].max()), 'argmax_rank': int(allr[:, i].argmax())}
                    for i, k in enumerate(('embed', 'match', 'allgather_embed', 'match_local', 'allgather_packed', 'merge'))}

    # the reference's own call shapes (predictions.py:152-156 embeds ONE image per call; scripts/insight_face.py:112 evaluates at
    # batch 12): forward latency at small batches, same model and weights, HIP events over back-to-back forwards
    latency = None
    if pipe is None and compute == 'f32' and not args.no_latency:
        latency = {}
        for lb in (1, 8, 12, 32):                               # 12: scripts/insight_face.py:112, the reference's evaluation batch
            if lb > batch:
                continue
            xs = crops[:lb]
            ob = emb_buf[:lb]
            for _ in range(5):
                model.embed_into(xs, ob)
            reps = 30
            le = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            torch.cuda.synchronize()
            tw = time.perf_counter()
            le[0].record()
            for _ in range(reps):
                model.embed_into(xs, ob)
            le[1].record()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - tw) / reps * 1e3
            ms = le[0].elapsed_time(le[1]) / reps
            tf = model.flops_per_image * lb / (ms * 1e-3) / 1e12
            latency['b%d' % lb] = {'ms': ms, 'wall_ms': wall, 'faces_per_s': lb / (wall * 1e-3), 'tflops': tf,
                                   'frac': tf / PEAK_F32_MFMA_TFLOPS}
        latency['note'] = ('forward only, f32; ms = HIP events over 30 back-to-back forwards on the launch stream, wall_ms = host clock '
                           'around the same loop; frac = algorithmic TFLOP/s over the f32-MFMA peak')

    # north_star states its MFMA target "on the ResNet-100 embedding forward at batch 256": time that forward
    # too (same model, same lanes policy, HIP events on the launch stream), outside the step loop
    b256_ms = None
    if pipe is None and arch.startswith('iresnet') and batch > 256:
        reps = max(5, min(args.steps, 20))
        e256 = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(reps)]
        c256 = crops[:256]
        model.embed(c256)
        for r in range(reps):
            e256[r][0].record()
            model.embed(c256)
            e256[r][1].record()
        torch.cuda.synchronize()
        b256_ms = float(np.mean([a.elapsed_time(b) for a, b in e256]))

    if rank == 0:
        # the shader clock THIS box holds under back-to-back f32 MFMAs (boxes of the pool differ by a few percent)
        import ctypes
        from deep_insight_face import _native as N
        ghz, probe_tf = ctypes.c_double(0.0), ctypes.c_double(0.0)
        N.check(N.lib.dif_probe_mfma_clock(ctypes.byref(ghz), ctypes.byref(probe_tf), torch.cuda.current_stream().cuda_stream))
        conv_ghz = None
        if pipe is None:
            for _ in range(max(3, min(args.steps, 10))):      # the governor settles over tens of ms: measure under the steps' load
                model.embed(crops)
            conv_ghz = model.held_clock_ghz(crops)            # inside the conv kernels of one (single-lane) forward
            forwards[batch] = forwards.get(batch, 0) + 1
        flops_embed = model.flops_per_image * batch            # algorithmic: 2 * MACs of every conv/dense
        if pipe is not None:
            flops_embed += det.flops_per_image * batch
        achieved = flops_embed / (embed_ms * 1e-3) / 1e12
        prof = model.profile(crops)
        forwards[batch] = forwards.get(batch, 0) + 1
        det_ops = det.op_table() if pipe is not None else []
        is_conv = ('conv_', 'stem')                                # every convolution kernel family (conv.hip, stem.hip)
        conv_ms = sum(ms for _, k, _, ms in prof if k.startswith(is_conv))
        conv_flops = sum(2 * macs * batch for _, k, macs, _ in prof if k.startswith(is_conv))
        shares = kernel_shares(prof)
        t_mixed_ms, compulsory, hbm_bound = layer_rooflines(prof, model.op_traffic(), batch)
        traffic, traffic_src = measured_traffic(args.workload, batch)
        filt = shard.gallery.stat('filter_terms')                # 1: one bf16 term per operand, 2: two terms, 0: f32 rows
        # match roofline: max(MFMA time of the filter at ITS ceiling, the filter's gallery copy + probes streamed once)
        m_flops = 2.0 * world * batch * (hi - lo) * 512
        m_ceiling = {1: PEAK_BF16_MFMA_TFLOPS, 2: PEAK_BF16_MFMA_TFLOPS / 3.0, 0: PEAK_F32_MFMA_TFLOPS}[filt]   # bf16 MFMAs per product: 1 / 3
        m_bytes = (hi - lo) * (1024.0 if filt == 1 else 2048.0) + world * batch * 2048.0
        m_roof_ms = max(m_flops / (m_ceiling * 1e12), m_bytes / (HBM_COPY_TBS * 1e12)) * 1e3
        e_roof_ms = flops_embed / (PEAK_F32_MFMA_TFLOPS * 1e12) * 1e3
        out = {
            'metric': 'faces/sec embedding+match (112x112, 512-d)' if pipe is None else
                      'frames/sec detect+crop+embed+match (640x480 frames, one face per frame)',
            'value': world * batch * args.steps / elapsed,
            'unit': 'faces/s' if pipe is None else 'frames/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32' if compute == 'f32' else ('bf16x3 (f32 operands split into 3 bf16 terms, 6 MFMA products, f32 accumulate)'
                                                     if compute == 'bf16x3' else
                                                     'bf16x2 (f32 operands split into 2 bf16 terms, 3 MFMA products, f32 accumulate)'),
            'data': 'synthetic (uint8 crops seed 1234, He-normal weights seed 2024, unit-norm gallery seed 7)',
            'verified': verified,       # after the timed region: 8 planted enrolments found at their rows by one more step
            # what carried the collectives: under backend "nccl" (= RCCL on ROCm) rccl_ranks is the size of the communicator
            'backend': (dist.get_backend() if grouped else None),
            'rccl_ranks': (dist.get_world_size() if grouped and dist.get_backend() == 'nccl' else None),
            'collectives_per_step': (2 if grouped else 0),
            'config': {'workload': desc, 'arch': arch, 'head': head, 'batch_per_gpu': batch,
                       'global_batch': world * batch, 'gallery_rows': gallery_rows,
                       'gallery_rows_per_gpu': hi - lo, 'emd': 512, 'metric': 'cosine',
                       'parallelism': 'dp%d + gallery row-shard' % world, 'backend': args.backend if grouped else None},
            'forwards_in_process': {str(k): v for k, v in sorted(forwards.items())},
            'latency': latency,
            'phases_per_rank_ms': per_rank,
            'phases_ms': (dict({'embed': embed_ms, 'match': match_ms}, **sub) if pipe is None else
                          {'detect+crop': float(np.mean([ev[i][0].elapsed_time(det_ms[i]) for i in range(args.steps)])),
                           'detect+crop+embed': embed_ms, 'match': match_ms}),
            'roofline': {
                'bound': 'mfma',
                'kernel': '%s: %.1f %% of the GPU time of one profiled (single-lane) forward over %d launches; one launch group = '
                          'the %d conv launches of one %s forward at batch %d%s'
                          % (shares[0][0], 100 * shares[0][1], shares[0][2],
                             sum(1 for _, k, _, _ in prof if k.startswith(is_conv)), arch, batch,
                             '' if pipe is None else ' + the %d conv launches of the %s detector per chunk of 64 frames'
                             % (sum(1 for _, k, _ in det_ops if k.startswith(is_conv)), 'MTCNN' if mtcnn else 'YOLOv3-face')),
                'kernels': [{'kernel': k, 'share': round(sh, 4), 'launches': c} for k, sh, c in shares[:8]],
                'achieved': achieved, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': achieved / PEAK_F32_MFMA_TFLOPS, 'traffic': traffic,
                'traffic_unit': 'HBM bytes per forward (rocprofv3 PMC)',
                'traffic_source': (traffic_src + ' (committed profile of this command, not a measurement of this run)') if traffic_src else None,
                'compulsory_bytes': compulsory, 'traffic_ratio': (traffic / compulsory) if traffic else None,
                # per-launch mixed roofline: sum over launches of max(flops / f32-MFMA peak, compulsory bytes / measured HBM copy rate)
                'mixed': {'t_roof_ms': t_mixed_ms, 'frac_mixed': t_mixed_ms / embed_ms, 'hbm_rate_tbs': HBM_COPY_TBS,
                          'launches_bound_by_hbm': len(hbm_bound), 'which': hbm_bound[:24]},
                'clock_note': 'peak is the 2.4 GHz figure.  conv_clock_ghz = shader clock held INSIDE the convolution kernels of one '
                              'single-lane forward on this box (s_memtime / s_memrealtime over every block: dif_net_embed_clock); '
                              'mfma_loop_clock_ghz / mfma_loop_tflops = what a register-only loop of the same MFMA instruction '
                              'holds and sustains on this box (dif_probe_mfma_clock)',
                'conv_clock_ghz': conv_ghz, 'mfma_loop_clock_ghz': ghz.value, 'mfma_loop_tflops': probe_tf.value,
                'frac_of_peak_at_conv_clock': achieved / (PEAK_F32_MFMA_TFLOPS * conv_ghz / 2.4) if conv_ghz else None,
                'algorithmic_flops_per_forward': flops_embed,
                'forward_ms_hip_events': embed_ms,
                'conv_only': {'ms': conv_ms, 'tflops': conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms else None},
                'match': {'ms': match_ms, 'tflops': m_flops / (match_ms * 1e-3) / 1e12,
                          'kernel': ('match_g1_kernel (probes resident in LDS, gallery copy in MFMA-fragment order streamed into registers)'
                                     if shard.gallery.stat('frag_copy') else
                                     {1: 'match_b1_kernel', 2: 'match_bd_kernel / match_tile_kernel', 0: 'match_tile_kernel'}[filt]),
                          'filter': {1: 'bf16 (one term per operand, one v_mfma_f32_32x32x16_bf16 per 16 k; candidates within the '
                                        'proven bound re-ranked in the reference arithmetic)',
                                     2: 'bf16x2 (two bf16 terms per operand, three v_mfma_f32_32x32x16_bf16 per 16 k)',
                                     0: 'f32 MFMA'}[filt],
                          'ceiling_tflops': m_ceiling, 'algorithmic_bytes': m_bytes,
                          'frac_of_two_term_ceiling': m_flops / (PEAK_BF16_MFMA_TFLOPS / 3.0 * 1e12) * 1e3 / match_ms,
                          'hbm_floor_ms': m_bytes / (HBM_COPY_TBS * 1e12) * 1e3, 'mfma_floor_ms': m_flops / (m_ceiling * 1e12) * 1e3,
                          't_roof_ms': m_roof_ms, 'frac': m_roof_ms / match_ms,
                          'achieved_gbs': m_bytes / (match_ms * 1e-3) / 1e9},
                'combined': {'t_roof_ms': e_roof_ms + m_roof_ms, 'step_ms': elapsed / args.steps * 1e3,
                             'frac': (e_roof_ms + m_roof_ms) / (elapsed / args.steps * 1e3),
                             'note': '(t_roof of the embedding forward at the f32-MFMA peak + t_roof of the match) / measured step'},
            },
        }
        if compute != 'f32':
            # honest denominators: the bf16 MFMA peak, against which 6 (3) MFMA flops are spent per algorithmic flop
            nprod = 6.0 if compute == 'bf16x3' else 3.0
            out['roofline']['peak_note'] = ('frac above is against the f32-MFMA peak for comparison with the float32 path; the '
                                            'mode runs on the bf16 MFMA (2500 TFLOP/s dense), %d products per algorithmic '
                                            'multiply-add: ceiling 2500/%d = %.1f TFLOP/s algorithmic' % (nprod, nprod, 2500.0 / nprod))
            out['roofline']['frac_of_%s_ceiling' % compute] = achieved / (2500.0 / nprod)
            out['roofline']['bf16_mfma_tflops_issued'] = achieved * nprod
        if b256_ms is not None:
            a256 = model.flops_per_image * 256 / (b256_ms * 1e-3) / 1e12
            out['roofline']['b256'] = {'forward_ms_hip_events': b256_ms, 'achieved': a256, 'peak': PEAK_F32_MFMA_TFLOPS,
                                       'unit': 'TFLOP/s', 'frac': a256 / PEAK_F32_MFMA_TFLOPS,
                                       'faces_per_s_embed_only': 256 / (b256_ms * 1e-3),
                                       'traffic': measured_traffic('r100', 256)[0],
                                       'traffic_source': measured_traffic('r100', 256)[1],
                                       'note': 'north_star target configuration: IResNet-100 forward at batch 256'}
        if world == 1 and pipe is None and compute == 'f32' and arch.startswith('iresnet') and not args.no_throughput_mode:
            # the split-bf16 THROUGHPUT modes on the same crops, weights and batch (never the headline: `value` above is float32,
            # the reference's arithmetic).  Each tier is timed and put through the float32 gates on this very batch: cosine gap
            # to the float32 embeddings < 1e-5, every pairwise arccos distance of the batch within 1e-5 of float32's, and every
            # probe naming the same gallery row (the step's own gallery + the 8 planted enrolments).  The block reports the
            # fastest tier that passes all three.
            import math
            ef = emb_buf.double()
            ef = ef / ef.norm(dim=1, keepdim=True)
            pd_ref = torch.arccos((ef @ ef.t()).clamp(-1, 1)) / math.pi
            idx_ref = idx_v.clone()
            tiers = {}
            for tier, desc in (('bf16x3', 'three bf16 terms per f32 operand, six bf16 MFMA products'),
                               ('bf16x2', 'two bf16 terms per f32 operand (hi + mid), three bf16 MFMA products')):
                b3 = DifEmbedder(arch, head, 512, (112, 112, 3), max_batch=batch, compute=tier)
                b3.set_weights(model.get_weights())
                b3.set_input_transform(scale=1 / 255.)
                e3 = b3.embed(crops)
                for _ in range(3):
                    b3.embed(crops)
                reps = max(3, min(args.steps, 10))
                tv = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                tv[0].record()
                for _ in range(reps):
                    b3.embed(crops)
                tv[1].record()
                torch.cuda.synchronize()
                out['forwards_in_process']['%s_%d' % (tier, batch)] = 4 + reps
                ms3 = tv[0].elapsed_time(tv[1]) / reps
                e3d = e3.double()
                e3d = e3d / e3d.norm(dim=1, keepdim=True)
                gap = float((1 - (ef * e3d).sum(1)).max())
                pdd = float((torch.arccos((e3d @ e3d.t()).clamp(-1, 1)) / math.pi - pd_ref).abs().max())
                idx3, _ = shard.match(e3, 1)
                same_rows = int((idx3 == idx_ref).sum())
                ok = gap < 1e-5 and pdd < 1e-5 and same_rows == batch
                tiers[tier] = {
                    'compute': '%s: %s, f32 accumulation; 3x3 / stride 1 layers from 64 channels up, the other layers stay float32' % (tier, desc),
                    'forward_ms_hip_events': ms3, 'faces_per_s_embed_only': batch / (ms3 * 1e-3),
                    'faces_per_s_with_the_float32_runs_match': batch / ((ms3 + match_ms) * 1e-3),
                    'algorithmic_tflops': flops_embed / (ms3 * 1e-3) / 1e12, 'speedup_vs_float32_forward': embed_ms / ms3,
                    'mfma_products_per_multiply_add': 6 if tier == 'bf16x3' else 3,
                    'frac_of_its_bf16_ceiling': flops_embed / (ms3 * 1e-3) / 1e12 / (PEAK_BF16_MFMA_TFLOPS / (6.0 if tier == 'bf16x3' else 3.0)),
                    'max_cosine_gap_to_float32_embeddings': gap, 'max_pairwise_distance_diff_vs_float32': pdd,
                    'probes_naming_the_float32_row': '%d / %d (gallery of %d rows)' % (same_rows, batch, gallery_rows),
                    'passes_the_float32_gates': ok}
                b3.close()
            passing = [t for t in tiers if tiers[t]['passes_the_float32_gates']]
            best = min(passing, key=lambda t: tiers[t]['forward_ms_hip_events']) if passing else None
            out['throughput_mode'] = dict(tiers[best], tier=best) if best else {'tier': None}
            out['throughput_mode']['gates'] = 'cosine gap < 1e-5, all pairwise distances within 1e-5, same gallery row for every probe -- vs the float32 embeddings of this run'
            out['throughput_mode']['tiers'] = tiers
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = (cpu_baseline(arch, head, gallery_rows) if pipe is None else
                                   (cpu_baseline_frames_mtcnn(gallery_rows, det.get_weights()) if mtcnn else
                                    cpu_baseline_frames(gallery_rows, dp)))
        if json_fd is None:
            print(json.dumps(out), flush=True)
        else:
            os.write(json_fd, (json.dumps(out) + '\n').encode())
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
