"""Copies the round's rocprofv3 summaries from gpurun_out/r05/ into profiles/ and prints the cross-checks
the bench line's `roofline` rests on: per forward, the sum of the conv kernel durations in the kernel-trace stats
(one lane: no overlap) vs the HIP-event forward time reported by bench.py, and the frac recomputed from them."""
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'gpurun_out', 'r05')
PROF = os.path.join(ROOT, 'profiles')
R = 'r05'
PEAK = 157.3


def jl(path):
    with open(path) as fh:
        for line in fh:
            if line.startswith('{'):
                return json.loads(line)
    raise RuntimeError('no JSON line in ' + path)


def conv_ms(stats_csv):
    ns = calls = 0
    for r in csv.DictReader(open(stats_csv)):
        if any(k in r['Name'] for k in ('conv_igemm_kernel', 'conv_pipe_kernel', 'conv_bdp_kernel', 'conv_tn_kernel', 'conv_t2_kernel', 'conv_sk_kernel', 'conv_skp_kernel', 'conv_sk_reduce_kernel', 'conv_mt_kernel',
                                         'stem_mfma_kernel', 'stem3x3_kernel')):
            ns += float(r['TotalDurationNs'])
            calls += int(r['Calls'])
    return ns / 1e6, calls


def cp(src, dst):
    shutil.copy(os.path.join(OUT, src), os.path.join(PROF, dst))


def tool(name, args, dst):
    with open(os.path.join(PROF, dst), 'w') as fh:
        subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', name)] + args, stdout=fh)


def main():
    lines = []
    cp('ks_default/p_kernel_stats.csv', R + '_default_kernel_stats.csv')
    cp('ks_default.json', R + '_default_bench_profiled.json')
    cp('ks_default_1lane/p_kernel_stats.csv', R + '_default_1lane_kernel_stats.csv')
    cp('ks_default_1lane.json', R + '_default_1lane_bench_profiled.json')
    cp('ks_r50_1lane/p_kernel_stats.csv', R + '_r50_1lane_kernel_stats.csv')
    for w in ('default', 'default_1lane', 'r100', 'r50', 'r50_1lane', 'r100_arc', 'r100_1m_bf16x3', 'r100_1m_bf16x2', 'frames', 'frames_mtcnn', 'default_fc'):
        cp('bench_%s.json' % w, '%s_%s_bench.json' % (R, w))
    for src, dst in (('layers_r100.txt', 'r100_b256_layers.txt'), ('layers_r50.txt', 'r50_b256_layers.txt'),
                     ('layers_r100_bf16x2.txt', 'r100_b256_bf16x2_layers.txt'), ('layers_yolov3.txt', 'yolov3_b64_layers.txt'),
                     ('latency.txt', 'latency.txt'), ('batch_sweep.txt', 'batch_sweep.txt'), ('bf_tier_gates.txt', 'bf_tier_gates.txt'),
                     ('match_ab.txt', 'match_filter_ab.txt'), ('match_pmc_table.txt', 'match_wave_states.txt'),
                     ('match_traffic.txt', 'match_fabric_traffic.txt'), ('launch_modes.txt', 'launch_modes.txt'),
                     ('gallery_set.txt', 'gallery_set.txt'), ('trace_r100_b1.txt', 'r100_b1_block_trace.txt'),
                     ('trace_r100_b8.txt', 'r100_b8_block_trace.txt'), ('trace_r50_b1.txt', 'r50_b1_block_trace.txt')):
        cp(src, R + '_' + src.replace(src, dst))
    for a, short in (('iresnet100', 'r100'), ('resnet', 'r50')):
        for b in (1, 8, 32):
            cp('layers_%s_b%d.txt' % (a, b), '%s_%s_b%d_layers.txt' % (R, short, b))
            if os.path.exists(os.path.join(OUT, 'ksb_%s_%d.txt' % (a, b))):
                cp('ksb_%s_%d.txt' % (a, b), '%s_%s_b%d_kernel_stats.txt' % (R, short, b))
    if os.path.exists(os.path.join(OUT, 'ks_frames_mtcnn.txt')):
        cp('ks_frames_mtcnn.txt', R + '_frames_mtcnn_kernel_stats.txt')
    # forwards per profiled run, in units of the workload's batch: bench.py reports them (`forwards_in_process`:
    # steps + warmup, the per-layer profile, the warm-up and the stamped forward of the clock measurement, and the
    # batch-256 forwards of the default workload)
    def equiv_forwards(tag, batch):
        f = jl(os.path.join(OUT, tag + '.json'))['forwards_in_process']
        return sum(int(b) * n for b, n in f.items() if b.isdigit()) / float(batch)
    tool('pmc_traffic.py', [os.path.join(OUT, 'pf_default/p_counter_collection.csv'),
                            os.path.join(OUT, 'pw_default/p_counter_collection.csv'), '%g' % equiv_forwards('pf_default', 512)],
         R + '_r100_1m_b512_hbm_traffic.json')
    tool('pmc_traffic.py', [os.path.join(OUT, 'pf_r100/p_counter_collection.csv'),
                            os.path.join(OUT, 'pw_r100/p_counter_collection.csv'), '%g' % equiv_forwards('pf_r100', 256)],
         R + '_r100_b256_hbm_traffic.json')
    tool('pmc_traffic.py', [os.path.join(OUT, 'pf_r50/p_counter_collection.csv'),
                            os.path.join(OUT, 'pw_r50/p_counter_collection.csv'), '%g' % equiv_forwards('pf_r50', 256)],
         R + '_r50_b256_hbm_traffic.json')
    tool('pmc_mfma.py', [os.path.join(OUT, 'pm_default/p_counter_collection.csv')], R + '_r100_1m_mfma_util.json')
    tool('pmc_mfma.py', [os.path.join(OUT, 'pm_bf16x2/p_counter_collection.csv')], R + '_r100_1m_bf16x2_mfma_util.json')
    for src, dst in (('trace_r100_f32.txt', '_r100_b256_block_trace.txt'), ('trace_r100_bf16x2.txt', '_r100_b256_bf16x2_block_trace.txt'),
                     ('trace_r50_f32.txt', '_r50_b256_block_trace.txt'), ('place_r100_f32.txt', '_r100_b256_block_placement.txt')):
        if not os.path.exists(os.path.join(OUT, src)):
            continue
        with open(os.path.join(OUT, src)) as fh, open(os.path.join(PROF, R + dst), 'w') as out:
            out.writelines(l for l in fh if l.startswith(('trace ', 'place ', '      pipelined')))

    # cross-check 1: default workload, ONE lane: every forward of the profiled process, in batch-512 equivalents
    one = jl(os.path.join(OUT, 'ks_default_1lane.json'))
    ms, calls = conv_ms(os.path.join(OUT, 'ks_default_1lane/p_kernel_stats.csv'))
    fw = equiv_forwards('ks_default_1lane', 512)
    per_fw = ms / fw
    flops = one['roofline']['algorithmic_flops_per_forward']
    lines.append('default workload (IResNet-100, 512 faces, 1M gallery), ONE lane, under rocprofv3: %d conv launches, '
                 'sum of conv kernel durations %.2f ms = %.2f ms per batch-512 forward (kernel-trace stats) vs HIP-event forward '
                 '%.2f ms in the same run; frac from the stats %.4f vs bench %.4f'
                 % (calls, ms, per_fw, one['roofline']['forward_ms_hip_events'], flops / (per_fw * 1e-3) / 1e12 / PEAK,
                    one['roofline']['frac']))
    # the match filter kernel of the same profiled run: per-launch duration from the stats vs the HIP-event match phase
    for r in csv.DictReader(open(os.path.join(OUT, 'ks_default_1lane/p_kernel_stats.csv'))):
        if 'match_g1_kernel' in r['Name'] or 'match_b1_kernel' in r['Name'] or 'match_bd_kernel' in r['Name']:
            lines.append('%s in that run: %s calls, %.1f us average (kernel-trace stats) vs match phase %.3f ms by HIP events '
                         '(the phase also holds probe_prep1 / finish / exact)'
                         % (r['Name'].split('(')[0].split('::')[-1], r['Calls'], float(r['AverageNs']) / 1e3, one['phases_ms']['match']))
    d = jl(os.path.join(OUT, 'bench_default.json'))
    dp = jl(os.path.join(OUT, 'ks_default.json'))
    lines.append('default workload, default executor (two lanes): un-profiled %.0f faces/s, forward %.2f ms, frac %.4f (b256: '
                 '%.2f ms, frac %.4f), match %.2f ms; the same command under rocprofv3 --kernel-trace: %.0f faces/s, frac %.4f'
                 % (d['value'], d['roofline']['forward_ms_hip_events'], d['roofline']['frac'],
                    d['roofline']['b256']['forward_ms_hip_events'], d['roofline']['b256']['frac'], d['phases_ms']['match'],
                    dp['value'], dp['roofline']['frac']))
    m = d['roofline']['match']
    lines.append('match roofline of that line: %.3f ms, %.1f TFLOP/s of the %.0f ceiling = %.3f (MFMA floor %.3f ms, HBM floor %.3f ms); '
                 'combined (t_roof embed + t_roof match) / step = %.4f; throughput_mode: %s %.2f ms = %.3fx'
                 % (m['ms'], m['tflops'], m['ceiling_tflops'], m['frac'], m['mfma_floor_ms'], m['hbm_floor_ms'],
                    d['roofline']['combined']['frac'], d['throughput_mode']['tier'], d['throughput_mode']['forward_ms_hip_events'],
                    d['throughput_mode']['speedup_vs_float32_forward']))
    # cross-check 2: ResNet-50V2 one lane
    r50 = jl(os.path.join(OUT, 'bench_r50_1lane.json'))
    ms, calls = conv_ms(os.path.join(OUT, 'ks_r50_1lane/p_kernel_stats.csv'))
    fw50 = equiv_forwards('ks_r50_1lane', 256)
    lines.append('r50 (configs[1]), ONE lane: %d conv launches per forward, %.3f ms of conv kernels per forward (stats) vs HIP-event '
                 'forward %.3f ms (un-profiled run), frac %.4f' % (round(calls / fw50), ms / fw50, r50['roofline']['forward_ms_hip_events'],
                                                                  r50['roofline']['frac']))
    for w in ('r100', 'r50', 'r100_arc', 'r100_1m_bf16x3', 'r100_1m_bf16x2', 'frames', 'frames_mtcnn', 'default_fc'):
        b = jl(os.path.join(OUT, 'bench_%s.json' % w))
        lines.append('%-16s %8.0f %s  step %.2f ms  phases %s  frac(f32 peak) %.4f'
                     % (w, b['value'], b['unit'], b['ms_per_step'], json.dumps(b['phases_ms']), b['roofline']['frac']))
    for w, f in (('r100_1m b512', R + '_r100_1m_b512_hbm_traffic.json'), ('r100 b256', R + '_r100_b256_hbm_traffic.json'),
                 ('r50 b256', R + '_r50_b256_hbm_traffic.json')):
        t = json.load(open(os.path.join(PROF, f)))['total']
        lines.append('fabric traffic %-13s %.2f GB per forward (read %.2f + write %.2f)'
                     % (w, t['conv_hbm_bytes_per_forward'] / 1e9, t['conv_read_bytes_per_forward'] / 1e9,
                        t['conv_write_bytes_per_forward'] / 1e9))
    print('\n'.join(lines))
    with open(os.path.join(PROF, R + '_summary.txt'), 'w') as fh:
        fh.write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    main()
