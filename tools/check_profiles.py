"""Copies the latest rocprofv3 summaries from gpurun_out/ into profiles/ and prints the
cross-check the bench line's `roofline` rests on: per forward, the sum of the conv kernel
durations in the kernel-trace stats vs the HIP-event forward time reported by bench.py."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'gpurun_out')
PROF = os.path.join(ROOT, 'profiles')
ROUND = 'r01'


def latest(pattern):
    files = sorted(glob.glob(os.path.join(OUT, pattern)), key=os.path.getmtime)
    return files[-1] if files else None


def main():
    lines = []
    for w, forwards_ks in (('r50', 14), ('r100', 14)):
        ks = latest('ks1_%s/*/*_kernel_stats.csv' % w)          # single-lane run: durations do not overlap
        ks2 = latest('ks_%s/*/*_kernel_stats.csv' % w)          # default two-lane run
        shutil.copy(ks2, os.path.join(PROF, '%s_%s_b256_kernel_stats_2lanes.csv' % (ROUND, w)))
        shutil.copy(os.path.join(OUT, 'bench1_%s.json' % w), os.path.join(PROF, '%s_%s_b256_bench_1lane.json' % (ROUND, w)))
        bench1 = json.load(open(os.path.join(OUT, 'bench1_%s.json' % w)))
        pf = latest('pf_%s/*/*_counter_collection.csv' % w)
        pw = latest('pw_%s/*/*_counter_collection.csv' % w)
        shutil.copy(ks, os.path.join(PROF, '%s_%s_b256_kernel_stats.csv' % (ROUND, w)))
        shutil.copy(os.path.join(OUT, 'layers_%s.txt' % w), os.path.join(PROF, '%s_%s_b256_layers.txt' % (ROUND, w)))
        shutil.copy(os.path.join(OUT, 'bench_%s.json' % w), os.path.join(PROF, '%s_%s_b256_bench.json' % (ROUND, w)))
        pm = latest('pm_%s/*/*_counter_collection.csv' % w)
        if pm:
            with open(os.path.join(PROF, '%s_%s_b256_mfma_util.json' % (ROUND, w)), 'w') as fh:
                subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'pmc_mfma.py'), pm], stdout=fh)
        with open(os.path.join(PROF, '%s_%s_b256_hbm_traffic.json' % (ROUND, w)), 'w') as fh:
            subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'pmc_traffic.py'), pf, pw, '5'], stdout=fh)
        conv_ns, conv_calls = 0.0, 0
        for r in csv.DictReader(open(ks)):
            if 'conv_igemm_kernel' in r['Name'] or 'conv_pipe_kernel' in r['Name']:
                conv_ns += float(r['TotalDurationNs'])
                conv_calls += int(r['Calls'])
        bench = json.load(open(os.path.join(OUT, 'bench_%s.json' % w)))
        traffic = json.load(open(os.path.join(PROF, '%s_%s_b256_hbm_traffic.json' % (ROUND, w))))['total']
        lines.append('%-4s ONE lane (DIF_STREAMS=1): %d conv launches/forward, sum of conv kernel durations %.3f ms/forward '
                     '(rocprofv3 kernel-trace) vs HIP-event forward %.3f ms (%.1f TFLOP/s, frac %.3f)\n'
                     '     default executor: HIP-event forward %.3f ms, %.1f TFLOP/s, frac %.3f, %.0f faces/s incl. match | '
                     'fabric traffic %.1f GB/forward'
                     % (w, conv_calls // forwards_ks, conv_ns / forwards_ks / 1e6,
                        bench1['roofline']['forward_ms_hip_events'], bench1['roofline']['achieved'], bench1['roofline']['frac'],
                        bench['roofline']['forward_ms_hip_events'], bench['roofline']['achieved'], bench['roofline']['frac'],
                        bench['value'], traffic['conv_hbm_bytes_per_forward'] / 1e9))
    print('\n'.join(lines))
    with open(os.path.join(PROF, '%s_summary.txt' % ROUND), 'w') as fh:
        fh.write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    main()
