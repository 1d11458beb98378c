"""Per-kernel sums of every counter found in one or more rocprofv3 PMC passes of the same command
(`--pmc A B C ...`, one pass per counter group, never together with tracing).

    python tools/pmc_table.py <pass1/p_counter_collection.csv> [<pass2/...csv> ...]

Kernels are keyed by name with template arguments (the conv kernels differ by them); a counter's value is summed over
the kernel's dispatches of its pass and printed per dispatch, so passes with equal dispatch counts line up.
"""
import collections
import csv
import sys


def short(name):
    name = name.replace('void ', '').replace('dif::', '')
    return name.split('(')[0][:58]


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    counters = []
    for path in sys.argv[1:]:
        for r in csv.DictReader(open(path)):
            k = short(r['Kernel_Name'])
            c = r['Counter_Name']
            if c not in counters:
                counters.append(c)
            agg[k][c] += float(r['Counter_Value'])
            key = (path, r['Dispatch_Id'])
            if key not in seen[(k, c)]:
                seen[(k, c)].add(key)
            if c == counters[0]:
                agg[k]['_ns'] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    print('%-58s %6s %9s ' % ('kernel', 'disp', 'us/disp') + ' '.join('%14s' % c[-14:] for c in counters))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]['_ns']):
        n = len(seen[(k, counters[0])]) or 1
        if a['_ns'] / n < 20000:
            continue
        print('%-58s %6d %9.1f ' % (k, n, a['_ns'] / n / 1e3) +
              ' '.join('%14.4g' % (a[c] / max(1, len(seen[(k, c)]))) for c in counters))
    derived(agg, seen)


def derived(agg, seen):
    """Wave-state fractions where the counters for them were collected (SQ_* wave counters tick in quad-cycles)."""
    need = ('SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE')
    print()
    print('%-58s %9s %9s %9s %9s %11s %11s' % ('kernel', 'resident', 'wait_any', 'wait_inst', 'active', 'mfma/disp', 'mfma/life'))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]['_ns']):
        if any(c not in a for c in need) or a['SQ_WAVE_CYCLES'] == 0:
            continue
        n = {c: max(1, len(seen[(k, c)])) for c in need}
        v = {c: a[c] / n[c] for c in need}
        if v['GRBM_GUI_ACTIVE'] / 8 < 40000:
            continue
        cyc = v['GRBM_GUI_ACTIVE'] / 8.0                    # shader cycles of one dispatch
        wave_cyc = v['SQ_WAVE_CYCLES'] * 4.0                # wave-cycles of one dispatch
        simd_busy = v['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0  # matrix-pipe cycles per SIMD
        # resident: mean number of waves per SIMD over the dispatch; mfma/life: pipe cycles per SIMD over the cycles a SIMD
        # holds four waves' worth of wave-time (= how busy the pipe is while the waves are there)
        print('%-58s %9.2f %9.3f %9.3f %9.3f %11.3f %11.3f' % (
            k, wave_cyc / 1024.0 / cyc, v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES'], v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES'],
            v['SQ_ACTIVE_INST_ANY'] / v['SQ_WAVE_CYCLES'], simd_busy / cyc, simd_busy / (wave_cyc / 1024.0 / 4.0)))


if __name__ == '__main__':
    main()
