"""MFMA-pipe utilisation per kernel family from a rocprofv3 PMC pass
(`--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`) of `bench.py`.

SQ_VALU_MFMA_BUSY_CYCLES sums, over all SIMDs, the cycles their matrix pipe is busy (64 per
v_mfma_f32_32x32x2_f32: MI355X_MICROARCH.md "Per-instruction cycle constants");
GRBM_GUI_ACTIVE is summed over the 8 XCDs, so kernel cycles = GRBM_GUI_ACTIVE / 8 and

    MFMA busy fraction (of GRBM cycles) = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles * 1024 SIMDs).

Caveat (MI355X_MICROARCH.md "DVFS give-back"): GRBM_GUI_ACTIVE / 8 / duration reads HIGH on dispatches
shorter than a few ms -- it gives 2.4-2.5 GHz here while s_memtime / s_memrealtime inside the same
kernels shows the shader clock held at 2.03 GHz (profiles/r01_conv_trace.txt).  The clock-free number
is `mfma_busy_cycles_per_simd_per_ns` (= busy GHz): divided by 2.4 it is the fraction of the 157.3 TFLOP/s
peak the issued MFMAs amount to (padding and tile tails included); divided by the held clock (2.03)
it is how busy the matrix pipe really is.

    python tools/pmc_mfma.py <counter_collection.csv>
"""
import collections
import csv
import json
import sys

SIMDS = 256 * 4


def main():
    rows = collections.defaultdict(dict)
    for r in csv.DictReader(open(sys.argv[1])):
        d = rows[r['Dispatch_Id']]
        d['name'] = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('dif::', '')
        d['ns'] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        d[r['Counter_Name']] = float(r['Counter_Value'])
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for d in rows.values():
        if 'SQ_VALU_MFMA_BUSY_CYCLES' not in d or 'GRBM_GUI_ACTIVE' not in d:
            continue
        a = agg[d['name']]
        a['launches'] += 1
        a['mfma_busy'] += d['SQ_VALU_MFMA_BUSY_CYCLES']
        a['cycles'] += d['GRBM_GUI_ACTIVE'] / 8.0
        a['ns'] += d['ns']
    out = {}
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]['ns']):
        if a['mfma_busy'] == 0:
            continue
        busy = a['mfma_busy'] / (a['cycles'] * SIMDS)
        ghz = a['cycles'] / a['ns']
        busy_ghz = a['mfma_busy'] / SIMDS / a['ns']
        out[k] = {'launches': int(a['launches']), 'total_ms': a['ns'] / 1e6,
                  'mfma_busy_cycles_per_simd_per_ns': busy_ghz,
                  'issued_mfma_frac_of_157.3TF_peak': busy_ghz / 2.4,
                  'mfma_pipe_busy_at_held_clock_2.03GHz': busy_ghz / 2.03,
                  'grbm_clock_ghz_unreliable_on_short_dispatches': ghz, 'mfma_busy_frac_of_grbm_cycles': busy}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
