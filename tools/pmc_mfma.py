"""MFMA-pipe utilisation per kernel family from a rocprofv3 PMC pass
(`--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`) of `bench.py`.

SQ_VALU_MFMA_BUSY_CYCLES sums, over all SIMDs, the cycles their matrix pipe is busy (64 per
v_mfma_f32_32x32x2_f32: MI355X_MICROARCH.md "Per-instruction cycle constants");
GRBM_GUI_ACTIVE is summed over the 8 XCDs, so kernel cycles = GRBM_GUI_ACTIVE / 8 and

    MFMA busy fraction (of GRBM cycles) = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles * 1024 SIMDs).

Clock: GRBM_GUI_ACTIVE / 8 / duration is the shader clock over the dispatch (it reads high on dispatches much
shorter than 0.3 ms: MI355X_MICROARCH.md "DVFS give-back"); on the 0.5 ms convolution launches it agrees with the
in-kernel s_memtime / s_memrealtime ratio that bench.py records (dif_net_embed_clock: 2.34-2.38 GHz on round 2's
boxes; round 1's box held 2.03 GHz).  `mfma_busy_cycles_per_simd_per_ns` is clock-free: divided by 2.4 it is the
fraction of the 157.3 TFLOP/s peak the issued MFMAs amount to (padding and tile tails included);
`mfma_busy_frac_of_grbm_cycles` is how busy the matrix pipe is at the clock the dispatch ran at.

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
                  'grbm_clock_ghz': ghz, 'mfma_busy_frac_of_grbm_cycles': busy}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
