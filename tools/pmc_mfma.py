"""MFMA-pipe utilisation per kernel family from a rocprofv3 PMC pass
(`--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`) of `bench.py`.

SQ_VALU_MFMA_BUSY_CYCLES sums, over all SIMDs, the cycles their matrix pipe is busy (64 per
v_mfma_f32_32x32x2_f32: MI355X_MICROARCH.md "Per-instruction cycle constants");
GRBM_GUI_ACTIVE is summed over the 8 XCDs, so kernel cycles = GRBM_GUI_ACTIVE / 8 and

    MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles * 1024 SIMDs).

The clock the chip actually held is kernel cycles / kernel duration; the busy fraction times that
clock over 2.4 GHz is the fraction of the 157.3 TFLOP/s peak the issued MFMAs amount to (it
includes the zero-padded K columns and tile tails, which algorithmic FLOPs do not).

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
        out[k] = {'launches': int(a['launches']), 'total_ms': a['ns'] / 1e6, 'mfma_busy_frac': busy,
                  'clock_ghz': ghz, 'issued_mfma_frac_of_157.3TF_peak': busy * ghz / 2.4}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
