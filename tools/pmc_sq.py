"""Aggregates a rocprofv3 --pmc counter_collection.csv per kernel family: every counter summed over the
dispatches, plus per-nanosecond rates per SIMD (clock-free, see tools/pmc_mfma.py).

    python tools/pmc_sq.py <counter_collection.csv> [substring of the kernel names to keep]
"""
import collections
import csv
import json
import sys


def main():
    keep = sys.argv[2] if len(sys.argv) > 2 else ''
    rows = collections.defaultdict(dict)
    for r in csv.DictReader(open(sys.argv[1])):
        d = rows[r['Dispatch_Id']]
        d['name'] = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('dif::', '')
        d['ns'] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        d.setdefault('c', {})[r['Counter_Name']] = float(r['Counter_Value'])
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for d in rows.values():
        if keep and keep not in d['name']:
            continue
        a = agg[d['name']]
        a['launches'] += 1
        a['ns'] += d['ns']
        for k, v in d['c'].items():
            a[k] += v
    out = {}
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1]['ns']):
        o = {'launches': int(a['launches']), 'total_ms': a['ns'] / 1e6}
        for k, v in a.items():
            if k in ('launches', 'ns'):
                continue
            o[k] = v
            o[k + '_per_simd_per_ns'] = v / 1024.0 / a['ns']
        out[name] = o
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
