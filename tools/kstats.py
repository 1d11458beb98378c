"""Condenses a rocprofv3 --kernel-trace --stats directory: per kernel name (template arguments kept) calls, total and
average microseconds, share.  python tools/kstats.py <dir> [forwards]"""
import csv
import glob
import sys

d = sys.argv[1]
fw = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = sorted(glob.glob(d + '/**/*kernel_stats.csv', recursive=True))
if not f:
    raise SystemExit('no *kernel_stats.csv under ' + d)
rows = list(csv.DictReader(open(f[-1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('%-100s %8s %10s %9s %6s' % ('kernel', 'calls/fw', 'us/forward', 'avg us', '%'))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
    t = float(r['TotalDurationNs'])
    print('%-100s %8.1f %10.1f %9.2f %6.1f' % (r['Name'][:100], int(r['Calls']) / fw, t / fw / 1e3, float(r['AverageNs']) / 1e3, 100 * t / tot))
print('TOTAL %.1f us per forward' % (tot / fw / 1e3))
