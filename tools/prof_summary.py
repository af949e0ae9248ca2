#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV per training iteration: tools/prof_summary.py <dir> <n_iterations>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
n = float(sys.argv[2])
rows = list(csv.DictReader(open(f)))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print(f'total kernel time per iteration: {tot / n / 1e3:.1f} us')
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{r['Name'][:58]:58s} calls/it {int(r['Calls']) / n:6.1f} avg {float(r['AverageNs']) / 1e3:8.1f} us  per-iter {int(r['TotalDurationNs']) / n / 1e3:8.1f} us")
