#!/usr/bin/env python3
"""One pass of a rocprofv3 --kernel-trace run as a timeline: tools/pass_timeline.py <dir> [which]
Lists every kernel between two consecutive adam_kernel launches (the `which`-th pass from the end) with its
duration and the gap before it, then the per-kernel totals."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('adam_kernel')]
a, b = adam[-which - 1], adam[-which]
seg = rows[a + 1:b + 1]
t0 = int(rows[a]['End_Timestamp'])
prev = t0
tot = collections.defaultdict(lambda: [0, 0.0])
gaps = 0.0
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:40]
    gap = (s - prev) / 1e3
    gaps += max(gap, 0)
    print(f'{(s - t0) / 1e3:9.1f} us  +{gap:6.1f}  {(e - s) / 1e3:8.1f} us  {name}  grid {r.get("Grid_Size_X", "?")}')
    tot[name][0] += 1; tot[name][1] += (e - s) / 1e3
    prev = max(prev, e)
print(f'pass wall {(prev - t0) / 1e3:.1f} us, {len(seg)} kernels, idle gaps {gaps:.1f} us')
for k, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f'{k:42s} launches {n:3d} total {t:9.1f} us')
