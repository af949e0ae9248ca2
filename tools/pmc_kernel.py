#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc run per (kernel, grid size): tools/pmc_kernel.py <dir> [kernel-substring]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
sub = sys.argv[2] if len(sys.argv) > 2 else ''
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if sub not in k:
        continue
    key = (k, r['Grid_Size'])
    agg[key][r['Counter_Name']] += float(r['Counter_Value'])
    did = (r['Dispatch_Id'], r['Counter_Name'])
    if r['Counter_Name'] == 'SQ_WAVE_CYCLES' or len(agg[key]) == 1:
        pass
    if (r['Dispatch_Id']) not in seen:
        seen.add(r['Dispatch_Id']); cnt[key] += 1
for key, c in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0)):
    n = cnt[key]
    print(key[0][:34], 'grid', key[1], 'launches', n, ' '.join(f'{k}={v / n:.3g}' for k, v in sorted(c.items())))
