#!/usr/bin/env python3
"""Aggregate two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into HBM bytes per launch per kernel:
tools/pmc_traffic.py <dir_fetch> <dir_write> <out.json> "<command>"   (counters are in KiB-ish units of 1024 B... see note)"""
import collections, csv, glob, json, sys


def per_kernel(d, counter):
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        k = r['Kernel_Name'].split('(')[0]
        tot[k] += float(r['Counter_Value'])
        cnt[k] += 1
    return {k: dict(launches=cnt[k], avg_kb_per_launch=round(tot[k] / cnt[k], 2)) for k in tot}


fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
out = dict(command=sys.argv[4],
           note='Raw counters x 1024 B. MI355X_MICROARCH.md: on gfx950 FETCH_SIZE under-reports wide (16 B/lane) streaming reads by '
                '2x; the GEMM kernels load 4 B/lane, a width the guide calls uncalibrated, so the raw value is reported uncorrected. '
                'WRITE_SIZE is exact for streaming stores.',
           per_kernel=dict(FETCH_SIZE=fetch, WRITE_SIZE=write))
for k in ('gemm_kernel', 'gemm_mfma_kernel'):
    if k in fetch:
        fb, wb = fetch[k]['avg_kb_per_launch'] * 1024, write.get(k, dict(avg_kb_per_launch=0))['avg_kb_per_launch'] * 1024
        out[k] = dict(fetch_bytes_per_launch=fb, write_bytes_per_launch=wb, hbm_bytes_per_launch=fb + wb)
# whole-pass traffic: every kernel's bytes summed, per training iteration.  Kernels that only run while the benchmark sets up
# (torch's fills of the freshly allocated workspace, runtime buffer copies of the uploads) are listed separately: they are
# not part of an iteration and would otherwise be spread over however few iterations a counter run happens to have
def is_setup(name):
    return name.startswith('void at::native') or name.startswith('__amd_rocclr')


def per_iteration(fetch, write, clips):
    # passes = launches of loss_tail_kernel (one per pass, nothing else launches it), counted PER RUN: the two counter runs
    # repeat their timed region until 0.25 s have passed and need not make the same number of passes
    it_f, it_w = float(clips) * fetch['loss_tail_kernel']['launches'], float(clips) * write['loss_tail_kernel']['launches']
    def total(tab, setup):
        return sum(v['launches'] * v['avg_kb_per_launch'] for k, v in tab.items() if is_setup(k) == setup) * 1024
    f, w = total(fetch, False) / it_f, total(write, False) / it_w
    return dict(iterations_fetch_run=it_f, iterations_write_run=it_w, fetch_bytes=f, write_bytes=w, hbm_bytes=f + w,
                setup_bytes_whole_run=total(fetch, True) + total(write, True),
                note='all kernels of the timed loop (model, loss, optimizer) / iterations of the same run; raw counters; '
                     'setup_bytes_whole_run = one-time fills / uploads before the loop (torch fill kernels, runtime copies), not included')


if len(sys.argv) > 5:
    out['per_iteration'] = per_iteration(fetch, write, sys.argv[5])
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print({k: out[k] for k in out if k.startswith('gemm')})
