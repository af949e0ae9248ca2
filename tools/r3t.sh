set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; T=${1:-r3t}
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p1 -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --batched-clips 0 --surface-steps 0 --audio-steps 0 > $O/${T}_bench_under_rocprof.json 2> $O/${T}_p1.err; echo "rc=$?"
python tools/pass_timeline.py $O/p1 > $O/${T}_pass_timeline_one_clip.txt 2>&1
python tools/prof_summary.py $O/p1 220 40 > $O/${T}_kernel_summary_per_iteration.txt
rm -rf $O/p1
timeout -k 10 400 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --surface-steps 0 --audio-steps 0 > $O/${T}_bench.json 2> $O/${T}_bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open('gpurun_out/${T}_bench.json').read().strip().splitlines()[-1])
b = d.get('batched') or {}
print('one clip: %.0f it/s %.3f ms launches %s' % (d['value'], d['ms_per_step'], d['config'].get('launches_per_pass')))
if b: print('batched: %.0f clip-it/s, %.2f ms/pass' % (b['value'], b['ms_per_pass']))
PY
