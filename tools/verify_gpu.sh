# One gpurun call that checks a tree end to end on an MI355X: GPU test suite, smoke(), the default bench line.
# Usage: gpurun --timeout 1200 -- bash tools/verify_gpu.sh <tag>   (writes gpurun_out/<tag>_*)
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; T=${1:-verify}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/${T}_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 600 python bench.py > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open('gpurun_out/${T}_bench_default.json').read().strip().splitlines()[-1])
b = d.get('batched') or {}
print('one clip: %.0f it/s %.3f ms upload %s launches %s steps %s' % (d['value'], d['ms_per_step'], d.get('value_with_upload'), d['config'].get('launches_per_pass'), d['steps']))
print('roofline', d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('traffic_per_iteration'))
print('cpu', d.get('cpu_baseline'))
print('surface', (d.get('surface') or {}).get('value'), (d.get('surface') or {}).get('fused_value'))
ae = d.get("audio_extension") or {}
print("audio ext", ae.get("value"), ae.get("error"))
if b: print('batched: %.0f clip-it/s, %.2f ms/pass' % (b['value'], b['ms_per_pass']))
PY
