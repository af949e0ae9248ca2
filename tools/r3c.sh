set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r3c}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/${T}_tests.log
timeout -k 10 300 python tools/step_profile.py 4 16 4 64 > gpurun_out/${T}_steps64.txt 2>&1; echo "prof64 rc=$?"; tail -1 gpurun_out/${T}_steps64.txt
sort -k3 -n -r gpurun_out/${T}_steps64.txt | awk '$3+0>60' | head -24
timeout -k 10 400 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/${T}_bench.err
python - <<PY
import json
d = json.loads(open('gpurun_out/${T}_bench.json').read().strip().splitlines()[-1])
b = d.get('batched') or {}
print('one clip: %.0f it/s %.3f ms upload %s launches %s' % (d['value'], d['ms_per_step'], d.get('value_with_upload'), d['config'].get('launches_per_pass')))
print('surface', (d.get('surface') or {}).get('value'), (d.get('surface') or {}).get('fused_value'))
ae = d.get("audio_extension") or {}
print("audio ext", ae.get("value"), ae.get("stft"), ae.get("gram"), ae.get("error"))
if b: print('batched: %.0f clip-it/s, %.2f ms/pass' % (b['value'], b['ms_per_pass']))
PY
