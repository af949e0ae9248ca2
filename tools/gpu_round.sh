#!/bin/bash
# Developer helper for one gpurun call: GPU parity tests, the default bench line, the 64-clip step profile.
# Usage (from the repo root on the GPU box): bash tools/gpu_round.sh <tag> [tests|bench|prof ...]
set -o pipefail
TAG=${1:-x}; shift
WHAT=${*:-tests bench prof}
mkdir -p gpurun_out
for w in $WHAT; do
  case $w in
    tests) timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/${TAG}_tests.log ;;
    bench) timeout -k 10 300 python bench.py --steps 200 --warmup 20 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/${TAG}_bench.json').read().strip().splitlines()[-1])
    b = d.get('batched') or {}
    print('one clip: %.0f it/s, %.3f ms; roofline %s frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac']))
    s_ = d.get("surface") or {}
    if s_: print("surface: %.0f it/s, %.3f ms" % (s_["value"], s_["ms_per_step"]))
    if b: print('batched: %.0f clip-it/s, %.2f ms/pass; %s %.1f TF frac %.3f' % (b['value'], b['ms_per_pass'], b['roofline']['kernel'], b['roofline']['achieved'], b['roofline']['frac']))
    print('cpu', d.get('cpu_baseline', {}).get('value'))
except Exception as e:
    print('bench parse failed', e)
PY
    ;;
    prof) timeout -k 10 300 python tools/step_profile.py 4 16 4 64 > gpurun_out/${TAG}_steps64.txt 2>&1; echo "prof64 rc=$?"; tail -1 gpurun_out/${TAG}_steps64.txt
          timeout -k 10 120 python tools/step_profile.py > gpurun_out/${TAG}_steps1.txt 2>&1; echo "prof1 rc=$?"; tail -1 gpurun_out/${TAG}_steps1.txt ;;
  esac
done
