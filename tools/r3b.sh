set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3b_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r3b_tests.log
timeout -k 10 300 python tools/step_profile.py 4 16 4 64 > gpurun_out/r3b_steps64_w2.txt 2>&1; echo "prof64 rc=$?"; tail -1 gpurun_out/r3b_steps64_w2.txt
# same kernels at 3 waves per SIMD (spilling build) for comparison
cd music-style-transfer_amd && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPSA_BWD_MINW=3 -c csrc/notes.hip -o build/notes.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libmst_amd.so build/gemm.o build/lstm.o build/combine.o build/notes.o build/loss_optim.o build/plan.o && cd ..
timeout -k 10 300 python tools/step_profile.py 4 16 4 64 > gpurun_out/r3b_steps64_w3.txt 2>&1; echo "prof64w3 rc=$?"; tail -1 gpurun_out/r3b_steps64_w3.txt
grep -E "psa_notes|rowlin|segred|loss" gpurun_out/r3b_steps64_w2.txt gpurun_out/r3b_steps64_w3.txt
cd music-style-transfer_amd && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c csrc/notes.hip -o build/notes.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libmst_amd.so build/gemm.o build/lstm.o build/combine.o build/notes.o build/loss_optim.o build/plan.o && cd ..
timeout -k 10 400 python bench.py --steps 200 --warmup 20 > gpurun_out/r3b_bench.json 2> gpurun_out/r3b_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r3b_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3b_bench.json').read().strip().splitlines()[-1])
b = d.get('batched') or {}
print('one clip: %.0f it/s %.3f ms spread %s upload %s' % (d['value'], d['ms_per_step'], d.get('spread'), d.get('value_with_upload')))
print('surface', (d.get('surface') or {}).get('value'), (d.get('surface') or {}).get('fused_value'))
if b: print('batched: %.0f clip-it/s, %.2f ms/pass; %s' % (b['value'], b['ms_per_pass'], b.get('spread')))
PY
