set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r3e}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/${T}_tests.log
timeout -k 10 400 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/${T}_bench.err
python - <<PY
import json
d = json.loads(open('gpurun_out/${T}_bench.json').read().strip().splitlines()[-1])
b = d.get('batched') or {}
print('one clip: %.0f it/s %.3f ms upload %s launches %s' % (d['value'], d['ms_per_step'], d.get('value_with_upload'), d['config'].get('launches_per_pass')))
print('surface', (d.get('surface') or {}))
ae = d.get("audio_extension") or {}
print("audio ext", ae.get("value"), ae.get("stft"), ae.get("gram"), ae.get("error"))
if b: print('batched: %.0f clip-it/s, %.2f ms/pass' % (b['value'], b['ms_per_pass']))
PY
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --share-device --steps 40 --warmup 4 > gpurun_out/${T}_rehearsal_dp2_gloo.json 2> gpurun_out/${T}_rehearsal_dp2.err; echo "dp2 rc=$?"; tail -c 600 gpurun_out/${T}_rehearsal_dp2_gloo.json
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --backend gloo --share-device --tile-bars --steps 10 --warmup 2 > gpurun_out/${T}_rehearsal_tile2_gloo.json 2> gpurun_out/${T}_rehearsal_tile2.err; echo "tile2 rc=$?"; tail -c 900 gpurun_out/${T}_rehearsal_tile2_gloo.json
timeout -k 10 300 python bench.py --tile-bars --steps 20 --warmup 3 > gpurun_out/${T}_bench_tile_bars.json 2> /dev/null; echo "tile1 rc=$?"; tail -c 900 gpurun_out/${T}_bench_tile_bars.json
