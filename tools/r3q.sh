set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; T=${1:-r3q}
A="--steps 400 --warmup 40 --no-cpu-baseline --batched-clips 0 --surface-steps 0 --audio-steps 0 --no-roofline"
for g in per-lane joint; do
echo "== graphs $g"; timeout -k 10 200 python bench.py $A --graphs $g > gpurun_out/${T}_$g.json 2> gpurun_out/${T}_$g.err; echo "rc=$?"; python -c "import json;d=json.loads(open('gpurun_out/${T}_$g.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['spread'],d.get('value_with_upload'),d['config']['final_total_loss'])"
done
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --share-device --steps 40 --warmup 4 > gpurun_out/${T}_dp2.json 2> gpurun_out/${T}_dp2.err; echo "dp2 rc=$?"; tail -c 700 gpurun_out/${T}_dp2.json
