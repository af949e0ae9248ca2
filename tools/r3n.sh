set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; T=${1:-r3n}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/${T}_tests.log
A="--steps 200 --warmup 20 --no-cpu-baseline --batched-clips 0 --surface-steps 0 --audio-steps 0 --no-roofline"
for br in 0 1; do
echo "== no graph, branches=$br"; MST_BRANCHES=$br timeout -k 10 200 python -X faulthandler bench.py $A --no-graph > gpurun_out/${T}_a$br.json 2> gpurun_out/${T}_a$br.err; echo "rc=$?"; python -c "import json;d=json.loads(open('gpurun_out/${T}_a$br.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])"
done
echo "== graph, branches=0"; timeout -k 10 200 python -X faulthandler bench.py $A > gpurun_out/${T}_b.json 2> gpurun_out/${T}_b.err; echo "rc=$?"; python -c "import json;d=json.loads(open('gpurun_out/${T}_b.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])"
