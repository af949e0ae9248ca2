set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; T=${1:-r3s}
for f in 0 2 1; do
echo "== dense_flavour $f"; MST_DENSE_FLAVOUR=$f timeout -k 10 300 python bench.py --clips-per-gpu 64 --steps 640 --warmup 64 --no-cpu-baseline --no-roofline > gpurun_out/${T}_$f.json 2> gpurun_out/${T}_$f.err; echo "rc=$?"; python -c "import json;d=json.loads(open('gpurun_out/${T}_$f.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step']*64)"
done
MST_DENSE_FLAVOUR=2 timeout -k 10 300 python tools/step_profile.py 4 16 4 64 > gpurun_out/${T}_steps64_dense2.txt 2>&1; tail -1 gpurun_out/${T}_steps64_dense2.txt
