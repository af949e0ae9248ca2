# per-kernel SQ counters of the 64-clip pass (wave-parked / issue-stall / active split, LDS conflicts, MFMA busy)
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-sq}
export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/${T}_a -- python bench.py --clips-per-gpu 64 --steps 128 --warmup 64 --no-cpu-baseline --no-roofline --audio-steps 0 > /dev/null 2> gpurun_out/${T}_a.err; echo "a rc=$?"
python tools/pmc_kernel.py gpurun_out/${T}_a > gpurun_out/${T}_a.txt 2>&1
timeout -k 10 500 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/${T}_b -- python bench.py --clips-per-gpu 64 --steps 128 --warmup 64 --no-cpu-baseline --no-roofline --audio-steps 0 > /dev/null 2> gpurun_out/${T}_b.err; echo "b rc=$?"
python tools/pmc_kernel.py gpurun_out/${T}_b > gpurun_out/${T}_b.txt 2>&1
rm -rf gpurun_out/${T}_a gpurun_out/${T}_b
head -12 gpurun_out/${T}_a.txt
