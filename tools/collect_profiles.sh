#!/bin/bash
# Collect the round's profiles on the GPU box (one gpurun call): rocprofv3 kernel stats of the bench commands, the two
# PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs as MI355X_MICROARCH.md prescribes), step profiles.
# Usage: bash tools/collect_profiles.sh r03   (writes gpurun_out/<tag>_*; copy what you want judged into profiles/)
set -o pipefail
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
ONE="bench.py --steps 200 --warmup 20 --no-cpu-baseline --batched-clips 0 --surface-steps 0 --audio-steps 0"
B64="bench.py --clips-per-gpu 64 --steps 640 --warmup 64 --no-cpu-baseline"
run() { echo "== $*" >&2; timeout -k 10 400 "$@"; echo "rc=$?" >&2; }
run rocprofv3 --kernel-trace --stats --output-format csv -d $O/p1 -- python $ONE > $O/${TAG}_bench_under_rocprof.json 2> $O/${TAG}_p1.err
cp $O/p1/*/*kernel_stats.csv $O/${TAG}_rocprofv3_kernel_stats.csv
python tools/prof_summary.py $O/p1 220 40 > $O/${TAG}_kernel_summary_per_iteration.txt
python tools/pass_timeline.py $O/p1 > $O/${TAG}_pass_timeline_one_clip.txt 2>&1
run rocprofv3 --kernel-trace --stats --output-format csv -d $O/p64 -- python $B64 > $O/${TAG}_bench_64_under_rocprof.json 2> $O/${TAG}_p64.err
cp $O/p64/*/*kernel_stats.csv $O/${TAG}_rocprofv3_kernel_stats_64_clips.csv
python tools/pass_timeline.py $O/p64 > $O/${TAG}_pass_timeline_64_clips.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  run rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pm1_$c -- python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-graph --no-roofline --batched-clips 0 --surface-steps 0 --audio-steps 0 > /dev/null 2> $O/${TAG}_pm1_$c.err
  run rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pm64_$c -- python bench.py --clips-per-gpu 64 --steps 128 --warmup 64 --no-cpu-baseline --no-roofline > /dev/null 2> $O/${TAG}_pm64_$c.err
done
python tools/pmc_traffic.py $O/pm1_FETCH_SIZE $O/pm1_WRITE_SIZE $O/${TAG}_pmc_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-graph --no-roofline --batched-clips 0 --surface-steps 0 --audio-steps 0 (two separate passes)" 1
python tools/pmc_traffic.py $O/pm64_FETCH_SIZE $O/pm64_WRITE_SIZE $O/${TAG}_pmc_hbm_traffic_64_clips.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --clips-per-gpu 64 --steps 128 --warmup 64 --no-cpu-baseline --no-roofline (two separate passes; per_iteration = per clip-iteration)" 64
# bench lines kept beside the profiles: the default line, 32 / 64 clips per launch, the tiled long clip, VALU : MFMA per GEMM launch
run python bench.py --steps 200 --warmup 20 > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
run python bench.py --clips-per-gpu 32 --steps 640 --warmup 64 --no-cpu-baseline > $O/${TAG}_bench_clips32.json 2> /dev/null
run python bench.py --clips-per-gpu 64 --steps 1280 --warmup 128 --no-cpu-baseline > $O/${TAG}_bench_clips64.json 2> /dev/null
run python bench.py --tile-bars --steps 40 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench_tile_bars.json 2> /dev/null
run rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmcv -- python bench.py --clips-per-gpu 64 --steps 128 --warmup 64 --no-cpu-baseline --no-roofline > /dev/null 2> $O/${TAG}_pmcv.err
python tools/pmc_kernel.py $O/pmcv > $O/${TAG}_pmc_valu_mfma_64_clips.txt 2>&1
python tools/step_profile.py > $O/${TAG}_step_profile_single_iteration.txt 2>&1
python tools/step_profile.py 4 16 4 64 > $O/${TAG}_step_profile_64_clips.txt 2>&1
# audio extension (not reference parity): kernel stats of its STFT / Gram / iteration kernels
run rocprofv3 --kernel-trace --stats --output-format csv -d $O/pa -- python tools/audio_profile.py > $O/${TAG}_audio_profile.log 2>&1
cp $O/pa/*/*kernel_stats.csv $O/${TAG}_rocprofv3_kernel_stats_audio_extension.csv
# N > 1 rehearsals on this one GPU (gloo, every rank on cuda:0): data parallel and the tiled long clip
export HSA_ENABLE_IPC_MODE_LEGACY=0
run python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --share-device --steps 40 --warmup 4 > $O/${TAG}_rehearsal_dp2_gloo_share_device.json 2> /dev/null
run python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --backend gloo --share-device --tile-bars --steps 10 --warmup 2 > $O/${TAG}_rehearsal_tile2_gloo_share_device.json 2> /dev/null
rm -rf $O/p1 $O/p64 $O/pm1_* $O/pm64_* $O/pmcv $O/pa
ls -la $O | grep ${TAG}_ | head -30
