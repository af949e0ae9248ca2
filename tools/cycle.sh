#!/bin/bash
# developer loop: rebuild the interpreter library, run the CPU parity suite, (optionally) ASan, rebuild for gfx950
set -e
cd /root/repo
tests/hipsim/build.sh 2>&1 | grep -E "error|built" | head -8
timeout 1500 python -m pytest tests/test_sim_parity.py -x -q 2>&1 | tail -2
if [ "$1" = "asan" ]; then
  ASAN=1 tests/hipsim/build.sh >/dev/null 2>&1
  LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 timeout 1500 python tests/hipsim/asan_check.py 2>&1 | tail -1
fi
python __graft_entry__.py | tail -1
