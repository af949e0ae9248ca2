#!/bin/bash
# Build the product's .hip sources against the hipsim CPU interpreter (TEST TOOLING ONLY).
# Output: tests/hipsim/libmst_sim.so (+ _asan variant with ASAN=1). Never shipped, never loaded
# by the product package.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/../../music-style-transfer_amd/csrc"
CXX=/opt/rocm/lib/llvm/bin/clang++
[ -x "$CXX" ] || CXX=clang++
OUT="$HERE/libmst_sim.so"
FLAGS="-O2"
if [ "$ASAN" = "1" ]; then OUT="$HERE/libmst_sim_asan.so"; FLAGS="-O1 -g -fsanitize=address -shared-libasan"; fi
OBJS=""
for f in gemm lstm combine notes loss_optim plan audio conv lin; do
  $CXX -x c++ -std=c++17 $FLAGS -fPIC -I"$HERE" -Wall -Wno-unused-function -Wno-unknown-pragmas -Wno-unused-variable \
      -c "$SRC/$f.hip" -o "$HERE/$f.sim.o" &
done
wait
for f in gemm lstm combine notes loss_optim plan audio conv lin; do OBJS="$OBJS $HERE/$f.sim.o"; done
$CXX -shared $FLAGS -o "$OUT" $OBJS
echo "built $OUT"
