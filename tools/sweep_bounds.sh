#!/bin/bash
# On the GPU box: rebuild the library with different tuning macros and bench each (A/B in one session).
# usage: tools/sweep_bounds.sh "<-D flags for variant 1>" "<-D flags for variant 2>" ...
set -e
SRC=video-fingerprinting_amd/csrc/offmark_kernels.hip
LIB=video-fingerprinting_amd/offmark/_lib/liboffmark_hip.so
cp $LIB /tmp/lib_orig.so
for cfg in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared $cfg $SRC -o $LIB
  echo "== $cfg" >> gpurun_out/sweep.log
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline $BENCH_ARGS >> gpurun_out/sweep.log 2>>gpurun_out/sweep.err
done
cp /tmp/lib_orig.so $LIB
