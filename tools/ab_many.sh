#!/bin/bash
# Bench several prebuilt libraries in one GPU session, interleaved: tools/ab_many.sh <rounds> <lib>...
LIB=video-fingerprinting_amd/offmark/_lib/liboffmark_hip.so
R=$1; shift
cp $LIB /tmp/lib_orig.so
for r in $(seq 1 $R); do for v in "$@"; do cp $v $LIB; echo "== $v" >> gpurun_out/sweep.log; python bench.py --steps 20 --warmup 3 --no-cpu-baseline $BENCH_ARGS >> gpurun_out/sweep.log 2>>gpurun_out/sweep.err; done; done
cp /tmp/lib_orig.so $LIB
