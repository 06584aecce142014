#!/bin/bash
# A/B two prebuilt libraries in one GPU session, interleaved: tools/ab_libs.sh <libA> <libB> [rounds]
LIB=video-fingerprinting_amd/offmark/_lib/liboffmark_hip.so
cp $LIB /tmp/lib_orig.so
for r in $(seq 1 ${3:-3}); do for v in $1 $2; do cp $v $LIB; echo "== $v" >> gpurun_out/sweep.log; python bench.py --steps 20 --warmup 3 --no-cpu-baseline $BENCH_ARGS >> gpurun_out/sweep.log 2>>gpurun_out/sweep.err; done; done
cp /tmp/lib_orig.so $LIB
