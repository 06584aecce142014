#!/bin/bash
# usage: tools/prof.sh <tag> [bench args]   (run on the GPU box through gpurun; writes under gpurun_out/prof_<tag>)
# One kernel-trace pass and four PMC passes of the same bench command (separate passes: no trace domains with --pmc),
# then tools/make_profile_summary.py turns them into the three files that get copied into profiles/:
#   <tag>_kernel_stats.csv, <tag>_pmc_summary.txt, <tag>_traffic.json (stamped with the hash of the kernel sources).
set -e
TAG=${1:-r2}
shift || true
OUT=$PWD/gpurun_out/prof_$TAG
ROOT=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 5 --warmup 2 --preheat-ms 0 --no-cpu-baseline --no-kernel-events --no-extras $*"
# the kernel-trace pass runs enough steps for its per-kernel AVERAGE to describe the device at its operating clocks (a 5-step run from idle
# sits on the clock ramp: round 2's 8-call averages were 6-8 % above the bench line's in-run event averages), bench.py's own pre-heat included
# (its launches are in the average: the same kernels on the same data); counters do not care and skip the pre-heat
TRACE="python3 $ROOT/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-kernel-events --no-extras $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $TRACE > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o pmc -- $BENCH > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1
cd $ROOT
python3 tools/make_profile_summary.py $OUT $TAG "$TRACE (kernel trace); $BENCH (counters)"
