#!/bin/bash
# usage: tools/ladder_pmc.sh <tag>   (on the GPU box, through gpurun)
# Timing table of tools/bin/probe_ladder, then PMC passes of its "pmc" mode (two launches of every rung) in both tile orders.
# Counters in their own passes, no trace domains beside --pmc.  Output: gpurun_out/ladder_<tag>/
set -e
TAG=${1:-r6}
ROOT=$PWD
OUT=$ROOT/gpurun_out/ladder_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 $ROOT/tools/bin/probe_ladder 300 6 10 > $OUT/timing.txt 2> $OUT/timing.err
cd /tmp
rocprofv3 -L > $OUT/counters_available.txt 2>&1 || true
for ORDER in 8 0; do
  P="$ROOT/tools/bin/probe_ladder pmc $ORDER 300"
  timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/o${ORDER}_sq -o pmc -- $P > $OUT/o${ORDER}_sq.log 2>&1 || echo "pass sq order $ORDER failed"
  timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/o${ORDER}_tcc -o pmc -- $P > $OUT/o${ORDER}_tcc.log 2>&1 || echo "pass tcc order $ORDER failed"
  timeout -k 10 300 rocprofv3 --pmc TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum SQ_INSTS_VMEM_WR --output-format csv -d $OUT/o${ORDER}_stall -o pmc -- $P > $OUT/o${ORDER}_stall.log 2>&1 || echo "pass stall order $ORDER failed"
done
cd $ROOT
python3 tools/ladder_summary.py $OUT > $OUT/summary.txt 2> $OUT/summary.err || echo "summary failed"
tail -3 $OUT/timing.err
