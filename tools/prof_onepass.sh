set -e
OUT=$PWD/gpurun_out/prof_op
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 $PWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events --no-extras"
cd /tmp
for mode in two one; do
  A=""; [ $mode = one ] && A="--onepass 0"
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/${mode}_sq -o pmc -- $B $A > $OUT/${mode}_sq.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/${mode}_sq2 -o pmc -- $B $A > $OUT/${mode}_sq2.log 2>&1
done
