#!/bin/bash
# SQ counters of the dense 32x32 weight-gradient kernel (the top kernel of the step) collected INSIDE the training step
# (bench.py --no-extras) and stand-alone (tools/bench_conv.py 24), in separate small passes, ON THE GPU BOX:
#   bash tools/pmc_wgrad_instep.sh r06_c
# (rocprofv3 --pmc serialises the dispatches: "in the step" means the step's data, cache state and launch order, not its overlap)
TAG=${1:-r06}; PAT=${2:-'conv_mfma_wgrad_kernel<3, 2, 1, 1, true, false, true>'}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
: > $OUT/wgrad_dense_pmc.txt
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  for WHERE in step alone; do
    rm -rf $OUT/p
    if [ $WHERE = step ]; then
      rocprofv3 --pmc $SET --kernel-trace --output-format rocpd -d $OUT/p -o p -- python $ROOT/bench.py --steps 20 --warmup 10 --no-extras --no-cpu-baseline > /dev/null 2>$OUT/err_$WHERE$i.txt
    else
      rocprofv3 --pmc $SET --kernel-trace --output-format rocpd -d $OUT/p -o p -- env BENCH_FUSED=1 python $ROOT/tools/bench_conv.py 24 > /dev/null 2>$OUT/err_$WHERE$i.txt
    fi
    echo "== $WHERE: $SET" >> $OUT/wgrad_dense_pmc.txt
    python $ROOT/tools/pmc_summary.py $(find $OUT/p -name "*.db" | head -1) | grep -F -A5 "$PAT" >> $OUT/wgrad_dense_pmc.txt
    rm -rf $OUT/p
  done
done
cat $OUT/wgrad_dense_pmc.txt
