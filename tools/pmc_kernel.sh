#!/bin/bash
# SQ counters of one layer's kernels (tools/bench_conv.py <layer>), in separate small passes (ON THE GPU BOX):
#   bash tools/pmc_kernel.sh 25 'conv_wgrad_b3|conv_mfma_b3'      (layer index of tools/bench_conv.py, kernel-name regex)
LAYER=${1:-25}; PAT=${2:-wgrad_b3}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_kernel; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rm -rf $OUT/p$i
  rocprofv3 --pmc $SET --kernel-trace --output-format rocpd -d $OUT/p$i -o p -- python $ROOT/tools/bench_conv.py $LAYER > /dev/null 2>$OUT/err$i.txt
  python $ROOT/tools/pmc_summary.py $(find $OUT/p$i -name "*.db" | head -1) | grep -A4 -E "$PAT"
  rm -rf $OUT/p$i
done
