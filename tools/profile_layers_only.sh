#!/bin/bash
# parts (D) and (E) of tools/profile_round.sh only (ON THE GPU BOX): matrix-pipe occupancy of selected layers' kernels and the
# per-layer stand-alone timings:   gpurun -- 'bash tools/profile_layers_only.sh r02_k'
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
db() { find $1 -name "*.db" | head -1; }
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format rocpd -d $OUT/pmc_mfma -o p -- python $ROOT/tools/bench_conv.py 6,7,16,17,18,24,25,26,27 > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py $(db $OUT/pmc_mfma) > $OUT/pmc_mfma.txt
python $ROOT/tools/bench_conv.py > $OUT/per_layer_conv_microbench.log 2>&1
rm -rf $OUT/pmc_mfma
