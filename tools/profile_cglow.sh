#!/bin/bash
# the conditional-Glow part of tools/profile_round.sh alone: gpurun -- 'bash tools/profile_cglow.sh r05_h'
TAG=${1:-r05}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format rocpd -d $OUT/cglow -o cglow -- \
    python $ROOT/bench.py --leg cglow --no-cpu-baseline > $OUT/cglow_bench.json 2> $OUT/cglow_bench.err
D=$(find $OUT/cglow -name "*.db" | head -1)
python $ROOT/tools/rocprof_summary.py $D > $OUT/cglow_kernel_stats.csv
python $ROOT/tools/timeline.py $D > $OUT/cglow_step_timeline.txt
rm -rf $OUT/cglow
tail -2 $OUT/cglow_step_timeline.txt
