#!/bin/bash
# kernel trace of the drop-in loop (tools/dropin_phases.py): timeline of one step, ON THE GPU BOX
TAG=${1:-r06_dropin}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
DROPIN_INNER=0 rocprofv3 --kernel-trace --stats --output-format rocpd -d $OUT/db -o dropin -- python $ROOT/tools/dropin_phases.py 60 > $OUT/dropin_phases.log 2>&1
D=$(find $OUT/db -name "*.db" | head -1)
python $ROOT/tools/timeline.py $D 400 index_elementwise first > $OUT/dropin_step_timeline.txt 2>&1 || python $ROOT/tools/timeline.py $D 400 gather first > $OUT/dropin_step_timeline.txt
python $ROOT/tools/rocprof_summary.py $D > $OUT/dropin_kernel_stats.csv
rm -rf $OUT/db
grep -v amdgpu $OUT/dropin_phases.log; tail -3 $OUT/dropin_step_timeline.txt; grep -v "pdes::" $OUT/dropin_step_timeline.txt | head -60
