#!/bin/bash
# Same-box A/B of two builds of libpdes_hip.so.  Different gpurun boxes differ by ~2 % in step time, the same box
# repeats to ~0.1 %, so only comparisons inside ONE gpurun call mean anything at the 1 % level.
#   mkdir -p ab; build variant A; cp pde_surrogate_amd/libpdes_hip.so ab/lib_old.so; build B; cp ... ab/lib_new.so
#   gpurun -- 'bash tools/archive/ab_bench.sh'        (ab/ is git-ignored but travels to the box)
run() { python bench.py --steps 150 --warmup 30 --no-cpu-baseline 2>&1 | grep -o "\"ms_per_step\": [0-9.]*"; }
for v in old new old new; do cp ab/lib_$v.so pde_surrogate_amd/libpdes_hip.so; echo $v; run; done
