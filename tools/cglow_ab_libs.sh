#!/bin/bash
# separate-process A/B of library builds on the conditional-Glow leg (ab/lib_<name>.so): bash tools/cglow_ab_libs.sh new prev new prev
cp pde_surrogate_amd/libpdes_hip.so /tmp/lib_keep.so
for v in "$@"; do
  cp ab/lib_$v.so pde_surrogate_amd/libpdes_hip.so
  printf "%-8s " $v
  python bench.py --leg cglow --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done
cp /tmp/lib_keep.so pde_surrogate_amd/libpdes_hip.so
