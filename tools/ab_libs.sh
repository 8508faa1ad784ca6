#!/bin/bash
# separate-process A/B of builds of the library on ONE box: bash tools/ab_libs.sh main nomfma main nomfma ...
# (ab/lib_<name>.so; restores ab/lib_main.so at the end).  PDES_* knobs in the environment apply to every run.
for v in "$@"; do
  cp ab/lib_$v.so pde_surrogate_amd/libpdes_hip.so
  printf "%s: " $v
  python bench.py --steps 200 --warmup 60 --no-cpu-baseline --no-extras 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*" | head -1
done
cp ab/lib_main.so pde_surrogate_amd/libpdes_hip.so
