#!/bin/bash
# separate-process A/B of builds of the library on ONE box, loss-path kernels at B = 16384: bash tools/ab_loss_libs.sh main plain main plain
for v in "$@"; do
  cp ab/lib_$v.so pde_surrogate_amd/libpdes_hip.so
  printf "%s: " $v
  python tools/bench_loss.py 2>/dev/null | grep '"batch": 16384' | python -c "
import sys, json
print(' '.join('%s %.1f' % (json.loads(l)['variant'][5:12], json.loads(l)['us_per_launch']) for l in sys.stdin))"
done
cp ab/lib_main.so pde_surrogate_amd/libpdes_hip.so
