#!/bin/bash
# separate-process A/B of HIP runtime environment knobs on ONE box (they are read when the runtime initialises, so
# they cannot be flipped inside a process): the default bench step, alternating, three rounds
#   bash tools/archive/ab_hipenv.sh "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" ...
cd "$(dirname "$0")/.."
for r in 1 2 3; do
  for cfg in "$@"; do
    v=$(env $cfg python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 |
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
    echo "round $r  [$cfg]  ms/step, samples/s: $v"
  done
done
