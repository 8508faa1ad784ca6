#!/bin/bash
# separate-process A/B of environment knobs on the conditional-Glow leg: bash tools/cglow_ab.sh "X=1" "PDES_FIN_ONLOAD=0" ...
for kv in "$@"; do
  printf "%-22s " $kv
  env $kv python bench.py --leg cglow --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done
