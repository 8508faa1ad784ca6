#!/bin/bash
out=gpurun_out/hunt_${1:-f}; mkdir -p $out
n=${2:-12}
for k in $(seq 1 $n); do
  python3 tools/diag/stall_hunt.py 45 > $out/limit$k.log 2>&1
  PDES_HOST_THREADS=0 python3 tools/diag/stall_hunt.py 45 > $out/nolimit$k.log 2>&1
done
for v in limit nolimit; do
  echo "== $v: runs $(ls $out/$v*.log | wc -l), throttled $(grep -l "'nr_throttled': [1-9]" $out/$v*.log | wc -l), with a stalled step $(grep -l '^step ' $out/$v*.log | wc -l)"
done
grep -h "^step " $out/*.log | head; grep -h "affinity of main" $out/limit1.log $out/nolimit1.log
