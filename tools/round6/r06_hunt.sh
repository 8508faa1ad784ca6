#!/bin/bash
out=gpurun_out/hunt_${1:-a}; mkdir -p $out
n=${2:-16}
lscpu | grep -i "numa\|socket\|model name" > $out/lscpu.txt
for k in $(seq 1 $n); do
  python3 tools/diag/stall_hunt.py 45 > $out/base$k.log 2>&1
  HUNT_PIN=0 python3 tools/diag/stall_hunt.py 45 > $out/nopin$k.log 2>&1
  HUNT_MEMPOL=1 python3 tools/diag/stall_hunt.py 45 > $out/mempol$k.log 2>&1
done
cat $out/lscpu.txt
grep -H "host stalls\|numa_balancing\|mempolicy" $out/*.log | sed 's/.*hunt_[a-z]*\///' | cut -c1-330
