#!/bin/bash
out=gpurun_out/hunt_${1:-d}; mkdir -p $out
n=${2:-12}
for k in $(seq 1 $n); do
  HUNT_LINK=1 python3 tools/diag/stall_hunt.py 45 > $out/link$k.log 2>&1
  HIP_FORCE_DEV_KERNARG=0 python3 tools/diag/stall_hunt.py 45 > $out/hostarg$k.log 2>&1
done
grep -H "host stalls\|link log\|stall wall" $out/*.log | sed 's/.*hunt_[a-z]*\///' | cut -c1-400
