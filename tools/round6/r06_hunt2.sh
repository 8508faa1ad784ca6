#!/bin/bash
out=gpurun_out/hunt_${1:-c}; mkdir -p $out
n=${2:-12}
ps -eo pid,ppid,etimes,pcpu,args --sort=-pcpu | head -40 > $out/ps.txt
( while true; do rocm-smi --showpower --showclocks --showuse --showmemuse --json > /dev/null 2>&1; done ) &
SMI=$!
for k in $(seq 1 $n); do python3 tools/diag/stall_hunt.py 45 > $out/smi$k.log 2>&1; done
kill $SMI; wait $SMI 2>/dev/null
for k in $(seq 1 $n); do python3 tools/diag/stall_hunt.py 45 > $out/quiet$k.log 2>&1; done
cat $out/ps.txt | cut -c1-200
grep -H "host stalls" $out/*.log | sed 's/.*hunt_[a-z]*\///' | cut -c1-250
