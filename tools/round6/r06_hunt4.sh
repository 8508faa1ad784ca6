#!/bin/bash
out=gpurun_out/hunt_${1:-e}; mkdir -p $out
n=${2:-12}
cat /sys/fs/cgroup/cpu.max > $out/cpumax.txt 2>&1; nproc >> $out/cpumax.txt; uptime >> $out/cpumax.txt
cat /proc/sys/kernel/sched_latency_ns /sys/kernel/debug/sched/latency_ns >> $out/cpumax.txt 2>&1
for k in $(seq 1 $n); do
  python3 tools/diag/stall_hunt.py 45 > $out/base$k.log 2>&1
done
cat $out/cpumax.txt
grep -h "^step \|threads by\|cgroup" $out/*.log | sort | uniq -c | sort -rn | cut -c1-420 | head -40
