#!/bin/bash
# the driver's short command as fresh processes: contract window vs steady state, control-group throttling, host stalls
tag=${1:-a}; n=${2:-8}
out=gpurun_out/stall_$tag; mkdir -p $out
for k in $(seq 1 $n); do
  $ENVX python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/run$k.json 2> $out/run$k.err
done
python3 - <<'PY' $out $n
import json, sys, os
out, n = sys.argv[1], int(sys.argv[2])
for k in range(1, n + 1):
    try:
        d = json.loads(open(os.path.join(out, f'run{k}.json')).read().strip().splitlines()[-1])
        h = d['timed_steps_host_enqueue_ms']
        print(f'run{k}', 'ms_per_step', d['ms_per_step'], 'steady', (d.get('steady_state') or {}).get('ms_per_step'),
              'max host', max(h[1:]), 'max gpu', max(d['timed_steps_ms'][1:]), 'throttle', d['cgroup_cpu_throttled_during_steps'],
              'threads', d['host_threads'].get('threads'))
    except Exception as e:
        print(f'run{k}', 'failed', e)
PY
