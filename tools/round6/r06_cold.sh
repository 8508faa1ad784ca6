#!/bin/bash
# the driver's command as the FIRST GPU work of a fresh box, as its own process, then again (warm box),
# then the per-step series tool.  Everything lands in gpurun_out/cold_<tag>/.
tag=${1:-a}
out=gpurun_out/cold_$tag
mkdir -p $out
rocm-smi --showclocks --showperflevel --showpower > $out/smi_before.txt 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/run1.json 2> $out/run1.err
rocm-smi --showclocks --showperflevel --showpower > $out/smi_after1.txt 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/run2.json 2> $out/run2.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/run3.json 2> $out/run3.err
python3 tools/step_series.py 60 > $out/step_series.log 2>&1
python3 - <<'PY' $out
import json, sys, os
out = sys.argv[1]
for r in ('run1', 'run2', 'run3'):
    try:
        d = json.loads(open(os.path.join(out, r + '.json')).read().strip().splitlines()[-1])
        print(r, 'ms_per_step', d['ms_per_step'], 'steady', (d.get('steady_state') or {}).get('ms_per_step'))
        print('  warm', d['warmup_steps_ms'])
        print('  timed', d['timed_steps_ms'])
        print('  host', d['timed_steps_host_enqueue_ms'])
    except Exception as e:
        print(r, 'failed', e)
PY
