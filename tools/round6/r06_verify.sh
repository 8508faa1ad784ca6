#!/bin/bash
# round-6 checkpoint on a fresh box: the driver's bench command FIRST (cold box), then the GPU suite, then the default bench.
tag=${1:-r06_d}
out=gpurun_out/$tag
mkdir -p $out
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err
python3 -m pytest tests -m gpu -x -q -p no:cacheprovider > $out/gpu_suite.txt 2>&1
tail -3 $out/gpu_suite.txt
python3 - <<'PY' $out
import json, sys, os
d = json.loads(open(os.path.join(sys.argv[1], 'bench_driver.json')).read().strip().splitlines()[-1])
print('driver window ms_per_step', d['ms_per_step'], 'value', d['value'], 'steady', (d.get('steady_state') or {}).get('ms_per_step'))
print('timed', d['timed_steps_ms'])
print('roofline', d['roofline'])
PY
