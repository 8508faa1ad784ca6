#!/bin/bash
# the driver's round-end sequence on one fresh box: GPU suite (-x -q), smoke(), the bench command; which .so files the python
# processes mapped
mkdir -p gpurun_out/r06_rehearsal
python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/r06_rehearsal/gpu_suite.txt 2>&1; grep -E "passed|failed|^FAILED" gpurun_out/r06_rehearsal/gpu_suite.txt | tail -3
python3 -c "
import __graft_entry__ as g
g.smoke()
import re
print('mapped:', sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'libpdes' in l or 'oracle' in l)))
" 2>&1 | grep -E "smoke|mapped"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_rehearsal/bench.json 2> gpurun_out/r06_rehearsal/bench.err
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r06_rehearsal/bench.json').read().strip().splitlines()[-1])
print('bench', d['metric'], d['value'], d['unit'], d['ms_per_step'], 'steady', d['steady_state']['ms_per_step'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['sample'][:60])
print('timed', d['timed_steps_ms'])
print('keys', sorted(d.keys()))
PY
