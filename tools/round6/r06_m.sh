#!/bin/bash
# smoke() and the default bench.py (no flags), timed; the driver's form right after
mkdir -p gpurun_out/r06_m
( time python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) 2>&1 | grep -E "smoke|real"
( time python3 bench.py > gpurun_out/r06_m/bench_default.json 2> gpurun_out/r06_m/bench_default.err ) 2>&1 | grep real
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_m/bench_driver.json 2> gpurun_out/r06_m/bench_driver.err ) 2>&1 | grep real
python3 - <<'PY'
import json
for f in ('bench_default', 'bench_driver'):
    d = json.loads(open(f'gpurun_out/r06_m/{f}.json').read().strip().splitlines()[-1])
    print(f, d['ms_per_step'], d['value'], 'steady', (d.get('steady_state') or {}).get('ms_per_step'), 'roofline', d['roofline']['frac'], 'traffic', d['roofline']['traffic'],
          'gc', d.get('python_gc_collections_during_steps'), 'dropin', d['dropin'].get('ms_per_step'), d['dropin'].get('ms_per_step_without_loss_item'), 'cpu', d['cpu_baseline'].get('value'))
    print('  variants', {k: v['frac'] for k, v in d['roofline_variants'].items() if isinstance(v, dict) and 'frac' in v})
PY
tail -25 gpurun_out/r06_m/bench_default.err | grep "\[bench\]" | tail -20
