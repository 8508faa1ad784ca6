#!/bin/bash
# host stalls per 300 steps under different runtime settings (fresh process each)
out=gpurun_out/stall2; mkdir -p $out
run() { tag=$1; shift; env "$@" PDES_BENCH_TRACE=1 python3 bench.py --gpus 1 --steps 75 --warmup 5 --no-extras --no-cpu-baseline > $out/$tag.json 2> $out/$tag.err; }
for k in 1 2 3 4; do run base$k X=1; done
for k in 1 2 3 4; do run nosig$k PDES_FORK_SIGNAL=0; done
for k in 1 2 3 4; do run gc$k PDES_BENCH_GCFREEZE=1; done
python3 - <<'PY' $out
import json, sys, os, glob
out = sys.argv[1]
for f in sorted(glob.glob(out + '/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        h = d['timed_steps_host_enqueue_ms']; g = d['timed_steps_ms']
        ser = d.get('host_phase_series_ms_fwd_loss_bwd') or []
        big = [(i, s_) for i, s_ in enumerate(ser) if max(s_) > 2.0 and i > 0]
        print(os.path.basename(f), 'ms', d['ms_per_step'], 'stalls>2ms', [(i, round(v, 1)) for i, v in enumerate(h) if v > 2.0 and i > 0],
              'gpu>1.9', [(i, round(v, 2)) for i, v in enumerate(g) if v > 1.9 and i > 0], 'phases', big, 'gc', d['python_gc_collections_during_steps'])
    except Exception as e:
        print(f, 'failed', e)
PY
