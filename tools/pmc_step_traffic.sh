#!/bin/bash
# HBM traffic of ONE training step, kernel by kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only
# beside them) over `bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline`; the loss-kernel roofline launches are left out.
#   gpurun -- 'bash tools/pmc_step_traffic.sh r06_z'      -> gpurun_out/<tag>/step_traffic.txt
TAG=${1:-r06}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_step_$C
  rocprofv3 --pmc $C --kernel-trace --output-format rocpd -d $OUT/pmc_step_$C -o p -- python $ROOT/bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline > $OUT/pmc_step_run.log 2>&1
  python $ROOT/tools/pmc_summary.py $(find $OUT/pmc_step_$C -name "*.db" | head -1) > $OUT/step_traffic_$C.txt
  rm -rf $OUT/pmc_step_$C
done
python - $OUT <<'PY' > $OUT/step_traffic.txt
import re, sys, os
out = sys.argv[1]
def load(c):
    res, k = {}, None
    for line in open(os.path.join(out, f'step_traffic_{c}.txt')):
        if line.startswith('    '):
            m = re.match(r'\s+(\S+)\s+mean\s+([\d.]+)\s+median\s+([\d.]+)\s+n=(\d+)', line)
            if m and m.group(1) == c:
                res[k] = (float(m.group(2)), int(m.group(4)))
        elif line.strip():
            k = line.strip()
    return res
f, w = load('FETCH_SIZE'), load('WRITE_SIZE')
steps = None
for k, (m, n) in f.items():
    if 'adam_kernel' in k:
        steps = n
rows = []
for k in sorted(set(f) | set(w)):
    if 'darcy_loss_kernel' in k or 'sobel' in k:
        continue
    n = f.get(k, w.get(k))[1]
    if steps and n % steps:
        continue
    rd = f.get(k, (0, 0))[0] * 1024 * 2 * n / steps      # FETCH_SIZE: KiB, doubled per the gfx950 note (16-byte coalesced reads)
    wr = w.get(k, (0, 0))[0] * 1024 * n / steps
    rows.append((rd + wr, rd, wr, n // steps, k))
rows.sort(reverse=True)
tr, tw = sum(r[1] for r in rows), sum(r[2] for r in rows)
print(f'# HBM traffic per training step (bs 32), {steps} steps: read {tr / 1e6:.1f} MB (FETCH_SIZE x 2: the gfx950 correction for 16-byte coalesced reads -- an UPPER bound for kernels with narrower reads), written {tw / 1e6:.1f} MB, total {(tr + tw) / 1e6:.1f} MB')
print('# MB_total,MB_read,MB_written,launches_per_step,kernel')
for t, rd, wr, n, k in rows:
    print(f'{t / 1e6:.2f},{rd / 1e6:.2f},{wr / 1e6:.2f},{n},{k[:120]}')
PY
head -12 $OUT/step_traffic.txt
