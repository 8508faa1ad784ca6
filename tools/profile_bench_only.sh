#!/bin/bash
# rocprofv3 evidence of one round, run ON THE GPU BOX from the repo root:
#   gpurun -- 'bash tools/profile_bench_only.sh r02_k'   (part (A) of tools/profile_round.sh only: the bench command under the kernel trace)
# writes summaries under gpurun_out/<tag>/ (copy the ones to be judged into profiles/).  Counter passes (--pmc) run
# separately from the kernel-trace/stats pass, each with --kernel-trace only (MI355X_MICROARCH.md).
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
db() { find $1 -name "*.db" | head -1; }

# (A) the bench command: kernel trace + stats, step timeline, loss-kernel launches
rocprofv3 --kernel-trace --stats --output-format rocpd -d $OUT/bench -o bench -- \
    python $ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/bench.err
D=$(db $OUT/bench)
python $ROOT/tools/rocprof_summary.py $D > $OUT/kernel_stats.csv
python $ROOT/tools/timeline.py $D > $OUT/step_timeline.txt
python $ROOT/tools/rocprof_summary.py $D --by-grid darcy_loss > $OUT/loss_kernel_by_batch.csv
python $ROOT/tools/rocprof_summary.py $D --per-launch "darcy_loss_kernel<64, true, false, false, false>" 8388608 > $OUT/loss_kernel_per_launch.csv
python - $OUT/loss_kernel_per_launch.csv > $OUT/loss_kernel_sustained.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
last = rows[104:204]                    # bench.py `roofline`: 1 + 3 burst + 100 warm-up launches, then the 100 timed ones (the later 200 belong to roofline_variants)
d = [float(r['duration_us']) for r in last]
avg = sum(d) / len(d)
b = 114688 * 16384
print(f'# darcy_loss_kernel<64,true,false>, B = 16384: dispatches 104 .. 203 of {len(rows)} of `bench.py` (the HIP-event timed ones, after 100+ back-to-back warm-up launches)')
print(f'# avg_us,{avg:.3f},min_us,{min(d):.3f},max_us,{max(d):.3f},algorithmic_bytes,{b},GBps,{b / avg / 1e3:.1f},frac_of_8TBps,{b / avg / 1e3 / 8000:.4f}')
print('launch,start_us,duration_us')
for r in last:
    print(f"{r['launch']},{r['start_us']},{r['duration_us']}")
PY

# (B) HBM traffic of the loss kernel (B = 16384): FETCH_SIZE and WRITE_SIZE in separate passes
rm -rf $OUT/bench
