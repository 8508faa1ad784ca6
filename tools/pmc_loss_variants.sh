#!/bin/bash
# HBM traffic of every loss-path kernel SURVEY 8(d) prices (tools/bench_loss.py VARIANTS) at B = 16,384, ON THE GPU BOX:
#   gpurun -- 'bash tools/pmc_loss_variants.sh r06_a'
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (they do not fit one pass; --kernel-trace only beside them,
# MI355X_MICROARCH.md), one process per variant and counter.  Writes gpurun_out/<tag>/loss_variants_pmc.json (copy it to
# profiles/<tag>_loss_variants_pmc.json) and the per-variant summary text.
TAG=${1:-r06}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
: > $OUT/loss_variants_pmc.txt
for V in loss_fwd_bwd loss_fwd_only loss_nonlinear_fwd_bwd sobel_grad sobel_grad_adjoint; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/pmc_${V}_$C
    rocprofv3 --pmc $C --kernel-trace --output-format rocpd -d $OUT/pmc_${V}_$C -o p -- python $ROOT/tools/bench_loss.py pmc $V > $OUT/pmc_run.log 2>&1
    echo "== $V $C" >> $OUT/loss_variants_pmc.txt
    python $ROOT/tools/pmc_summary.py $(find $OUT/pmc_${V}_$C -name "*.db" | head -1) --json $V $C >> $OUT/loss_variants_pmc.txt
    rm -rf $OUT/pmc_${V}_$C
  done
done
python $ROOT/tools/pmc_summary.py --collect $OUT/loss_variants_pmc.txt > $OUT/loss_variants_pmc.json
cat $OUT/loss_variants_pmc.json
