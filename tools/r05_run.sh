#!/bin/bash
# one GPU-box session of round 5: bash tools/r05_run.sh <tag> <steps...>   (steps: suite ab conv bench)
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export PDES_REQUIRE_GPU=1
for S in "$@"; do
  case $S in
    suite) timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/gpu_suite.txt 2>&1; echo "suite rc=$?" | tee -a $OUT/status.txt; tail -3 $OUT/gpu_suite.txt ;;
    quick) timeout 900 python -m pytest tests/test_loss_gpu.py tests/test_densed_gpu.py -m gpu -x -q -p no:cacheprovider -k "dropin or g23 or g11 or variants or determinism or g2_g3" > $OUT/gpu_quick.txt 2>&1; echo "quick rc=$?" | tee -a $OUT/status.txt; tail -3 $OUT/gpu_quick.txt ;;
    ab) timeout 600 python tools/ab_env.py $AB_ARGS > $OUT/ab_env.log 2>&1; echo "ab rc=$?" | tee -a $OUT/status.txt; cat $OUT/ab_env.log | grep -v Warning ;;
    conv) for M in 1 2; do PDES_WGRAD_MTW=$M timeout 300 python tools/bench_conv.py 1,2,3,4,5,6,19,20,21,22,23,24 > $OUT/bench_conv_mtw$M.log 2>&1; done; echo "conv rc=$?" | tee -a $OUT/status.txt; paste -d'\n' $OUT/bench_conv_mtw1.log $OUT/bench_conv_mtw2.log | grep wgrad | cut -c1-60,100-130 ;;
    bench) timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/status.txt; python - $OUT/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value', 'ms_per_step')}, 'roofline', d['roofline']['frac'])
for k in ('config4_channelized', 'dropin', 'segment_graphs', 'dp1_rccl'):
    print(k, d.get(k))
print('cglow', {k: d.get('cglow_reverse_kl', {}).get(k) for k in ('ms_per_step', 'samples_per_s', 'error')})
print('cpu', {k: d.get('cpu_baseline', {}).get(k) for k in ('value', 'cores', 'cpus_allowed', 'loss_only_samples_per_s')})
PY
      tail -12 $OUT/bench.err ;;
    loss) timeout 900 python -m pytest tests/test_loss_gpu.py -m gpu -x -q -p no:cacheprovider > $OUT/gpu_loss.txt 2>&1; echo "loss rc=$?" | tee -a $OUT/status.txt; tail -5 $OUT/gpu_loss.txt ;;
    gensizes) timeout 600 python tools/bench_loss_generic.py 2>&1 | grep -v amdgpu.ids > $OUT/loss_kernel_generic_sizes.log; echo "gensizes rc=$?" | tee -a $OUT/status.txt; cat $OUT/loss_kernel_generic_sizes.log ;;
    cglow) timeout 900 python -m pytest tests/test_cglow_gpu.py -m gpu -x -q -p no:cacheprovider > $OUT/gpu_cglow.txt 2>&1; echo "cglow rc=$?" | tee -a $OUT/status.txt; tail -5 $OUT/gpu_cglow.txt ;;
    cglowb) for i in 1 2; do timeout 600 python bench.py --leg cglow --no-cpu-baseline 2> $OUT/cglow_bench.err | tee $OUT/cglow_bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k: d.get(k) for k in ('ms_per_step','samples_per_s','dispatches_per_step')})"; done ;;
    benchq) timeout 600 python bench.py --no-extras --no-cpu-baseline > $OUT/bench_quick.json 2> $OUT/bench_quick.err; echo "benchq rc=$?" | tee -a $OUT/status.txt; python -c "
import json,sys; d=json.loads(open('$OUT/bench_quick.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])" ;;
  esac
done
