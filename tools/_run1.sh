export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/r01_i
O=$R/gpurun_out/r01_i
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null
tail -c 600 $O/bench.json
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof -o p -- python $R/bench.py --no-cpu-baseline > $O/bench_prof.json 2> /tmp/prof.err < /dev/null
f=$(find /tmp/prof -name "*.db" 2>/dev/null | head -1)
echo "db=$f"
if [ -n "$f" ]; then
  timeout 60 python $R/tools/rocprof_summary.py $f > $O/kernel_stats.csv < /dev/null
  timeout 60 python $R/tools/rocprof_summary.py $f --by-grid darcy_loss_kernel > $O/loss_kernel_by_batch.csv < /dev/null
  timeout 60 python $R/tools/timeline.py $f > $O/step_timeline.txt < /dev/null
  head -5 $O/kernel_stats.csv | cut -c1-150; tail -2 $O/step_timeline.txt
else tail -5 /tmp/prof.err; fi
cd $R
timeout 200 python tools/bench_conv.py > $O/per_layer_conv_microbench.log 2>&1 < /dev/null
tail -1 $O/per_layer_conv_microbench.log
