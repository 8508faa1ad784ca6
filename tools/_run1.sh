export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/ab_env.py PDES_FIN_EARLY 0 1 2>&1 < /dev/null | grep "ms/step" > gpurun_out/knobs.log
cat gpurun_out/knobs.log
