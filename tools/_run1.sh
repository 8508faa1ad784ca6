export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 500 python -m pytest tests -m gpu -x -q 2>&1 < /dev/null | tail -4 > gpurun_out/tests.log
cat gpurun_out/tests.log
for v in 1 2; do timeout 100 python tools/ab_env.py PDES_DUMMY 1 2>&1 < /dev/null | grep "ms/step"; done
