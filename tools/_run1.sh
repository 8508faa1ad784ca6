export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_densed_gpu.py tests/test_trainer_gpu.py -x -q 2>&1 < /dev/null | tail -12 > gpurun_out/tests.log
cat gpurun_out/tests.log
timeout 200 python tools/ab_env.py PDES_FIN_SPLIT 0 1 2>&1 < /dev/null | grep "ms/step"
