export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python tools/ab_env.py PDES_EVENT_SCOPE system device 2>&1 < /dev/null | grep "ms/step"
timeout 300 python -m pytest tests/test_densed_gpu.py tests/test_trainer_gpu.py -x -q 2>&1 < /dev/null | tail -2
