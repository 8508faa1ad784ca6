export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_densed_gpu.py tests/test_trainer_gpu.py -x -q 2>&1 < /dev/null | tail -3 > gpurun_out/p1_tests.log
timeout 60 python tools/bench_conv.py 6,7,17 2>&1 < /dev/null | grep "Trans" > gpurun_out/p1_micro.log
timeout 200 python tools/ab_env.py PDES_MFMA_1X1 0 1 > gpurun_out/p1_ab.log 2>&1 < /dev/null
cat gpurun_out/p1_tests.log gpurun_out/p1_micro.log; grep -v amdgpu.ids gpurun_out/p1_ab.log
