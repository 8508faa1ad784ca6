export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 500 python -m pytest tests -m gpu -x -q 2>&1 < /dev/null | tail -4 > gpurun_out/tests.log
cat gpurun_out/tests.log
PDES_FUSE_MAXHW=256 timeout 200 python tools/ab_env.py PDES_FUSE_FINALIZE 0 1 2>&1 < /dev/null | grep "ms/step"
