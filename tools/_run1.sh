export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python tools/ab_env.py PDES_MFMA_1X1W 0 1 > gpurun_out/p1_ab.log 2>&1 < /dev/null
timeout 200 python tools/ab_env.py PDES_MFMA_1X1 0 1 >> gpurun_out/p1_ab.log 2>&1 < /dev/null
grep -v amdgpu.ids gpurun_out/p1_ab.log
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | head -6
