export TMPDIR=/tmp
mkdir -p gpurun_out
for r in 1 2 3; do for v in 1 4; do echo "SPI=$v"; PDES_1X1W_SPI=$v timeout 100 python tools/ab_env.py PDES_MFMA_1X1W 1 2>&1 < /dev/null | grep "ms/step"; done; done > gpurun_out/p1_ab.log 2>&1
cat gpurun_out/p1_ab.log
