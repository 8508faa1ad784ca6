#!/bin/bash
# round-6 checkpoint f: tile-size experiments (PDES_MFMA_MT2 = 1 / 2 / 3), PMC traffic of the loss-path kernels, GPU suite
mkdir -p gpurun_out/r06_f
python tools/ab_env.py PDES_MFMA_MT2 1 2 3 2>&1 | grep -v amdgpu > gpurun_out/r06_f/ab_mt.log
for v in 1 2 3; do echo "== PDES_MFMA_MT2=$v"; PDES_MFMA_MT2=$v python tools/bench_conv.py 1,2,3,4,5,6,19,20,21,22,23,24 2>&1 | grep -v amdgpu | cut -c1-150; done > gpurun_out/r06_f/layers_mt.log
cat gpurun_out/r06_f/ab_mt.log
bash tools/pmc_loss_variants.sh r06_f > gpurun_out/r06_f_pmc.log 2>&1
python3 -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r06_f/gpu_suite.txt 2>&1
grep -E "passed|failed|emulation|^FAILED" gpurun_out/r06_f/gpu_suite.txt | tail -8
