#!/bin/bash
# round-6 checkpoint e: PMC traffic of the loss-path kernels on the current sources, then the GPU suite
bash tools/pmc_loss_variants.sh r06_e > gpurun_out/r06_e_pmc.log 2>&1
mkdir -p gpurun_out/r06_e
python3 -m pytest tests -m gpu -x -q -s -p no:cacheprovider > gpurun_out/r06_e/gpu_suite.txt 2>&1
grep -E "passed|failed|emulation" gpurun_out/r06_e/gpu_suite.txt | tail -5
