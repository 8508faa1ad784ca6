#!/bin/bash
# round-6 checkpoint h: PMC traffic of the loss-path kernels (current fingerprint), the whole GPU suite twice
mkdir -p gpurun_out/r06_h
bash tools/pmc_loss_variants.sh r06_h > gpurun_out/r06_h_pmc.log 2>&1
python3 -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r06_h/gpu_suite.txt 2>&1
grep -E "passed|failed|emulation|^FAILED" gpurun_out/r06_h/gpu_suite.txt | tail -6
python3 -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r06_h/gpu_suite2.txt 2>&1
grep -E "passed|failed|^FAILED" gpurun_out/r06_h/gpu_suite2.txt | tail -4
