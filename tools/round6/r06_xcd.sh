#!/bin/bash
# XCD-aware workgroup mapping (PDES_XCD_MAP): parity, per-layer timing and the step, same process / same box
python3 -m pytest tests/test_densed_gpu.py tests/test_b3_adversarial_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/xcd_tests.txt 2>&1; grep -E "passed|failed" gpurun_out/xcd_tests.txt
for v in 1 0 1 0; do echo "== PDES_XCD_MAP=$v"; PDES_XCD_MAP=$v python tools/bench_conv.py ${1:-18,25,26,27} 2>&1 | grep -v amdgpu | cut -c1-150; done
python tools/ab_env.py PDES_XCD_MAP 1 0 2>&1 | grep -v amdgpu
