#!/bin/bash
# MT = 2 forward of the 16x16 dense layers: per-layer timing, parity, step A/B (same process: tools/ab_env.py)
python tools/bench_conv.py 9,10,11,12,13,14,15,16 2>&1 | grep -v amdgpu | cut -c1-110
PDES_MFMA_MT2=0 python tools/bench_conv.py 9,10,11,12,13,14,15,16 2>&1 | grep -v amdgpu | cut -c1-110
python -m pytest tests/test_densed_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error"
for i in 1 2 3; do
  for v in 1 0; do printf "MT2=$v: "; PDES_MFMA_MT2=$v python bench.py --steps 200 --warmup 60 --no-cpu-baseline --no-extras 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*" | head -1; done
done
