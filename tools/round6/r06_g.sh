#!/bin/bash
# round-6 checkpoint g: the row-band kernel with compile-time geometry (PDES_BAND_FIXED=1, default) against the run-time plan
mkdir -p gpurun_out/r06_g
python3 -m pytest tests/test_loss_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for v in 1 0; do echo "== PDES_BAND_FIXED=$v"; PDES_BAND_FIXED=$v python tools/bench_loss_generic.py 2>&1 | grep -v amdgpu; done | tee gpurun_out/r06_g/band_fixed_ab.log
