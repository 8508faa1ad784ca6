#!/bin/bash
# compile ONE csrc/*.hip with the library's flags and print the resource usage of the kernels whose mangled name matches $2
#   bash tools/hipcc_one.sh conv_mfma_wgrad wgrad_kernelILi3E
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/pde_surrogate_amd/csrc/$1.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -munsafe-fp-atomics \
  $PDES_EXTRA_FLAGS -Rpass-analysis=kernel-resource-usage -c $SRC -o $ROOT/pde_surrogate_amd/csrc/$1.o > /tmp/hipcc_one.log 2>&1
rc=$?
grep -E "error|warning: v" -A6 /tmp/hipcc_one.log | head -40
python3 - "$2" <<'PY'
import re, sys
t = open('/tmp/hipcc_one.log').read()
pat = sys.argv[1] if len(sys.argv) > 1 else ''
for b in t.split('Function Name: ')[1:]:
    name = b.split('\n')[0]
    if pat and pat not in name:
        continue
    g = lambda k: (re.search(k + r': (\d+)', b) or [None, '?'])[1]
    print(name[:100], 'VGPR', g('VGPRs'), 'AGPR', g('AGPRs'), 'SGPR', g('SGPRs'), 'scratch', g(r'ScratchSize \[bytes/lane\]'),
          'occ', g(r'Occupancy \[waves/SIMD\]'), 'LDS', g(r'LDS Size \[bytes/block\]'))
PY
exit $rc
