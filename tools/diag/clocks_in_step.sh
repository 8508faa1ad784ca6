#!/bin/bash
# shader clock and package power WHILE the fused step runs (one rocm-smi sample every ~0.3 s beside tools/step_series.py-like stepping)
python3 - <<'PY' &
import contextlib, io, time, torch
from pde_surrogate_amd.models.codec import DenseED
from pde_surrogate_amd.train import MixedResidualTrainer
from pde_surrogate_amd.utils.data import grf_kle_fields
dev = torch.device('cuda:0')
torch.manual_seed(1)
with contextlib.redirect_stdout(io.StringIO()):
    net = DenseED(1, 3, 64, [6, 8, 6])
tr = MixedResidualTrainer(net, 32, 64, lr=1e-3, device=dev)
x = torch.from_numpy(grf_kle_fields(32, cache_dir='/tmp')).to(dev)
t0 = time.time()
n = 0
while time.time() - t0 < 8.0:
    for _ in range(100):
        tr.step(x, 1e-4)
    n += 100
torch.cuda.synchronize()
print('steps', n, 'ms/step', (time.time() - t0) / n * 1e3, flush=True)
PY
sleep 3.5
for i in 1 2 3 4 5 6 7 8; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | tr '\n' ' '; echo; sleep 0.3; done
wait
