#!/usr/bin/env python
"""GPU time of every training step from process start (one event per step, no synchronisation in between): is the window the
driver times -- steps 6 .. 25 of a fresh process -- the steady state?   python tools/step_series.py [steps]"""
import contextlib
import io
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pde_surrogate_amd.models.codec import DenseED
from pde_surrogate_amd.train import MixedResidualTrainer
from pde_surrogate_amd.utils.data import grf_kle_fields

N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
B = 32
dev = torch.device('cuda:0')
torch.manual_seed(1)
with contextlib.redirect_stdout(io.StringIO()):
    model = DenseED(1, 3, 64, [6, 8, 6], growth_rate=16, init_features=48)
tr = MixedResidualTrainer(model, B, 64, lr=1e-3, weight_bound=10.0, device=dev, use_graph=False)
data = torch.from_numpy(grf_kle_fields(512, cache_dir='/tmp')).to(dev)
batches = [data[i * B:(i + 1) * B].contiguous() for i in range(512 // B)]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
torch.cuda.synchronize()
ev[0].record()
for i in range(N):
    tr.step(batches[i % len(batches)], 1e-4)
    ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
for lo in range(0, N, 10):
    print(f'steps {lo + 1:3d}-{lo + 10:3d}: ' + ' '.join(f'{t:6.3f}' for t in ms[lo:lo + 10]), flush=True)
print(f'steps 6-25 mean {sum(ms[5:25]) / 20:.4f}   steps {N - 39}-{N} mean {sum(ms[-40:]) / 40:.4f}')


def series(tag, n=30):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    e[0].record()
    for i in range(n):
        tr.step(batches[i % len(batches)], 1e-4)
        e[i + 1].record()
    torch.cuda.synchronize()
    m = [e[i].elapsed_time(e[i + 1]) for i in range(n)]
    print(f'{tag}: ' + ' '.join(f'{t:5.3f}' for t in m[:16]) + f' | 1-20 mean {sum(m[:20]) / 20:.4f}', flush=True)


if len(sys.argv) > 2:
    import time
    from pde_surrogate_amd.models import darcy
    Kb = torch.exp(0.5 * torch.randn(16384, 1, 64, 64, device=dev))
    yb = torch.randn(16384, 3, 64, 64, device=dev)
    torch.cuda.synchronize()
    series('back to back      ')
    time.sleep(0.1)
    series('after 100 ms idle ')
    time.sleep(0.1)
    for _ in range(150):                       # ~55 ms of loss-kernel launches, not waited for
        darcy.darcy_loss_launch(Kb, yb, (1, 1, 10, 10), True)
    time.sleep(0.05)                           # (a host stall the GPU works through)
    series('idle, then filler ')
    time.sleep(0.02)
    series('after 20 ms idle  ')
    time.sleep(0.005)
    series('after 5 ms idle   ')
