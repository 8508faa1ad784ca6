#!/usr/bin/env python
"""host enqueue time vs GPU time of the conditional-Glow reverse-KL step: the loop is timed to the point where the host
has enqueued everything, and again after the device has finished"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from pde_surrogate_amd.models.glow_msc import MultiScaleCondGlow
from pde_surrogate_amd.train import ReverseKLTrainer
from pde_surrogate_amd.utils.data import grf_kle_fields

dev = torch.device('cuda:0')
B = 32
data = torch.from_numpy(grf_kle_fields(B, 32, 100, cache_dir='/tmp')).to(dev)
torch.manual_seed(1)
np.random.seed(1)
with contextlib.redirect_stdout(io.StringIO()):
    net = MultiScaleCondGlow(32, 1, 3, [3, 4, 4], [6, 6, 6], LUdecompose=True).to(dev).train()
tr = ReverseKLTrainer(net, B, 32, device=dev)
for _ in range(20):
    tr.step(data, 1e-4)
torch.cuda.synchronize()
for n in (1, 5, 50):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.step(data, 1e-4)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{n:3d} steps: host enqueue {(t1 - t0) / n * 1e3:.3f} ms/step, until the device is done {(t2 - t0) / n * 1e3:.3f} ms/step')
