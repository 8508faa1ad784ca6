#!/usr/bin/env python
"""Is the training step host-bound?  Time to ENQUEUE k steps (no synchronisation) against the time until the GPU has
finished them: python tools/archive/host_slack.py"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pde_surrogate_amd.models.codec import DenseED
from pde_surrogate_amd.train import MixedResidualTrainer
from pde_surrogate_amd.utils.data import grf_kle_fields

dev = torch.device('cuda:0')
torch.manual_seed(1)
with contextlib.redirect_stdout(io.StringIO()):
    model = DenseED(1, 3, 64, [6, 8, 6])
tr = MixedResidualTrainer(model, 32, 64, lr=1e-3, device=dev)
data = torch.from_numpy(grf_kle_fields(512, cache_dir='/tmp')).to(dev)
idx = [torch.arange(i * 32, (i + 1) * 32, device=dev) for i in range(16)]
for i in range(50):
    tr.load_batch(data, idx[i % 16]); tr.step(None, 1e-4)
torch.cuda.synchronize()
for k in (5, 20, 50):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(k):
        tr.load_batch(data, idx[i % 16]); tr.step(None, 1e-4)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{k:3d} steps: host enqueue {1e3 * (t1 - t0) / k:.3f} ms/step, until the GPU is done {1e3 * (t2 - t0) / k:.3f} ms/step')
