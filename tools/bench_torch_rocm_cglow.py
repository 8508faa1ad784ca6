#!/usr/bin/env python
"""Context measurement (NOT part of bench.py): the reverse-KL training step of the default conditional Glow written with
stock PyTorch-ROCm ops (F.conv2d / F.batch_norm via MIOpen, torch.cat, chunk, sigmoid, autograd, torch.optim.Adam) on the
same MI355X -- what the reference's own code path costs on this GPU.  The functional restatement in oracle/glow.py is run
on CUDA tensors (it is device agnostic); nothing from /root/reference is used.  Prints samples/s at bs = 32."""
import contextlib
import io
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import glow as oglow
from pde_surrogate_amd.models.glow_msc import MultiScaleCondGlow
from pde_surrogate_amd.utils.data import grf_kle_fields


def main():
    dev = torch.device('cuda:0')
    B, steps, warm = 32, 30, 8
    torch.manual_seed(1)
    np.random.seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        net = MultiScaleCondGlow(32, 1, 3, [3, 4, 4], [6, 6, 6], LUdecompose=True)
    sd = {k: v.detach().clone().to(dev) for k, v in net.state_dict().items()}
    keys = oglow.param_keys(sd)
    for k in keys:
        sd[k].requires_grad_(True)
    x = torch.from_numpy(grf_kle_fields(B, 32, 100, cache_dir='/tmp')).to(dev)
    shapes = oglow.latent_shapes(sd, 3, 32)
    out = {}
    for tag, kw in (('eager_adam', {}), ('fused_adam', {'fused': True})):
        opt = torch.optim.Adam([sd[k] for k in keys], lr=1e-3, **kw)

        def step():
            opt.zero_grad(set_to_none=True)
            eps = [torch.randn((B,) + s, device=dev) for s in shapes]
            loss = oglow.reverse_kl_loss(sd, x, eps, 150.0, 50.0, True)[0]
            loss.backward()
            opt.step()
        for _ in range(warm):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out[tag] = {'samples_per_s': round(B / dt, 1), 'ms_per_step': round(dt * 1e3, 2)}
    print(json.dumps({'stock_pytorch_rocm_cglow_reverse_kl_bs32': out}))


if __name__ == '__main__':
    main()
