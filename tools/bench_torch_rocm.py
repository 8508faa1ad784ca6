#!/usr/bin/env python
"""Context measurement (NOT part of bench.py): the same training step written with stock PyTorch-ROCm
ops (nn.Conv2d / nn.BatchNorm2d via MIOpen, torch.cat, F.interpolate, conv2d-based Sobel, autograd,
torch.optim.Adam) on the same MI355X -- i.e. what the reference's own code path costs on this GPU.
The network is rebuilt from this package's layer plan with torch.nn modules; nothing from
/root/reference or oracle/ is used.  Prints samples/s at bs=32 (eager, and with torch's fused Adam)."""
import contextlib
import io
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
import torch.nn.functional as F

from pde_surrogate_amd.models.codec import _plan_densed


class TorchDenseED(nn.Module):
    def __init__(self, blocks=(6, 8, 6), growth=16, init=48, imsize=64):
        super().__init__()
        self.specs, _ = _plan_densed(list(blocks), growth, init, 1, 3, imsize)
        self.convs = nn.ModuleList([nn.Conv2d(s.cin, s.cout, s.k, s.stride, s.pad, bias=False) for s in self.specs])
        self.bns = nn.ModuleList([nn.BatchNorm2d(s.cin) if s.norm else nn.Identity() for s in self.specs])

    def forward(self, x):
        bufs = {'in': x}
        for s, conv, bn in zip(self.specs, self.convs, self.bns):
            z = bufs[s.src]
            if s.norm:
                z = F.relu(bn(z), inplace=True)
            if s.up:
                z = F.interpolate(z, scale_factor=2.0, mode='nearest')
            y = conv(z)
            bufs[s.dst] = torch.cat([bufs[s.dst], y], 1) if (s.dst == s.src) else y
        return bufs['out']


def sobel_loss(K, y, wb=10.0):
    n = y.shape[-1]
    dev = y.device
    vs = torch.tensor([[-1., 0., 1.], [-2., 0., 2.], [-1., 0., 1.]], device=dev).view(1, 1, 3, 3) / 8
    hs = vs.transpose(-1, -2)
    mod = torch.eye(n, device=dev)
    mod[0, 0], mod[1, 0], mod[-2, -1], mod[-1, -1] = 4, -1, -1, 4

    def gh(t):
        return torch.matmul(F.conv2d(F.pad(t, (1, 1, 1, 1), mode='replicate'), vs) * n, mod)

    def gv(t):
        return torch.matmul(mod.t(), F.conv2d(F.pad(t, (1, 1, 1, 1), mode='replicate'), hs) * n)
    u, s1, s2 = y[:, [0]], y[:, [1]], y[:, [2]]
    lc = ((s1 + K * gh(u)) ** 2 + (s2 + K * gv(u)) ** 2).mean()
    lt = ((gh(s1) + gv(s2)) ** 2).mean()
    ld = F.mse_loss(y[:, 0, :, 0], torch.ones_like(y[:, 0, :, 0])) + (y[:, 0, :, -1] ** 2).mean()
    ln = (y[:, 2, [0, -1], :] ** 2).mean()
    return lc + lt + wb * (ld + ln)


def run(B=32, steps=30, warmup=5, fused_adam=False):
    dev = torch.device('cuda:0')
    torch.manual_seed(1)
    net = TorchDenseED().to(dev).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=fused_adam)
    x = torch.exp(0.5 * torch.randn(B, 1, 64, 64, device=dev))
    def step():
        net.zero_grad()
        loss = sobel_loss(x, net(x))
        loss.backward()
        opt.step()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {'impl': 'pytorch-rocm eager (MIOpen), fused_adam=%s' % fused_adam, 'bs': B, 'ms_per_step': round(dt / steps * 1e3, 3),
            'samples_per_s': round(B * steps / dt, 1), 'torch': torch.__version__}


if __name__ == '__main__':
    for fa in (False, True):
        try:
            print(json.dumps(run(fused_adam=fa)), flush=True)
        except Exception as e:          # fused Adam may be unavailable
            print(json.dumps({'fused_adam': fa, 'error': str(e)[:200]}), flush=True)
