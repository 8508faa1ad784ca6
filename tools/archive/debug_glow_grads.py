#!/usr/bin/env python
"""Default conditional Glow on the GPU vs the CPU oracle (fp32 and fp64) on the G19 inputs: per-tensor gradient errors,
and for the worst tensors how many elements carry the error (isolated ReLU flips show up as a handful of elements).
Debugging aid, runs the oracle: not part of the product."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from glow_util import perturb_glow, reverse_kl                         # noqa: E402
from oracle import glow                                                # noqa: E402
from pde_surrogate_amd.models.glow_msc import MultiScaleCondGlow       # noqa: E402


def main():
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'G19_cglow_default.npz'))
    torch.manual_seed(1)
    np.random.seed(1)
    net = MultiScaleCondGlow(32, 1, 3, [3, 4, 4], [6, 6, 6], LUdecompose=True)
    perturb_glow(net, torch.Generator().manual_seed(13), 0.4)
    sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    ref = {}
    for dt in (torch.float32, torch.float64):
        sd = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
        keys = glow.param_keys(sd)
        for k in keys:
            sd[k].requires_grad_(True)
        x = torch.from_numpy(g['x']).to(dt)
        eps = [torch.from_numpy(g[f'eps{i}']).to(dt) for i in range(2)]
        loss = glow.reverse_kl_loss(sd, x, eps, 150.0, 50.0, True)[0]
        loss.backward()
        ref[dt] = {k: sd[k].grad.double() for k in keys}
    dev = torch.device('cuda:0')
    net = net.to(dev).train()
    x = torch.from_numpy(g['x']).to(dev)
    eps = [torch.from_numpy(g[f'eps{i}']).to(dev) for i in range(2)]
    loss = reverse_kl(net, x, eps, 150.0, 50.0)[0]
    loss.backward()
    rows = []
    for k, p in net.named_parameters():
        mine = p.grad.double().cpu()
        r32, r64 = ref[torch.float32][k], ref[torch.float64][k]
        n = float(r64.norm())
        e = (mine - r64).abs()
        big = int((e > 1e-3 * n / max(mine.numel(), 1) ** 0.5 * 10).sum())
        rows.append((float((mine - r64).norm()) / n, float((r32 - r64).norm()) / n, float((mine - r32).norm()) / n, big,
                     mine.numel(), k))
    rows.sort(reverse=True)
    print('gpu-vs-fp64  cpu32-vs-fp64  gpu-vs-cpu32  #elements>10x-rms-tolerance  numel  name')
    for r in rows[:25]:
        print(f'{r[0]:.2e}  {r[1]:.2e}  {r[2]:.2e}  {r[3]:6d} {r[4]:8d}  {r[5]}')
    print('median gpu-vs-fp64', float(np.median([r[0] for r in rows])), ' median cpu32-vs-fp64', float(np.median([r[1] for r in rows])))


if __name__ == '__main__':
    main()
