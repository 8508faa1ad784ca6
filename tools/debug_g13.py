#!/usr/bin/env python
"""Which gradient tensors of the G13 (bilinear, default net, B = 4 white-noise fields) pass deviate from the reference,
and by how much against the fp64 oracle norms (GPU box): python tools/debug_g13.py"""
import contextlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pde_surrogate_amd.models.codec import DenseED
from pde_surrogate_amd.models.darcy import darcy_mixed_residual_loss

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/G13_bilinear.npz'))
dev = torch.device('cuda:0')
torch.manual_seed(1)
with contextlib.redirect_stdout(io.StringIO()):
    net = DenseED(1, 3, 64, [6, 8, 6], upsample='bilinear').to(dev).train()
x = torch.from_numpy(g['x']).to(dev)
y = net(x)
loss = darcy_mixed_residual_loss(x, y, 10.0)[0]
loss.backward()
names = [k for k, _ in net.named_parameters()]
norms = np.array([float(p.grad.double().norm()) for _, p in net.named_parameters()])
d32 = np.abs(norms - g['grad_norms']) / g['grad_norms']
d64 = np.abs(norms - g['grad_norms_fp64']) / g['grad_norms_fp64']
for i in np.argsort(-d32)[:12]:
    print(f'{names[i]:50s} vs ref fp32 {d32[i]:.2e}  vs fp64 {d64[i]:.2e}')
print('output rel-L2 vs reference', float(np.linalg.norm(y.detach().cpu().numpy() - g['y']) / np.linalg.norm(g['y'])))
