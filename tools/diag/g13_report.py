#!/usr/bin/env python
"""G13 (default DenseED with bilinear upsampling) against the reference's stored gradients: every tensor beyond 1e-3 with the
share of its squared deviation in the two worst channels.   PDES_MFMA_MT2=0/1 python tools/diag/g13_report.py"""
import contextlib, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from conftest import golden, load_seeded, rel_l2
from pde_surrogate_amd.models.codec import DenseED
from pde_surrogate_amd.models.darcy import darcy_mixed_residual_loss
dev = torch.device('cuda:0')
g = golden('G13_bilinear.npz')
torch.manual_seed(1)
with contextlib.redirect_stdout(io.StringIO()):
    net = DenseED(1, 3, 64, [6, 8, 6], growth_rate=16, init_features=48, upsample='bilinear')
load_seeded(net, 'densed_seed1')
net = net.to(dev).train()
x = torch.from_numpy(g['x']).to(dev)
y = net(x)
print('output rel-L2', rel_l2(y.detach().cpu().numpy(), g['y']))
loss = darcy_mixed_residual_loss(x, y, 10.0)[0]
loss.backward()
gr = dict(net.named_parameters())
for k in g.files:
    if not k.startswith('grad/'):
        continue
    got, want = gr[k[5:]].grad.cpu().numpy(), g[k]
    e = rel_l2(got, want)
    if e < 1e-3:
        continue
    d = (got - want).astype(np.float64)
    per = d ** 2 if d.ndim == 1 else (d ** 2).sum(axis=tuple(i for i in range(d.ndim) if i != (1 if d.ndim == 4 else 0)))
    top = np.argsort(-per)[:2]
    print(f'  {e:.2e} {k[5:]:50s} worst channels {top.tolist()} hold {100 * per[top].sum() / per.sum():.0f} %')
