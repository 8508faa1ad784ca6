#!/usr/bin/env python
"""per-channel view of the G13 BatchNorm-bias gradients that deviate (GPU box): python tools/archive/debug_g13b.py"""
import contextlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from pde_surrogate_amd.models.codec import DenseED
from pde_surrogate_amd.models.darcy import darcy_mixed_residual_loss

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests/golden/G13_bilinear.npz'))
dev = torch.device('cuda:0')
x = torch.from_numpy(g['x']).to(dev)
torch.manual_seed(1)
with contextlib.redirect_stdout(io.StringIO()):
    net = DenseED(1, 3, 64, [6, 8, 6], upsample='bilinear').to(dev).train()
y = net(x)
darcy_mixed_residual_loss(x, y, 10.0)[0].backward()
eng = net._engine(x)
P = dict(net.named_parameters())
for name in ('features.DecBlock1.denselayer3.norm1', 'features.DecBlock1.denselayer4.norm1', 'features.DecBlock1.denselayer1.norm1'):
    for wb in ('bias', 'weight'):
        k = f'{name}.{wb}'
        got, want = P[k].grad.cpu().numpy().astype(np.float64), g['grad/' + k].astype(np.float64)
        d = np.abs(got - want)
        idx = np.argsort(-d)[:4]
        print(k, 'rel', np.linalg.norm(got - want) / np.linalg.norm(want), 'norm(want)', np.linalg.norm(want))
        for i in idx:
            print(f'   ch {i:3d} got {got[i]: .6e} want {want[i]: .6e} diff {got[i] - want[i]: .3e}  gamma {float(P[name + ".weight"][i]): .4f} beta {float(P[name + ".bias"][i]): .4f}')
# the raw activations of the DecBlock1 buffer: per-channel mean / std over (B, H, W)
buf = [k for k, s in zip(eng.X, eng.X.values()) if s.shape[1] == 200 and s.shape[2] == 16][0]
X = eng.X[buf].double()
m, sd = X.mean((0, 2, 3)).cpu().numpy(), X.std((0, 2, 3)).cpu().numpy()
print('DecBlock1 buffer', buf, 'per-channel std: min', sd[:72].min(), 'argmin', int(sd[:72].argmin()), 'median', float(np.median(sd[:72])))
for i in (np.argsort(sd[:72])[:4]):
    print(f'   ch {i} mean {m[i]:.4e} std {sd[i]:.4e}')
