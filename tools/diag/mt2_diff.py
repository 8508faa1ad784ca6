#!/usr/bin/env python
"""PDES_MFMA_MT2 = 0 against 1 in ONE process: activations buffer by buffer, the statistics arena, every gradient tensor.
    python tools/diag/mt2_diff.py [B] [nearest|bilinear] [noise|grf]"""
import contextlib, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pde_surrogate_amd import _lib
from pde_surrogate_amd.models.codec import DenseED
from pde_surrogate_amd.models.darcy import darcy_mixed_residual_loss
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
up = sys.argv[2] if len(sys.argv) > 2 else 'bilinear'
kind = sys.argv[3] if len(sys.argv) > 3 else 'noise'
dev = torch.device('cuda:0')
torch.manual_seed(1)
with contextlib.redirect_stdout(io.StringIO()):
    net = DenseED(1, 3, 64, [6, 8, 6], growth_rate=16, init_features=48, upsample=up).to(dev).train()
if kind == 'g13':
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from conftest import golden, load_seeded
    g = golden('G13_bilinear.npz')
    load_seeded(net, 'densed_seed1')
    x = torch.from_numpy(g['x']).to(dev)
elif kind == 'noise':
    x = torch.exp(0.5 * torch.randn(B, 1, 64, 64, device=dev))
else:
    from pde_surrogate_amd.utils.data import grf_kle_fields
    x = torch.from_numpy(grf_kle_fields(B, cache_dir='/tmp')).to(dev)
res = {}
for v in (0, 1):
    _lib.set_option('PDES_MFMA_MT2', v)
    net.zero_grad()
    y = net(x)
    eng = [e for pool in net._engines.values() for e in (pool if isinstance(pool, list) else [pool])][0]
    X = {k: t.clone() for k, t in eng.X.items()}
    arena = eng.arena.clone()
    darcy_mixed_residual_loss(x, y, 10.0)[0].backward()
    res[v] = (y.detach().clone(), X, arena, {k: p.grad.clone() for k, p in net.named_parameters()})
y0, X0, a0, g0 = res[0]
y1, X1, a1, g1 = res[1]
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))
print('output', rel(y1, y0))
for k in X0:
    print('buffer %-8s rel %.2e  max abs %.2e  bitwise equal %s' % (k, rel(X1[k], X0[k]), float((X1[k] - X0[k]).abs().max()), bool(torch.equal(X1[k], X0[k]))))
print('arena rel', rel(a1, a0), 'max abs', float((a1 - a0).abs().max()))
rows = sorted(((rel(g1[k], g0[k]), k) for k in g0), reverse=True)
print('gradients: worst', [(f'{e:.2e}', k) for e, k in rows[:6]], '| median %.2e' % float(np.median([e for e, _ in rows])))
if kind == 'g13':
    from conftest import rel_l2
    for v, gr in ((0, g0), (1, g1)):
        errs = sorted(((rel_l2(gr[k[5:]].cpu().numpy(), g[k]), k[5:]) for k in g.files if k.startswith('grad/')), reverse=True)
        print('MT2=%d vs the reference: worst' % v, [(f'{e:.2e}', k) for e, k in errs[:4]], '| median %.2e' % float(np.median([e for e, _ in errs])))
