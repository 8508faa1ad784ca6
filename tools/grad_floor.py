#!/usr/bin/env python
"""Rounding floor of the parameter gradients (GPU box): default DenseED, same weights and input, gradients from
  (a) the HIP path, (b) the CPU oracle in fp32 (= the arithmetic of the reference), (c) the CPU oracle in fp64,
per tensor rel-L2 of (a) and (b) against (c).  Usage: grad_floor.py [nearest|bilinear] [B]"""
import contextlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import codec as oc, train as ot
from pde_surrogate_amd.models.codec import DenseED
from pde_surrogate_amd.models.darcy import darcy_mixed_residual_loss
from pde_surrogate_amd.utils.data import grf_kle_fields

up = sys.argv[1] if len(sys.argv) > 1 else 'nearest'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device('cuda:0')
kind = sys.argv[3] if len(sys.argv) > 3 else 'grf'
if kind == 'grf':
    x = torch.from_numpy(grf_kle_fields(B, n_kle=128, seed=3, cache_dir='/tmp'))
else:                                  # white-noise log-permeability (the G2/G6/G13 fixtures' input family)
    x = torch.from_numpy(np.exp(0.5 * np.random.default_rng(5).standard_normal((B, 1, 64, 64))).astype(np.float32))
rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
torch.manual_seed(1)
with contextlib.redirect_stdout(io.StringIO()):
    net = DenseED(1, 3, 64, [6, 8, 6], upsample=up)
sd0 = {k: v.clone() for k, v in net.state_dict().items()}
res = {}
for tag, dt in (('cpu32', torch.float32), ('cpu64', torch.float64)):
    sd = {k: (v.detach().clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    tr = ot.CpuTrainer(sd, [6, 8, 6], upsample=up)
    _, loss, _ = tr.forward_loss(x.to(dt), True)
    loss.backward()
    res[tag] = {k: sd[k].grad.numpy().astype(np.float64) for k in tr.keys}
if not torch.cuda.is_available():
    print('cpu part ok'); sys.exit(0)
net = net.to(dev).train()
xd = x.to(dev)
loss, *_ = darcy_mixed_residual_loss(xd, net(xd), 10.0)
loss.backward()
res['hip'] = {k: p.grad.cpu().numpy() for k, p in net.named_parameters()}
rows = [(rel(res['hip'][k], res['cpu64'][k]), rel(res['cpu32'][k], res['cpu64'][k]), k) for k in res['cpu64']]
h, c = np.array([r[0] for r in rows]), np.array([r[1] for r in rows])
print(f'{up} B={B} {kind} B3={os.environ.get("PDES_MFMA_B3", "1")}: HIP vs fp64   max {h.max():.2e} median {np.median(h):.2e} | CPU fp32 vs fp64   max {c.max():.2e} median {np.median(c):.2e}')
for r in sorted(rows, reverse=True)[:8]:
    print(f'  hip {r[0]:.2e}  cpu32 {r[1]:.2e}  {r[2]}')
nh = np.array([np.linalg.norm(res['hip'][k]) for k in res['cpu64']]); n64 = np.array([np.linalg.norm(res['cpu64'][k]) for k in res['cpu64']])
n32 = np.array([np.linalg.norm(res['cpu32'][k]) for k in res['cpu64']])
print(f'  norms: HIP vs fp64 max {np.abs(nh / n64 - 1).max():.2e} | CPU fp32 vs fp64 max {np.abs(n32 / n64 - 1).max():.2e}')
