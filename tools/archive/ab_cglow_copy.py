"""same-process A/B of a plan flag of pde_surrogate_amd.models.glow_msc (two models, alternating):
    python tools/archive/ab_cglow_copy.py [_MERGE_COPY | _FUSE_COPY_FINALIZE]"""
import contextlib, io, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from pde_surrogate_amd.models import glow_msc
from pde_surrogate_amd.train import ReverseKLTrainer
from pde_surrogate_amd.utils.data import grf_kle_fields
dev = torch.device('cuda:0'); B = 32
data = torch.from_numpy(grf_kle_fields(B, 32, 100, cache_dir='/tmp')).to(dev)
trs = {}
import sys as _s
ATTR = _s.argv[1] if len(_s.argv) > 1 else '_MERGE_COPY'
for merge in (True, False):
    setattr(glow_msc, ATTR, merge)
    torch.manual_seed(1); np.random.seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        net = glow_msc.MultiScaleCondGlow(32, 1, 3, [3, 4, 4], [6, 6, 6], LUdecompose=True).to(dev).train()
    trs[merge] = ReverseKLTrainer(net, B, 32, device=dev)
    for _ in range(15): trs[merge].step(data, 1e-4)
res = {True: [], False: []}
for r in range(4):
    for m in (True, False):
        tr = trs[m]
        for _ in range(5): tr.step(data, 1e-4)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40): tr.step(data, 1e-4)
        torch.cuda.synchronize(); res[m].append((time.perf_counter() - t0) / 40 * 1e3)
for m in res: print(ATTR, m, ' '.join(f'{t:.3f}' for t in res[m]))
