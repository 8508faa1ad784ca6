#!/usr/bin/env python
"""Where the drop-in loop body (train_codec_mixed_residual.py:224-240 of the reference, verbatim, on this build's modules)
spends its time beside the fused trainer: host milliseconds and GPU milliseconds (HIP events) per phase -- batch gather,
zero_grad, forward, the three loss functions, loss.backward(), torch.optim.Adam.step().   python tools/dropin_phases.py [steps]"""
import contextlib
import io
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pde_surrogate_amd.models.codec import DenseED
from pde_surrogate_amd.models.darcy import (conv_constitutive_constraint as constitutive_constraint,
                                            conv_continuity_constraint as continuity_constraint,
                                            conv_boundary_condition as boundary_condition)
from pde_surrogate_amd.utils.data import grf_kle_fields
from pde_surrogate_amd.utils.image_gradient import SobelFilter
from pde_surrogate_amd import parallel

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = 32
dev = torch.device('cuda:0')
parallel.limit_host_threads()
torch.manual_seed(1)
with contextlib.redirect_stdout(io.StringIO()):
    model = DenseED(1, 3, 64, [6, 8, 6], growth_rate=16, init_features=48).to(dev)
kw, opt_mod = {}, torch.optim
if os.environ.get('DROPIN_ADAM') == 'flat':             # `from pde_surrogate_amd import optim` instead of torch.optim
    from pde_surrogate_amd import optim as opt_mod
elif os.environ.get('DROPIN_ADAM'):                     # 'foreach' | 'fused' (default: torch's choice + this build's auto-fused hook)
    kw = {os.environ['DROPIN_ADAM']: True}
optimizer = opt_mod.Adam(model.parameters(), lr=1e-3, weight_decay=0.0, **kw)
sobel_filter = SobelFilter(64, correct=True, device=dev)
data = torch.from_numpy(grf_kle_fields(4096, cache_dir='/tmp')).to(dev)
perm = torch.randperm(4096, generator=torch.Generator().manual_seed(1)).to(dev)
model.train()
PH = ['gather', 'zero_grad', 'forward', 'loss', 'backward', 'adam']
host = {p: 0.0 for p in PH}
gpu = {p: 0.0 for p in PH}


def body(i, record):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(PH) + 1)] if record else None
    t = [0.0] * (len(PH) + 1)

    def mark(k):
        if record:
            ev[k].record()
            t[k] = time.perf_counter()
    mark(0)
    lo = (i * B) % (4096 - B + 1)
    input = data[perm[lo:lo + B]]
    mark(1)
    model.zero_grad()
    mark(2)
    output = model(input)
    mark(3)
    loss_pde = constitutive_constraint(input, output, sobel_filter) + continuity_constraint(output, sobel_filter)
    loss_dirichlet, loss_neumann = boundary_condition(output)
    loss = loss_pde + (loss_dirichlet + loss_neumann) * 10.0
    mark(4)
    loss.backward()
    mark(5)
    optimizer.step()
    mark(6)
    return ev, t


# finer: host time inside the two autograd nodes' backward (the rest of loss.backward() is the autograd engine itself:
# thread hand-over, 82 AccumulateGrad nodes)
from pde_surrogate_amd.models import codec as _codec, darcy as _darcy
inner = {'net_backward': 0.0, 'loss_backward': 0.0, 'n': 0}
_nb, _lb = _codec._NetFn.backward, _darcy._Terms.backward


def _net_backward(ctx, gy):
    t0 = time.perf_counter()
    r = _nb(ctx, gy)
    inner['net_backward'] += time.perf_counter() - t0
    inner['n'] += 1
    return r


def _loss_backward(ctx, g):
    t0 = time.perf_counter()
    r = _lb(ctx, g)
    inner['loss_backward'] += time.perf_counter() - t0
    return r


if os.environ.get('DROPIN_INNER', '1') == '1':
    _codec._NetFn.backward = staticmethod(_net_backward)
    _darcy._Terms.backward = staticmethod(_loss_backward)
for i in range(30):
    body(i, False)
inner.update(net_backward=0.0, loss_backward=0.0, n=0)
torch.cuda.synchronize()
# (a) un-instrumented rate
t0 = time.perf_counter()
for i in range(N):
    body(i, False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'drop-in loop without loss.item(): {1e3 * (t2 - t0) / N:.4f} ms per step (host enqueue {1e3 * (t1 - t0) / N:.4f} ms per step)')
# (b) per phase, bursts of 10 steps behind a synchronise (the host's own time, not queue back-pressure)
recs = []
for r in range(N // 10):
    torch.cuda.synchronize()
    for i in range(10):
        recs.append(body(r * 10 + i, True))
torch.cuda.synchronize()
for ev, t in recs:
    for k, p in enumerate(PH):
        host[p] += t[k + 1] - t[k]
        gpu[p] += ev[k].elapsed_time(ev[k + 1]) * 1e-3
n = len(recs)
print('phase        host ms   gpu-interval ms')
for p in PH:
    print(f'{p:10s} {1e3 * host[p] / n:9.4f} {1e3 * gpu[p] / n:12.4f}')
print(f'{"sum":10s} {1e3 * sum(host.values()) / n:9.4f} {1e3 * sum(gpu.values()) / n:12.4f}')
print('optimizer:', type(optimizer).__module__, {k: optimizer.param_groups[0].get(k) for k in ('foreach', 'fused', 'capturable')}, '| params', len(list(model.parameters())))
if inner['n']:
    print(f"inside loss.backward(): _NetFn.backward {1e3 * inner['net_backward'] / inner['n']:.4f} ms, _Terms.backward "
          f"{1e3 * inner['loss_backward'] / inner['n']:.4f} ms per step (host)")
