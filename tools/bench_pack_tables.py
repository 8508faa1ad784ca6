#!/usr/bin/env python
"""where the per-step weight-packing launch (pdes_pack_all2, ~30 us at the head of every step) spends its time: each table
alone, HIP events.   python tools/bench_pack_tables.py"""
import contextlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pde_surrogate_amd import _lib
from pde_surrogate_amd.models.codec import DenseED

dev = torch.device('cuda:0')
torch.manual_seed(1)
with contextlib.redirect_stdout(io.StringIO()):
    net = DenseED(1, 3, 64, [6, 8, 6]).to(dev).train()
x = torch.exp(0.5 * torch.randn(32, 1, 64, 64, device=dev))
net(x)
L, st = _lib.lib(), _lib.stream_ptr()
T = {'direct': (net._pack_table, net._pack_n, net._pack_max), 'mfma': (net._mpack_table, net._mpack_n, net._mpack_max),
     'up': (net._upack_table, net._upack_n, net._upack_max), 'b3': (net._bpack_table, net._bpack_n, net._bpack_max),
     'b3up': (net._bupack_table, net._bupack_n, net._bupack_max)}


def run(names):
    a = []
    mx = 1
    for k in ('direct', 'mfma', 'up', 'b3', 'b3up'):
        t, n, m = T[k]
        on = k in names and n
        a += [t.data_ptr() if on else None, n if on else 0]
        if on:
            mx = max(mx, m)
    rc = L.pdes_pack_all2(*a, mx, st)
    assert rc == 0, rc


def timeit(names, iters=50):
    for _ in range(5):
        run(names)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run(names)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


print('all tables      %6.1f us' % timeit(list(T)))
for k in T:
    print('%-8s n=%3d   %6.1f us   (max elements per image %d)' % (k, T[k][1], timeit([k]), T[k][2]))
print('direct + mfma   %6.1f us' % timeit(['direct', 'mfma']))
print('up + b3 + b3up  %6.1f us' % timeit(['up', 'b3', 'b3up']))
print('all but direct  %6.1f us' % timeit(['mfma', 'up', 'b3', 'b3up']))
print('mfma + b3 + b3up %5.1f us' % timeit(['mfma', 'b3', 'b3up']))
