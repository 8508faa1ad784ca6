#!/usr/bin/env python
"""timing experiments on the row-band loss kernel (a library built with PDES_EXTRA_FLAGS=-DPDES_TUNE): flags bit 4096 = no
arithmetic, 8192 = no output copy (general path).   python tools/bench_loss_tune.py 65 16384"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pde_surrogate_amd import _lib

dev = torch.device('cuda:0')
L, st, ctx = _lib.lib(), _lib.stream_ptr(), _lib.context(dev)
n, B = int(sys.argv[1]), int(sys.argv[2])
K = torch.exp(0.5 * torch.randn(B, 1, n, n, device=dev))
y = torch.randn(B, 3, n, n, device=dev)
g = torch.empty_like(y)
for flags, name in ((0, 'full'), (4096, 'no arithmetic'), (8192, 'no output copy'), (4096 | 8192, 'staging only'), (16384, 'forward only')):
    gy = None if flags & 16384 else g
    f = flags & ~16384
    part = torch.empty(_lib.loss_partial_rows(B, n, n, f & 31), 4, device=dev)

    def run():
        rc = L.pdes_darcy_loss(ctx, K.data_ptr(), y.data_ptr(), gy.data_ptr() if gy is not None else None, part.data_ptr(), None, B, n, n,
                               1.0, 1.0, 10.0, 10.0, f, 0.0, 0.0, st)
        assert rc == 0, rc
    for _ in range(30):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 30
    print(f'n={n} B={B} {name:16s} {us:9.1f} us   {7 * n * n * 4 * B / us / 1e3 / 8000:.3f} of 8 TB/s (7 planes)', flush=True)
