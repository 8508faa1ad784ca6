#!/usr/bin/env python
"""HBM rate of the any-size loss kernel (csrc/darcy_loss_generic.hip) next to the specialised one, HIP events, working
sets beyond the Infinity Cache: python tools/bench_loss_generic.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pde_surrogate_amd import _lib

dev = torch.device('cuda:0')
L, st, ctx = _lib.lib(), _lib.stream_ptr(), _lib.context(dev)


def rate(n, B, flags, iters=30, warm=30):
    K = torch.exp(0.5 * torch.randn(B, 1, n, n, device=dev))
    y = torch.randn(B, 3, n, n, device=dev)
    g = torch.empty_like(y)
    part = torch.empty(_lib.loss_partial_rows(B, n, n, flags), 4, device=dev)

    def run():
        rc = L.pdes_darcy_loss(ctx, K.data_ptr(), y.data_ptr(), g.data_ptr(), part.data_ptr(), None, B, n, n, 1.0, 1.0, 10.0, 10.0,
                               flags, 0.0, 0.0, st)
        assert rc == 0, rc
    for _ in range(warm):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    by = 7 * n * n * 4 * B
    return us, by / us / 1e3


# flags: 4 = correct=False, 16 = PDES_LOSS_GENERIC (never the specialisation), 8 = PDES_LOSS_TILED (the tile kernel of rounds 1-3)
CASES = [(64, 16384, 0, 'specialised 64x64'), (64, 16384, 16, 'row bands 64x64'), (64, 16384, 4, 'row bands 64x64 (correct=False)'),
         (64, 16384, 24, 'tiles 64x64'),
         (65, 16384, 0, "row bands 65x65"), (65, 16384, 8, "tiles 65x65"), (66, 16384, 0, "row bands 66x66"), (96, 7168, 0, "row bands 96x96"),
         (128, 4096, 0, 'row bands 128x128'), (128, 4096, 8, 'tiles 128x128'),
         (48, 28672, 0, 'row bands 48x48'), (48, 28672, 8, 'tiles 48x48'),
         (100, 6552, 0, 'row bands 100x100'), (129, 3936, 0, 'row bands 129x129'), (130, 3872, 0, 'row bands 130x130'),
         (131, 3816, 0, 'row bands 131x131'), (200, 1640, 0, 'row bands 200x200'),
         (256, 1024, 0, 'row bands 256x256'), (256, 1024, 8, 'tiles 256x256'),
         (32, 65536, 0, 'specialised 32x32'), (32, 65536, 16, 'row bands 32x32'),
         (64, 32, 16, 'row bands 64x64 at the training batch'), (64, 32, 0, 'specialised 64x64 at the training batch'),
         (65, 32, 0, 'row bands 65x65 at batch 32'), (128, 32, 0, 'row bands 128x128 at batch 32'), (128, 32, 8, 'tiles 128x128 at batch 32')]
if len(sys.argv) > 1:
    CASES = [c for c in CASES if any(a in c[3] for a in sys.argv[1:])]
for n, B, flags, name in CASES:
    us, gbs = rate(n, B, flags)
    print(f'{name:40s} B={B:6d}  {us:9.1f} us  {gbs:8.1f} GB/s  {gbs / 8000:.3f} of 8 TB/s', flush=True)
