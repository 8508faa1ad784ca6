#!/usr/bin/env python
"""Time the fused Sobel+Darcy-residual kernel with HIP events at several batch sizes.
Algorithmic bytes: 7 planes x 64*64*4 B = 114,688 B/sample fwd+bwd; 4 planes fwd-only."""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pde_surrogate_amd.models import darcy
from pde_surrogate_amd import _lib


def time_loss(B, n=64, iters=50, bwd=True):
    dev = torch.device('cuda:0')
    K = torch.exp(0.5 * torch.randn(B, 1, n, n, device=dev))
    y = torch.randn(B, 3, n, n, device=dev)
    g = torch.empty_like(y) if bwd else None
    part = torch.empty(B, 4, device=dev)
    L = _lib.lib()
    st = _lib.stream_ptr()
    def run():
        rc = L.pdes_darcy_loss(_lib.context(dev), K.data_ptr(), y.data_ptr(), g.data_ptr() if bwd else None, part.data_ptr(), None,
                               B, n, n, 1.0, 1.0, 10.0, 10.0, 0, 0.0, 0.0, st)
        assert rc == 0, rc
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    planes = 7 if bwd else 4
    gbs = planes * n * n * 4 * B / (us * 1e-6) / 1e9
    return dict(B=B, bwd=bwd, us=round(us, 2), GBps=round(gbs, 1), frac_8TBps=round(gbs / 8000, 4))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'pmc':      # one configuration, few launches: for rocprofv3 --pmc
        print(json.dumps(time_loss(16384, bwd=True, iters=5)), flush=True)
        sys.exit(0)
    for bwd in (True, False):
        for B in (32, 256, 2048, 16384):
            print(json.dumps(time_loss(B, bwd=bwd, iters=200 if B <= 256 else 30)), flush=True)
