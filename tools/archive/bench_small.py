#!/usr/bin/env python
"""stand-alone timing (HIP events, one descriptor at a time) of the conditional Glow's convolutions on the 8x8 level:
forward / data gradient / weight gradient through the C ABI, for option values given on the command line
    python tools/archive/bench_small.py PDES_MFMA_SMALL 1 2 0"""
import contextlib
import ctypes
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from pde_surrogate_amd import _lib
from pde_surrogate_amd.models.codec import ConvDesc
from pde_surrogate_amd.models.glow_msc import MultiScaleCondGlow


def main():
    key, values = sys.argv[1], sys.argv[2:]
    dev = torch.device('cuda:0')
    torch.manual_seed(1)
    np.random.seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        net = MultiScaleCondGlow(32, 1, 3, [3, 4, 4], [6, 6, 6], LUdecompose=True).to(dev).train()
    x = torch.rand(32, 1, 32, 32, device=dev) + 0.5
    y, logp = net.generate(x)
    (y.sum() + logp.sum()).backward()                       # fills every buffer the backward kernels read
    eng = net._engine(x)
    if not hasattr(eng, '_reduce_n'):
        eng._plan_wgrad_scratch()
    L, st = _lib.lib(), _lib.stream_ptr()
    picks = [i for i, s in enumerate(net._specs) if s.kind == 'conv' and eng.buf_hw[s.dst] == (8, 8) and s.k == 3][:4] + \
            [i for i, s in enumerate(net._specs) if s.kind == 'conv' and eng.buf_hw[s.dst] == (8, 8) and s.cout == 24][:1]
    sz = ctypes.sizeof(ConvDesc)
    for v in values:
        _lib.set_option(key, v)
        for i in picks:
            s = net._specs[i]
            d = ctypes.byref(eng.descs, i * sz)
            row = []
            for name, fn in (('fwd', L.pdes_conv_forward), ('dgrad', L.pdes_conv_backward_data), ('wgrad', L.pdes_conv_backward_weight)):
                for _ in range(20):
                    _lib.check(fn(eng.ctx, d, 1, st), name)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(200):
                    fn(eng.ctx, d, 1, st)
                e1.record()
                torch.cuda.synchronize()
                row.append(f'{name} {e0.elapsed_time(e1) / 200 * 1e3:6.1f} us')
            print(f'{key}={v}  {s.conv[-60:]:60s} {s.cin:4d}->{s.cout:3d}  ' + '  '.join(row))


if __name__ == '__main__':
    main()
