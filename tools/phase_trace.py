#!/usr/bin/env python
"""Developer tool: phase timing inside conv_mfma_kernel.  Needs a trace build:
    PDES_EXTRA_FLAGS=-DPDES_TRACE python -m pde_surrogate_amd.build --force
One workgroup in the middle of the grid stamps s_memrealtime (100 MHz) at phase boundaries."""
import contextlib, ctypes, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pde_surrogate_amd import _lib
from pde_surrogate_amd.models.codec import DenseED


def main(layers, B=32):
    dev = torch.device('cuda:0')
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        net = DenseED(1, 3, 64, [6, 8, 6]).to(dev).train()
    x = torch.exp(0.5 * torch.randn(B, 1, 64, 64, device=dev))
    y = net(x)
    eng = net._engine(x)
    eng.backward(torch.randn_like(y))
    torch.cuda.synchronize()
    L, st = _lib.lib(), _lib.stream_ptr()
    L.pdes_debug_trace.argtypes = [ctypes.c_void_p]
    buf = (ctypes.c_ulonglong * 16)()
    for i in layers:
        s, d = net._specs[i], eng.descs[i]
        ref = ctypes.byref(d)
        for name, fn in (('fwd', L.pdes_conv_forward), ('dgrad', L.pdes_conv_backward_data)):
            for _ in range(3):
                fn(eng.ctx, ref, 1, st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(eng.ctx, ref, 1, st); e1.record()
            torch.cuda.synchronize()
            L.pdes_debug_trace(buf)
            t = list(buf)
            u = lambda a, b: (t[b] - t[a]) * 0.01
            print(f'{i:2d} {s.conv:30s} {name:5s} kernel {e0.elapsed_time(e1)*1e3:6.1f}us | geom+cf {u(0,1):5.2f} first-load {u(1,2):5.2f} '
                  f'commit0 {u(2,3):5.2f} loop {u(3,4):6.2f} [mfma {t[8]*0.01:6.2f} commit {t[9]*0.01:6.2f} barrier {t[10]*0.01:6.2f}] '
                  f'combine {u(4,5):5.2f} epilogue {u(5,6):5.2f} total {u(0,6):6.2f}', flush=True)


if __name__ == '__main__':
    main([int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else [6, 16, 24, 25])
