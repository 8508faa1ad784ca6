#!/usr/bin/env python
"""Per-layer micro-benchmark of the DenseED convolution kernels (HIP-event timed): forward,
data gradient and weight gradient of every layer of the default net at batch B.
Prints us and achieved TFLOP/s (2*Cout*Cin*k*k*Hout*Wout*B flops per pass)."""
import contextlib
import ctypes
import io
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pde_surrogate_amd import _lib
from pde_surrogate_amd.models.codec import DenseED


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main(B=32, only=None):
    dev = torch.device('cuda:0')
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        net = DenseED(1, 3, 64, [6, 8, 6]).to(dev).train()
    x = torch.exp(0.5 * torch.randn(B, 1, 64, 64, device=dev))
    y = net(x)
    eng = net._engine(x)
    gy = torch.randn_like(y)
    eng.backward(gy)          # fills T/G buffers so every backward kernel has sane inputs
    torch.cuda.synchronize()
    L, st = _lib.lib(), _lib.stream_ptr()
    tot = [0.0, 0.0, 0.0]
    for i, s in enumerate(net._specs):
        if only is not None and i not in only:
            continue
        d = eng.descs[i]
        if os.environ.get('BENCH_FUSED') == '1' and d.fin_tstats and s.norm is not None and s.k == 3 and s.stride == 1 \
                and not s.up and s.cout <= 16:
            d.g_fused = 1          # time the finalize-on-load variants (values are meaningless here, timing is not)
        ref = ctypes.byref(d)
        fl = 2.0 * s.cout * s.cin * s.k * s.k * d.Hout * d.Wout * B
        t_f = timeit(lambda: L.pdes_conv_forward(eng.ctx, ref, 1, st))
        t_w = timeit(lambda: L.pdes_conv_backward_weight(eng.ctx, ref, 1, st))
        t_d = timeit(lambda: L.pdes_conv_backward_data(eng.ctx, ref, 1, st)) if s.norm is not None else 0.0
        tot[0] += t_f; tot[1] += t_w; tot[2] += t_d
        print(f'{i:2d} {s.conv:34s} {s.cin:3d}->{s.cout:3d} k{s.k} s{s.stride} up{s.up} {d.Hout:2d}x{d.Wout:2d} '
              f'fwd {t_f:7.1f}us {fl / t_f / 1e6:6.1f}TF | wgrad {t_w:7.1f}us {fl / t_w / 1e6:6.1f}TF | '
              f'dgrad {t_d:7.1f}us {(fl / t_d / 1e6 if t_d else 0):6.1f}TF', flush=True)
    print('sum fwd %.1f us, wgrad %.1f us, dgrad %.1f us' % tuple(tot))


if __name__ == '__main__':
    only = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else None
    main(only=only)
