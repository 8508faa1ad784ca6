#!/usr/bin/env python
"""The loss-path kernels SURVEY 8(d) gives algorithmic bytes for, HIP-event timed on the stream they are launched on.

    variant                  kernel (csrc/darcy_loss.hip)                    algorithmic bytes per unit
    loss_fwd_bwd             darcy_loss_kernel<64, BWD>                      7 planes x 64*64*4 B = 114,688 B / sample
    loss_fwd_only            darcy_loss_kernel<64, fwd>  (the test() path)   4 planes             =  65,536 B / sample
    loss_nonlinear_fwd_bwd   darcy_loss_kernel<64, BWD, NONLIN>              7 planes             = 114,688 B / sample
    sobel_grad               sobel_grad_kernel<64>    (grad_h + grad_v)      3 planes             =  49,152 B / plane
    sobel_grad_adjoint       sobel_adjoint_kernel<64> (their adjoint)        3 planes             =  49,152 B / plane

`python tools/bench_loss.py`                 the table at B = 32 / 256 (cache-resident, launch-bound) / 16,384 (HBM regime)
`python tools/bench_loss.py pmc <variant>`   a few launches of ONE variant at B = 16,384 for `rocprofv3 --pmc` (tools/pmc_loss_variants.sh)
bench.py imports `time_variant` for the `roofline_variants` object of its line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pde_surrogate_amd import _lib

PLANE = 64 * 64 * 4
HBM_PEAK_GBPS = 8000.0
# variant -> (bytes per unit, what a unit is, the kernel's name in a rocprofv3 trace)
VARIANTS = {
    'loss_fwd_bwd': (7 * PLANE, 'sample', 'darcy_loss_kernel<64, true, false, false, false>'),
    'loss_fwd_only': (4 * PLANE, 'sample', 'darcy_loss_kernel<64, false, false, false, false>'),
    'loss_nonlinear_fwd_bwd': (7 * PLANE, 'sample', 'darcy_loss_kernel<64, true, true, false, false>'),
    'sobel_grad': (3 * PLANE, 'plane', 'sobel_grad_kernel<64>'),
    'sobel_grad_adjoint': (3 * PLANE, 'plane', 'sobel_adjoint_kernel<64>'),
}
# the stand-alone Sobel variants process the three channels (u, sigma1, sigma2) of B samples: 3 B planes per launch
UNITS_PER_SAMPLE = {'sobel_grad': 3, 'sobel_grad_adjoint': 3}


def make_launch(variant, B, dev):
    """-> (callable that enqueues ONE launch on the current stream, units per launch)"""
    L, st, ctx = _lib.lib(), _lib.stream_ptr(), _lib.context(dev)
    if variant.startswith('loss_'):
        K = torch.exp(0.5 * torch.randn(B, 1, 64, 64, device=dev))
        y = torch.randn(B, 3, 64, 64, device=dev)
        g = torch.empty_like(y) if variant != 'loss_fwd_only' else None
        part = torch.empty(B, 4, device=dev)
        nl = 1 if variant == 'loss_nonlinear_fwd_bwd' else 0
        gp = g.data_ptr() if g is not None else None

        def run(_keep=(K, y, g, part)):
            rc = L.pdes_darcy_loss(ctx, K.data_ptr(), y.data_ptr(), gp, part.data_ptr(), None, B, 64, 64,
                                   1.0, 1.0, 10.0, 10.0, nl, 0.1 if nl else 0.0, 0.1 if nl else 0.0, st)
            assert rc == 0, rc
        return run, B
    n = B * UNITS_PER_SAMPLE[variant]
    a = torch.randn(n, 1, 64, 64, device=dev)
    b = torch.randn(n, 1, 64, 64, device=dev)
    c = torch.empty(n, 1, 64, 64, device=dev)
    if variant == 'sobel_grad':
        def run(_keep=(a, b, c)):
            rc = L.pdes_sobel_grad(a.data_ptr(), b.data_ptr(), c.data_ptr(), n, 64, 64, 1, st)
            assert rc == 0, rc
    else:
        def run(_keep=(a, b, c)):
            rc = L.pdes_sobel_grad_adjoint(a.data_ptr(), b.data_ptr(), c.data_ptr(), n, 64, 64, 1, st)
            assert rc == 0, rc
    return run, n


def time_variant(variant, B, iters, warmup=10, dev=None):
    """average microseconds per launch over `iters` back-to-back launches behind `warmup` untimed ones (HIP events on the
    launch stream) -> dict with the roofline figures"""
    dev = dev or torch.device('cuda:0')
    run, units = make_launch(variant, B, dev)
    for _ in range(warmup):
        run()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize(dev)
    us = e0.elapsed_time(e1) * 1e3 / iters
    per_unit, unit, kernel = VARIANTS[variant]
    nbytes = per_unit * units
    gbs = nbytes / (us * 1e-6) / 1e9
    return {'variant': variant, 'kernel': kernel, 'batch': B, 'units_per_launch': units, 'unit': unit,
            'algorithmic_bytes_per_unit': per_unit, 'algorithmic_bytes_per_launch': nbytes, 'us_per_launch': round(us, 2),
            'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBPS, 'frac': round(gbs / HBM_PEAK_GBPS, 4), 'launches_timed': iters,
            'warmup_launches': warmup}


def source_fingerprint():
    """sha256 over the sources the loss-path kernels are built from: a counter file measured on another build of these
    kernels is stale (bench.py refuses it)"""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    # (the five measured kernels are all in darcy_loss.hip; darcy_generic.h: wmul / sqrt_f; the any-size kernels are not measured)
    for f in ('darcy_loss.hip', 'darcy_generic.h', 'pdes_common.h'):
        with open(os.path.join(root, 'pde_surrogate_amd', 'csrc', f), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'pmc':      # one variant, few launches: for rocprofv3 --pmc
        v = sys.argv[2] if len(sys.argv) > 2 else 'loss_fwd_bwd'
        print(json.dumps(time_variant(v, 16384, iters=10, warmup=2)), flush=True)
        sys.exit(0)
    for v in VARIANTS:
        for B, it, w in ((32, 200, 20), (256, 200, 20), (2048, 50, 20), (16384, 100, 100)):
            print(json.dumps(time_variant(v, B, it, w)), flush=True)
