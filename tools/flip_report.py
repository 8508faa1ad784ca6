#!/usr/bin/env python
"""Where do the matrix-core and the VALU convolution paths differ at a given batch?  Runs the comparison of
tests/test_densed_gpu.py::test_mfma_kernels_match_direct_kernels for several seeds and prints, for the worst gradient
tensors, how much of the squared deviation sits in the single worst channel (a ReLU flip moves one channel of the BatchNorm
gradients of one layer; a wrong tile / split plan moves everything).   python tools/flip_report.py 64 5 6 7"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    from pde_surrogate_amd import _lib
    import test_densed_gpu as T
    B = int(sys.argv[1])
    dev = torch.device('cuda:0')
    for seed in [int(a) for a in sys.argv[2:]] or [5]:
        _lib.set_option('PDES_CONV_IMPL', 'direct')
        y0, l0, g0 = T._run_default(dev, B=B, seed=seed)
        _lib.set_option('PDES_CONV_IMPL', 'auto')
        y1, l1, g1 = T._run_default(dev, B=B, seed=seed)
        print('B %d seed %d: output rel-L2 %.2e  loss rel %.2e' % (
            B, seed, float((y1 - y0).norm() / y0.norm()), abs(l1 - l0) / abs(l0)))
        rows = []
        for k in g0:
            d = (g1[k] - g0[k]).double()
            e = float(d.norm() / g0[k].double().norm())
            # channel axis: dim 0 for BatchNorm vectors, dim 1 (input channel) for convolution weights
            if d.dim() == 1:
                per = d ** 2
            else:
                per = (d ** 2).sum(dim=(0, 2, 3))
            top = float(per.max() / per.sum()) if float(per.sum()) > 0 else 0.0
            rows.append((e, k, int(per.argmax()), top))
        for e, k, c, top in sorted(rows, reverse=True)[:6]:
            print('   %.2e  %-50s worst channel %3d holds %.0f %% of the squared deviation' % (e, k, c, 100 * top))


if __name__ == '__main__':
    main()
