#!/usr/bin/env python
"""same-process A/B of library options on the conditional-Glow reverse-KL step (boxes differ by up to 16 %):
    python tools/archive/ab_cglow.py PDES_MFMA_SMALL 1 2 [rounds]"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from pde_surrogate_amd import _lib
from pde_surrogate_amd.models.glow_msc import MultiScaleCondGlow
from pde_surrogate_amd.train import ReverseKLTrainer
from pde_surrogate_amd.utils.data import grf_kle_fields


def main():
    key, values = sys.argv[1], sys.argv[2:4]
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    dev = torch.device('cuda:0')
    B = 32
    data = torch.from_numpy(grf_kle_fields(B, 32, 100, cache_dir='/tmp')).to(dev)
    torch.manual_seed(1)
    np.random.seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        net = MultiScaleCondGlow(32, 1, 3, [3, 4, 4], [6, 6, 6], LUdecompose=True).to(dev).train()
    tr = ReverseKLTrainer(net, B, 32, device=dev)
    for _ in range(15):
        tr.step(data, 1e-4)
    res = {v: [] for v in values}
    for r in range(rounds):
        for v in values:
            _lib.set_option(key, None if v == 'default' else v)
            for _ in range(5):
                tr.step(data, 1e-4)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(40):
                tr.step(data, 1e-4)
            torch.cuda.synchronize()
            res[v].append((time.perf_counter() - t0) / 40 * 1e3)
    for v in values:
        print(f'{key}={v}: ms/step', ' '.join(f'{t:.3f}' for t in res[v]), ' median', f'{np.median(res[v]):.3f}')


if __name__ == '__main__':
    main()
