#!/usr/bin/env python
"""Same-process decomposition of the training step with the COMPONENT-TIMING build of the library (-DPDES_TIMING_KNOBS:
pdes_backward2 reads the environment variable PDES_TIMING per call; wrong numbers, right launch structure):
    PDES_EXTRA_FLAGS=-DPDES_TIMING_KNOBS python -m pde_surrogate_amd.build --force      (never ship that build)
    python tools/ab_timing.py 0 1 3 4 8 12 ...        -> ms per step for each value of PDES_TIMING, interleaved rounds
bits: 1 no weight-gradient kernels (fork events kept), 2 no fork events, 4 finalize as one workgroup, 8 no finalize launch,
16 no data-gradient kernels, 32 neither weight-gradient kernel nor fork for the dense (finalize-on-load) layers"""
import contextlib
import io
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pde_surrogate_amd.models.codec import DenseED
from pde_surrogate_amd.train import MixedResidualTrainer
from pde_surrogate_amd.utils.data import grf_kle_fields


def main(values, rounds=3, steps=150, warm=20, B=32):
    dev = torch.device('cuda:0')
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        model = DenseED(1, 3, 64, [6, 8, 6], growth_rate=16, init_features=48)
    tr = MixedResidualTrainer(model, B, 64, lr=1e-3, weight_bound=10.0, device=dev, use_graph=False)
    data = torch.from_numpy(grf_kle_fields(512, cache_dir='/tmp')).to(dev)
    batches = [data[i * B:(i + 1) * B].contiguous() for i in range(512 // B)]
    res = {v: [] for v in values}
    for r in range(rounds):
        for v in values:
            os.environ['PDES_TIMING'] = v
            for i in range(warm):
                tr.step(batches[i % len(batches)], 1e-6)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                tr.step(batches[i % len(batches)], 1e-6)
            torch.cuda.synchronize()
            res[v].append((time.perf_counter() - t0) / steps * 1e3)
    for v in values:
        print(f'PDES_TIMING={v}: ' + ' '.join(f'{t:.4f}' for t in res[v]) + f'  | min {min(res[v]):.4f} ms/step', flush=True)


if __name__ == '__main__':
    main(sys.argv[1:] or ['0', '1', '3', '4', '8'])
