#!/usr/bin/env python
"""Same-process A/B of run-time knobs (options of the library's context, pdes_context_set_option):
    python tools/ab_env.py PDES_MFMA_1X1 0 1 2 3        -> ms per training step for each value, interleaved rounds
    python tools/ab_env.py PDES_MFMA_NG 1 2 -- PDES_FEW_R 2 4    several knobs, one after the other ('-' = unset)
Different gpurun boxes differ by ~2 %, one process repeats to ~0.1 %, so knob decisions are made here."""
import contextlib
import io
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pde_surrogate_amd import _lib
from pde_surrogate_amd.models.codec import DenseED
from pde_surrogate_amd.train import MixedResidualTrainer
from pde_surrogate_amd.utils.data import grf_kle_fields


def main(groups, rounds=3, steps=150, warm=20, B=32):
    dev = torch.device('cuda:0')
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        model = DenseED(1, 3, 64, [6, 8, 6], growth_rate=16, init_features=48)
    tr = MixedResidualTrainer(model, B, 64, lr=1e-3, weight_bound=10.0, device=dev, use_graph=False)
    data = torch.from_numpy(grf_kle_fields(512, cache_dir='/tmp')).to(dev)
    batches = [data[i * B:(i + 1) * B].contiguous() for i in range(512 // B)]
    for name, values in groups:
        res = {v: [] for v in values}
        for r in range(rounds):
            for v in values:
                _lib.set_option(name, None if v == '-' else v)
                if hasattr(tr.eng, '_reduce_n'):
                    del tr.eng._reduce_n              # re-plan the split-K scratch under the new options
                for i in range(warm):
                    tr.step(batches[i % len(batches)], 1e-4)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(steps):
                    tr.step(batches[i % len(batches)], 1e-4)
                torch.cuda.synchronize()
                res[v].append((time.perf_counter() - t0) / steps * 1e3)
        _lib.set_option(name, None)
        for v in values:
            print(f'{name}={v}: ' + ' '.join(f'{t:.4f}' for t in res[v]) + f'  | min {min(res[v]):.4f} ms/step', flush=True)


if __name__ == '__main__':
    groups, cur = [], []
    for a in sys.argv[1:] + ['--']:
        if a == '--':
            if cur:
                groups.append((cur[0], cur[1:]))
            cur = []
        else:
            cur.append(a)
    main(groups)
