#!/usr/bin/env python
"""Same-process A/B: the training step on torch's default stream vs on a HIGH-priority stream (the weight-gradient streams
stay at the lowest priority): does queue priority protect the finalize -> data-gradient chain from the kernels beside it?"""
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


def main(rounds=3, steps=150, warm=20, B=32):
    dev = torch.device('cuda:0')
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        model = DenseED(1, 3, 64, [6, 8, 6], growth_rate=16, init_features=48)
    tr = MixedResidualTrainer(model, B, 64, lr=1e-3, weight_bound=10.0, device=dev, use_graph=False)
    data = torch.from_numpy(grf_kle_fields(512, cache_dir='/tmp')).to(dev)
    batches = [data[i * B:(i + 1) * B].contiguous() for i in range(512 // B)]
    lo, hi = torch.cuda.Stream.priority_range()
    print('priority range (least, greatest):', lo, hi)
    streams = {'default': None, 'high': torch.cuda.Stream(dev, priority=hi), 'normal_new': torch.cuda.Stream(dev, priority=0)}
    res = {k: [] for k in streams}
    for r in range(rounds):
        for name, s in streams.items():
            ctx = torch.cuda.stream(s) if s is not None else contextlib.nullcontext()
            torch.cuda.synchronize()
            with ctx:
                for i in range(warm):
                    tr.step(batches[i % len(batches)], 1e-6)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(steps):
                    tr.step(batches[i % len(batches)], 1e-6)
                torch.cuda.synchronize()
            res[name].append((time.perf_counter() - t0) / steps * 1e3)
    for k, v in res.items():
        print(f'{k}: ' + ' '.join(f'{t:.4f}' for t in v) + f'  | min {min(v):.4f} ms/step', flush=True)


if __name__ == '__main__':
    main()
