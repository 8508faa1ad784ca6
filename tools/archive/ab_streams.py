#!/usr/bin/env python
"""Same-process A/B of the number of weight-gradient streams (a model attribute, not a library option):
    python tools/archive/ab_streams.py [rounds]    -> ms per training step with 1 and 2 side streams, interleaved rounds"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pde_surrogate_amd.models.codec import DenseED
from pde_surrogate_amd.train import MixedResidualTrainer
from pde_surrogate_amd.utils.data import grf_kle_fields


def main(rounds=3, steps=150, warm=20, B=32):
    dev = torch.device('cuda:0')
    data = torch.from_numpy(grf_kle_fields(512, cache_dir='/tmp')).to(dev)
    batches = [data[i * B:(i + 1) * B].contiguous() for i in range(512 // B)]
    trs = {}
    for k in (1, 2):
        torch.manual_seed(1)
        with contextlib.redirect_stdout(io.StringIO()):
            model = DenseED(1, 3, 64, [6, 8, 6])
        model.wgrad_streams = k          # (the default is 2)
        trs[k] = MixedResidualTrainer(model, B, 64, lr=1e-3, device=dev)
    res = {k: [] for k in trs}
    for r in range(rounds):
        for k, tr in trs.items():
            for i in range(warm):
                tr.step(batches[i % len(batches)], 1e-4)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                tr.step(batches[i % len(batches)], 1e-4)
            torch.cuda.synchronize()
            res[k].append((time.perf_counter() - t0) / steps * 1e3)
    for k in trs:
        print(f'wgrad streams {k}: ' + ' '.join(f'{t:.4f}' for t in res[k]) + f'  | min {min(res[k]):.4f} ms/step', flush=True)
    # same arithmetic either way
    a = torch.cat([p.detach().reshape(-1) for p in trs[1].model.parameters()])
    b = torch.cat([p.detach().reshape(-1) for p in trs[2].model.parameters()])
    print('max |param difference| after the same number of steps:', float((a - b).abs().max()))


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
