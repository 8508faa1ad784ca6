#!/usr/bin/env python
"""ONE launch mode of the training step per process (streams of several trainers in one process share hardware queues
and distort each other): python tools/archive/ab_segments.py <eager|segments|forward> [PDES_SEG_MAX] -- prints ms/step
(GPU-bound, long unsynchronised run), the host's enqueue time (bursts of 10 behind a synchronise) and a checksum of the
parameters after the run (equal across modes: same kernels, same order).  tools/archive/ab_segments.sh alternates the modes."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pde_surrogate_amd.models.codec import DenseED
from pde_surrogate_amd.train import MixedResidualTrainer
from pde_surrogate_amd.utils.data import grf_kle_fields

mode = sys.argv[1] if len(sys.argv) > 1 else 'eager'
dev = torch.device('cuda:0')
data = torch.from_numpy(grf_kle_fields(512, cache_dir='/tmp')).to(dev)
idx = [torch.arange(i * 32, (i + 1) * 32, device=dev) for i in range(16)]
torch.manual_seed(1)
with contextlib.redirect_stdout(io.StringIO()):
    model = DenseED(1, 3, 64, [6, 8, 6])
tr = MixedResidualTrainer(model, 32, 64, lr=1e-3, device=dev, use_graph=False if mode == 'eager' else mode)


def run(k):
    for i in range(k):
        tr.load_batch(data, idx[i % 16]); tr.step(None, 1e-4)


run(300)
torch.cuda.synchronize()
tag = f"{mode}{'/' + os.environ['PDES_SEG_MAX'] if os.environ.get('PDES_SEG_MAX') else ''}{'/splitw' if os.environ.get('PDES_SEG_SPLITW') == '1' else ''}"
res = []
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(200)
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 200)
enq = []
for r in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(10)
    enq.append((time.perf_counter() - t0) / 10)
    torch.cuda.synchronize()
print(f'{tag:24s} ms/step {" ".join(f"{v * 1e3:.4f}" for v in res)}   host enqueue {sorted(enq)[2] * 1e3:.3f} ms/step   '
      f'checksum {float(tr.flat.double().sum()):.10e}', flush=True)
