#!/usr/bin/env python
"""config 5 closure (Decoder [8, 6], nonlinear Darcy, B = 1) in a FRESH process: evaluations per second as one hipGraph
replay and as eager launches (three streams), each alone.   python tools/bench_solver.py [graph|eager]"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pde_surrogate_amd.models.codec import Decoder
from pde_surrogate_amd.solver import ResidualClosure
from pde_surrogate_amd.utils.data import grf_kle_fields

dev = torch.device('cuda:0')
K = torch.from_numpy(grf_kle_fields(9, n_kle=1024, cache_dir='/tmp')[[8]]).to(dev)
torch.manual_seed(0)
z = (torch.randn(1, 1, 16, 16) * 0.5).to(dev)
for mode in (sys.argv[1:] or ['graph', 'eager']):
    with contextlib.redirect_stdout(io.StringIO()):
        net = Decoder(1, 3, [8, 6]).to(dev).train()
    clo = ResidualClosure(net, z, K, 10.0, True, 0.1, 0.1, use_graph=mode == 'graph')
    for _ in range(20):
        float(clo())
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(300):
            float(clo())
        best = max(best, 300 / (time.perf_counter() - t0))
    print(f'{mode:6s} {best:8.1f} closure evaluations per second (loss read back on the host each time)', flush=True)
    from pde_surrogate_amd.lbfgs import FlatLBFGS
    opt = FlatLBFGS(net._flat, net._gscratch, lr=0.5, max_iter=20, history_size=50)
    opt.step(clo)
    n0 = clo.n_calls
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(15):
        opt.step(clo)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'{mode:6s} FlatLBFGS {15 / dt:6.2f} epochs per second, {(clo.n_calls - n0) / dt:7.1f} closure evaluations per second', flush=True)
