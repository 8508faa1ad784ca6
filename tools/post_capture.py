#!/usr/bin/env python
"""What slows the eager step down after a hipGraph capture in the same process?

One process, one eager MixedResidualTrainer (default DenseED, B = 32); its step is timed (best of 3 x 100 steps) after each
of a list of treatments, in the order given on the command line (default: all):
    base      nothing
    streams   eight more low-priority HIP streams exist and have run a kernel each (a second / third trainer's side streams)
    tgraph    a torch.cuda.graph capture + 10 replays of a trivial kernel (what the config-5 solver does)
    pgraph    this trainer switched to 'forward' (pdes_graph capture of the forward pass), 20 steps, switched back
    trim      hipDeviceGraphMemTrim + empty_cache
    trainer2  a second trainer (own net, own side streams) has run 20 eager steps and is kept alive
    seg2      a second trainer in 'segments' mode has run 20 steps and is kept alive
    drop2     the second trainers are deleted
    fresh     build a NEW trainer of the timed kind now and time it (then drop it)
With `cglow` as the FIRST argument the timed step is the conditional-Glow reverse-KL step (the leg the slowdown was seen on:
5.33 -> 6.30 ms after the config-5 solver leg of bench.py), and these treatments exist too:
    solver        bench.config5_timing: hipGraph closure, eager closure, torch.optim.LBFGS / FlatLBFGS epochs, autograd closure
    solver_graph  only a ResidualClosure with use_graph=True, 100 evaluations
    solver_eager  only a ResidualClosure with use_graph=False, 100 evaluations
    lbfgs         four torch.optim.LBFGS epochs on an eager closure
    autograd      100 autograd closures on the drop-in modules
"""
import contextlib
import ctypes
import gc
import io
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from pde_surrogate_amd.models.codec import DenseED
    from pde_surrogate_amd.train import MixedResidualTrainer
    from pde_surrogate_amd.utils.data import grf_kle_fields
    dev = torch.device('cuda:0')
    data = torch.from_numpy(grf_kle_fields(512, cache_dir='/tmp')).to(dev)
    perm = torch.randperm(512, device=dev)

    def make(mode=False):
        torch.manual_seed(1)
        with contextlib.redirect_stdout(io.StringIO()):
            m = DenseED(1, 3, 64, [6, 8, 6])
        return MixedResidualTrainer(m, 32, 64, lr=1e-3, device=dev, use_graph=mode)

    def run(tr, n):
        for i in range(n):
            lo = (i * 32) % (512 - 32)
            tr.load_batch(data, perm[lo:lo + 32])
            tr.step(None, 1e-4)

    def timed(tr, n=100, reps=3):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            run(tr, n)
            torch.cuda.synchronize(dev)
            best = min(best, (time.perf_counter() - t0) / n * 1e3)
        return best

    def make_cglow():
        import numpy as np
        from pde_surrogate_amd.models.glow_msc import MultiScaleCondGlow
        from pde_surrogate_amd.train import ReverseKLTrainer
        d32 = torch.from_numpy(grf_kle_fields(128, 32, 100, cache_dir='/tmp')).to(dev)
        torch.manual_seed(1)
        np.random.seed(1)
        with contextlib.redirect_stdout(io.StringIO()):
            net = MultiScaleCondGlow(32, 1, 3, [3, 4, 4], [6, 6, 6], LUdecompose=True).to(dev).train()
        tr = ReverseKLTrainer(net, 32, 32, lr=1.5e-3, weight_bound=50.0, beta=150.0, device=dev)
        tr._d32 = d32
        return tr

    cglow = len(sys.argv) > 1 and sys.argv[1] == 'cglow'
    if cglow:
        del sys.argv[1]
        T1 = make_cglow()

        def run(tr, n):                                       # noqa: F811
            for i in range(n):
                tr.step(tr._d32[(i % 4) * 32:(i % 4 + 1) * 32], 1e-4)
        run(T1, 30)
    else:
        T1 = make()
        run(T1, 150)
    keep = []

    def solver_bits(which):
        from pde_surrogate_amd.models.codec import Decoder
        from pde_surrogate_amd.models.darcy import darcy_mixed_residual_loss
        from pde_surrogate_amd.solver import ResidualClosure
        K = torch.from_numpy(grf_kle_fields(9, n_kle=1024, cache_dir='/tmp')[[8]]).to(dev)
        z = (torch.randn(1, 1, 16, 16) * 0.5).to(dev)
        with contextlib.redirect_stdout(io.StringIO()):
            net = Decoder(1, 3, [8, 6]).to(dev).train()
        if which in ('solver_graph', 'solver_eager', 'lbfgs'):
            clo = ResidualClosure(net, z, K, 10.0, True, 0.1, 0.1, use_graph=which == 'solver_graph')
            if which == 'lbfgs':
                opt = torch.optim.LBFGS(net.parameters(), lr=0.5, max_iter=20, history_size=50)
                for _ in range(4):
                    opt.step(clo)
            else:
                for _ in range(100):
                    float(clo())
            keep.append(clo)
        else:
            for _ in range(100):
                net.zero_grad()
                loss = darcy_mixed_residual_loss(K, net(z), 10.0, True, 0.1, 0.1)[0]
                loss.backward()
                float(loss)
        keep.append(net)
        torch.cuda.synchronize(dev)

    todo = sys.argv[1:] or ['base', 'streams', 'base', 'tgraph', 'base', 'pgraph', 'trim', 'trainer2', 'seg2', 'drop2', 'base']
    for what in todo:
        if what == 'streams':
            least = torch.cuda.Stream.priority_range()[0]
            for _ in range(8):
                s = torch.cuda.Stream(dev, priority=least)
                with torch.cuda.stream(s):
                    torch.zeros(16, device=dev).add_(1)
                keep.append(s)
            torch.cuda.synchronize(dev)
        elif what == 'tgraph':
            a = torch.zeros(1024, device=dev)
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream(dev)
            with torch.cuda.stream(s):
                a.add_(1)
                torch.cuda.synchronize(dev)
                with torch.cuda.graph(g, stream=s):
                    a.add_(1)
            for _ in range(10):
                g.replay()
            torch.cuda.synchronize(dev)
            keep.append(g)
        elif what == 'pgraph':
            T1.set_launch_mode('forward')
            run(T1, 20)
            torch.cuda.synchronize(dev)
            print('   (forward-graph mode itself: %.4f ms per step)' % timed(T1, 100, 2))
            T1.set_launch_mode(False)
            gc.collect()
        elif what == 'trim':
            hip = ctypes.CDLL('libamdhip64.so')
            print('   hipDeviceGraphMemTrim ->', hip.hipDeviceGraphMemTrim(0))
            gc.collect()
            torch.cuda.empty_cache()
        elif what == 'trainer2':
            t2 = make()
            run(t2, 20)
            torch.cuda.synchronize(dev)
            keep.append(t2)
        elif what == 'seg2':
            t2 = make('segments')
            run(t2, 20)
            torch.cuda.synchronize(dev)
            keep.append(t2)
        elif what == 'solver':
            sys.path.insert(0, ROOT)
            import bench
            bench.config5_timing(dev, n=100)
        elif what in ('solver_graph', 'solver_eager', 'lbfgs', 'autograd'):
            solver_bits(what)
        elif what == 'fresh':                                   # a NEW trainer (new side streams from torch's pool) is timed
            t3 = make_cglow() if cglow else make()
            run(t3, 30)
            print('   a fresh trainer built now: %.4f ms per step' % timed(t3, 40 if cglow else 100), flush=True)
            del t3
            gc.collect()
        elif what == 'drop2':
            keep[:] = [k for k in keep if not isinstance(k, MixedResidualTrainer)]
            gc.collect()
            torch.cuda.empty_cache()
        print('%-12s eager step %.4f ms' % (what, timed(T1, 40 if cglow else 100)), flush=True)


if __name__ == '__main__':
    main()
