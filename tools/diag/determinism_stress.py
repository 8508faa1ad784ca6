"""Stress the bitwise reproducibility of the fused step's gradients (fixed-order reductions): the same input through
forward + loss + backward N times, every gradient buffer compared bit for bit with the first; mismatches are reported per
parameter tensor (which localises a racy kernel).  Three legs: one trainer; two trainers alternating (they share the device's
weight-gradient streams and the context's events -- the shape of the data-parallel emulation in the GPU tests); three Adam steps
from the same initial parameters, repeated.
    gpurun -- 'python tools/diag/determinism_stress.py [N]'"""
import contextlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pde_surrogate_amd.models.codec import DenseED
from pde_surrogate_amd.train import MixedResidualTrainer
from pde_surrogate_amd.utils.data import grf_kle_fields

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device('cuda:0')
B = 32
data = torch.from_numpy(grf_kle_fields(6 * B, n_kle=64, cache_dir='/tmp')).to(dev)


def make():
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        net = DenseED(1, 3, 64, blocks=[6, 8, 6]).to(dev).train()
    return net, MixedResidualTrainer(net, B, 64, lr=1e-3, device=dev)


def report(net, ref, got, tag):
    names = [k for k, _ in net.named_parameters()]
    bad = []
    for k, p, off in zip(names, net._params, net._offsets):
        a, b = ref[off:off + p.numel()], got[off:off + p.numel()]
        if not torch.equal(a, b):
            d = (a - b).abs()
            bad.append((k, int((d > 0).sum()), float(d.max()), float((a - b).norm() / a.norm().clamp_min(1e-30))))
    print(f'  {tag}: {len(bad)} tensors differ:', bad[:6], flush=True)


def grads(tr, x):
    tr.x_static.copy_(x)
    tr.gflat.zero_()
    tr._grad_clean = False
    tr._compute()
    return tr.gflat.clone(), tr.terms.clone()


net, tr = make()
ref, tref = grads(tr, data[:B])
n_bad = 0
for i in range(N):
    g, t = grads(tr, data[:B])
    if not (torch.equal(g, ref) and torch.equal(t, tref)):
        n_bad += 1
        report(net, ref, g, f'one trainer, repeat {i}')
print(f'leg 1 (one trainer, {N} repeats): {n_bad} mismatching runs', flush=True)

net2, tr2 = make()
ref2, _ = grads(tr2, data[B:2 * B])
n_bad = 0
for i in range(N):
    ga, _ = grads(tr, data[:B])
    gb, _ = grads(tr2, data[B:2 * B])
    if not torch.equal(ga, ref):
        n_bad += 1
        report(net, ref, ga, f'alternating, trainer 0, repeat {i}')
    if not torch.equal(gb, ref2):
        n_bad += 1
        report(net2, ref2, gb, f'alternating, trainer 1, repeat {i}')
print(f'leg 2 (two trainers alternating, {N} repeats each): {n_bad} mismatching runs', flush=True)

init = tr.flat.clone()
final = None
n_bad = 0
for i in range(max(N // 5, 20)):
    tr.flat.copy_(init)
    tr.exp_avg.zero_(); tr.exp_avg_sq.zero_(); tr.step_count = 0
    tr.gflat.zero_(); tr._grad_clean = False
    for s in range(3):
        tr.step(data[s * B:(s + 1) * B], 1e-3)
    torch.cuda.synchronize()
    f = tr.flat.clone()
    if final is None:
        final = f
    elif not torch.equal(f, final):
        n_bad += 1
        report(net, final, f, f'3 Adam steps, repeat {i}')
print(f'leg 3 (3 Adam steps from the same start, {max(N // 5, 20)} repeats): {n_bad} mismatching runs', flush=True)
