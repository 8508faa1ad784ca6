"""host time of optimizer.step() in the drop-in loop with a synchronise per step: plain torch.optim.Adam (retargeted by the global
hook) against pde_surrogate_amd.optim.Adam, both orders; counts the flat-kernel launches"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pde_surrogate_amd import _lib, optim as poptim
from pde_surrogate_amd.models.codec import DenseED
from pde_surrogate_amd.models.darcy import darcy_mixed_residual_loss
dev = torch.device('cuda:0')
x = torch.exp(0.3 * torch.randn(32, 1, 64, 64, device=dev))
calls = [0]
L = _lib.lib()
real = L.pdes_adam_step_host
class Spy:
    def __getattr__(self, k):
        if k == 'pdes_adam_step_host':
            def f(*a):
                calls[0] += 1
                return real(*a)
            return f
        return getattr(L, k)
_lib_lib = _lib.lib
def leg(mod, tag, n=60):
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        net = DenseED(1, 3, 64, [6, 8, 6]).to(dev).train()
    opt = mod.Adam(net.parameters(), lr=1e-3)
    t_step, t_all = [], []
    calls[0] = 0
    _lib.lib = lambda: Spy()
    for i in range(n):
        t0 = time.perf_counter()
        net.zero_grad()
        loss = darcy_mixed_residual_loss(x, net(x), 10.0)[0]
        loss.backward()
        for g in opt.param_groups:
            g['lr'] = 1e-3
        t1 = time.perf_counter()
        opt.step()
        t2 = time.perf_counter()
        v = loss.item()
        t3 = time.perf_counter()
        t_step.append((t2 - t1) * 1e3); t_all.append((t3 - t0) * 1e3)
    _lib.lib = _lib_lib
    med = lambda a: sorted(a)[len(a) // 2]
    print(f'{tag}: {type(opt).__module__}.{type(opt).__name__} flat launches {calls[0]} of {n}; step() host median {med(t_step[10:]):.3f} ms, '
          f'first five {[round(t, 2) for t in t_step[:5]]}; whole iteration median {med(t_all[10:]):.3f} ms', flush=True)
leg(torch.optim, 'plain 1')
leg(poptim, 'redirected 1')
leg(torch.optim, 'plain 2')
leg(poptim, 'redirected 2')
