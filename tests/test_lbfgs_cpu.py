"""FlatLBFGS (pde_surrogate_amd/lbfgs.py) against torch.optim.LBFGS -- the optimiser the reference's solver uses
(solve_conv_mixed_residual.py:124: lr 0.5, max_iter 20, history_size 50, no line search).  Same algorithm and stopping
rules, inner products summed in another order: the iterates must agree closely on problems small and well-conditioned
enough for that to be meaningful, including after the history ring wraps around."""
import numpy as np
import pytest
import torch

from pde_surrogate_amd.lbfgs import FlatLBFGS


def _problem(kind, n, seed):
    g = torch.Generator().manual_seed(seed)
    if kind == 'quadratic':
        A = torch.randn(n, n, generator=g, dtype=torch.float64)
        A = A @ A.t() / n + 0.5 * torch.eye(n, dtype=torch.float64)
        b = torch.randn(n, generator=g, dtype=torch.float64)
        return lambda x: 0.5 * x @ A @ x - b @ x
    scale = torch.linspace(1.0, 3.0, n - 1, dtype=torch.float64)
    return lambda x: ((1 - x[:-1]) ** 2).sum() + (scale * (x[1:] - x[:-1] ** 2) ** 2).sum()     # Rosenbrock-like


@pytest.mark.parametrize('kind,n,hist,epochs', [('quadratic', 40, 50, 6), ('quadratic', 60, 5, 8), ('rosen', 12, 50, 10),
                                                ('rosen', 30, 4, 12)])
def test_iterates_match_torch_lbfgs(kind, n, hist, epochs):
    f = _problem(kind, n, 0)
    x0 = 0.1 * torch.randn(n, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    # torch.optim.LBFGS
    xa = x0.clone().requires_grad_(True)
    opt = torch.optim.LBFGS([xa], lr=0.5, max_iter=20, history_size=hist)

    def closure_a():
        opt.zero_grad()
        loss = f(xa)
        loss.backward()
        return loss
    # FlatLBFGS on a flat parameter and a flat gradient buffer written in place by the closure
    xb, gb = x0.clone(), torch.zeros(n, dtype=torch.float64)

    def closure_b():
        xr = xb.detach().clone().requires_grad_(True)
        loss = f(xr)
        loss.backward()
        gb.copy_(xr.grad)
        return loss.detach()
    flat = FlatLBFGS(xb, gb, lr=0.5, max_iter=20, history_size=hist)
    for ep in range(epochs):
        la = float(opt.step(closure_a))
        lb = float(flat.step(closure_b))
        assert abs(la - lb) <= 1e-7 * max(1.0, abs(la)), (ep, la, lb)
        np.testing.assert_allclose(xb.numpy(), xa.detach().numpy(), rtol=1e-6, atol=1e-8, err_msg=f'epoch {ep}')
    assert opt.state[xa]['func_evals'] == flat.func_evals
    assert float(f(xb)) < float(f(x0))


def test_fp32_buffers_and_early_exit():
    f = _problem('quadratic', 20, 3)
    x = torch.zeros(20)
    g = torch.zeros(20)

    def closure():
        xr = x.detach().double().requires_grad_(True)
        loss = f(xr)
        loss.backward()
        g.copy_(xr.grad.float())
        return loss.detach().float()
    opt = FlatLBFGS(x, g, lr=1.0, max_iter=50, history_size=10)
    l0 = float(opt.step(closure))
    l1 = float(opt.step(closure))
    assert l1 < l0 and float(g.abs().max()) < 1e-3
    n = opt.func_evals
    opt.step(closure)                                    # already converged: a step evaluates once or a few times and stops
    assert opt.func_evals - n <= 3
    with pytest.raises(ValueError):
        FlatLBFGS(torch.zeros(3, 3), torch.zeros(3, 3))
