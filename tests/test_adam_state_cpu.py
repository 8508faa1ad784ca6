"""CPU: the fused trainers' Adam moments <-> torch.optim.Adam.state_dict(): a --mode fused checkpoint
must load with the reference's `optimizer.load_state_dict(checkpoint['optimizer_state_dict'])`
(train_cglow_reverse_kl.py:281-289 there) and the reverse; round-2 flat checkpoints still load."""
from types import SimpleNamespace

import pytest
import torch

from pde_surrogate_amd.train import adam_state_dict, load_adam_state, to_torch_adam_state


def _trainer(step):
    torch.manual_seed(0)
    shapes = [(4, 3, 3, 3), (4,), (4,), (2, 4, 1, 1)]
    params = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    # flat layout = the engine's: NOT in parameters() order (BatchNorm tensors first, then the convolution weights)
    order = [1, 2, 0, 3]
    offsets, off = [0] * 4, 0
    for i in order:
        offsets[i] = off
        off += params[i].numel()
    model = SimpleNamespace(_params=params, _offsets=offsets)
    return SimpleNamespace(model=model, exp_avg=torch.randn(off), exp_avg_sq=torch.rand(off), step_count=step,
                           lr=1e-3, betas=(0.9, 0.999), eps=1e-8, wd=0.0)


def test_fused_state_loads_into_torch_adam_and_back():
    tr = _trainer(7)
    sd = adam_state_dict(tr)
    opt = torch.optim.Adam(tr.model._params, lr=5e-4)
    opt.load_state_dict(sd)                                   # what the reference / --mode dropin does on resume
    for p, off in zip(tr.model._params, tr.model._offsets):
        st = opt.state[p]
        assert int(st['step']) == 7
        assert torch.equal(st['exp_avg'].reshape(-1), tr.exp_avg[off:off + p.numel()])
        assert torch.equal(st['exp_avg_sq'].reshape(-1), tr.exp_avg_sq[off:off + p.numel()])
    for p in tr.model._params:                                # the loaded optimizer steps
        p.grad = torch.ones_like(p)
    opt.step()
    tr2 = _trainer(0)
    tr2.exp_avg.zero_(); tr2.exp_avg_sq.zero_()
    assert load_adam_state(tr2, opt.state_dict())             # a torch state_dict (e.g. written by the reference) -> flat
    assert tr2.step_count == 8
    for p, off in zip(tr.model._params, tr2.model._offsets):              # (same layout; the optimizer holds tr's parameters)
        assert torch.equal(tr2.exp_avg[off:off + p.numel()], opt.state[p]['exp_avg'].reshape(-1))


def test_round2_flat_format_still_loads_and_converts():
    tr = _trainer(3)
    flat = {'exp_avg': tr.exp_avg.clone(), 'exp_avg_sq': tr.exp_avg_sq.clone(), 'step': 3}
    tr2 = _trainer(0)
    tr2.exp_avg.zero_()
    assert load_adam_state(tr2, flat) and tr2.step_count == 3 and torch.equal(tr2.exp_avg, tr.exp_avg)
    opt = torch.optim.Adam(tr.model._params)
    opt.load_state_dict(to_torch_adam_state(flat, tr.model))
    p, off = tr.model._params[0], tr.model._offsets[0]
    assert torch.equal(opt.state[p]['exp_avg'].reshape(-1), tr.exp_avg[off:off + p.numel()])
    # the flat format carries no hyper-parameters and load_state_dict installs the groups it is handed: with the loading
    # optimizer passed in, ITS --weight-decay / betas / eps survive the resume
    opt = torch.optim.Adam(tr.model._params, lr=3e-4, weight_decay=1e-2, betas=(0.8, 0.99), eps=1e-6)
    opt.load_state_dict(to_torch_adam_state(flat, tr.model, opt))
    g = opt.param_groups[0]
    assert (g['lr'], g['weight_decay'], tuple(g['betas']), g['eps']) == (3e-4, 1e-2, (0.8, 0.99), 1e-6)
    assert torch.equal(opt.state[p]['exp_avg'].reshape(-1), tr.exp_avg[off:off + p.numel()])


def test_unstepped_and_foreign_states():
    tr = _trainer(0)
    sd = adam_state_dict(tr)
    assert sd['state'] == {} and not load_adam_state(_trainer(0), sd)
    other = torch.optim.Adam([torch.nn.Parameter(torch.randn(3))])
    other.param_groups[0]['params'][0].grad = torch.ones(3)
    other.step()
    with pytest.raises(ValueError):
        load_adam_state(_trainer(0), other.state_dict())
