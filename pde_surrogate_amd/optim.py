"""`torch.optim.Adam` for the drop-in loop (reference train_codec_mixed_residual.py:151-152, :239).

The reference writes `optim.Adam(model.parameters(), lr=..., weight_decay=...)` and `optimizer.step()`.  On this build's
modules the 82 parameters are views of ONE flat fp32 buffer and the 82 gradients autograd stores are views of ONE flat
buffer in the same layout, so the step is one launch of the flat HIP kernel (`pdes_adam_step_host`, ~10 us) instead of
torch's per-tensor machinery: the default (`foreach`) implementation costs the host 1.8 ms per step on this net, the
`fused=True` one 0.27 ms of host time and 0.26 ms of GPU time (three multi-tensor launches over 82 small tensors) --
tools/dropin_phases.py, EXPERIMENTS.md round 6.

Two ways in, both without touching the loop body:
  * `from pde_surrogate_amd import optim` in place of `import torch.optim as optim` (the same import
    redirection as for the model / loss modules): `optim.Adam` below -- a subclass of `torch.optim.Adam` that takes the flat
    path whenever it can prove it applies and is torch's own optimiser otherwise (any parameter list, any option);
  * nothing at all: importing `pde_surrogate_amd.models.codec` registers a global optimiser step pre-hook that turns a
    plain `torch.optim.Adam` over exactly one such network's parameters, at its first step, into the subclass below
    (PDES_ADAM_AUTO_FUSED=1: only selects `fused=True` of the same class; =0 leaves it alone).

Everything else of `torch.optim` is re-exported unchanged.
"""
import ctypes
import math
import os

import torch
from torch.optim import *                       # noqa: F401,F403  (the module stands in for torch.optim)
from torch.optim import Adam as _TorchAdam

from . import _lib


def _owner(params):
    """the _HipNet whose flat buffer holds exactly `params` (in order), or None"""
    if not params:
        return None
    net = getattr(params[0], '_pdes_owner', None)
    net = net() if net is not None else None
    if net is None or getattr(net, '_flat', None) is None or len(params) != len(net._params):
        return None
    if any(a is not b for a, b in zip(params, net._params)):
        return None
    return net


class Adam(_TorchAdam):
    """torch.optim.Adam with a one-launch step for the parameters of a HIP DenseED / Decoder (see the module docstring).
    The flat path is taken when: one parameter group holding exactly the network's parameters, amsgrad / maximize /
    capturable / differentiable off, and every `.grad` is the view autograd received from the network's backward (same base
    buffer, the parameter's own offset).  Anything else -- a clipped or replaced gradient, a second group, another model --
    runs torch's implementation on the same state."""

    def __init__(self, params, *args, **kwargs):
        super().__init__(params, *args, **kwargs)
        self._flat_net = None
        self._flat_tries = 0                    # the network flattens its parameters at its first forward: looked for lazily
        self._flat_state = None                 # (exp_avg, exp_avg_sq, step tensor shared by every parameter's state, flat)
        self._hyper = (ctypes.c_float * 8)()

    def _find_net(self):
        if self._flat_net is None and self._flat_tries < 4 and len(self.param_groups) == 1:
            self._flat_tries += 1
            g = self.param_groups[0]
            if not (g.get('amsgrad') or g.get('maximize') or g.get('capturable') or g.get('differentiable')
                    or g.get('foreach') or g.get('fused')):
                self._flat_net = _owner(g['params'])
        return self._flat_net

    # -- state ------------------------------------------------------------------------------------------------------------
    def _make_flat_state(self, net):
        flat = net._flat
        m, v = torch.zeros_like(flat), torch.zeros_like(flat)
        step = torch.tensor(0.0, dtype=torch.float32)
        for p, off in zip(net._params, net._offsets):
            st = self.state[p]
            n = p.numel()
            if 'exp_avg' in st:                  # torch's implementation has stepped before (or a state_dict was loaded)
                m[off:off + n].copy_(st['exp_avg'].reshape(-1))
                v[off:off + n].copy_(st['exp_avg_sq'].reshape(-1))
                step.fill_(float(st['step']))
            st['step'] = step
            st['exp_avg'] = m[off:off + n].view(p.shape)
            st['exp_avg_sq'] = v[off:off + n].view(p.shape)
        self._flat_state = (m, v, step, flat)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._flat_state = None                 # the loaded tensors are torch's own: re-flattened by the next flat step

    def _flat_grad(self, net):
        """address of the flat buffer the 82 `.grad` tensors tile, or None when they do not: autograd keeps the views the
        network's backward returned (it detaches them -- `_base` is gone -- but does not copy), so gradient i sits 4 * offset_i
        bytes behind the first one exactly when nobody replaced it"""
        p0, off0 = net._params[0], net._offsets[0]
        g0 = p0.grad
        if g0 is None:
            return None
        base = g0.data_ptr() - 4 * off0
        for p, off in zip(net._params, net._offsets):
            g = p.grad
            if g is None or g.data_ptr() != base + 4 * off or g.dtype != torch.float32 or g.shape != p.shape or not g.is_contiguous():
                return None
        return base

    # -- step -------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        net = self._find_net()
        gflat = self._flat_grad(net) if (net is not None and closure is None) else None
        if gflat is None or net._flat is None or (self._flat_state is not None and self._flat_state[3] is not net._flat):
            if self._flat_state is not None:     # torch's path keeps a step count per parameter: un-share it
                s = self._flat_state[2]
                for p in net._params:
                    self.state[p]['step'] = s.clone()
                self._flat_state = None
            return super().step(closure)
        if any(('step' in self.state[p]) and self.state[p]['step'].device != torch.device('cpu') for p in net._params[:1]):
            return super().step(closure)         # (a loaded fused / capturable state keeps its step count on the device)
        if self._flat_state is None:
            self._make_flat_state(net)
        m, v, step, flat = self._flat_state
        g = self.param_groups[0]
        step += 1
        k = int(step)
        b1, b2 = g['betas']
        lr = float(g['lr'])
        h = self._hyper
        h[0], h[1], h[2], h[3], h[4] = lr, b1, b2, g['eps'], g['weight_decay']
        h[5], h[6] = 1.0 - b1 ** k, math.sqrt(1.0 - b2 ** k)
        with _lib.device_guard(flat.device):
            rc = _lib.lib().pdes_adam_step_host(flat.data_ptr(), gflat, m.data_ptr(), v.data_ptr(), h, 1.0, 0,
                                                flat.numel(), _lib.stream_ptr(flat.device))
        _lib.check(rc, 'pdes_adam_step_host')
        # what an in-place update does for autograd: the version counters move, so that a step taken between a forward and
        # its backward is still caught (_NetFn.backward compares them)
        torch.autograd.graph.increment_version(net._params)
        return None


_hook_handle = None


def _auto_fused_hook(optimizer, args, kwargs):
    """global step pre-hook: a plain torch.optim.Adam over exactly one HIP network's parameters, about to take its FIRST
    step with neither `foreach` nor `fused` chosen.  PDES_ADAM_AUTO_FUSED (read here, per optimiser):
      2 (default)  the object becomes a `pde_surrogate_amd.optim.Adam` (the subclass above: same state, same state_dict, still
                   an instance of torch.optim.Adam).  THIS step is already dispatched to torch's implementation and runs it
                   once; every later step is one launch of the flat kernel while the gradients are the backward's own views,
                   torch's implementation otherwise -- exactly what `import pde_surrogate_amd.optim as optim` gives
      1            `fused=True` of the same class (0.27 ms of host time per step instead of 1.8; round 6's first form)
      0            nothing"""
    if type(optimizer) is not _TorchAdam or getattr(optimizer, '_pdes_checked', False):
        return None
    optimizer._pdes_checked = True
    mode = os.environ.get('PDES_ADAM_AUTO_FUSED', '2')
    if mode == '0' or len(optimizer.param_groups) != 1:
        return None
    g = optimizer.param_groups[0]
    if g.get('foreach') is not None or g.get('fused') is not None or g.get('capturable') or g.get('differentiable'):
        return None
    if optimizer.state or _owner(g['params']) is None:       # (already stepped: its state is in the foreach layout)
        return None
    if mode == '1' or g.get('amsgrad') or g.get('maximize'):
        g['fused'] = True
        return None
    try:
        optimizer.__class__ = Adam
        optimizer._flat_net, optimizer._flat_tries, optimizer._flat_state = None, 0, None
        optimizer._hyper = (ctypes.c_float * 8)()
        patch = getattr(optimizer, '_patch_step_function', None)
        if patch is not None:                    # (torch wraps `step` of an optimiser's CLASS at construction: do it for ours;
            patch()                              #  without it the subclass's step simply runs unwrapped -- no hooks, same update)
    except Exception:                            # never let an acceleration break the user's optimiser: back to the plain class
        optimizer.__class__ = _TorchAdam
        g['fused'] = True
    return None


def install_auto_fused_hook():
    """idempotent; called when pde_surrogate_amd.models.codec is imported"""
    global _hook_handle
    if _hook_handle is None:
        from torch.optim.optimizer import register_optimizer_step_pre_hook
        _hook_handle = register_optimizer_step_pre_hook(_auto_fused_hook)
    return _hook_handle
