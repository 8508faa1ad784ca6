"""Darcy-flow physics losses for ConvNet surrogates on MI355X -- drop-in for the ConvNet half of
the reference's models/darcy.py (:147-233), backed by ONE fused HIP kernel
(csrc/darcy_loss.hip, C ABI `pdes_darcy_loss`).

Same names, argument meaning and return values as the reference:
    conv_constitutive_constraint(input, output, sobel_filter)                  darcy.py:162-176
    conv_constitutive_constraint_nonlinear(input, output, sobel_filter, b1, b2) darcy.py:179-191
    conv_continuity_constraint(output, sobel_filter, use_tb=True)               darcy.py:210-224
    conv_boundary_condition(output) -> (loss_dirichlet, loss_neumann)           darcy.py:226-233
All results are 0-dim tensors differentiable wrt `output` (callers just do loss.backward()).

`darcy_mixed_residual_loss` is the additional fused fast path: the whole loss of
train_codec_mixed_residual.py:228-232 and its gradient wrt `output` in a single launch.

There is no CPU path: tensors must live on the GPU and the HIP library must be built.
"""
import torch

from .. import _lib


def _check_fields(input, output):
    _lib.require_cuda(input, output)
    if output.dim() != 4 or output.shape[1] != 3:
        raise ValueError(f'output must be (B, 3, H, W): u, sigma1, sigma2; got {tuple(output.shape)}')
    B, _, H, W = output.shape
    if input is not None and tuple(input.shape) != (B, 1, H, W):
        raise ValueError(f'input must be (B, 1, H, W) = {(B, 1, H, W)}; got {tuple(input.shape)}')
    if input is not None and input.dtype != torch.float32 or output.dtype != torch.float32:
        raise RuntimeError('the HIP loss kernel computes in fp32: pass fp32 tensors')
    return B, H, W


def _check_sobel(sobel_filter, H):
    """-> the filter's `correct` flag (image_gradient.py:26); the kernels apply its stencils themselves"""
    if sobel_filter is None:
        return True
    n = getattr(sobel_filter, 'imsize', H)
    if n != H:
        raise ValueError(f'sobel_filter was built for imsize {n}, fields are {H}')
    return bool(getattr(sobel_filter, 'correct', True))


# cross-checks (tests, tools/bench_loss_generic.py): bits OR-ed into every launch's flags -- 16 = PDES_LOSS_GENERIC (the
# any-size kernels at 16 / 32 / 64 too), 8 = PDES_LOSS_TILED (the tile kernel instead of the row-band kernel)
EXTRA_FLAGS = 0


_ZERO_K = {}                      # (device index, B, H, W) -> zeros: the conductivity of launches that take none


def _zero_k(y):
    """the all-zero conductivity plane that conv_continuity_constraint / conv_boundary_condition launch with (neither term
    reads K): allocated once per shape and device instead of a fill launch per call"""
    key = (y.device.index, y.shape[0], y.shape[2], y.shape[3])
    z = _ZERO_K.get(key)
    if z is None:
        if len(_ZERO_K) > 16:
            _ZERO_K.clear()
        z = _ZERO_K[key] = torch.zeros((y.shape[0], 1, y.shape[2], y.shape[3]), device=y.device, dtype=torch.float32)
    return z


def darcy_loss_launch(K, y, weights, want_grad, nonlinear=False, beta1=0.0, beta2=0.0, use_tb=True, correct=True):
    """Raw launch: returns (terms[5] = {total, const, cont, dir, neu} device tensor, grad_y or None).
    use_tb=False: the continuity term leaves out rows 0 and H-1 (darcy.py:224); correct=False: the gradients of
    SobelFilter(correct=False) (image_gradient.py:72-75).  Any square field size.
    `weights`: four Python floats (passed by value), or a DEVICE tensor of four floats (read by the kernels:
    pdes_darcy_loss_dw -- no host synchronisation, which is what an autograd backward needs)."""
    B, H, W = _check_fields(K, y)
    if K is None:
        K = _zero_k(y)
    K = K.detach().contiguous()
    y = y.detach().contiguous()
    flags = (1 if nonlinear else 0) | (0 if use_tb else 2) | (0 if correct else 4) | EXTRA_FLAGS
    partials = torch.empty((_lib.loss_partial_rows(B, H, W, flags), 4), device=y.device, dtype=torch.float32)
    terms = torch.empty(5, device=y.device, dtype=torch.float32)
    grad = torch.empty_like(y) if want_grad else None
    with _lib.device_guard(y.device):
        if torch.is_tensor(weights):
            wd = weights.detach()
            if wd.device != y.device or wd.dtype != torch.float32 or wd.numel() != 4 or not wd.is_contiguous():
                wd = wd.to(device=y.device, dtype=torch.float32).reshape(4).contiguous()
            rc = _lib.lib().pdes_darcy_loss_dw(_lib.context(y.device), _lib.ptr(K), _lib.ptr(y), _lib.ptr(grad),
                                               _lib.ptr(partials), _lib.ptr(terms), B, H, W, _lib.ptr(wd), flags,
                                               float(beta1), float(beta2), _lib.stream_ptr(y.device))
        else:
            w = [float(v) for v in weights]
            rc = _lib.lib().pdes_darcy_loss(_lib.context(y.device), _lib.ptr(K), _lib.ptr(y), _lib.ptr(grad),
                                            _lib.ptr(partials), _lib.ptr(terms), B, H, W, w[0], w[1], w[2], w[3],
                                            flags, float(beta1), float(beta2),
                                            _lib.stream_ptr(y.device))
    _lib.check(rc, 'pdes_darcy_loss')
    return terms, grad


class _MixedResidual(torch.autograd.Function):
    """loss and dL/dy in one launch; backward is a scale by the upstream scalar."""

    @staticmethod
    def forward(ctx, K, y, weight_bound, nonlinear, beta1, beta2, correct=True):
        need = y.requires_grad
        terms, grad = darcy_loss_launch(K, y, (1.0, 1.0, weight_bound, weight_bound), need,
                                        nonlinear, beta1, beta2, True, correct)
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(terms)
        return terms[0].clone(), terms

    @staticmethod
    def backward(ctx, g_loss, _g_terms):
        (grad,) = ctx.saved_tensors
        return None, grad * g_loss, None, None, None, None, None


def darcy_mixed_residual_loss(input, output, weight_bound=10.0, nonlinear=False, beta1=0.0, beta2=0.0, correct=True):
    """Fused loss of train_codec_mixed_residual.py:228-232 (`correct`: the SobelFilter's flag, :155 passes True).

    Returns (loss, loss_pde, loss_dirichlet, loss_neumann): `loss` is differentiable wrt `output`,
    the other three are detached 0-dim tensors for logging."""
    loss, terms = _MixedResidual.apply(input, output, float(weight_bound), bool(nonlinear),
                                       float(beta1), float(beta2), bool(correct))
    return loss, terms[1] + terms[2], terms[3], terms[4]


class _Terms(torch.autograd.Function):
    """All four loss terms from one forward-only launch; backward = one launch whose per-term weights are the upstream
    gradients (exactly autograd's linear combination), read by the kernels from DEVICE memory: no host
    synchronisation inside `loss.backward()`."""

    @staticmethod
    def forward(ctx, K, y, nonlinear, beta1, beta2, use_tb=True, correct=True):
        terms, _ = darcy_loss_launch(K, y, (1.0, 1.0, 1.0, 1.0), False, nonlinear, beta1, beta2, use_tb, correct)
        ctx.save_for_backward(K, y)
        ctx.cfg = (nonlinear, beta1, beta2, use_tb, correct)
        return terms[1:5].clone()

    @staticmethod
    def backward(ctx, g):
        K, y = ctx.saved_tensors
        _forget(ctx)
        _, grad = darcy_loss_launch(K, y, g, True, *ctx.cfg)
        return None, grad, None, None, None, None, None


# The reference's loop body calls three loss functions on one (input, output) pair (train_codec_mixed_residual.py:228-231:
# constitutive + continuity, then the boundary conditions).  Every launch of the fused kernel computes all four terms, so
# the calls that follow the first one on the same `output` return entries of the SAME autograd result: one forward
# launch, and -- because autograd then sums the upstream gradients of that one node -- one backward launch instead of
# three of each.  One pair is remembered per thread; it is dropped when its backward runs, or replaced by the next pair.
import threading
import weakref

_last = {}                      # thread id -> _Shared (autograd's backward runs on another thread: a plain dict + lock)
_last_lock = threading.Lock()


class _Shared:
    # `terms` is a STRONG reference, and has to be: in `constitutive(...) + continuity(...)` the slices die with the sum, and
    # the boundary call two lines later must still find the 4-vector.  It keeps K, y and the graph above `output` alive until
    # the pair's backward runs, the next pair replaces it, or `forget_shared_loss()` is called (one entry per thread;
    # INTEGRATION.md section 1).
    __slots__ = ('out_ref', 'out_version', 'k_ptr', 'k_version', 'nonlinear', 'betas', 'use_tb', 'correct', 'terms', 'node_ref')


def forget_shared_loss():
    """drop the remembered (input, output) pair of the calling thread -- e.g. after an evaluation loop that ran with grad
    mode on and never called backward (the entry would otherwise hold that last graph until the next loss call)"""
    with _last_lock:
        _last.pop(threading.get_ident(), None)


def _share_node(output):
    """one shared node only where it cannot change what autograd allows.  For a LEAF `output` the reference's three loss
    functions build three independent graphs, and `.backward()` on each of them in turn is legal without retain_graph: there
    every call gets its own node.  For a network's output (a non-leaf) separate backward calls without retain_graph fail in
    the reference as well (the network's own graph is freed by the first), so sharing the loss node costs nothing.
    PDES_SHARE_LOSS_NODE=0 turns the sharing off altogether."""
    import os
    return output.grad_fn is not None and os.environ.get('PDES_SHARE_LOSS_NODE', '1') != '0'


def _forget(ctx):
    """the backward of a remembered pair has run (on whichever thread): drop it"""
    with _last_lock:
        for tid, e in list(_last.items()):
            if e.node_ref() is ctx or e.node_ref() is None:
                del _last[tid]


def _terms(input, output, need, nonlinear=False, beta1=0.0, beta2=0.0, use_tb=True, correct=True):
    """the four terms of `output` (and `input`, for the constitutive term) as ONE differentiable 4-vector.
    need: which of its settings this caller depends on -- 'const' (K, law, correct), 'cont' (use_tb, correct), 'bound'"""
    tid = threading.get_ident()
    share = _share_node(output) or not (torch.is_grad_enabled() and output.requires_grad)   # (no graph: nothing to share but the launch)
    e = _last.get(tid) if share else None
    terms = e.terms if e is not None else None
    if terms is not None and e.out_ref() is output and e.out_version == output._version and \
            terms.requires_grad == (torch.is_grad_enabled() and output.requires_grad):
        if need == 'bound':
            return terms
        if need == 'cont' and e.use_tb == use_tb and e.correct == correct:
            return terms
        if need == 'const' and e.k_ptr is not None and input is not None and e.k_ptr == input.data_ptr() and \
                e.k_version == input._version and e.nonlinear == nonlinear and e.betas == (beta1, beta2) and e.correct == correct:
            return terms
    t = _Terms.apply(input, output, nonlinear, beta1, beta2, use_tb, correct)
    if not share:
        return t
    e = _Shared()
    e.out_ref, e.out_version = weakref.ref(output), output._version
    e.k_ptr, e.k_version = (input.data_ptr(), input._version) if input is not None else (None, None)
    e.nonlinear, e.betas, e.use_tb, e.correct = nonlinear, (beta1, beta2), use_tb, correct
    e.terms = t
    # (the ctx handed to _Terms.backward IS the node t.grad_fn: `_forget` compares identities)
    e.node_ref = weakref.ref(t.grad_fn) if t.grad_fn is not None else (lambda: None)
    with _last_lock:
        if len(_last) > 64:     # threads that came and went
            _last.clear()
        _last[tid] = e
    return t


def conv_constitutive_constraint(input, output, sobel_filter):
    """sigma = -K grad(u): mean[(sigma1 + K u_x)^2 + (sigma2 + K u_y)^2]   (darcy.py:162-176)"""
    correct = _check_sobel(sobel_filter, output.shape[-1])
    return _terms(input, output, 'const', False, 0.0, 0.0, True, correct)[0]


def conv_constitutive_constraint_nonlinear(input, output, sobel_filter, beta1, beta2):
    """-K grad(u) = sigma + beta1 sqrt(K) sigma^2 + beta2 K sigma^3          (darcy.py:179-191)"""
    correct = _check_sobel(sobel_filter, output.shape[-1])
    return _terms(input, output, 'const', True, float(beta1), float(beta2), True, correct)[0]


def conv_continuity_constraint(output, sobel_filter, use_tb=True):
    """div(sigma) = 0: mean[(d sigma1/dx + d sigma2/dy)^2]                    (darcy.py:210-224)"""
    correct = _check_sobel(sobel_filter, output.shape[-1])
    return _terms(None, output, 'cont', False, 0.0, 0.0, bool(use_tb), correct)[1]


def conv_boundary_condition(output):
    """(loss_dirichlet, loss_neumann): u=1 left, u=0 right, sigma2=0 top/bottom (darcy.py:226-233)"""
    t = _terms(None, output, 'bound')
    return t[2], t[3]
