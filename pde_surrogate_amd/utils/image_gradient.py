"""SobelFilter on MI355X -- drop-in for the reference's utils/image_gradient.py:24-92.

grad_h / grad_v are HIP kernels (`pdes_sobel_grad`, and `pdes_sobel_grad_adjoint` for autograd, `pdes_sobel5_*` for
filter_size=5) instead of pad + conv2d + matmul, for any square imsize >= 2 and both values of `correct`
(csrc/darcy_loss.hip: 16 / 32 / 64 specialisations; csrc/darcy_loss_generic.hip: everything else).
"""
import numpy as np
import torch

from .. import _lib


def _launch_grad(image, want_h, want_v, correct, filter_size=3):
    _lib.require_cuda(image)
    if image.dim() != 4 or image.shape[1] != 1:
        raise ValueError(f'image must be (B, 1, H, W); got {tuple(image.shape)}')
    if image.dtype != torch.float32:
        raise RuntimeError('the HIP Sobel kernel computes in fp32')
    B, _, H, W = image.shape
    x = image.detach().contiguous()
    gh = torch.empty_like(x) if want_h else None
    gv = torch.empty_like(x) if want_v else None
    fn = _lib.lib().pdes_sobel_grad if filter_size == 3 else _lib.lib().pdes_sobel5_grad
    with _lib.device_guard(x.device):
        rc = fn(_lib.ptr(x), _lib.ptr(gh), _lib.ptr(gv), B, H, W, 1 if correct else 0, _lib.stream_ptr(x.device))
    _lib.check(rc, 'pdes_sobel_grad')
    return gh, gv


class _Grad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, horizontal, correct, filter_size=3):
        ctx.horizontal, ctx.correct, ctx.filter_size = horizontal, correct, filter_size
        gh, gv = _launch_grad(image, horizontal, not horizontal, correct, filter_size)
        return gh if horizontal else gv

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        B, _, H, W = g.shape
        out = torch.empty_like(g)
        a, b = (g, None) if ctx.horizontal else (None, g)
        fn = _lib.lib().pdes_sobel_grad_adjoint if ctx.filter_size == 3 else _lib.lib().pdes_sobel5_grad_adjoint
        with _lib.device_guard(g.device):
            rc = fn(_lib.ptr(a), _lib.ptr(b), _lib.ptr(out), B, H, W, 1 if ctx.correct else 0, _lib.stream_ptr(g.device))
        _lib.check(rc, 'pdes_sobel_grad_adjoint')
        return out, None, None, None


class SobelFilter(object):
    """Same constructor and methods as the reference (image_gradient.py:24-92)."""

    def __init__(self, imsize, correct=True, device='cpu'):
        self.imsize = imsize
        self.correct = correct
        self.device = device
        # kept for attribute compatibility with reference users (image_gradient.py:43-46)
        modifier = np.eye(imsize)
        modifier[0:2, 0] = np.array([4, -1])
        modifier[-2:, -1] = np.array([-1, 4])
        self.modifier = torch.tensor(modifier, dtype=torch.float32, device=device)

    def _check(self, filter_size):
        if filter_size not in (3, 5):
            raise ValueError(f'filter_size must be 3 or 5 (image_gradient.py:62-67); got {filter_size}')

    def grad_h(self, image, filter_size=3):
        """image gradient along the horizontal direction (x axis), (B,1,H,W) -> (B,1,H,W)"""
        self._check(filter_size)
        return _Grad.apply(image, True, self.correct, filter_size)

    def grad_v(self, image, filter_size=3):
        self._check(filter_size)
        return _Grad.apply(image, False, self.correct, filter_size)
