"""Test-time metrics and the data-driven loss on the device (csrc/metrics.hip).

  TestMetrics  NRMSE (relative l2) and R^2 of the reference's test() -- train_codec_mixed_residual.py:180-183,
               :196-197 and train_codec_max_likelihood.py:173-176,189-190 -- accumulated in device memory over the
               test batches and read with ONE host sync (the reference syncs and concatenates every batch).
  mse_loss     F.mse_loss(output, target) with its backward (train_codec_max_likelihood.py:170,203-204).
"""
import ctypes

import numpy as np
import torch

from . import _lib


class TestMetrics:
    __test__ = False         # not a pytest class

    def __init__(self, n_channels, device):
        self.C = int(n_channels)
        self.dev = torch.device(device)
        if self.dev.type != 'cuda':
            raise RuntimeError('TestMetrics runs on an MI355X only (no CPU fallback)')
        self.accum = torch.zeros(2 * self.C + 1, device=self.dev, dtype=torch.float64)
        self._scratch = {}

    def reset(self):
        self.accum.zero_()

    def update(self, output, target):
        """accumulate one batch; (B, C, H, W) fp32 device tensors.  Enqueue-only (no host sync)."""
        _lib.require_cuda(output, target)
        if output.shape != target.shape or output.dim() != 4 or output.shape[1] != self.C:
            raise ValueError(f'expected two (B, {self.C}, H, W) tensors; got {tuple(output.shape)} and {tuple(target.shape)}')
        if output.dtype != torch.float32 or target.dtype != torch.float32:
            raise RuntimeError('the metric kernels compute in fp32')
        B, C, H, W = output.shape
        per = self._scratch.get(B)
        if per is None:
            per = self._scratch[B] = torch.empty((B, C, 2), device=self.dev, dtype=torch.float32)
        with _lib.device_guard(self.dev):
            rc = _lib.lib().pdes_test_metrics(_lib.ptr(output.detach()), _lib.ptr(target.detach()), _lib.ptr(per),
                                              _lib.ptr(self.accum), B, C, H * W, _lib.stream_ptr(self.dev))
        _lib.check(rc, 'pdes_test_metrics')
        return per

    def result(self, y_variation):
        """(nrmse per channel, r2 per channel) as numpy arrays -- the one host sync of the test pass"""
        a = self.accum.cpu().numpy()
        n = a[2 * self.C]
        nrmse = (a[:self.C] / n).astype(np.float32)
        r2 = 1 - a[self.C:2 * self.C] / np.asarray(y_variation, np.float64)
        return nrmse, r2


def mse_launch(output, target, want_grad, loss_accum=None):
    """raw launch: (loss 1-element device tensor, grad or None)"""
    _lib.require_cuda(output, target)
    if output.shape != target.shape:
        raise ValueError(f'output {tuple(output.shape)} and target {tuple(target.shape)} differ')
    if output.dtype != torch.float32 or target.dtype != torch.float32:
        raise RuntimeError('the MSE kernel computes in fp32')
    o, t = output.detach().contiguous(), target.detach().contiguous()
    n = o.numel()
    L = _lib.lib()
    partials = torch.empty(L.pdes_mse_partials(n), device=o.device, dtype=torch.float64)
    loss = torch.empty(1, device=o.device, dtype=torch.float32)
    grad = torch.empty_like(o) if want_grad else None
    with _lib.device_guard(o.device):
        rc = L.pdes_mse_loss(_lib.ptr(o), _lib.ptr(t), _lib.ptr(grad), _lib.ptr(partials), _lib.ptr(loss),
                             _lib.ptr(loss_accum), n, _lib.stream_ptr(o.device))
    _lib.check(rc, 'pdes_mse_loss')
    return loss, grad


class _Mse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, output, target):
        loss, grad = mse_launch(output, target, output.requires_grad)
        ctx.save_for_backward(grad)
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None


def mse_loss(output, target):
    """drop-in for F.mse_loss(output, target) (mean reduction), differentiable wrt `output`"""
    return _Mse.apply(output, target)
