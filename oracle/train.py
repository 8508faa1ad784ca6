"""Oracle (test infrastructure): host-side schedule, metrics and the full CPU training step.

Restates
  * OneCycleScheduler.step / annealing_*  -- reference utils/practices.py:6-35
  * the training step                     -- reference train_codec_mixed_residual.py:224-240
    (zero_grad, forward, loss, backward, one-cycle LR, Adam step)
  * test-time metrics NRMSE / R^2         -- reference train_codec_mixed_residual.py:180-197
  * y_variation                           -- reference utils/load.py:28-30
It is also the CPU baseline that bench.py times on the GPU box's host cores ("port").
Pinned by tests/golden/G7 (8-step trajectory), G8 (LR values), G9 (metrics).
"""
import math
import numpy as np
import torch

from . import codec, darcy


def one_cycle_lr(pct, lr_max, div_factor=25.0, pct_start=0.3):
    """practices.py:28-35: linear lr_max/div -> lr_max on [0,pct_start], then cosine to lr_low/1e4."""
    lr_low = lr_max / div_factor
    if pct <= pct_start:
        return lr_low + (pct / pct_start) * (lr_max - lr_low)
    t = (pct - pct_start) / (1 - pct_start)
    end = lr_low / 1e4
    return end + (lr_max - end) / 2 * (math.cos(math.pi * t) + 1)


def y_variation(y):
    """load.py:28-30: sum over (n,h,w) of (y - mean_n y)^2, per channel."""
    y = np.asarray(y)
    return ((y - y.mean(0, keepdims=True)) ** 2).sum(axis=(0, 2, 3))


def test_metrics(outputs, targets, y_var):
    """train_codec_mixed_residual.py:180-197 -> (nrmse per channel, r2 per channel)."""
    err2 = ((outputs - targets) ** 2).sum(axis=(-1, -2))           # (n, C)
    rel = np.sqrt(err2 / (targets ** 2).sum(axis=(-1, -2)))
    return rel.mean(0), 1 - err2.sum(0) / y_var


class CpuTrainer:
    """Functional restatement of the reference train loop body on PyTorch-CPU fp32."""

    def __init__(self, sd, blocks, imsize=64, lr=1e-3, weight_decay=0.0, weight_bound=10.0,
                 lr_div=2.0, lr_pct=0.3, upsample='nearest'):
        self.sd = sd
        self.blocks, self.imsize, self.upsample = list(blocks), imsize, upsample
        self.weight_bound = weight_bound
        self.lr_max, self.lr_div, self.lr_pct = lr, lr_div, lr_pct
        self.keys = codec.param_keys(sd)
        for k in self.keys:
            sd[k].requires_grad_(True)
        self.opt = torch.optim.Adam([sd[k] for k in self.keys], lr=lr, weight_decay=weight_decay)

    def forward_loss(self, K, training=True):
        y = codec.densed_forward(self.sd, K, self.blocks, self.imsize, training, self.upsample)
        loss, lc, lt, ld, ln = darcy.mixed_residual_loss(K, y, self.weight_bound)
        return y, loss, (lc, lt, ld, ln)

    def step(self, K, pct=None):
        """one minibatch: returns (loss, lr, output)."""
        self.opt.zero_grad(set_to_none=True)
        y, loss, _ = self.forward_loss(K, True)
        loss.backward()
        lr = self.lr_max if pct is None else one_cycle_lr(pct, self.lr_max, self.lr_div, self.lr_pct)
        for g in self.opt.param_groups:
            g['lr'] = lr
        self.opt.step()
        return float(loss.detach()), lr, y.detach()

    def step_mse(self, K, target, pct=None):
        """train_codec_max_likelihood.py:197-211: the same loop body with F.mse_loss(output, target)."""
        self.opt.zero_grad(set_to_none=True)
        y = codec.densed_forward(self.sd, K, self.blocks, self.imsize, True, self.upsample)
        loss = torch.nn.functional.mse_loss(y, target)
        loss.backward()
        lr = self.lr_max if pct is None else one_cycle_lr(pct, self.lr_max, self.lr_div, self.lr_pct)
        for g in self.opt.param_groups:
            g['lr'] = lr
        self.opt.step()
        return float(loss.detach()), lr

    def grads(self):
        return {k: self.sd[k].grad for k in self.keys}
