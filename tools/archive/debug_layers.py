#!/usr/bin/env python
"""Layer-by-layer comparison of the HIP DenseED against the CPU oracle (diagnostic; GPU box).
Prints rel-L2 of every convolution's raw output and of every parameter gradient."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.nn.functional as F

from oracle import codec as oc, darcy as od
from pde_surrogate_amd.models.codec import DenseED
from pde_surrogate_amd.models.darcy import darcy_mixed_residual_loss


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def poison(dev, gib=3):
    """fill freshly allocated device memory with NaN and release it to the caching allocator, so any
    read of an uninitialised buffer shows up as NaN instead of (usually) zero"""
    ts = [torch.full((256 << 20,), float('nan'), device=dev) for _ in range(gib)]
    del ts


def main(blocks=(1, 1, 1), growth=4, init=8, imsize=16, B=4, decoder=False):
    dev = torch.device('cuda:0')
    poison(dev)
    torch.manual_seed(0)
    if decoder:
        from pde_surrogate_amd.models.codec import Decoder
        net = Decoder(1, 3, list(blocks), growth, init)
    else:
        net = DenseED(1, 3, imsize, list(blocks), growth, init)
    with torch.no_grad():
        for k, v in net.state_dict().items():
            if 'norm' in k and k.endswith('.weight'):
                v.copy_(1 + 0.2 * torch.randn_like(v))
            if 'norm' in k and k.endswith('.bias'):
                v.copy_(0.1 * torch.randn_like(v))
    sd = {k: v.clone().double() for k, v in net.state_dict().items()}
    x = torch.exp(0.5 * torch.randn(B, 1, imsize, imsize))
    K = torch.exp(0.5 * torch.randn(B, 1, 64, 64)) if decoder else x
    # oracle with hooks: record raw conv outputs in order
    recs = []
    orig = F.conv2d

    def rec_conv(*a, **k):
        o = orig(*a, **k)
        recs.append(o.detach())
        return o
    F.conv2d = rec_conv
    keys = oc.param_keys(sd)
    for k in keys:
        sd[k].requires_grad_(True)
    if decoder:
        yo = oc.decoder_forward(sd, x.double(), list(blocks), True)
    else:
        yo = oc.densed_forward(sd, x.double(), list(blocks), imsize, True)
    F.conv2d = orig
    loss_o = od.mixed_residual_loss(K.double(), yo, 10.0)[0]
    loss_o.backward()

    net = net.to(dev).train()
    xd = x.to(dev)
    y = net(xd)
    eng = net._engine(xd)
    for i, s in enumerate(net._specs):
        buf = eng.X[s.dst][:, s.dst_coff:s.dst_coff + s.cout].cpu().numpy()
        print(f'fwd {i:2d} {s.conv:36s} rel={rel(buf, recs[i].numpy()):.3e}')
    loss, *_ = darcy_mixed_residual_loss(K.to(dev), y, 10.0)
    print('loss', float(loss.detach()), float(loss_o), abs(float(loss.detach()) - float(loss_o)) / float(loss_o))
    loss.backward()
    for name, p in reversed(list(net.named_parameters())):
        print(f'grad {name:52s} rel={rel(p.grad.cpu().numpy(), sd[name].grad.numpy()):.3e}')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'default':
        main((6, 8, 6), 16, 48, 64, int(sys.argv[2]) if len(sys.argv) > 2 else 4)
    elif len(sys.argv) > 1 and sys.argv[1] == 'decoder':
        main((8, 6), 16, 48, 16, 1, decoder=True)
    else:
        main()
