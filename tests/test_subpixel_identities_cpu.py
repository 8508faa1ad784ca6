"""The sub-pixel (parity) decomposition of nearest-x2 upsampling + 3x3 convolution that the HIP kernels of the two
`_Transition` up-layers use (reference models/codec.py:24-30, :130-150: UpsamplingNearest2d + Conv2d(3, padding=1)):
    out[2y + py][2x + px] = sum_{a,b} Weff_p[a][b] . z[y + a + py - 1][x + b + px - 1]
with Weff_p[a][b] = the 3x3 taps that land on position (a, b) of parity p's 2x2 kernel (csrc/pack_kernels.h: weff),
and its two adjoints (csrc/conv_mfma_wgrad_b3.hip: dWeff -> dW fold; csrc/conv_mfma_b3.hip B3_UPBWD) -- checked in
fp64 against torch autograd of the plain formulation.  Pins the ALGEBRA on the CPU; the kernels themselves are compared
with the reference goldens in the `-m gpu` tests."""
import numpy as np
import torch
import torch.nn.functional as F


def _rows(parity, pos):          # 3x3 rows (or columns) that land on position `pos` of parity `parity`: pack_kernels.h weff_mask
    return {(0, 0): [0], (0, 1): [1, 2], (1, 0): [0, 1], (1, 1): [2]}[(parity, pos)]


def _weff(W, py, px, a, b):
    return sum(W[:, :, ky, kx] for ky in _rows(py, a) for kx in _rows(px, b))


def _shift(z, dy, dx):           # z[..., y + dy, x + dx] with zeros outside the map
    B, C, H, Wd = z.shape
    zp = F.pad(z, (1, 1, 1, 1))
    return zp[:, :, 1 + dy:1 + dy + H, 1 + dx:1 + dx + Wd]


def _setup(seed=0, B=2, Cin=5, Cout=4, H=6, Wd=8):
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(B, Cin, H, Wd, generator=g, dtype=torch.float64, requires_grad=True)
    W = torch.randn(Cout, Cin, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    out = F.conv2d(F.interpolate(z, scale_factor=2, mode='nearest'), W, padding=1)
    gout = torch.randn(out.shape, generator=g, dtype=torch.float64)
    gz, gW = torch.autograd.grad(out, (z, W), gout)
    return z.detach(), W.detach(), out.detach(), gout, gz, gW


def test_forward_is_four_parity_2x2_convolutions():
    z, W, out, *_ = _setup()
    for py in range(2):
        for px in range(2):
            acc = 0
            for a in range(2):
                for b in range(2):
                    acc = acc + torch.einsum('oc,bchw->bohw', _weff(W, py, px, a, b), _shift(z, a + py - 1, b + px - 1))
            assert torch.allclose(acc, out[:, :, py::2, px::2], rtol=1e-12, atol=1e-12)


def test_weight_gradient_folds_sixteen_products_into_nine_taps():
    """conv_wgrad_b3_up_kernel: dWeff[p][a][b] = <G_p, z shifted by (a + py - 1, b + px - 1)>, then
    dW[ky][kx] = sum_{py,px} dWeff[(py,px)][a(ky,py)][b(kx,px)], a(k, 0) = (0 if k == 0 else 1), a(k, 1) = (1 if k == 2 else 0)"""
    z, W, out, gout, gz, gW = _setup(seed=1)
    dWeff = {}
    for py in range(2):
        for px in range(2):
            Gp = gout[:, :, py::2, px::2]
            for a in range(2):
                for b in range(2):
                    dWeff[(py, px, a, b)] = torch.einsum('bohw,bchw->oc', Gp, _shift(z, a + py - 1, b + px - 1))
    pos = lambda k, p: (0 if k == 0 else 1) if p == 0 else (1 if k == 2 else 0)
    dW = torch.zeros_like(W)
    for ky in range(3):
        for kx in range(3):
            for py in range(2):
                for px in range(2):
                    dW[:, :, ky, kx] += dWeff[(py, px, pos(ky, py), pos(kx, px))]
    assert torch.allclose(dW, gW, rtol=1e-12, atol=1e-12)
    # and the fold is consistent with the forward map: tap (ky, kx) lands on position pos(ky, py) of parity py
    for py in range(2):
        for a in range(2):
            assert sorted(k for k in range(3) if pos(k, py) == a) == _rows(py, a)


def test_data_gradient_gathers_four_taps_per_parity():
    """B3_UPBWD: dz[y][x] = sum_p sum_{a,b} Weff_p[a][b]^T . G_p[y - a - py + 1][x - b - px + 1]"""
    z, W, out, gout, gz, gW = _setup(seed=2)
    dz = torch.zeros_like(z)
    for py in range(2):
        for px in range(2):
            Gp = gout[:, :, py::2, px::2]
            for a in range(2):
                for b in range(2):
                    dz = dz + torch.einsum('oc,bohw->bchw', _weff(W, py, px, a, b), _shift(Gp, -(a + py - 1), -(b + px - 1)))
    assert torch.allclose(dz, gz, rtol=1e-12, atol=1e-12)


def test_k_tail_split_is_exact():
    """PDES_B3_TAIL: the contraction over K channels = (whole 32-channel chunks) + (<= 4 tail channels), trivially -- and the
    counts the kernels rely on for the default network"""
    for k, chunks, tail in ((196, 6, 4), (98, 3, 2), (100, 3, 4)):
        assert k == 32 * chunks + tail and tail <= 4
    x, w = np.random.default_rng(0).standard_normal((2, 98)), np.random.default_rng(1).standard_normal(98)
    assert np.allclose(x @ w, x[:, :96] @ w[:96] + x[:, 96:] @ w[96:])
