"""Layer-level adversarial test of the bf16 x3 split kernel (csrc/conv_mfma_b3.hip: the 196 -> 98 3x3 layer on
v_mfma_f32_16x16x32_bf16, both operands split into hi/mid/lo bf16 terms, six of the nine cross products kept) against
an fp64 convolution -- VERDICT r1 weak #4.  The claim under test: the kernel is fp32-EQUIVALENT, i.e. its error against
fp64 is of the size of the exact-f32 pipe's (v_mfma_f32_16x16x4_f32, option PDES_MFMA_B3=0) on the same data, also
where the dropped cross terms could matter:
  * wide dynamic range across the 196 * 9 = 1764-long contraction (1e-6 .. 1e3 in the activations, 1e-2 .. 1e2 in
    the weights),
  * post-ReLU sparsity (90 % exact zeros),
  * mantissas sitting on bf16 rounding boundaries (low 16 bits 0x7fff / 0x8000 / 0x8001: the hi/mid/lo split carries),
  * cancellation (alternating-sign weights: the sum is ~1e-3 of sum |w||x|; the error is measured against the latter).
Error measures: rel-L2 of the output, and the componentwise-scaled maximum max_ij |err_ij| / (|w| * |x|)_ij in units of
2^-24 (the fp32 rounding unit)."""
import contextlib
import ctypes
import io

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

U = 2.0 ** -24


@pytest.fixture(scope='module')
def rig():
    """the default DenseED's engine at B = 2 and the descriptor of its bf16-split layer, with an identity BatchNorm in
    front (gamma 1, beta 0, statistics {0, n (1 - eps)}), so that the kernel's operand is exactly the buffer content"""
    from pde_surrogate_amd.models.codec import DenseED
    dev = torch.device('cuda:0')
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        net = DenseED(1, 3, 64, [6, 8, 6]).to(dev).train()
    x = torch.exp(0.5 * torch.randn(2, 1, 64, 64, device=dev))
    with torch.no_grad():
        net(x)
    eng = net._engine(x)
    i = [k for k, s in enumerate(net._specs) if s.conv == 'LastTransUp.conv1'][0]
    s, d = net._specs[i], eng.descs[i]
    assert (s.cin, s.cout, s.k) == (196, 98, 3) and d.wb_fwd and d.wb_bwd          # the bf16-split layer
    one, zero, _ = net._identity_bn(dev, s.cin)
    d.gamma, d.beta = one.data_ptr(), zero.data_ptr()
    n = 2 * 32 * 32
    eng.arena.zero_()
    off = eng.stat_off[s.src]
    stats = torch.zeros(2 * s.cin, dtype=torch.float64, device=dev)
    stats[1::2] = n * (1.0 - 1e-5)
    eng.arena[off:off + 2 * s.cin] = stats
    d.out_stats = None
    return dict(net=net, eng=eng, i=i, s=s, d=d, dev=dev, conv=net.features.LastTransUp.conv1)


def _inputs(kind, rng, shape_x, shape_w):
    if kind == 'wide':
        x = np.exp(rng.uniform(-14, 7, shape_x))
        w = rng.standard_normal(shape_w) * np.exp(rng.uniform(-4.6, 4.6, shape_w))
    elif kind == 'sparse':
        x = np.exp(rng.uniform(-3, 3, shape_x)) * (rng.uniform(size=shape_x) < 0.1)
        w = rng.standard_normal(shape_w) * 0.05
    elif kind == 'boundary':
        def on_boundary(shape, signed):
            m = rng.integers(0, 1 << 7, shape).astype(np.uint32) << 16        # bf16 mantissa bits
            low = rng.choice(np.array([0x7FFF, 0x8000, 0x8001, 0x00FF, 0xFF80], np.uint32), shape)
            e = rng.integers(120, 134, shape).astype(np.uint32) << 23
            bits = e | m | low
            if signed:
                bits = bits | (rng.integers(0, 2, shape).astype(np.uint32) << 31)
            return bits.view(np.float32).astype(np.float64)
        x, w = on_boundary(shape_x, False), on_boundary(shape_w, True) * 0.02
    else:                                   # cancellation
        x = 1.0 + 1e-3 * rng.standard_normal(shape_x)
        w = np.where(rng.integers(0, 2, shape_w) > 0, 1.0, -1.0) * (1.0 + 1e-3 * rng.standard_normal(shape_w))
    return x.astype(np.float32), w.astype(np.float32)


def _errors(got, ref, scale):
    e = got.astype(np.float64) - ref
    return float(np.linalg.norm(e) / np.linalg.norm(ref)), float(np.max(np.abs(e) / scale) / U)


@pytest.mark.parametrize('kind', ['wide', 'sparse', 'boundary', 'cancel'])
def test_bf16x3_forward_and_data_gradient_are_fp32_equivalent(rig, option, kind):
    from pde_surrogate_amd import _lib
    import torch.nn.functional as F
    eng, s, d, dev, conv = rig['eng'], rig['s'], rig['d'], rig['dev'], rig['conv']
    L, st = _lib.lib(), _lib.stream_ptr()
    rng = np.random.default_rng({'wide': 1, 'sparse': 2, 'boundary': 3, 'cancel': 4}[kind])
    x, w = _inputs(kind, rng, (2, 196, 32, 32), (98, 196, 3, 3))
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(w))
    eng.X[s.src].copy_(torch.from_numpy(x))
    xd, wd = torch.from_numpy(x).double(), torch.from_numpy(w).double()
    ref = F.conv2d(xd, wd, padding=1).numpy()
    scale = F.conv2d(xd.abs(), wd.abs(), padding=1).numpy() + 1e-300
    # data gradient: T_in = conv^T(g) (identity BatchNorm, operand > 0 -> mask 1; zeros of `sparse` mask their own T)
    g = (rng.standard_normal((2, 98, 32, 32)) * np.exp(rng.uniform(-3, 3, (2, 98, 32, 32)))).astype(np.float32)
    gd = torch.from_numpy(g).double()
    ref_t = F.conv_transpose2d(gd, wd, padding=1).numpy()
    scale_t = F.conv_transpose2d(gd.abs(), wd.abs(), padding=1).numpy() + 1e-300
    mask = (x > 0)
    res = {}
    for tag, b3 in (('b3', '1'), ('f32', '0')):
        option('PDES_MFMA_B3', b3)
        rig['net']._pack_weights()
        _lib.check(L.pdes_conv_forward(eng.ctx, ctypes.byref(d), 1, st), 'forward')
        out = eng.X[s.dst][:, s.dst_coff:s.dst_coff + 98].cpu().numpy()
        eng.T[s.dst][:, s.dst_coff:s.dst_coff + 98].copy_(torch.from_numpy(g))
        saved = (d.fin_tstats, d.t_accumulate)
        d.t_accumulate = 0
        _lib.check(L.pdes_conv_backward_data(eng.ctx, ctypes.byref(d), 1, st), 'backward_data')
        d.t_accumulate = saved[1]
        t = eng.T[s.src][:, :196].cpu().numpy()
        res[tag] = (_errors(out, ref, scale), _errors(t * mask, ref_t * mask, scale_t))
    (f_b3, t_b3), (f_32, t_32) = res['b3'], res['f32']
    print(f'{kind:9s} forward rel-L2 b3 {f_b3[0]:.2e} f32 {f_32[0]:.2e} | max scaled err (ulp of sum|w||x|) b3 {f_b3[1]:.2f} '
          f'f32 {f_32[1]:.2f} || dgrad rel-L2 b3 {t_b3[0]:.2e} f32 {t_32[0]:.2e} | b3 {t_b3[1]:.2f} f32 {t_32[1]:.2f}')
    for (b3e, f32e) in ((f_b3, f_32), (t_b3, t_32)):
        assert b3e[0] <= 2.0 * f32e[0] + 1e-7                # normwise: within 2x of the exact-f32 pipe
        assert b3e[1] <= max(2.0 * f32e[1], 4.0)             # componentwise: a few fp32 rounding units of sum |w||x|
