"""Layer-level adversarial test of EVERY kernel of the bf16 x3 family (csrc/conv_mfma_b3.hip, conv_mfma_b3_up.hip,
conv_mfma_wgrad_b3.hip: the wide 3x3 layers on v_mfma_f32_16x16x32_bf16, both operands split into hi/mid/lo bf16 terms,
six of the nine cross products kept; forward, data gradient and weight gradient, plain and sub-pixel forms, with and
without the f32 K-tail path) against fp64 convolutions -- VERDICT r1 weak #4, r2 weak #2.  The claim under test: each
kernel is fp32-EQUIVALENT, i.e. its error against fp64 is of the size of the exact-f32 pipe's
(v_mfma_f32_16x16x4_f32: the kernel's bit of PDES_MFMA_B3 cleared) on the same data, also where the dropped cross terms could matter:
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


LAYERS = {           # name -> (convolution, Cin, Cout, input map size, nearest-x2 in front of the 3x3)
    'wide': ('LastTransUp.conv1', 196, 98, 32, False),
    'up32': ('LastTransUp.conv2', 98, 49, 32, True),
    'up16': ('TransUp1.conv2', 100, 100, 16, True),
}
# every kernel of the bf16 x3 family: (layer, pass, its bit of the option PDES_MFMA_B3 -- cleared, the exact-f32 pipe runs)
KERNELS = {
    'conv_mfma_b3_kernel<FWD> 196->98': ('wide', 'fwd', 1),
    'conv_mfma_b3_kernel<BWD> 196->98': ('wide', 'dgrad', 1),
    'conv_wgrad_b3_kernel 196->98': ('wide', 'wgrad', 2),
    'conv_b3_up_fwd_kernel 98->49': ('up32', 'fwd', 4),
    'conv_mfma_b3_kernel<UPBWD> 98->49': ('up32', 'dgrad', 8),
    'conv_mfma_b3_kernel<UPBWD> 100->100': ('up16', 'dgrad', 8),
    'conv_wgrad_b3_up_kernel<32> 98->49': ('up32', 'wgrad', 16),
    'conv_wgrad_b3_up_kernel<16> 100->100': ('up16', 'wgrad', 16),
}
B = 2


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


def _rig(layer):
    """a fresh default DenseED engine at B = 2 (the split-K plan of a weight gradient depends on the options in force
    when it is made) and the descriptor of `layer`, with an identity BatchNorm in front (gamma 1, beta 0, statistics
    {0, n (1 - eps)}), so that the kernels' operand is exactly the buffer content"""
    from pde_surrogate_amd.models.codec import DenseED
    dev = torch.device('cuda:0')
    conv_name, cin, cout, hw, up = LAYERS[layer]
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        net = DenseED(1, 3, 64, [6, 8, 6]).to(dev).train()
    x = torch.exp(0.5 * torch.randn(B, 1, 64, 64, device=dev))
    with torch.no_grad():
        net(x)
    eng = net._engine(x)
    eng._plan_wgrad_scratch()
    i = [k for k, s in enumerate(net._specs) if s.conv == conv_name][0]
    s, d = net._specs[i], eng.descs[i]
    assert (s.cin, s.cout, s.k, bool(s.up)) == (cin, cout, 3, up) and eng.buf_hw[s.src] == (hw, hw)
    one, zero, _ = net._identity_bn(dev, s.cin)
    d.gamma, d.beta = one.data_ptr(), zero.data_ptr()
    eng.arena.zero_()
    off = eng.stat_off[s.src]
    stats = torch.zeros(2 * s.cin, dtype=torch.float64, device=dev)
    stats[1::2] = B * hw * hw * (1.0 - 1e-5)
    eng.arena[off:off + 2 * s.cin] = stats
    d.out_stats = None
    conv = net.features
    for part in conv_name.split('.'):
        conv = conv._modules[part]
    return dict(net=net, eng=eng, i=i, s=s, d=d, dev=dev, conv=conv, up=up, hw=hw)


def _reference(x, w, g, up):
    """fp64 forward, data gradient and weight gradient of [nearest x2 +] 3x3 convolution, and the sums of absolute
    products |w| * |x| each result is a rounding of (the componentwise error scale)"""
    import torch.nn.functional as F

    def run(xa, wa, ga):
        xa = xa.clone().requires_grad_(True)
        wa = wa.clone().requires_grad_(True)
        z = F.interpolate(xa, scale_factor=2, mode='nearest') if up else xa
        out = F.conv2d(z, wa, padding=1)
        gx, gw = torch.autograd.grad(out, (xa, wa), ga)
        return out.detach().numpy(), gx.numpy(), gw.numpy()
    xd, wd, gd = (torch.from_numpy(a).double() for a in (x, w, g))
    ref = run(xd, wd, gd)
    scale = run(xd.abs(), wd.abs(), gd.abs())
    return ref, tuple(a + 1e-300 for a in scale)


def _run_kernel(rig, what, x, w, g):
    from pde_surrogate_amd import _lib
    eng, s, d, conv, net = rig['eng'], rig['s'], rig['d'], rig['conv'], rig['net']
    L, st = _lib.lib(), _lib.stream_ptr()
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(w))
    eng.X[s.src][:, :s.cin].copy_(torch.from_numpy(x))
    net._pack_weights()
    ref = ctypes.byref(d)
    if what == 'fwd':
        _lib.check(L.pdes_conv_forward(eng.ctx, ref, 1, st), 'forward')
        return eng.X[s.dst][:, s.dst_coff:s.dst_coff + s.cout].cpu().numpy()
    eng.T[s.dst][:, s.dst_coff:s.dst_coff + s.cout].copy_(torch.from_numpy(g))
    if what == 'dgrad':
        saved = d.t_accumulate
        d.t_accumulate = 0
        _lib.check(L.pdes_conv_backward_data(eng.ctx, ref, 1, st), 'backward_data')
        d.t_accumulate = saved
        return eng.T[s.src][:, :s.cin].cpu().numpy()
    gw = net._grad_view[s.conv + '.weight']
    gw.zero_()
    _lib.check(L.pdes_conv_backward_weight(eng.ctx, ref, 1, st), 'backward_weight')
    row = eng._reduce_index[rig['i']]
    assert row >= 0 and d.ws_defer                      # split-K partials, reduced in a fixed order
    _lib.check(L.pdes_wgrad_reduce_all(eng._reduce_table.data_ptr() + 24 * row, 1, s.cout * s.cin * 9, st), 'reduce')
    return gw.cpu().numpy().copy()


# the f32 K-tail path exists in the forward / data-gradient kernels of the 196- and 98-channel layers only (196 = 6 x 32 + 4,
# 98 = 3 x 32 + 2; the weight gradients contract over pixels, and the 100 -> 100 data gradient has K = 100 output channels...)
CASES = [(k, '1') for k in KERNELS] + [(k, '0') for k, (layer, what, _) in KERNELS.items() if what != 'wgrad' and layer != 'up16']


@pytest.mark.parametrize('kind', ['wide', 'sparse', 'boundary', 'cancel'])
@pytest.mark.parametrize('kernel,tail', CASES)
def test_bf16x3_kernels_are_fp32_equivalent(option, kernel, kind, tail):
    """all six kernels of the bf16 x3 family (eight layer / pass combinations) x four adversarial input families x the
    f32 K-tail path on and off: error against fp64 within 2x of the exact-f32 pipe's on the same data, and at most 4
    fp32 rounding units of sum |w| |x| componentwise"""
    layer, what, bit = KERNELS[kernel]
    conv_name, cin, cout, hw, up = LAYERS[layer]
    ho = 2 * hw if up else hw
    rng = np.random.default_rng({'wide': 1, 'sparse': 2, 'boundary': 3, 'cancel': 4}[kind] + 10 * list(KERNELS).index(kernel))
    x, w = _inputs(kind, rng, (B, cin, hw, hw), (cout, cin, 3, 3))
    g = (rng.standard_normal((B, cout, ho, ho)) * np.exp(rng.uniform(-3, 3, (B, cout, ho, ho)))).astype(np.float32)
    refs, scales = _reference(x, w, g, up)
    k = {'fwd': 0, 'dgrad': 1, 'wgrad': 2}[what]
    # data gradient: T_in = mask * conv^T(g) under the identity BatchNorm (operand > 0 -> mask 1; the exact zeros of
    # `sparse` mask their own T)
    mask = (x > 0) if what == 'dgrad' else 1.0
    res, raw = {}, {}
    option('PDES_B3_TAIL', tail)
    for tag, v in (('b3', '31'), ('f32', str(31 - bit))):
        option('PDES_MFMA_B3', v)
        rig = _rig(layer)
        raw[tag] = _run_kernel(rig, what, x, w, g)
        res[tag] = _errors(raw[tag] * mask, refs[k] * mask, scales[k])
        del rig
    (b3, f32) = res['b3'], res['f32']
    print(f'{kernel:40s} {kind:9s} tail {tail}: rel-L2 b3 {b3[0]:.2e} f32 {f32[0]:.2e} | max err in ulp of sum|w||x| '
          f'b3 {b3[1]:.2f} f32 {f32[1]:.2f}')
    assert not np.array_equal(raw['b3'], raw['f32']), 'the option did not change the kernel: is the bf16 path taken?'
    assert b3[0] <= 2.0 * f32[0] + 1e-7                # normwise: within 2x of the exact-f32 pipe
    # componentwise: a few fp32 rounding units of sum |w||x|.  The MAXIMUM over 10^5 .. 10^6 entries of a rounding random
    # walk fluctuates between two correct summation orders (measured: 5.48 vs 2.70 units on the 2,048-pixel contraction of
    # the widest weight gradient with 90 % zeros, rel-L2 within 2x): 3x of the f32 pipe's maximum, or 4 units
    assert b3[1] <= max(3.0 * f32[1], 4.0)
