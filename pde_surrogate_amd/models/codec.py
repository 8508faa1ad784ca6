"""DenseED / Decoder on MI355X -- drop-in for the reference's models/codec.py (:210-370).

Same constructor arguments, ``forward`` contract, ``model_size`` / ``forward_test`` and the same
``state_dict`` keys (checkpoints interchange with the reference), but ``forward``/``backward`` run
hand-written HIP kernels through the C ABI (include/pdes_hip.h) instead of aten ops:

  * every parameter lives in ONE flat fp32 buffer (one RCCL all-reduce, one Adam launch);
  * a dense block owns one (B, Ctot, H, W) buffer, layers write their 16 channels in place
    (no ``torch.cat``), BatchNorm+ReLU(+nearest x2) are applied on the fly when a convolution
    loads its operand, batch statistics are accumulated by the producing convolution;
  * backward keeps one accumulator T per activation buffer (see include/pdes_hip.h) so the 27
    BatchNorm backward passes cost no extra pass over the activations.

There is no CPU implementation: inputs must be CUDA (ROCm) tensors.
"""
import ctypes
import os

import torch
import torch.nn as nn

from .. import _lib
from .. import optim as _optim
from ..utils.misc import module_size  # noqa: F401  (the reference's codec.py:14-21)

_optim.install_auto_fused_hook()      # a plain torch.optim.Adam over a HIP network's parameters takes its fused implementation

_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float


class ConvDesc(ctypes.Structure):
    """mirror of `pdes_conv_desc` (include/pdes_hip.h) -- field order and types must match"""
    _fields_ = [
        ('B', _I), ('Cin', _I), ('Cout', _I), ('Hin', _I), ('Win', _I), ('Hout', _I), ('Wout', _I),
        ('ksize', _I), ('stride', _I), ('pad', _I), ('upsample', _I),
        ('x', _P), ('x_ctot', _I), ('has_bn', _I), ('eval_mode', _I), ('eps', _F),
        ('gamma', _P), ('beta', _P), ('x_stats', _P), ('run_mean', _P), ('run_var', _P),
        ('w', _P), ('w_fwd', _P), ('w_bwd', _P), ('cout_pad', _I), ('cin_pad', _I),
        ('wm_fwd', _P), ('wm_bwd', _P), ('wu_fwd', _P), ('wu_bwd', _P), ('wb_fwd', _P), ('wb_bwd', _P),
        ('out', _P), ('out_ctot', _I), ('out_coff', _I), ('out_stats', _P), ('fin_xstats', _P), ('fin_tstats', _P),
        ('g', _P), ('g_ctot', _I), ('g_coff', _I), ('g_fused', _I),
        ('t_in', _P), ('t_accumulate', _I), ('final_c0', _I), ('final_c1', _I),
        ('t_stats', _P), ('bn_grad', _P), ('dw', _P), ('ws', _P), ('ws_bytes', ctypes.c_longlong),
        ('ws_defer', _I), ('nrep', _I), ('rep_stride', ctypes.c_longlong), ('wbu_fwd', _P),
        ('g_add', _P), ('x2', _P), ('x2_ctot', _I), ('t2', _P), ('p0', _P), ('p1', _P), ('acc', _P), ('flags', _I),
        ('wbu_bwd', _P),
        ('coef', _P), ('fin_coef', _P),
        ('h', _P), ('h_ctot', _I), ('th', _P), ('acc2', _P), ('cst', _P),
    ]


class PackItem(ctypes.Structure):
    _fields_ = [('w', _P), ('w_fwd', _P), ('w_bwd', _P), ('Cout', _I), ('Cin', _I), ('kk', _I),
                ('cout_pad', _I), ('cin_pad', _I)]


class MfmaPackItem(ctypes.Structure):
    _fields_ = [('w', _P), ('wm_fwd', _P), ('wm_bwd', _P), ('Cout', _I), ('Cin', _I), ('kk', _I)]


class UpPackItem(ctypes.Structure):
    _fields_ = [('w', _P), ('wu_fwd', _P), ('wu_bwd', _P), ('Cout', _I), ('Cin', _I)]


class B3PackItem(ctypes.Structure):
    _fields_ = [('w', _P), ('wb_fwd', _P), ('wb_bwd', _P), ('Cout', _I), ('Cin', _I)]


class B3UpPackItem(ctypes.Structure):
    _fields_ = [('w', _P), ('wbu_fwd', _P), ('wbu_bwd', _P), ('Cout', _I), ('Cin', _I)]


class ReduceItem(ctypes.Structure):
    _fields_ = [('part', _P), ('dw', _P), ('n', _I), ('nsplit', _I)]


class BnItem(ctypes.Structure):
    _fields_ = [('x_stats', _P), ('bn_grad', _P), ('run_mean', _P), ('run_var', _P), ('dgamma', _P),
                ('dbeta', _P), ('num_batches_tracked', _P), ('C', _I), ('count', _I)]


def _pad16(n):
    return (n + 15) // 16 * 16


# ------------------------------------------------------------------------------------------------
# network description: a list of convolution specs in forward order
UP_NEAREST, UP_BILINEAR_OP, OP_CHANNEL_MASK = 1, 2, 3     # pdes_conv_desc.upsample (include/pdes_hip.h)


class _ConvSpec:
    """one descriptor of the chain: a convolution with the BatchNorm+ReLU that precedes it -- or, with conv = None and
    up = UP_BILINEAR_OP, the bilinear x2 resampling of relu(bn(x)) into its own buffer; the convolution after such an op
    has no BatchNorm module of its own and reads the resampled buffer through an identity BatchNorm (`fake_bn`)"""
    __slots__ = ('conv', 'norm', 'cin', 'cout', 'k', 'stride', 'pad', 'up', 'src', 'dst', 'dst_coff', 'scale', 'fake_bn')

    def __init__(self, conv, norm, cin, cout, k, stride, pad, up, src, dst, dst_coff, scale, fake_bn=False):
        self.conv, self.norm = conv, norm          # module paths under `features`
        self.cin, self.cout, self.k, self.stride, self.pad, self.up = cin, cout, k, stride, pad, up
        self.src, self.dst, self.dst_coff = src, dst, dst_coff   # activation buffer ids
        self.scale = scale                          # (num, den): input-buffer size = imsize*num/den
        self.fake_bn = fake_bn

    @property
    def bn(self):
        """the kernels apply a BatchNorm+ReLU on operand load (a real module's, or the identity one)"""
        return self.norm is not None or self.fake_bn


def _plan_dropout(specs, bufs, drop):
    """nn.Dropout2d after the convolution just planned (reference codec.py:70-71, :111-120, :134-150, :172-173): a
    channel-mask op in place on the channels that convolution wrote"""
    if drop:
        s = specs[-1]
        specs.append(_ConvSpec(None, None, s.cout, s.cout, 0, 1, 0, OP_CHANNEL_MASK, s.dst, s.dst, s.dst_coff, bufs[s.dst][1]))


def _plan_up_conv(specs, bufs, t, conv, norm, cin, cout, mid, nxt, res, nres, upsample):
    """BN-ReLU -> x2 upsampling -> conv3x3 (reference codec.py:137-146 / :174-181)"""
    if upsample == 'nearest':
        specs.append(_ConvSpec(conv, norm, cin, cout, 3, 1, 1, UP_NEAREST, mid, nxt, 0, res))
        return
    up = mid + 'u'                                   # the resampled activation: (B, cin, 2H, 2W)
    bufs[up] = [cin, nres]
    specs.append(_ConvSpec(None, norm, cin, cin, 0, 1, 0, UP_BILINEAR_OP, mid, up, 0, res))
    specs.append(_ConvSpec(conv, None, cin, cout, 3, 1, 1, 0, up, nxt, 0, nres, fake_bn=True))


def _plan_block(specs, bufs, name, c, n_layers, growth, src, scale, drop=False, bottleneck=False, bn_size=8):
    """dense block: buffer `src` has room for all channels; layer j reads [0,c) writes [c,c+g).
    bottleneck (reference codec.py:55-62): layers with more than bn_size * growth inputs first reduce to
    bn_size * growth channels with a 1x1 convolution (norm1/conv1) into a buffer of their own, then norm2/conv2 3x3"""
    for j in range(1, n_layers + 1):
        p = f'{name}.denselayer{j}'
        if bottleneck and c > bn_size * growth:
            mid = f'{src}_{name}_bk{j}'
            bufs[mid] = [bn_size * growth, scale]
            specs.append(_ConvSpec(p + '.conv1', p + '.norm1', c, bn_size * growth, 1, 1, 0, 0, src, mid, 0, scale))
            specs.append(_ConvSpec(p + '.conv2', p + '.norm2', bn_size * growth, growth, 3, 1, 1, 0, mid, src, c, scale))
        else:
            specs.append(_ConvSpec(p + '.conv1', p + '.norm1', c, growth, 3, 1, 1, 0, src, src, c, scale))
        _plan_dropout(specs, bufs, drop)
        c += growth
    return c


def _plan_densed(blocks, growth, init_features, in_channels, out_channels, imsize, upsample='nearest', drop=False,
                 bottleneck=False, bn_size=8):
    """stage order and channel bookkeeping of DenseED (reference codec.py:229-293)"""
    if len(blocks) > 1 and len(blocks) % 2 == 0:
        raise ValueError('length of blocks must be an odd number, but got {}'.format(len(blocks)))
    enc, dec = blocks[:len(blocks) // 2], blocks[len(blocks) // 2:]
    specs, bufs = [], {}          # bufs: id -> [channels, (num, den)]
    pad = 3 if imsize % 2 == 0 else 2
    res = (1, 2)                  # feature-map size relative to the image, as a fraction
    bufs['in'] = [in_channels, (1, 1)]
    cur, c = 'b0', init_features
    bufs[cur] = [c + (enc[0] if enc else dec[0]) * growth, res]
    specs.append(_ConvSpec('In_conv', None, in_channels, init_features, 7, 2, pad, 0, 'in', cur, 0, (1, 1)))
    nb = 1
    for i, n in enumerate(enc, 1):
        c = _plan_block(specs, bufs, f'EncBlock{i}', c, n, growth, cur, res, drop, bottleneck, bn_size)
        t = f'TransDown{i}'
        mid, nxt = f'b{nb}', f'b{nb + 1}'
        nb += 2
        bufs[mid] = [c // 2, res]
        specs.append(_ConvSpec(t + '.conv1', t + '.norm1', c, c // 2, 1, 1, 0, 0, cur, mid, 0, res))
        _plan_dropout(specs, bufs, drop)
        nres = (res[0], res[1] * 2)
        following = enc[i] if i < len(enc) else dec[0]
        bufs[nxt] = [c // 2 + following * growth, nres]
        specs.append(_ConvSpec(t + '.conv2', t + '.norm2', c // 2, c // 2, 3, 2, 1, 0, mid, nxt, 0, res))
        _plan_dropout(specs, bufs, drop)
        cur, c, res = nxt, c // 2, nres
    for i, n in enumerate(dec, 1):
        c = _plan_block(specs, bufs, f'DecBlock{i}', c, n, growth, cur, res, drop, bottleneck, bn_size)
        if i < len(dec):
            t = f'TransUp{i}'
            mid, nxt = f'b{nb}', f'b{nb + 1}'
            nb += 2
            bufs[mid] = [c // 2, res]
            specs.append(_ConvSpec(t + '.conv1', t + '.norm1', c, c // 2, 1, 1, 0, 0, cur, mid, 0, res))
            _plan_dropout(specs, bufs, drop)
            nres = (res[0] * 2, res[1])
            bufs[nxt] = [c // 2 + dec[i] * growth, nres]
            _plan_up_conv(specs, bufs, t, t + '.conv2', t + '.norm2', c // 2, c // 2, mid, nxt, res, nres, upsample)
            _plan_dropout(specs, bufs, drop)
            cur, c, res = nxt, c // 2, nres
    _plan_last(specs, bufs, cur, c, res, out_channels, nb, upsample, drop)
    return specs, bufs


def _plan_last(specs, bufs, cur, c, res, out_channels, nb, upsample='nearest', drop=False):
    """last decoding (reference codec.py:163-188)"""
    t = 'LastTransUp'
    m1, m2 = f'b{nb}', f'b{nb + 1}'
    bufs[m1] = [c // 2, res]
    specs.append(_ConvSpec(t + '.conv1', t + '.norm1', c, c // 2, 3, 1, 1, 0, cur, m1, 0, res))
    _plan_dropout(specs, bufs, drop)                  # (the reference drops only after conv1 here, codec.py:172-173)
    nres = (res[0] * 2, res[1])
    bufs[m2] = [c // 4, nres]
    _plan_up_conv(specs, bufs, t, t + '.conv2', t + '.norm2', c // 2, c // 4, m1, m2, res, nres, upsample)
    bufs['out'] = [out_channels, nres]
    specs.append(_ConvSpec(t + '.conv3', t + '.norm3', c // 4, out_channels, 5, 1, 2, 0, m2, 'out', 0, nres))


def _plan_decoder(blocks, growth, init_features, dim_latent, out_channels, upsample='nearest', drop=False):
    """Decoder (reference codec.py:326-354); sizes are relative to the latent map (16x16 -> 64x64)"""
    specs, bufs = [], {}
    res = (1, 1)
    bufs['in'] = [dim_latent, res]
    cur, c, nb = 'b0', init_features, 1
    bufs[cur] = [c + blocks[0] * growth, res]
    specs.append(_ConvSpec('conv0', None, dim_latent, init_features, 3, 1, 1, 0, 'in', cur, 0, res))
    for i, n in enumerate(blocks, 1):
        c = _plan_block(specs, bufs, f'DecBlock{i}', c, n, growth, cur, res, drop)
        if i < len(blocks):
            t = f'TransUp{i}'
            mid, nxt = f'b{nb}', f'b{nb + 1}'
            nb += 2
            bufs[mid] = [c // 2, res]
            specs.append(_ConvSpec(t + '.conv1', t + '.norm1', c, c // 2, 1, 1, 0, 0, cur, mid, 0, res))
            _plan_dropout(specs, bufs, drop)
            nres = (res[0] * 2, res[1])
            bufs[nxt] = [c // 2 + blocks[i] * growth, nres]
            _plan_up_conv(specs, bufs, t, t + '.conv2', t + '.norm2', c // 2, c // 2, mid, nxt, res, nres, upsample)
            _plan_dropout(specs, bufs, drop)
            cur, c, res = nxt, c // 2, nres
    _plan_last(specs, bufs, cur, c, res, out_channels, nb, upsample, drop)
    return specs, bufs


def activation(name):
    """reference codec.py:191-203"""
    if name in ['tanh', 'Tanh']:
        return nn.Tanh()
    if name in ['relu', 'ReLU']:
        return nn.ReLU(inplace=False)      # (not in place: the input is the output of a custom autograd node)
    if name in ['lrelu', 'LReLU']:
        return nn.LeakyReLU(inplace=False)
    if name in ['sigmoid', 'Sigmoid']:
        return nn.Sigmoid()
    if name in ['softplus', 'Softplus']:
        return nn.Softplus(beta=4)
    raise ValueError('Unknown activation function')


def _add_path(root, path, module):
    parts = path.split('.')
    m = root
    for p in parts[:-1]:
        if p not in m._modules:
            m.add_module(p, nn.Sequential())
        m = m._modules[p]
    m.add_module(parts[-1], module)


def _build_modules(features, specs):
    """parameter containers in the reference's creation order (same RNG stream, same state_dict keys).
    nn.Conv2d / nn.BatchNorm2d are used for their parameters and default init only -- their forward
    is never called."""
    for s in specs:
        if s.norm is not None:
            _add_path(features, s.norm, nn.BatchNorm2d(s.cin))
            # the reference inserts ReLU (and upsample) modules here; they hold no state
        if s.conv is not None:
            _add_path(features, s.conv, nn.Conv2d(s.cin, s.cout, s.k, s.stride, s.pad, bias=False))


def _get(root, path):
    m = root
    for p in path.split('.'):
        m = m._modules[p]
    return m


# ------------------------------------------------------------------------------------------------
class _EngineBase:
    """what every descriptor-chain engine shares: the split-K scratch plan of its weight gradients and the side streams"""

    def _chain_specs(self):
        return self.net._specs

    def _plan_wgrad_scratch(self):
        """per-layer split-K scratch so that ONE reduce launch finishes every weight gradient"""
        L = _lib.lib()
        self._wgrad_ws, items, mx = [], [], 0
        for i, s in enumerate(self._chain_specs()):
            d = self.descs[i]
            if s.conv is None:                         # an operator descriptor: no weights
                continue
            d.ws_bytes = 1 << 40                       # plan without a scratch limit
            ns, fl = _I(0), ctypes.c_longlong(0)
            rc = L.pdes_conv_wgrad_plan(self.ctx, ctypes.byref(d), ctypes.byref(ns), ctypes.byref(fl))
            if rc != 0:                                # generic kernel: atomics into dw, shared scratch unused
                d.ws, d.ws_bytes, d.ws_defer = self.net._ws.data_ptr(), self.net._ws.numel() * 4, 0
                continue
            buf = torch.empty(fl.value, device=self.dev, dtype=torch.float32)
            self._wgrad_ws.append(buf)
            d.ws, d.ws_bytes, d.ws_defer = buf.data_ptr(), fl.value * 4, 1
            it = ReduceItem()
            it.part, it.dw, it.n, it.nsplit = buf.data_ptr(), d.dw, s.cout * s.cin * s.k * s.k, ns.value
            items.append(it)
            mx = max(mx, it.n)
        self._reduce_n, self._reduce_max = len(items), mx
        # host map descriptor -> row of the reduce table (-1: no deferred scratch), for pdes_backward
        idx, k = [], 0
        for d in self.descs:
            if d.ws_defer:
                idx.append(k); k += 1
            else:
                idx.append(-1)
        self._reduce_index = (_I * len(idx))(*idx)
        if items:
            arr = (ReduceItem * len(items))(*items)
            self._reduce_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.dev)

    def _side_stream(self, which='a'):
        """second HIP stream for the weight gradients (one per network and device), at the LOWEST queue priority: free
        workgroup slots go to the finalize -> data-gradient chain on the main stream first, the weight gradients fill
        what is left"""
        key = self.dev if which == 'a' else (self.dev, which)
        side = self.net._side_streams.get(key)             # (a per-network override, ab_cumask.py of the earlier rounds (git history))
        if side is None:
            side = _DEVICE_SIDE_STREAMS.get(key)
        if side is None:
            try:
                least = torch.cuda.Stream.priority_range()[0]
            except Exception:
                least = 0
            side = _DEVICE_SIDE_STREAMS[key] = torch.cuda.Stream(self.dev, priority=least)
        return side



# The two weight-gradient streams are PER DEVICE, shared by every network / engine / trainer of the process.  Each new
# torch.cuda.Stream is another HIP stream, and the runtime multiplexes HIP streams onto a handful of hardware queues: a
# trainer built late in a process (after other trainers, solvers, torch's own pool streams) used to draw side streams that
# share a hardware queue with each other or with the main stream, and ran 4-16 % slower for it (5.21-5.34 vs 6.16 ms for the
# conditional-Glow step, 1.695 vs 1.768 ms for the DenseED step: tools/post_capture.py, profiles/r04_a_late_trainer_streams.log
# -- this, not the hipGraph capture before it, was the "slow after a capture" effect of round 3).  The first pair of a
# process is the one every benchmark of this repository has measured.
_DEVICE_SIDE_STREAMS = {}


class _Engine(_EngineBase):
    """Activation/accumulator buffers + kernel descriptors for one (batch, size, device)."""

    def __init__(self, net, B, hin, win):
        self.net, self.B = net, B
        dev = net._flat.device
        self.dev = dev
        self.ctx = _lib.context(dev)           # options + fork/join events of this device (include/pdes_hip.h)
        self.arena_clean = False               # True: the statistics arena is already zero (trainer's Adam kernel did it)
        self.busy = False                      # leased to an autograd forward whose backward has not run yet
        self.reserved = False                  # owned by a MixedResidualTrainer: never leased
        specs, bufs = net._specs, net._bufs

        def size_of(scale, base):
            v = base * scale[0]
            if v % scale[1]:
                raise ValueError(f'input size {base} is not divisible by {scale[1]}')
            return v // scale[1]

        self.buf_hw = {k: (size_of(sc, hin), size_of(sc, win)) for k, (c, sc) in bufs.items()}
        self.X, self.T = {}, {}
        n_stat = 0
        self.stat_off = {}
        for k, (c, sc) in bufs.items():
            h, w = self.buf_hw[k]
            # 'in' is an engine-owned copy of the input: backward (weight gradient of the first
            # convolution) reads it after the caller's tensor may have been freed
            self.X[k] = torch.empty((B, c, h, w), device=dev, dtype=torch.float32)
            if k not in ('in', 'out'):
                self.T[k] = torch.empty((B, c, h, w), device=dev, dtype=torch.float32)
            self.stat_off[k] = n_stat
            n_stat += 2 * c
        # fp64 arena: per buffer {sum x, sum x^2}, then per buffer {sum T, sum T xhat}, then per BN {dgamma, dbeta}
        self.n_xstat = n_stat
        bn_off, n_bn = {}, 0
        for s in specs:
            if s.bn:                                   # identity BatchNorms get a (never read) slot too
                bn_off[s.norm or ('identity:' + s.conv)] = n_bn
                n_bn += 2 * s.cin
        # NREP replicas of the whole arena spread same-address fp64 atomics (readers sum them)
        self.nrep, self.rep_stride = _lib.lib().pdes_stat_replicas(), 2 * n_stat + n_bn
        # ... followed by the {mean, invstd} table of every buffer channel (float2 = one double slot per channel): cleared
        # with the arena, filled by the first kernel of a step that sums a channel's replicas (pdes_conv_desc.coef)
        self.arena = torch.zeros(self.nrep * self.rep_stride + n_stat // 2, device=dev, dtype=torch.float64)
        a0 = self.arena.data_ptr()
        cf = lambda k: a0 + 8 * (self.nrep * self.rep_stride + self.stat_off[k] // 2)
        xs = lambda k: a0 + 8 * self.stat_off[k]
        ts = lambda k: a0 + 8 * (n_stat + self.stat_off[k])
        bg = lambda nm: a0 + 8 * (2 * n_stat + bn_off[nm])

        # Dropout2d channel masks: one flat buffer for every mask op (B x C each), and ones for eval mode
        self.mask_off, n_mask = {}, 0
        for i, s in enumerate(specs):
            if s.up == OP_CHANNEL_MASK:
                self.mask_off[i] = n_mask
                n_mask += B * s.cout
        self.drop_masks = torch.ones(max(n_mask, 1), device=dev)
        self.drop_ones = torch.ones(max(n_mask, 1), device=dev)
        n = len(specs)
        self.descs = (ConvDesc * n)()
        last_reader = {}
        for i, s in enumerate(specs):
            last_reader[s.src] = i
        consumed = {}
        pk = net._packed
        for i, s in enumerate(specs):
            d = self.descs[i]
            hi, wi = self.buf_hw[s.src]
            ho, wo = self.buf_hw[s.dst]
            d.B, d.Cin, d.Cout, d.Hin, d.Win, d.Hout, d.Wout = B, s.cin, s.cout, hi, wi, ho, wo
            d.ksize, d.stride, d.pad, d.upsample = s.k, s.stride, s.pad, s.up
            d.x = self.X[s.src].data_ptr()
            d.x_ctot = bufs[s.src][0]
            d.has_bn = 1 if s.bn else 0
            d.eval_mode = 0
            d.eps = 1e-5
            if s.conv is not None:
                conv = _get(net.features, s.conv)
                d.w = conv.weight.data_ptr()
                d.w_fwd, d.w_bwd = pk[s.conv][0].data_ptr(), pk[s.conv][1].data_ptr()
                d.dw = net._grad_view[s.conv + '.weight'].data_ptr()
            d.cout_pad, d.cin_pad = _pad16(s.cout), _pad16(s.cin)
            mf = net._packed_mfma.get(s.conv)
            d.wm_fwd = mf[0].data_ptr() if mf else None
            d.wm_bwd = mf[1].data_ptr() if mf and mf[1] is not None else None
            uf = net._packed_up.get(s.conv)
            d.wu_fwd = uf[0].data_ptr() if uf else None
            d.wu_bwd = uf[1].data_ptr() if uf else None
            b3 = net._packed_b3.get(s.conv)
            d.wb_fwd = b3[0].data_ptr() if b3 else None
            d.wb_bwd = b3[1].data_ptr() if b3 else None
            bu = net._packed_b3u.get(s.conv)
            d.wbu_fwd = bu[0].data_ptr() if bu is not None else None
            d.wbu_bwd = bu[1].data_ptr() if bu is not None else None
            d.ws, d.ws_bytes, d.ws_defer = net._ws.data_ptr(), net._ws.numel() * 4, 0
            d.nrep, d.rep_stride = self.nrep, self.rep_stride
            if s.bn:
                if s.norm is not None:
                    bn = _get(net.features, s.norm)
                    d.gamma, d.beta = bn.weight.data_ptr(), bn.bias.data_ptr()
                    d.run_mean, d.run_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                else:            # identity BatchNorm over a resampled buffer (csrc/upsample_bilinear.hip)
                    one, zero, var = net._identity_bn(dev, s.cin)
                    d.gamma, d.beta, d.run_mean, d.run_var = one.data_ptr(), zero.data_ptr(), zero.data_ptr(), var.data_ptr()
                d.x_stats = xs(s.src)
                d.coef = cf(s.src)
                d.t_in = self.T[s.src].data_ptr()
                d.t_stats = ts(s.src)
                d.bn_grad = bg(s.norm or ('identity:' + s.conv))
                d.t_accumulate = 0 if last_reader[s.src] == i else 1
                d.final_c0 = consumed.get(s.src, 0)
                d.final_c1 = s.cin
                consumed[s.src] = max(consumed.get(s.src, 0), s.cin)
            d.out = self.X[s.dst].data_ptr()
            d.out_ctot, d.out_coff = bufs[s.dst][0], s.dst_coff
            if s.up == OP_CHANNEL_MASK:
                d.w = self.drop_masks.data_ptr() + 4 * self.mask_off[i]
            if s.dst != 'out':
                d.out_stats = xs(s.dst)
                d.fin_xstats, d.fin_tstats = xs(s.dst), ts(s.dst)
                d.fin_coef = cf(s.dst)
                d.g = self.T[s.dst].data_ptr()      # finalised in place before use
                d.g_ctot, d.g_coff = bufs[s.dst][0], s.dst_coff
                if s.up == UP_BILINEAR_OP:          # its consumer's BatchNorm is the identity: T IS dL/d(out)
                    d.fin_xstats, d.fin_tstats = None, None
                if i + 1 < n and specs[i + 1].up == OP_CHANNEL_MASK:
                    # a Dropout2d op follows: IT accumulates the statistics of these channels and carries their
                    # BatchNorm-backward finalize; this convolution's `g` is then already dL/d(its own output)
                    d.out_stats, d.fin_xstats, d.fin_tstats = None, None, None
            else:
                d.out_stats = None
                d.g, d.g_ctot, d.g_coff = None, bufs[s.dst][0], 0
        self._out_stats = [d.out_stats for d in self.descs]
        # BatchNorm table (device) for running statistics and fp32 gamma/beta gradients
        items = []
        for s in specs:
            if s.norm is None:
                continue
            bn = _get(net.features, s.norm)
            hi, wi = self.buf_hw[s.src]
            it = BnItem()
            it.x_stats, it.bn_grad = xs(s.src), bg(s.norm)
            it.run_mean, it.run_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            it.dgamma = net._grad_view[s.norm + '.weight'].data_ptr()
            it.dbeta = net._grad_view[s.norm + '.bias'].data_ptr()
            it.num_batches_tracked = bn.num_batches_tracked.data_ptr()
            it.C, it.count = s.cin, B * hi * wi
            items.append(it)
        self.n_bn = len(items)
        self.max_c = max(it.C for it in items)
        arr = (BnItem * len(items))(*items)
        self.bn_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)

    # -- launches -------------------------------------------------------------------------------
    def forward(self, x, training, defer_running=False):
        """defer_running: the caller promises a backward(tail=...) for this forward, which updates the running
        statistics in its end-of-step launch (they are read by nothing in between)"""
        L, st = _lib.lib(), _lib.stream_ptr()
        net = self.net
        if x.data_ptr() != self.X['in'].data_ptr():
            self.X['in'].copy_(x)
        ev = 0 if training else 1
        for d, os_ in zip(self.descs, self._out_stats):
            d.eval_mode = ev
            d.out_stats = os_ if training else None     # eval: running statistics, nothing accumulated
        if self.mask_off:
            # nn.Dropout2d: a fresh Bernoulli(1 - p) channel mask scaled by 1/(1-p) per training forward; identity in eval
            src = self.drop_masks if training else self.drop_ones
            for i, off in self.mask_off.items():
                self.descs[i].w = src.data_ptr() + 4 * off
            if training:
                p = net.drop_rate
                inject = getattr(net, '_dropout_inject', None)      # tests: masks given in the reference's call order
                if inject is not None:
                    flat = torch.cat([torch.as_tensor(m, dtype=torch.float32).reshape(-1) for m in inject])
                    self.drop_masks.copy_(flat.to(self.dev))
                else:
                    self.drop_masks.bernoulli_(1.0 - p).mul_(1.0 / (1.0 - p))
        if training:
            if self.arena_clean:                       # cleared by the Adam kernel of the previous fused step
                self.arena_clean = False
            else:
                self.arena.zero_()
        net._pack_weights()
        _lib.check(L.pdes_conv_forward(self.ctx, self.descs, len(self.descs), st), 'pdes_conv_forward')
        if training and not defer_running:
            _lib.check(L.pdes_bn_update_running(self.bn_table.data_ptr(), self.n_bn, self.max_c,
                                                ctypes.c_float(0.1), self.nrep, self.rep_stride, st),
                       'pdes_bn_update_running')
        return self.X['out']

    def backward(self, grad_y, need_input_grad=False, tail=None, bucket_hook=None):
        """parameter gradients are ACCUMULATED into net._gflat (zero it first for plain gradients).
        tail = (update_running, partials, B, H, W, w_const, w_cont, w_dir, w_neu, terms, terms_accum): finish with
        pdes_step_tail (BatchNorm parameter gradients + running statistics + loss terms in one launch).
        bucket_hook: _lib.BucketHook called by pdes_backward once the widest layers' weight gradients are final on
        the weight-gradient stream (data-parallel bucket, train.py)"""
        L, st = _lib.lib(), _lib.stream_ptr()
        n = len(self.descs)
        if not hasattr(self, '_reduce_n'):
            self._plan_wgrad_scratch()
        self.descs[n - 1].g = grad_y.data_ptr()
        # Weight gradients hang off the finalize -> dgrad chain (nothing reads them before the single
        # reduce at the end), so pdes_backward runs them on a second HIP stream.  Not under hipGraph
        # capture: the runtime serialises forked graph branches with heavier barriers than it saves.
        side = None
        if self.net.wgrad_stream and not torch.cuda.is_current_stream_capturing():
            side = ctypes.c_void_p(self._side_stream().cuda_stream)
        rt = self._reduce_table.data_ptr() if self._reduce_n else None
        hook = ctypes.byref(bucket_hook) if (bucket_hook is not None and side is not None) else None
        side_b = None
        if side is not None and self.net.wgrad_streams == 2:
            side_b = ctypes.c_void_p(self._side_stream('b').cuda_stream)
        _lib.check(L.pdes_backward2(self.ctx, self.descs, n, st, side, side_b, rt, self._reduce_index, hook), 'pdes_backward')
        if tail is None:
            _lib.check(L.pdes_bn_param_grads(self.bn_table.data_ptr(), self.n_bn, self.max_c, self.nrep,
                                             self.rep_stride, st), 'pdes_bn_param_grads')
        else:
            upd, partials, B, H, W, w0, w1, w2, w3, terms, accum = tail
            _lib.check(L.pdes_step_tail(self.bn_table.data_ptr(), self.n_bn, self.max_c, ctypes.c_float(0.1),
                                        1 if upd else 0, _lib.ptr(partials), B, H, W, w0, w1, w2, w3, _lib.ptr(terms),
                                        _lib.ptr(accum), self.nrep, self.rep_stride, st), 'pdes_step_tail')


class PdesOp(ctypes.Structure):
    """mirror of `pdes_op` (include/pdes_hip.h)"""
    _fields_ = [('kind', _I), ('arg', _I), ('stream', _I)]


OP_LAUNCH, OP_RECORD, OP_WAIT, OP_HOOK = 0, 1, 2, 3


class StepProgram:
    """One training step of an engine as LINEAR hipGraphs + the fork / join events between them (csrc/step_graph.hip):
    graph 0 = weight packing + forward chain + loss launch on the main stream; per backward segment (a stage of the
    network: LastTransUp, DecBlock2, ...) the finalize -> data-gradient chain on the main stream and the segment's weight
    gradients on one of the two weight-gradient streams, released by ONE event per segment; the early split-K reduce (+
    the data-parallel bucket hook) behind the segment that completes 60 % of the weights; a last graph with the first
    layer's weight gradient, the remaining reduce and the end-of-step launch.  Replayed by one pdes_program_run call."""

    def __init__(self, eng, loss_launch, tail, grad_y, seg_max=None, forward_only=False, split_w=False):
        L = _lib.lib()
        self.eng, self._L, self.graphs = eng, L, []
        net, descs, specs = eng.net, eng.descs, eng._chain_specs()
        n = len(descs)
        if eng.mask_off:
            raise NotImplementedError('Dropout2d draws its channel masks with torch RNG launches: eager steps only')
        if not hasattr(eng, '_reduce_n'):
            eng._plan_wgrad_scratch()
        for d, os_ in zip(descs, eng._out_stats):
            d.eval_mode, d.out_stats = 0, os_
        descs[n - 1].g = grad_y.data_ptr()
        # segments = stages of the network (layers of one dense block / transition), optionally cut into chunks
        segs, start = [], 0
        stage = lambda i: (specs[i].conv or specs[i].norm or '').split('.')[0]
        for i in range(1, n + 1):
            if i == n or stage(i) != stage(start) or (seg_max and i - start >= seg_max):
                segs.append((start, i))
                start = i
        self.segments = segs
        per = [(specs[i].cout * specs[i].cin * specs[i].k * specs[i].k) if eng._reduce_index[i] >= 0 else 0 for i in range(n)]
        total, done, trigger = sum(per), 0, None
        cap = eng._side_stream()                     # idle while the program is built
        cap_ptr = ctypes.c_void_p(cap.cuda_stream)

        def capture(fn):
            torch.cuda.synchronize(eng.dev)
            with torch.cuda.stream(cap):
                _lib.check(L.pdes_graph_begin(cap_ptr), 'pdes_graph_begin')
                try:
                    fn(cap_ptr)
                finally:
                    g = ctypes.c_void_p()
                    rc = L.pdes_graph_end(cap_ptr, ctypes.byref(g))
                _lib.check(rc, 'pdes_graph_end')
            self.graphs.append(g)
            return len(self.graphs) - 1

        def fwd(st):
            net._pack_weights()
            _lib.check(L.pdes_conv_forward(eng.ctx, descs, n, st), 'pdes_conv_forward')
            loss_launch(st)
        ops = [(OP_LAUNCH, capture(fwd), 0)]
        ev = 0
        used = set()
        self.forward_only = forward_only
        for k, (lo, hi) in enumerate(reversed(segs) if not forward_only else ()):
            g_chain = capture(lambda st: _lib.check(L.pdes_backward_chain(eng.ctx, descs, lo, hi, st), 'pdes_backward_chain'))
            ops.append((OP_LAUNCH, g_chain, 0))
            wlo = max(lo, 1)                          # the first layer's weight gradient goes last, on the main stream
            if hi > wlo and any(specs[i].conv is not None for i in range(wlo, hi)):
                side = 1 + (k & 1)
                if split_w and hi - wlo >= 2:          # the segment's weight gradients alternate between BOTH streams
                    def w_half(par):
                        def fn(st):
                            for i in range(hi - 1, wlo - 1, -1):
                                if (hi - 1 - i) % 2 == par and specs[i].conv is not None:
                                    _lib.check(L.pdes_backward_weights(eng.ctx, descs, i, i + 1, st), 'pdes_backward_weights')
                        return fn
                    g_w, g_w2 = capture(w_half(0)), capture(w_half(1))
                    ops += [(OP_RECORD, ev, 0), (OP_WAIT, ev, side), (OP_LAUNCH, g_w, side), (OP_WAIT, ev, 3 - side),
                            (OP_LAUNCH, g_w2, 3 - side)]
                    used.add(3 - side)
                else:
                    g_w = capture(lambda st: _lib.check(L.pdes_backward_weights(eng.ctx, descs, wlo, hi, st), 'pdes_backward_weights'))
                    ops += [(OP_RECORD, ev, 0), (OP_WAIT, ev, side), (OP_LAUNCH, g_w, side)]
                ev += 1
                used.add(side)
                done += sum(per[wlo:hi])
                if trigger is None and total and 5 * done >= 3 * total and lo > 0:
                    rows = [eng._reduce_index[i] for i in range(lo, n) if eng._reduce_index[i] >= 0]
                    if rows and rows[0] > 0:
                        trigger = (lo, rows[0])
                        other = 3 - side
                        if other in used:             # the reduce reads partials written on both weight-gradient streams
                            ops += [(OP_RECORD, ev, other), (OP_WAIT, ev, side)]
                            ev += 1
                        first, cnt, mx = rows[0], eng._reduce_n - rows[0], max(per[lo:n])
                        g_r = capture(lambda st: _lib.check(L.pdes_wgrad_reduce_all(
                            eng._reduce_table.data_ptr() + 24 * first, cnt, mx, st), 'pdes_wgrad_reduce_all'))
                        ops += [(OP_LAUNCH, g_r, side), (OP_HOOK, lo, side)]
        for side in sorted(used):                     # join the weight-gradient streams into the main stream
            ops += [(OP_RECORD, ev, side), (OP_WAIT, ev, 0)]
            ev += 1
        self.early = trigger

        def end(st):
            if specs[0].conv is not None:
                _lib.check(L.pdes_backward_weights(eng.ctx, descs, 0, 1, st), 'pdes_backward_weights')
            cnt = trigger[1] if trigger else eng._reduce_n
            if cnt:
                mx = max(per[i] for i in range(n) if 0 <= eng._reduce_index[i] < cnt)
                _lib.check(L.pdes_wgrad_reduce_all(eng._reduce_table.data_ptr(), cnt, mx, st), 'pdes_wgrad_reduce_all')
            upd, partials, B, H, W, w0, w1, w2, w3, terms, accum = tail
            _lib.check(L.pdes_step_tail(eng.bn_table.data_ptr(), eng.n_bn, eng.max_c, ctypes.c_float(0.1), 1 if upd else 0,
                                        _lib.ptr(partials), B, H, W, w0, w1, w2, w3, _lib.ptr(terms), _lib.ptr(accum),
                                        eng.nrep, eng.rep_stride, st), 'pdes_step_tail')
        if not forward_only:
            ops.append((OP_LAUNCH, capture(end), 0))
        torch.cuda.synchronize(eng.dev)
        self.n_ops = len(ops)
        self._ops = (PdesOp * len(ops))(*[PdesOp(*o) for o in ops])
        self._graph_arr = (ctypes.c_void_p * len(self.graphs))(*[g.value for g in self.graphs])
        self._streams = (ctypes.c_void_p * 3)(None, eng._side_stream().cuda_stream, eng._side_stream('b').cuda_stream)
        self.nodes = [L.pdes_graph_nodes(g) for g in self.graphs]

    def run(self, hook=None):
        self._streams[0] = _lib.stream_ptr()
        rc = self._L.pdes_program_run(self.eng.ctx, self._graph_arr, len(self.graphs), self._streams, 3, self._ops,
                                      self.n_ops, ctypes.byref(hook) if hook is not None else None)
        _lib.check(rc, 'pdes_program_run')

    def close(self):
        for g in self.graphs:
            self._L.pdes_graph_destroy(g)
        self.graphs = []

    __del__ = close


class _Lease:
    """an engine held by one autograd forward until its backward has run -- or until autograd drops the graph
    (the lease is garbage collected with the function's ctx)"""

    def __init__(self, eng):
        self.eng = eng
        eng.busy = True

    def release(self):
        if self.eng is not None:
            self.eng.busy = False
            self.eng = None

    __del__ = release


class _NetFn(torch.autograd.Function):
    """forward/backward of the whole network as ONE autograd node.  The node owns an engine (activations, BatchNorm
    batch statistics, gradient accumulators) from its forward to its backward, so several forwards may be outstanding
    -- `(loss(model(x1)) + loss(model(x2))).backward()`, or an eval forward between a forward and its backward --
    exactly as with the reference's nn.Module."""

    @staticmethod
    def forward(ctx, x, net, grad_on, *params):
        eng = net._acquire(x)
        ctx.net, ctx.eng = net, eng
        ctx.trained = net.training
        ctx.pver = sum(p._version for p in params)
        with _lib.device_guard(x.device):
            y = eng.forward(x, net.training)
            y = y.clone()      # the engine's output buffer is reused by its next forward
        # the engine stays leased only when a backward can follow
        # (grad mode is always off inside Function.forward: the caller's mode comes in as `grad_on`)
        ctx.lease = _Lease(eng) if (grad_on and any(p.requires_grad for p in params)) else None
        return y

    @staticmethod
    def backward(ctx, gy):
        net, eng, lease = ctx.net, ctx.eng, ctx.lease
        if not ctx.trained:
            raise RuntimeError('backward through an eval-mode forward is not implemented (BatchNorm in eval mode '
                               'is only used under torch.no_grad() by the reference)')
        if lease is None or lease.eng is not eng:
            raise RuntimeError('backward through this forward a second time: its activations have been released '
                               '(retain_graph is not supported by the HIP DenseED)')
        if sum(p._version for p in net._params) != ctx.pver:
            lease.release()
            raise RuntimeError('a parameter of the network was modified in place between forward and backward '
                               '(the packed weight images no longer match the activations)')
        with _lib.device_guard(gy.device):
            net._pack_weights()       # another forward may have re-packed; the live weights are unchanged (checked above)
            net._gscratch.zero_()
            net._grad_dirty = True    # a fused trainer sharing this buffer must clear it before its next step
            eng.backward(gy.contiguous())
            # hand autograd views of ONE fresh copy (it may keep them as .grad; the scratch is reused)
            fresh = net._gscratch.clone()
        lease.release()
        grads = [fresh[off:off + p.numel()].view(p.shape) for p, off in zip(net._params, net._offsets)]
        return (None, None, None) + tuple(grads)


class _HipNet(nn.Module):
    """shared machinery of DenseED and Decoder (and, with the module tree at the top level, MultiScaleCondGlow)"""
    _prefix = 'features.'          # parameter-name prefix of the modules the specs name

    @property
    def _root(self):
        return self.features

    def _finish_init(self, specs, bufs, drop_rate, upsample, out_activation):
        if not (0.0 <= float(drop_rate) < 1.0):
            raise ValueError(f'drop_rate must be in [0, 1); got {drop_rate}')
        self.drop_rate = float(drop_rate)
        if upsample not in ('nearest', 'bilinear'):
            raise ValueError(f"upsample must be 'nearest' or 'bilinear' (reference codec.py:132-146); got {upsample!r}")
        self._specs, self._bufs = specs, bufs
        self.features = nn.Sequential()
        _build_modules(self.features, specs)
        # optional output activation (reference codec.py:289-290, :355-356; every script passes None): a stateless
        # elementwise module applied to the network output by torch, registered under the reference's module name
        self._out_act = None
        if out_activation is not None:
            self._out_act = activation(out_activation)
            self.features.add_module(out_activation, self._out_act)
        self._flat = None
        self._engines = {}
        self._side_streams = {}
        self._grad_dirty = False
        # weight gradients on a second HIP stream beside the finalize -> data-gradient chain (PDES_WGRAD_STREAM=0 in
        # the environment at construction, or this attribute, selects the single-stream form)
        self.wgrad_stream = os.environ.get('PDES_WGRAD_STREAM', '1') != '0'
        # two side streams: the weight gradients of successive layers are independent and each is sized for ~1 wave per
        # SIMD, so two of them fill the chip better beside the data-gradient chain (1.9553 -> 1.9406 ms per step,
        # same-process A/B, ab_streams.py of the earlier rounds (git history)); PDES_WGRAD_STREAMS=1 selects one
        self.wgrad_streams = 1 if os.environ.get('PDES_WGRAD_STREAMS', '2') == '1' else 2

    # -- flat parameter / gradient storage -------------------------------------------------------
    def _flatten(self, device):
        """move every parameter into one flat fp32 buffer on `device` and re-point .data at views.
        Layout: [every BatchNorm weight/bias | the convolution weights in layer order].  Backward finishes the
        convolution weights of the LAST layers first and the BatchNorm gradients last (pdes_step_tail), so the tail of
        the gradient buffer is final early: the data-parallel trainer all-reduces it as its first bucket while the
        rest of the backward pass still runs (train.py)."""
        named = list(self.named_parameters())
        total = sum(p.numel() for _, p in named)
        flat = torch.empty(total, device=device, dtype=torch.float32)
        gflat = torch.zeros(total, device=device, dtype=torch.float32)
        self._grad_view = {}
        conv_names = {self._prefix + sp.conv + '.weight' for sp in self._specs if sp.conv is not None}
        order = [i for i, (nm, _) in enumerate(named) if nm not in conv_names] + \
                [i for i, (nm, _) in enumerate(named) if nm in conv_names]
        offsets, off = [0] * len(named), 0
        for i in order:
            offsets[i] = off
            off += named[i][1].numel()
        views = []
        for (name, p), off in zip(named, offsets):
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
            key = name[len(self._prefix):]
            self._grad_view[key] = gflat[off:off + n].view(p.shape)
            views.append(self._grad_view[key])
        self._offsets = offsets
        # gradient-buffer offset of each layer's convolution weight (ascending in layer order)
        # (a resampling op has no weights: it takes the offset of the convolution that follows it)
        names = [nm for nm, _ in named]
        self._conv_off, nxt_off = [0] * len(self._specs), total
        for i in range(len(self._specs) - 1, -1, -1):
            sp = self._specs[i]
            if sp.conv is not None:
                nxt_off = offsets[names.index(self._prefix + sp.conv + '.weight')]
            self._conv_off[i] = nxt_off
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.data = m.running_mean.data.to(device).contiguous()
                m.running_var.data = m.running_var.data.to(device).contiguous()
                m.num_batches_tracked.data = m.num_batches_tracked.data.to(device)
        self._flat, self._gscratch, self._gscratch_views = flat, gflat, views
        self._params = [p for _, p in named]
        import weakref
        me = weakref.ref(self)
        for p in self._params:                  # pde_surrogate_amd.optim finds the flat buffers through its parameters
            p._pdes_owner = me
        # packed weight copies (zero padded once; the pack kernel rewrites the live part every forward)
        self._packed, items, mx = {}, [], 0
        self._identity = {}
        for s in self._specs:
            if s.conv is None:
                continue
            kk = s.k * s.k
            wf = torch.zeros(s.cin * kk * _pad16(s.cout), device=device)
            wb = torch.zeros(s.cout * kk * _pad16(s.cin), device=device)
            self._packed[s.conv] = (wf, wb)
            it = PackItem()
            it.w = _get(self._root, s.conv).weight.data_ptr()
            it.w_fwd, it.w_bwd = wf.data_ptr(), wb.data_ptr()
            it.Cout, it.Cin, it.kk, it.cout_pad, it.cin_pad = s.cout, s.cin, kk, _pad16(s.cout), _pad16(s.cin)
            items.append(it)
            mx = max(mx, s.cout * s.cin * kk)
        self._pack_items = {'direct': [(s.conv, it) for s, it in zip([q for q in self._specs if q.conv is not None], items)]}
        arr = (PackItem * len(items))(*items)
        self._pack_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
        self._pack_n, self._pack_max = len(items), mx
        # matrix-core weight images for the 1x1 / 3x3 stride-1 convolutions with >= 16 input channels
        self._packed_mfma, mitems, mmx = {}, [], 0
        for s in self._specs:
            raw3 = getattr(s, 'kind', None) == 'raw' and s.k == 3 and s.stride == 1      # conditional Glow: Conv2dZeros on raw features
            if s.conv is None or s.k not in (1, 3, 5) or not (s.bn or raw3) or (s.stride != 1 and not (s.stride == 2 and s.k == 3)):
                continue
            kk = s.k * s.k
            pad128 = lambda n: (n + 127) // 128 * 128            # N-tiles are padded to a multiple of 8
            nf = _pad16(s.cin) * kk * pad128(s.cout)            # = ksteps * kk * ntiles_padded * 64
            nb = _pad16(s.cout) * kk * pad128(s.cin)
            wf = torch.zeros(nf, device=device)
            wb = torch.zeros(nb, device=device)
            self._packed_mfma[s.conv] = (wf, wb)
            it = MfmaPackItem()
            it.w = _get(self._root, s.conv).weight.data_ptr()
            it.wm_fwd, it.wm_bwd = wf.data_ptr(), wb.data_ptr()
            it.Cout, it.Cin, it.kk = s.cout, s.cin, kk
            mitems.append(it)
            self._pack_items.setdefault('mfma', []).append((s.conv, it))
            mmx = max(mmx, nf, nb)
        self._mpack_n, self._mpack_max = len(mitems), mmx
        # effective 2x2 weight images of the nearest-x2 + 3x3 convolutions (sub-pixel decomposition)
        self._packed_up, uitems, umx = {}, [], 0
        pad128 = lambda n: (n + 127) // 128 * 128
        for s in self._specs:
            if not (s.up == UP_NEAREST and s.k == 3 and s.stride == 1 and s.norm is not None):
                continue
            nf, nb = _pad16(s.cin) * pad128(s.cout) * 16, _pad16(s.cout) * pad128(s.cin) * 16
            uf, ub = torch.zeros(nf, device=device), torch.zeros(nb, device=device)
            self._packed_up[s.conv] = (uf, ub)
            it = UpPackItem()
            it.w = _get(self._root, s.conv).weight.data_ptr()
            it.wu_fwd, it.wu_bwd, it.Cout, it.Cin = uf.data_ptr(), ub.data_ptr(), s.cout, s.cin
            uitems.append(it)
            self._pack_items.setdefault('up', []).append((s.conv, it))
            umx = max(umx, nf, nb)
        self._upack_n, self._upack_max = len(uitems), umx
        if uitems:
            arr = (UpPackItem * len(uitems))(*uitems)
            self._upack_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
        # three-way bf16 split images of the wide 3x3 layers (conv_mfma_b3.hip: fp32 accuracy on the bf16 matrix pipe)
        self._packed_b3, bitems, bmx = {}, [], 0
        for s in self._specs:
            if not (s.conv is not None and s.k == 3 and s.stride == 1 and not s.up and s.bn and s.cin >= 64 and s.cout >= 80):
                continue
            nf, nb = ctypes.c_longlong(0), ctypes.c_longlong(0)
            _lib.check(_lib.lib().pdes_b3_image_elems(s.cout, s.cin, ctypes.byref(nf), ctypes.byref(nb)), 'pdes_b3_image_elems')
            bf = torch.zeros(nf.value, device=device, dtype=torch.int16)
            bb = torch.zeros(nb.value, device=device, dtype=torch.int16)
            self._packed_b3[s.conv] = (bf, bb)
            it = B3PackItem()
            it.w = _get(self._root, s.conv).weight.data_ptr()
            it.wb_fwd, it.wb_bwd, it.Cout, it.Cin = bf.data_ptr(), bb.data_ptr(), s.cout, s.cin
            bitems.append(it)
            self._pack_items.setdefault('b3', []).append((s.conv, it))
            bmx = max(bmx, nf.value // 24, nb.value // 24)
        self._bpack_n, self._bpack_max = len(bitems), bmx
        if bitems:
            arr = (B3PackItem * len(bitems))(*bitems)
            self._bpack_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
        # split images of the effective sub-pixel weights of the nearest-x2 + 3x3 layers (conv_mfma_b3_up.hip)
        self._packed_b3u, buitems, bumx = {}, [], 0
        for s in self._specs:
            if not (s.up == UP_NEAREST and s.k == 3 and s.stride == 1 and s.norm is not None and s.cin >= 64 and s.cout >= 32):
                continue
            nf, nb = ctypes.c_longlong(0), ctypes.c_longlong(0)
            _lib.check(_lib.lib().pdes_b3up_image_elems(s.cout, s.cin, ctypes.byref(nf), ctypes.byref(nb)), 'pdes_b3up_image_elems')
            img = torch.zeros(nf.value, device=device, dtype=torch.int16)
            imgb = torch.zeros(nb.value, device=device, dtype=torch.int16)
            self._packed_b3u[s.conv] = (img, imgb)
            it = B3UpPackItem()
            it.w = _get(self._root, s.conv).weight.data_ptr()
            it.wbu_fwd, it.wbu_bwd, it.Cout, it.Cin = img.data_ptr(), imgb.data_ptr(), s.cout, s.cin
            buitems.append(it)
            self._pack_items.setdefault('b3up', []).append((s.conv, it))
            bumx = max(bumx, nf.value // 24, nb.value // 24)
        self._bupack_n, self._bupack_max = len(buitems), bumx
        if buitems:
            arr = (B3UpPackItem * len(buitems))(*buitems)
            self._bupack_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
        self._ws = torch.empty(8 << 20, device=device)       # 32 MiB split-K scratch (weight gradients)
        if mitems:
            arr = (MfmaPackItem * len(mitems))(*mitems)
            self._mpack_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
        self._engines = {}
        self._lean, self._lean_key, self._lean_keep = None, None, []

    def _identity_bn(self, device, c):
        """(ones, zeros, 1 - eps) vectors: gamma / beta = running_mean / running_var of the identity BatchNorm through
        which a convolution reads a resampled buffer"""
        t = self._identity.get(device)
        if t is None or t[0].numel() < c:
            n = max(c, 256)
            t = self._identity[device] = (torch.ones(n, device=device), torch.zeros(n, device=device),
                                          torch.full((n,), 1.0 - 1e-5, device=device))
        return t

    # image kinds of the five packing tables: (table, forward bit, backward bit of pdes_conv_image_use)
    _IMG_BITS = {'direct': 1 | 2, 'mfma': 4 | 8, 'up': 16 | 32, 'b3': 64 | 128, 'b3up': 256 | 512}

    def _lean_tables(self):
        """the packing tables WITHOUT the images no kernel of any live engine reads under the options in force
        (pdes_conv_image_use over every descriptor of every engine): for the default net on the matrix cores that drops the
        27 of its 28 VALU images (the query cannot know that the first layer's data gradient is never asked for), the f32
        images of the three wide layers and one f32 sub-pixel image -- 29.5 -> 20 us,
        EXPERIMENTS.md round 4.  Rebuilt when an engine is added or an option changes; older tables stay alive (captured
        hipGraphs hold their addresses, and they remain sufficient for the engines that existed at capture time)."""
        key = self._lean_state()
        if self._lean_key == key:
            return self._lean
        engines = [e for pool in self._engines.values() for e in pool]
        engines += list(getattr(self, '_fwd_engines', {}).values())       # (conditional Glow: the y -> z direction's own chains)
        use, L, mask = {}, _lib.lib(), ctypes.c_int(0)
        for e in engines:
            for sp, d in zip(e._chain_specs(), e.descs):
                if sp.conv is None:
                    continue
                _lib.check(L.pdes_conv_image_use(e.ctx, ctypes.byref(d), ctypes.byref(mask)), 'pdes_conv_image_use')
                use[sp.conv] = use.get(sp.conv, 0) | mask.value
        lean = {}
        dev = self._flat.device
        full_max = {'direct': self._pack_max, 'mfma': self._mpack_max, 'up': self._upack_max, 'b3': self._bpack_max,
                    'b3up': self._bupack_max}
        for kind, bits in self._IMG_BITS.items():
            items = [it for name, it in self._pack_items.get(kind, []) if use.get(name, bits) & bits]
            if items:
                arr = (type(items[0]) * len(items))(*items)
                t = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
                self._lean_keep.append(t)
                lean[kind] = (t, len(items), full_max[kind])
            else:
                lean[kind] = (None, 0, 0)
        args, mx = [], 1
        for kind in ('direct', 'mfma', 'up', 'b3', 'b3up'):
            t, n, m = lean[kind]
            args += [t.data_ptr() if n else None, n]
            mx = max(mx, m)
        self._lean_args = (args, mx, sum(lean[k][1] for k in lean))
        self._lean, self._lean_key = lean, key
        return lean

    def _lean_state(self):
        """(options epoch, engines alive): engines are only ever added to a flattened net"""
        return (_lib.OPTIONS_EPOCH, sum(len(p) for p in self._engines.values()), len(getattr(self, '_fwd_engines', ())))

    def _pack_weights(self):
        """rebuild the packed weight images from the live weights: one launch (direct, MFMA, sub-pixel, bf16-split tables),
        only the images some kernel reads (`_lean_tables`; PDES_PACK_ALL=1 in the environment: every image)"""
        lean_ok = (self._engines or getattr(self, '_fwd_engines', None)) and os.environ.get('PDES_PACK_ALL', '0') != '1'
        if lean_ok and self._lean_key != self._lean_state():
            if torch.cuda.is_current_stream_capturing():
                # building a table copies host memory: never inside a capture.  No table yet, or one made for other
                # options / fewer engines (an option changed or an engine was added since the last eager pack): the graph
                # being captured gets the FULL tables -- a stale lean table could leave out an image its kernels read
                lean_ok = False
            else:
                self._lean_tables()
        if lean_ok:
            args, mx, n_items = self._lean_args
            if n_items:
                _lib.check(_lib.lib().pdes_pack_all2(*args, mx, _lib.stream_ptr()), 'pdes_pack_all2')
            return
        mx = max(self._pack_max, self._mpack_max if self._mpack_n else 0, self._upack_max if self._upack_n else 0,
                 self._bpack_max if self._bpack_n else 0, self._bupack_max if self._bupack_n else 0)
        rc = _lib.lib().pdes_pack_all2(self._pack_table.data_ptr(), self._pack_n,
                                       self._mpack_table.data_ptr() if self._mpack_n else None, self._mpack_n,
                                       self._upack_table.data_ptr() if self._upack_n else None, self._upack_n,
                                       self._bpack_table.data_ptr() if self._bpack_n else None, self._bpack_n,
                                       self._bupack_table.data_ptr() if self._bupack_n else None, self._bupack_n,
                                       mx, _lib.stream_ptr())
        _lib.check(rc, 'pdes_pack_all2')

    def _is_flat(self, device):
        if self._flat is None or self._flat.device != device:
            return False
        base = self._flat.data_ptr()
        for p, off in zip(self._params, self._offsets):
            if p.data_ptr() != base + 4 * off:
                return False
        return True

    # engines per (batch, size): forwards whose backward has not run yet.  A guard against leaked graphs (each engine holds
    # the activations of a forward: 0.1 GB at B = 32), not a limit of the kernels: raise it per class or per instance
    # (`net.MAX_OUTSTANDING = 64`) or with PDES_MAX_OUTSTANDING in the environment when a loop really holds more forwards.
    MAX_OUTSTANDING = int(os.environ.get('PDES_MAX_OUTSTANDING', '8'))

    def _pool(self, x):
        if not self._is_flat(x.device):
            self._flatten(x.device)
        key = (x.shape[0], x.shape[2], x.shape[3])
        pool = self._engines.get(key)
        if pool is None:
            with _lib.device_guard(x.device):
                pool = self._engines[key] = [self._new_engine(key)]
        return key, pool

    def _new_engine(self, key):
        return _Engine(self, *key)

    def _engine(self, x):
        """the primary engine of this (batch, size) -- the one a MixedResidualTrainer drives directly"""
        return self._pool(x)[1][0]

    def _acquire(self, x):
        """an engine no outstanding autograd forward (and no trainer) owns; a new one when all are taken"""
        key, pool = self._pool(x)
        for eng in pool:
            if not eng.busy and not eng.reserved:
                return eng
        if len(pool) >= self.MAX_OUTSTANDING:
            raise RuntimeError(f'{len(pool)} forward passes of shape {key} are waiting for their backward: call '
                               'backward() (or drop the outputs / use torch.no_grad()) before running more, or raise '
                               'net.MAX_OUTSTANDING (PDES_MAX_OUTSTANDING) if the loop really holds that many')
        with _lib.device_guard(x.device):
            eng = self._new_engine(key)
        pool.append(eng)
        return eng

    def zero_grad(self, set_to_none=True):
        """nn.Module.zero_grad; the default (gradients to None) without the per-parameter checks of the generic loop --
        0.11 -> 0.03 ms of host time per step of the drop-in loop body (train_codec_mixed_residual.py:225 there)"""
        if not set_to_none or getattr(self, '_flat', None) is None:
            return super().zero_grad(set_to_none)
        for p in self._params:
            p.grad = None

    def forward(self, x):
        _lib.require_cuda(x)
        if x.dim() != 4 or x.shape[1] != self._bufs['in'][0]:
            raise ValueError(f'expected input (B, {self._bufs["in"][0]}, H, W); got {tuple(x.shape)}')
        if x.dtype != torch.float32:
            raise RuntimeError('the HIP kernels compute in fp32: pass an fp32 input')
        x = x.contiguous()
        self._engine(x)   # flattens parameters before autograd sees them
        y = _NetFn.apply(x, self, torch.is_grad_enabled(), *self._params)
        return y if self._out_act is None else self._out_act(y)

    def forward_test(self, x):
        print('input: {}'.format(x.data.size()))
        y = self.forward(x)
        eng = self._engine(x)
        seen = []
        convs = [s for s in self._specs if s.conv is not None]
        for s in convs:
            stage = s.conv.split('.')[0]
            if stage not in seen:
                seen.append(stage)
        for stage in seen:
            last = [s for s in convs if s.conv.split('.')[0] == stage][-1]
            c = last.dst_coff + last.cout
            h, w = eng.buf_hw[last.dst]
            print('{}: {}'.format(stage, torch.Size((x.shape[0], c, h, w))))
        return y

    @property
    def model_size(self):
        return module_size(self)

    def reset_parameters(self, verbose=False):
        for module in self.modules():
            if isinstance(module, (nn.Conv2d, nn.BatchNorm2d)):
                module.reset_parameters()
                if verbose:
                    print('Reset parameters in {}'.format(module))


class DenseED(_HipNet):
    def __init__(self, in_channels, out_channels, imsize, blocks, growth_rate=16,
                 init_features=48, drop_rate=0, bn_size=8, bottleneck=False,
                 out_activation=None, upsample='nearest'):
        """Dense Convolutional Encoder-Decoder Network (reference codec.py:210-293).

        Args are the reference's (`bn_size` only matters with `bottleneck=True`, as there).
        """
        super(DenseED, self).__init__()
        blocks = [int(b) for b in blocks]
        if upsample not in ('nearest', 'bilinear'):
            raise ValueError(f"upsample must be 'nearest' or 'bilinear' (reference codec.py:132-146); got {upsample!r}")
        specs, bufs = _plan_densed(blocks, growth_rate, init_features, in_channels, out_channels, imsize, upsample,
                                   drop=bool(drop_rate and drop_rate > 0), bottleneck=bool(bottleneck), bn_size=bn_size)
        self._finish_init(specs, bufs, drop_rate, upsample, out_activation)
        print('# params {}, # conv layers {}'.format(*self.model_size))


class Decoder(_HipNet):
    """Decoder to solve one PDE instance (reference codec.py:321-370)."""

    def __init__(self, dim_latent, out_channels, blocks, growth_rate=16, init_features=48,
                 drop_rate=0., upsample='nearest', out_activation=None):
        super(Decoder, self).__init__()
        blocks = [int(b) for b in blocks]
        if upsample not in ('nearest', 'bilinear'):
            raise ValueError(f"upsample must be 'nearest' or 'bilinear' (reference codec.py:344-347); got {upsample!r}")
        specs, bufs = _plan_decoder(blocks, growth_rate, init_features, dim_latent, out_channels, upsample,
                                    drop=bool(drop_rate and drop_rate > 0))
        self._finish_init(specs, bufs, drop_rate, upsample, out_activation)
