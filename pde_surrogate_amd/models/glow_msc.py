"""Multiscale conditional Glow on MI355X -- drop-in for the reference's models/glow_msc.py (MultiScaleCondGlow, :672-968)
on the path train_cglow_reverse_kl.py trains: ``generate`` (z ~ p(z|x) -> y with log p(y|x), :783-829) with its backward
pass, plus eval-mode ``generate`` / ``sample`` / ``predict`` and the inference-only y -> z direction (``forward``).

Same constructor arguments, ``state_dict`` keys and (under the same torch / numpy seeds) initial parameters as the
reference.  The whole of ``generate`` -- input encoder, top-latent prior, 18 reversible layers with their dense coupling
networks, unsqueezes, split priors, log-determinants -- is ONE descriptor chain of the C ABI (include/pdes_hip.h): the
convolutions run on the DenseED kernels (BatchNorm+ReLU fused into the operand load, dense-block buffers instead of
torch.cat), the flow operators are PDES_OP_* descriptors (csrc/flow_ops.hip), the backward pass is pdes_backward2 over
the same chain (weight gradients on two side streams), every parameter lives in one flat buffer (one Adam launch, one
all-reduce).  There is no CPU implementation.

Not built: ``flow_coupling='wide'`` (no reference script selects it), ``train_sampling=False`` (maximum-likelihood
training through the y -> z direction: that direction is inference-only here), squeeze factors other than 2,
``x_channels != 1`` (the reference's own encoder only works for 1: glow_msc.py:34-36 vs :495).
"""
import ctypes
import os

import numpy as np
import scipy.linalg
import torch
import torch.nn as nn

from .. import _lib
from .codec import BnItem, ConvDesc, _EngineBase, _HipNet, _Lease, _get, _pad16

_I = ctypes.c_int
_P = ctypes.c_void_p

OP_COPY, OP_BIAS_SCALE, OP_COUPLING, OP_MIX, OP_UNSQUEEZE, OP_GAUSS = 4, 5, 6, 7, 8, 9       # include/pdes_hip.h
FLOW_FORWARD, GAUSS_DETACH_LSD, MIX_COUPLED = 1, 2, 4
MAX_MIX_CHANNELS = 48
_FUSE_COPY_FINALIZE = True   # PDES_OP_COPY backward applies its channels' finalize on load (False: separate launch; A/B only)
_MERGE_COPY = True        # torch.cat((y1, cond), 1) as ONE two-source copy descriptor (False: one per source; A/B only)
_FUSE_COUPLING_MIX = True  # generate(): affine coupling + ActNorm / invertible 1x1 of a reversible layer as ONE descriptor
                          # (PDES_MIX_COUPLED: the coupling's output is never stored; False: two launches each way; A/B only)
_FOLD_ZEROS = True        # the coupling kernels apply the coupling net's Conv2dZeros epilogue (bias, exp(3 scale)) and its
                          # backward themselves (False: a PDES_OP_BIAS_SCALE launch in between, each way; A/B only)


class FlowItem(ctypes.Structure):
    """mirror of `pdes_flow_item`"""
    _fields_ = [('C', _I), ('HW', _I), ('lu', _I), ('l', _P), ('u', _P), ('log_s', _P), ('p', _P), ('sign_s', _P),
                ('weight', _P), ('an_weight', _P), ('an_bias', _P), ('W', _P), ('Winv', _P), ('logdet', _P), ('acc', _P),
                ('dl', _P), ('du', _P), ('dlog_s', _P), ('dweight', _P), ('dan_weight', _P), ('dan_bias', _P)]


# ------------------------------------------------------------------------------------------------
# parameter containers with the reference's names and initial values (their forward is never called)
class Conv2dZeros(nn.Module):
    """3x3 convolution + bias, then a per-channel exp(3 scale); everything starts at zero (glow_msc.py:237-252)"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=True)
        nn.init.zeros_(self.conv.weight)
        nn.init.zeros_(self.conv.bias)
        self.scale = nn.Parameter(torch.zeros(1, out_channels, 1, 1))


class ActNorm(nn.Module):
    """per-channel affine, identity at the start (glow_msc.py:50-96)"""

    def __init__(self, in_features):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(in_features, 1, 1))
        self.bias = nn.Parameter(torch.zeros(in_features, 1, 1))
        self.data_init = False
        self.data_initialized = False


def _random_rotation(c):
    """Q of the QR factorisation of a standard normal matrix, drawn from numpy's global stream (glow_msc.py:124, :184)"""
    q, _ = np.linalg.qr(np.random.randn(c, c))
    return q.astype(np.float32)


class InvertibleConv1x1(nn.Module):
    """plain parameterisation: the matrix itself (glow_msc.py:99-157)"""

    def __init__(self, in_channels, train_sampling=True):
        super().__init__()
        self.w_shape = (in_channels, in_channels)
        self.train_sampling = train_sampling
        self.weight = nn.Parameter(torch.from_numpy(_random_rotation(in_channels)))


class InvertibleConv1x1LU(nn.Module):
    """W = P (L + I) (U + diag(sign_s exp(log_s))) with a fixed permutation P (glow_msc.py:161-233)"""

    def __init__(self, in_channels, train_sampling=True):
        super().__init__()
        c = in_channels
        self.w_shape = (c, c)
        self.train_sampling = train_sampling
        perm, lower, upper = scipy.linalg.lu(_random_rotation(c))
        diag = np.diag(upper)
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        self.register_buffer('p', f32(perm))
        self.l = nn.Parameter(f32(lower))
        self.u = nn.Parameter(f32(np.triu(upper, k=1)))
        self.log_s = nn.Parameter(f32(np.log(np.abs(diag))))
        self.register_buffer('sign_s', f32(np.sign(diag)))
        self.register_buffer('l_mask', f32(np.tril(np.ones((c, c)), -1)))
        self.register_buffer('u_mask', f32(np.triu(np.ones((c, c)), 1)))
        self.register_buffer('eye', f32(np.eye(c)))


def _dense_layer(cin, growth):
    m = nn.Sequential()
    m.add_module('norm1', nn.BatchNorm2d(cin))
    m.add_module('conv1', nn.Conv2d(cin, growth, kernel_size=3, stride=1, padding=1, bias=False))
    return m


# ------------------------------------------------------------------------------------------------
class _Spec:
    """one descriptor of the chain.  kind: 'conv' (BatchNorm+ReLU in front), 'raw' (a convolution that reads its input as
    is), or an operator code"""

    def __init__(self, kind, src, dst, cin, cout, scale, dst_coff=0, conv=None, norm=None, k=0, stride=1, pad=0, **extra):
        self.kind, self.src, self.dst, self.cin, self.cout, self.scale, self.dst_coff = kind, src, dst, cin, cout, scale, dst_coff
        self.conv, self.norm, self.k, self.stride, self.pad = conv, norm, k, stride, pad
        self.up = 0 if kind in ('conv', 'raw') else kind
        self.fake_bn = False
        self.x = extra                       # operator specifics (second source, parameter paths, ...)

    @property
    def bn(self):
        return self.kind == 'conv'


def _plan_encoder(y_channels, enc_blocks, flow_blocks, growth, init_features, with_grad):
    """the input encoder and the top latent's prior (glow_msc.py:474-546): shared by both directions of the flow"""
    L = len(flow_blocks)
    specs, bufs = [], {}
    bufs['in'] = [1, (1, 1)]
    conds = []
    c = init_features
    res = (1, 1)
    cur = 'e1'
    bufs[cur] = [c + (enc_blocks[0] - 1) * growth, res]
    specs.append(_Spec(OP_COPY, 'in', cur, 1, 1, res, grad=None))
    specs.append(_Spec('raw', 'in', cur, 1, init_features - 1, res, dst_coff=1, conv='encoder.dense_block1.in_conv',
                       k=3, pad=1, grad=None))
    specs.append(_Spec(OP_BIAS_SCALE, cur, cur, init_features - 1, init_features - 1, res, dst_coff=1,
                       bias='encoder.dense_block1.in_conv.bias', scale_p=None))
    for i, n_layers in enumerate(enc_blocks, 1):
        blk = f'encoder.dense_block{i}'
        for j in range(1, (n_layers - 1 if i == 1 else n_layers) + 1):
            p = f'{blk}.denselayer{j}'
            specs.append(_Spec('conv', cur, cur, c, growth, res, dst_coff=c, conv=p + '.conv1', norm=p + '.norm1', k=3, pad=1))
            c += growth
        conds.append((cur, c, res))
        if i < len(enc_blocks):
            t = f'encoder.trans_down{i}'
            nres = (1, res[1] * 2)
            nxt = f'e{i + 1}'
            bufs[nxt] = [c // 2 + enc_blocks[i] * growth, nres]
            if i > 1:                                   # bottleneck transition (glow_msc.py:512-516, codec.py:105-118)
                mid = f'm{i}'
                bufs[mid] = [c // 2, res]
                specs.append(_Spec('conv', cur, mid, c, c // 2, res, conv=t + '.conv1', norm=t + '.norm1', k=1))
                specs.append(_Spec('conv', mid, nxt, c // 2, c // 2, res, conv=t + '.conv2', norm=t + '.norm2', k=3,
                                   stride=2, pad=1))
            else:
                specs.append(_Spec('conv', cur, nxt, c, c // 2, res, conv=t + '.conv1', norm=t + '.norm1', k=3, stride=2,
                                   pad=1))
            cur, c, res = nxt, c // 2, nres
    # channels of the flow at every level (glow_msc.py:878-896)
    C = {1: y_channels}
    for i in range(2, L + 1):
        C[i] = (C[i - 1] if i == 2 else C[i - 1] // 2) * 4
    if C[L] > MAX_MIX_CHANNELS:
        raise ValueError(f'{L} flow levels need {C[L]}-channel invertible 1x1 convolutions; the kernels hold up to '
                         f'{MAX_MIX_CHANNELS}')
    fres = {i: (1, 2 ** (i - 1)) for i in range(1, L + 1)}
    # ---- top latent (glow_msc.py:519-526, :806-811)
    bufs['top'] = [2 * C[L], res]
    specs.append(_Spec('raw', cur, 'top', c, 2 * C[L], res, conv='encoder.top_latent.conv', k=3, pad=1,
                       grad='D' if with_grad else None))
    specs.append(_Spec(OP_BIAS_SCALE, 'top', 'top', 2 * C[L], 2 * C[L], res, bias='encoder.top_latent.conv.bias',
                       scale_p='encoder.top_latent.scale'))
    return specs, bufs, conds, C, fres


def _coupling_net(specs, bufs, lay, z, nb, hb, n1, n2, cond, cc, r, growth, with_grad):
    """torch.cat((y1, cond), 1) -> three dense layers -> BatchNorm, ReLU, Conv2dZeros (glow_msc.py:274-293, :321, :339)"""
    cn = n1 + cc
    bufs[nb] = [cn + 3 * growth, r]
    bufs[hb] = [2 * n2, r]
    if _MERGE_COPY:
        specs.append(_Spec(OP_COPY, z, nb, n1, n1 + cc, r, grad='T' if with_grad else None, src2=cond,
                           grad2='D' if with_grad else None))
    else:
        specs.append(_Spec(OP_COPY, z, nb, n1, n1, r, grad='T' if with_grad else None))
        specs.append(_Spec(OP_COPY, cond, nb, cc, cc, r, dst_coff=n1, grad='D' if with_grad else None))
    cp = lay + '.coupling.coupling_nn'
    for k in range(1, 4):
        specs.append(_Spec('conv', nb, nb, cn + (k - 1) * growth, growth, r, dst_coff=cn + (k - 1) * growth,
                           conv=f'{cp}.denselayer{k}.conv1', norm=f'{cp}.denselayer{k}.norm1', k=3, pad=1))
    specs.append(_Spec('conv', nb, hb, cn + 3 * growth, 2 * n2, r, conv=cp + '.reduce.conv_zero.conv',
                       norm=cp + '.reduce.norm1', k=3, pad=1))
    zeros = dict(bias=cp + '.reduce.conv_zero.conv.bias', scale_p=cp + '.reduce.conv_zero.scale')
    if _FOLD_ZEROS:
        return zeros                 # -> the PDES_OP_COUPLING descriptor behind the net
    specs.append(_Spec(OP_BIAS_SCALE, hb, hb, 2 * n2, 2 * n2, r, **zeros))
    return {}


def _plan_glow(y_channels, enc_blocks, flow_blocks, lu, growth=16, init_features=48):
    """descriptors of generate(), in execution order, the activation buffers and the tables the engine needs"""
    L = len(flow_blocks)
    specs, bufs, conds, C, fres = _plan_encoder(y_channels, enc_blocks, flow_blocks, growth, init_features, True)
    mix, eps = [], {}
    res = fres[L]
    z = f'z{L}_{flow_blocks[L - 1]}'
    bufs[z] = [C[L], fres[L]]
    eps[L - 2] = f'eps{L - 2}'
    bufs[eps[L - 2]] = [C[L], fres[L]]
    specs.append(_Spec(OP_GAUSS, 'top', z, C[L], C[L], fres[L], prior='top', eps=eps[L - 2], flags=GAUSS_DETACH_LSD))
    # ---- the flow, z -> y (glow_msc.py:813-829)
    for i in range(L, 0, -1):
        blk = f'flow.revblock{i}'
        cond, cc, r = conds[i - 1]
        nl = flow_blocks[i - 1]
        Ci, n2 = C[i], C[i] // 2
        n1 = Ci - n2
        if 1 < i < L:                                   # Split.reverse (glow_msc.py:575-587)
            z = f'z{i}_{nl}'
            p = f'p{i}'
            bufs[p] = [Ci, r]
            pc = blk + '.split.latent_encoder.conv2d'
            specs.append(_Spec('raw', z, p, Ci // 2, Ci, r, conv=pc + '.conv', k=3, pad=1, grad='T'))
            specs.append(_Spec(OP_BIAS_SCALE, p, p, Ci, Ci, r, bias=pc + '.conv.bias', scale_p=pc + '.scale'))
            eps[i - 2] = f'eps{i - 2}'
            bufs[eps[i - 2]] = [Ci // 2, r]
            specs.append(_Spec(OP_GAUSS, p, z, Ci // 2, Ci // 2, r, dst_coff=Ci // 2, prior=p, eps=eps[i - 2], flags=0))
        for j in range(nl, 0, -1):
            lay = f'{blk}.revlayers.revlayer{j}'
            first = i == 1 and j == 1
            z = f'z{i}_{j}'
            nb, hb = f'n{i}_{j}', f'h{i}_{j}'
            zeros = _coupling_net(specs, bufs, lay, z, nb, hb, n1, n2, cond, cc, r, growth, True)
            if first:
                bufs['out'] = [Ci, r]
                specs.append(_Spec(OP_COUPLING, z, 'out', Ci, Ci, r, h=hb, **zeros))
            else:
                ub, zn = f'u{i}_{j}', f'z{i}_{j - 1}'
                if zn not in bufs:
                    bufs[zn] = [Ci, r]
                if _FUSE_COUPLING_MIX:
                    specs.append(_Spec(OP_MIX, z, zn, Ci, Ci, r, index=len(mix), flags=MIX_COUPLED, h=hb, **zeros))
                else:
                    bufs[ub] = [Ci, r]
                    specs.append(_Spec(OP_COUPLING, z, ub, Ci, Ci, r, h=hb, **zeros))
                    specs.append(_Spec(OP_MIX, ub, zn, Ci, Ci, r, index=len(mix)))
                mix.append((Ci, r, lay + '.norm', lay + '.conv1x1', i, j))
        if i > 1:                                       # Squeeze.reverse (glow_msc.py:629-636, :422-432)
            zt = f'z{i - 1}_{flow_blocks[i - 2]}'
            ct = C[i - 1]
            if zt not in bufs:
                bufs[zt] = [ct, fres[i - 1]]
            specs.append(_Spec(OP_UNSQUEEZE, f'z{i}_0', zt, Ci, Ci // 4, r))
    return specs, bufs, dict(C=C, L=L, mix=mix, eps=eps, conds=conds, inputs=['in'] + list(eps.values()), grad=True)


def _plan_glow_forward(y_channels, enc_blocks, flow_blocks, lu, mix_index, growth=16, init_features=48):
    """descriptors of the y -> z direction (MultiScaleCondGlow.forward, glow_msc.py:746-780), inference only: the same
    encoder, then per level squeeze, [ActNorm, inverse 1x1,] coupling forward, split with the log-probability of the
    factored-out half.  mix_index: (level, layer) -> row of the matrix table built for generate()"""
    L = len(flow_blocks)
    specs, bufs, conds, C, fres = _plan_encoder(y_channels, enc_blocks, flow_blocks, growth, init_features, False)
    bufs['yin'] = [C[1], (1, 1)]
    cur, ccur, eps = 'yin', C[1], {}
    for i in range(1, L + 1):
        blk = f'flow.revblock{i}'
        cond, cc, r = conds[i - 1]
        Ci, n2 = C[i], C[i] // 2
        n1 = Ci - n2
        if i > 1:                                        # Squeeze.forward on the half that continues (glow_msc.py:410-420)
            nxt = f'f{i}_0'
            bufs[nxt] = [Ci, r]
            specs.append(_Spec(OP_UNSQUEEZE, cur, nxt, Ci // 4, Ci, fres[i - 1], flags=FLOW_FORWARD))
            cur = nxt
        for j in range(1, flow_blocks[i - 1] + 1):
            lay = f'{blk}.revlayers.revlayer{j}'
            if not (i == 1 and j == 1):
                gb = f'g{i}_{j}'
                bufs[gb] = [Ci, r]
                specs.append(_Spec(OP_MIX, cur, gb, Ci, Ci, r, index=mix_index[(i, j)], flags=FLOW_FORWARD))
                cur = gb
            nb, hb, fb = f'n{i}_{j}', f'h{i}_{j}', f'f{i}_{j}'
            zeros = _coupling_net(specs, bufs, lay, cur, nb, hb, n1, n2, cond, cc, r, growth, False)
            bufs[fb] = [Ci, r]
            specs.append(_Spec(OP_COUPLING, cur, fb, Ci, Ci, r, h=hb, flags=FLOW_FORWARD, **zeros))
            cur = fb
        if 1 < i < L:                                    # Split.forward (glow_msc.py:561-573)
            p = f'p{i}'
            bufs[p] = [Ci, r]
            pc = blk + '.split.latent_encoder.conv2d'
            specs.append(_Spec('raw', cur, p, Ci // 2, Ci, r, conv=pc + '.conv', k=3, pad=1, grad=None))
            specs.append(_Spec(OP_BIAS_SCALE, p, p, Ci, Ci, r, bias=pc + '.conv.bias', scale_p=pc + '.scale'))
            eps[i - 2] = f'eps{i - 2}'
            bufs[eps[i - 2]] = [Ci // 2, r]
            specs.append(_Spec(OP_GAUSS, cur, eps[i - 2], Ci // 2, Ci // 2, r, prior=p, src_coff=Ci // 2, flags=FLOW_FORWARD))
    eps[L - 2] = f'eps{L - 2}'
    bufs[eps[L - 2]] = [C[L], fres[L]]
    specs.append(_Spec(OP_GAUSS, cur, eps[L - 2], C[L], C[L], fres[L], prior='top', src_coff=0, flags=FLOW_FORWARD))
    return specs, bufs, dict(C=C, L=L, mix=None, eps=eps, conds=conds, inputs=['in', 'yin'], grad=False, z=cur)


# ------------------------------------------------------------------------------------------------
class _GlowEngine(_EngineBase):
    """activation / gradient buffers, accumulator arena and descriptors of generate() for one (batch, size, device)"""

    def __init__(self, net, B, H, W, plan=None):
        self.net, self.B = net, B
        dev = net._flat.device
        self.dev = dev
        self.ctx = _lib.context(dev)
        self.busy = self.reserved = False
        L = _lib.lib()
        specs, bufs, meta = plan if plan is not None else (net._specs, net._bufs, net._meta)
        self.specs, self.meta = specs, meta
        self.has_grad = bool(meta['grad'])
        mixes = net._meta['mix']                      # the matrix table is per reversible layer: shared by both directions
        self.buf_hw = {}
        for k, (c, sc) in bufs.items():
            if H % sc[1] or W % sc[1] or ((H // sc[1]) * (W // sc[1])) % 4:
                raise ValueError(f'image size {H}x{W} does not divide down to the flow level 1/{sc[1]} (or leaves fewer '
                                 'than 4 pixels there)')
            self.buf_hw[k] = (H // sc[1], W // sc[1])
        f32 = dict(device=dev, dtype=torch.float32)
        inputs = set(meta['inputs'])
        self.X = {k: torch.empty((B, c) + self.buf_hw[k], **f32) for k, (c, sc) in bufs.items()}
        self.T = {}
        if self.has_grad:
            self.T = {k: torch.empty((B, c) + self.buf_hw[k], **f32) for k, (c, sc) in bufs.items()
                      if k not in inputs and k != 'out'}
        # gradients of buffers that are read both through BatchNorms and as they are (the encoder's features): one flat
        # allocation, cleared at the start of every backward pass
        direct = [s.src for s in specs if s.x.get('grad') == 'D'] + [s.x['src2'] for s in specs if s.x.get('grad2') == 'D']
        self.direct = sorted(set(direct))
        n_d = sum(bufs[k][0] * self.buf_hw[k][0] * self.buf_hw[k][1] * B for k in self.direct)
        self.Dflat = torch.zeros(max(n_d, 1), **f32)
        self.D, off = {}, 0
        for k in self.direct:
            n = bufs[k][0] * self.buf_hw[k][0] * self.buf_hw[k][1] * B
            self.D[k] = self.Dflat[off:off + n].view((B, bufs[k][0]) + self.buf_hw[k])
            off += n
        # ---- fp64 arena (replicated): per buffer {sum x, sum x^2} and {sum T, sum T xhat}; per BatchNorm {dgamma, dbeta};
        #      per bias/scale op {dbias, dscale}; per invertible 1x1 + ActNorm {dweight, dbias, dW}; log p per sample
        n_stat, self.stat_off = 0, {}
        for k, (c, sc) in bufs.items():
            self.stat_off[k] = n_stat
            n_stat += 2 * c
        bn_off, n_bn = {}, 0
        for s in specs:
            if s.bn:
                bn_off[s.norm] = n_bn
                n_bn += 2 * s.cin
        aux_off, n_aux = {}, 0
        for i, s in enumerate(specs):
            if s.kind == OP_BIAS_SCALE or (s.kind in (OP_COUPLING, OP_MIX) and s.x.get('bias')):
                aux_off[i] = n_aux
                n_aux += 2 * _aux_channels(s)
        mix_off, n_mix = [], 0
        for (c, *_rest) in mixes:
            mix_off.append(n_mix)
            n_mix += 2 * c + c * c
        base_bn, base_aux = 2 * n_stat, 2 * n_stat + n_bn
        base_mix = base_aux + n_aux
        base_logp = base_mix + n_mix
        self.nrep, self.rep_stride = L.pdes_stat_replicas(), base_logp + B
        self.arena = torch.zeros(self.nrep * self.rep_stride, device=dev, dtype=torch.float64)
        a0 = self.arena.data_ptr()
        xs = lambda k: a0 + 8 * self.stat_off[k]
        ts = lambda k: a0 + 8 * (n_stat + self.stat_off[k])
        self._logp_acc = a0 + 8 * base_logp
        self.logp = torch.empty(B, **f32)
        self.glogp = torch.zeros(B, **f32)
        # ---- invertible 1x1 convolutions + ActNorms: matrices, log-determinants, the device table of pdes_flow_prepare
        nm = len(mixes)
        self.Wtab = torch.zeros(max(sum(m[0] * m[0] for m in mixes), 1), **f32)
        self.Winv = torch.zeros_like(self.Wtab)
        self.logdet = torch.zeros(max(nm, 1), device=dev, dtype=torch.float64)
        items, woff = [], 0
        self._w_ptr = []
        gv = net._grad_view
        for m, (c, r, npath, cpath, _, _) in enumerate(mixes):
            an, cv = _get(net, npath), _get(net, cpath)
            hw = self.buf_hw_of(r, H, W)
            it = FlowItem()
            it.C, it.HW, it.lu = c, hw[0] * hw[1], 1 if net.LUdecompose else 0
            if net.LUdecompose:
                it.l, it.u, it.log_s = cv.l.data_ptr(), cv.u.data_ptr(), cv.log_s.data_ptr()
                it.p, it.sign_s = cv.p.data_ptr(), cv.sign_s.data_ptr()
                it.dl, it.du, it.dlog_s = (gv[cpath + '.l'].data_ptr(), gv[cpath + '.u'].data_ptr(),
                                           gv[cpath + '.log_s'].data_ptr())
            else:
                it.weight, it.dweight = cv.weight.data_ptr(), gv[cpath + '.weight'].data_ptr()
            it.an_weight, it.an_bias = an.weight.data_ptr(), an.bias.data_ptr()
            it.dan_weight, it.dan_bias = gv[npath + '.weight'].data_ptr(), gv[npath + '.bias'].data_ptr()
            it.W, it.Winv = self.Wtab.data_ptr() + 4 * woff, self.Winv.data_ptr() + 4 * woff
            it.logdet = self.logdet.data_ptr() + 8 * m
            it.acc = a0 + 8 * (base_mix + mix_off[m])
            self._w_ptr.append((it.W, it.Winv, it.acc))
            items.append(it)
            woff += c * c
        self.n_mix = nm
        if nm:
            self.flow_table = torch.frombuffer(bytearray(bytes((FlowItem * nm)(*items))), dtype=torch.uint8).to(dev)
        # ---- descriptors
        n = len(specs)
        self.descs = (ConvDesc * n)()
        bn_readers = {}
        for i, s in enumerate(specs):
            if s.bn:
                bn_readers.setdefault(s.src, []).append(i)
        consumed = {}
        # channels of a buffer that some BatchNorm reads: [0, bn_max); what lies above (the last dense layer's output of
        # the encoder's last block) has direct consumers only -- its gradient is the direct accumulator alone
        bn_max = {k: max(specs[i].cin for i in v) for k, v in bn_readers.items()}
        pk = net._packed
        self._train_stats = []
        for i, s in enumerate(specs):
            d = self.descs[i]
            hi, wi = self.buf_hw[s.src]
            ho, wo = self.buf_hw[s.dst]
            d.B, d.Cin, d.Cout, d.Hin, d.Win, d.Hout, d.Wout = B, s.cin, s.cout, hi, wi, ho, wo
            d.ksize, d.stride, d.pad, d.upsample = s.k, s.stride, s.pad, s.up
            d.x, d.x_ctot = self.X[s.src].data_ptr() + 4 * s.x.get('src_coff', 0) * hi * wi, bufs[s.src][0]
            d.flags = s.x.get('flags', 0)
            d.eps, d.nrep, d.rep_stride = 1e-5, self.nrep, self.rep_stride
            d.cout_pad, d.cin_pad = _pad16(s.cout), _pad16(s.cin)
            d.out, d.out_ctot, d.out_coff = self.X[s.dst].data_ptr(), bufs[s.dst][0], s.dst_coff
            dst_bn = s.dst in bn_readers                 # the destination is read through BatchNorms: statistics + finalize
            carried = i + 1 < n and specs[i + 1].kind == OP_BIAS_SCALE and specs[i + 1].dst == s.dst   # ... by the op behind it
            if s.dst in self.T:
                d.g, d.g_ctot, d.g_coff = self.T[s.dst].data_ptr(), bufs[s.dst][0], s.dst_coff
            if dst_bn and s.dst_coff >= bn_max[s.dst]:
                dst_bn = False
                if self.has_grad:
                    if s.dst not in self.D:
                        raise RuntimeError(f'channels [{s.dst_coff}, {s.dst_coff + s.cout}) of {s.dst} have no consumer')
                    d.g = self.D[s.dst].data_ptr()
            elif dst_bn and s.dst_coff + s.cout > bn_max[s.dst]:
                raise RuntimeError(f'{s.dst}: a layer\'s output is only partly read through BatchNorms')
            if dst_bn and not carried and s.kind != OP_COUPLING:
                d.out_stats = xs(s.dst)
                d.fin_xstats, d.fin_tstats = xs(s.dst), ts(s.dst)
                if s.dst in self.D:
                    d.g_add = self.D[s.dst].data_ptr()
            if s.kind in ('conv', 'raw'):
                conv = _get(net, s.conv)
                d.w = conv.weight.data_ptr()
                d.w_fwd, d.w_bwd = pk[s.conv][0].data_ptr(), pk[s.conv][1].data_ptr()
                d.dw = gv[s.conv + '.weight'].data_ptr()
                mf = net._packed_mfma.get(s.conv)
                d.wm_fwd = mf[0].data_ptr() if mf else None
                d.wm_bwd = mf[1].data_ptr() if mf and mf[1] is not None else None
                b3 = net._packed_b3.get(s.conv)
                d.wb_fwd = b3[0].data_ptr() if b3 else None
                d.wb_bwd = b3[1].data_ptr() if b3 else None
                d.ws, d.ws_bytes, d.ws_defer = net._ws.data_ptr(), net._ws.numel() * 4, 0
            if s.kind == 'conv':
                bn = _get(net, s.norm)
                d.has_bn = 1
                d.gamma, d.beta = bn.weight.data_ptr(), bn.bias.data_ptr()
                d.run_mean, d.run_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                d.x_stats, d.t_stats = xs(s.src), ts(s.src)
                d.t_in = self.T[s.src].data_ptr() if self.has_grad else None
                d.bn_grad = a0 + 8 * (base_bn + bn_off[s.norm])
                d.t_accumulate = 0 if bn_readers[s.src][-1] == i else 1
                d.final_c0, d.final_c1 = consumed.get(s.src, 0), s.cin
                consumed[s.src] = max(consumed.get(s.src, 0), s.cin)
            elif s.kind == 'raw':
                tgt = s.x.get('grad')
                if tgt is not None:                       # plain data gradient: accumulated behind the other consumers'
                    d.t_in = (self.D if tgt == 'D' else self.T)[s.src].data_ptr()
                    d.t_accumulate = 1
            elif s.kind == OP_COPY:
                tgt = s.x.get('grad')
                if tgt is not None:
                    d.t_in = (self.D if tgt == 'D' else self.T)[s.src].data_ptr()
                    d.t_accumulate = 1
                if d.fin_tstats and not d.g_add and _FUSE_COPY_FINALIZE:
                    d.g_fused = 1                     # the copy's backward applies the BatchNorm-backward finalize itself
                if s.x.get('src2'):
                    d.x2, d.x2_ctot = self.X[s.x['src2']].data_ptr(), bufs[s.x['src2']][0]
                    if s.x.get('grad2') == 'D':
                        d.t2 = self.D[s.x['src2']].data_ptr()
            elif s.kind == OP_BIAS_SCALE:
                d.p0 = _get_param(net, s.x['bias']).data_ptr()
                d.p1 = _get_param(net, s.x['scale_p']).data_ptr() if s.x['scale_p'] else None
                d.acc = a0 + 8 * (base_aux + aux_off[i])
            elif s.kind == OP_COUPLING:
                h = s.x['h']
                d.x2, d.x2_ctot = self.X[h].data_ptr(), bufs[h][0]
                d.acc, d.p1 = self._logp_acc, self.glogp.data_ptr()
                if s.x.get('bias'):                      # the folded Conv2dZeros epilogue of the coupling net
                    d.gamma = _get_param(net, s.x['bias']).data_ptr()
                    d.beta = _get_param(net, s.x['scale_p']).data_ptr()
                    d.bn_grad = a0 + 8 * (base_aux + aux_off[i])
                if self.has_grad:
                    d.t2 = self.T[h].data_ptr()
                    d.t_in, d.t_accumulate = self.T[s.src].data_ptr(), 0
            elif s.kind == OP_MIX:
                w_ptr, winv_ptr, acc = self._w_ptr[s.x['index']]
                an = _get(net, mixes[s.x['index']][2])
                d.x2, d.x2_ctot = (winv_ptr if d.flags & FLOW_FORWARD else w_ptr), s.cin
                d.p0, d.p1, d.acc = an.weight.data_ptr(), an.bias.data_ptr(), acc
                if self.has_grad:
                    d.t_in, d.t_accumulate = self.T[s.src].data_ptr(), 0
                if d.flags & MIX_COUPLED:                  # the affine coupling in front of it, in the same launch
                    h = s.x['h']
                    d.h, d.h_ctot = self.X[h].data_ptr(), bufs[h][0]
                    d.acc2, d.cst = self._logp_acc, self.glogp.data_ptr()
                    if s.x.get('bias'):
                        d.gamma = _get_param(net, s.x['bias']).data_ptr()
                        d.beta = _get_param(net, s.x['scale_p']).data_ptr()
                        d.bn_grad = a0 + 8 * (base_aux + aux_off[i])
                    if self.has_grad:
                        d.th = self.T[h].data_ptr()
            elif s.kind == OP_UNSQUEEZE:
                if self.has_grad:
                    d.t_in, d.t_accumulate = self.T[s.src].data_ptr(), 0
            elif s.kind == OP_GAUSS:
                pr = s.x['prior']
                d.x2, d.x2_ctot = self.X[pr].data_ptr(), bufs[pr][0]
                d.acc, d.p1 = self._logp_acc, self.glogp.data_ptr()
                if not (d.flags & FLOW_FORWARD):
                    d.p0 = self.X[s.x['eps']].data_ptr()
                if self.has_grad:
                    d.t2 = self.T[pr].data_ptr()
            self._train_stats.append(d.out_stats)
        self._last = n - 1
        # ---- tables of the end-of-step launches: true BatchNorms, and the {dbias, dscale} of the bias/scale ops
        items = []
        for s in specs:
            if not s.bn:
                continue
            bn = _get(net, s.norm)
            hi, wi = self.buf_hw[s.src]
            it = BnItem()
            it.x_stats, it.bn_grad = xs(s.src), a0 + 8 * (base_bn + bn_off[s.norm])
            it.run_mean, it.run_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            it.dgamma, it.dbeta = gv[s.norm + '.weight'].data_ptr(), gv[s.norm + '.bias'].data_ptr()
            it.num_batches_tracked = bn.num_batches_tracked.data_ptr()
            it.C, it.count = s.cin, B * hi * wi
            items.append(it)
        self.n_bn, self.max_c = len(items), max(it.C for it in items)
        self.bn_table = torch.frombuffer(bytearray(bytes((BnItem * len(items))(*items))), dtype=torch.uint8).to(dev)
        self._dummy = torch.zeros(max(s.cout for s in specs), **f32)      # dscale of a bias-only op lands here
        items = []
        for i, s in enumerate(specs):
            if i not in aux_off:
                continue
            it = BnItem()
            it.bn_grad = a0 + 8 * (base_aux + aux_off[i])
            it.dgamma = gv[s.x['bias']].data_ptr()
            it.dbeta = gv[s.x['scale_p']].data_ptr() if s.x['scale_p'] else self._dummy.data_ptr()
            it.C, it.count = _aux_channels(s), 1
            items.append(it)
        self.n_aux, self.max_aux = len(items), max(it.C for it in items)
        self.aux_table = torch.frombuffer(bytearray(bytes((BnItem * len(items))(*items))), dtype=torch.uint8).to(dev)

    @staticmethod
    def buf_hw_of(res, H, W):
        return H // res[1], W // res[1]

    def _chain_specs(self):
        return self.specs

    # -- launches -------------------------------------------------------------------------------
    def forward(self, x, eps_list, training):
        """generate(): -> (y, logp) views of engine buffers (valid until the next forward of this engine)"""
        feed = {'in': x}
        for k, name in self.meta['eps'].items():
            feed[name] = eps_list[k]
        self.run(feed, training)
        return self.X['out'], self.logp

    def run(self, feed, training, lo=0, hi=None, first=True, last=True):
        """execute descriptors [lo, hi) of the chain; `first` / `last`: with the launches that precede / follow the whole
        chain (a chain is run in pieces only by the data-dependent ActNorm initialisation)"""
        L, st = _lib.lib(), _lib.stream_ptr()
        net = self.net
        n = len(self.descs)
        hi = n if hi is None else hi
        if not first:
            if self.n_mix:
                _lib.check(L.pdes_flow_prepare(self.flow_table.data_ptr(), self.n_mix, 1, st), 'pdes_flow_prepare')
            self._launch(lo, hi, st)
            if last:
                self._finish(training, st)
            return
        for name, t in feed.items():
            if t is not None and t.data_ptr() != self.X[name].data_ptr():
                self.X[name].copy_(t)
        ev = 0 if training else 1
        for d, os_ in zip(self.descs, self._train_stats):
            d.eval_mode = ev
            d.out_stats = os_ if training else None
        self.arena.zero_()
        net._pack_weights()
        if self.n_mix:
            need_inv = 0 if (net.LUdecompose and self.has_grad) else 1       # the y -> z direction applies the inverse matrices
            _lib.check(L.pdes_flow_prepare(self.flow_table.data_ptr(), self.n_mix, need_inv, st), 'pdes_flow_prepare')
        self._launch(lo, hi, st)
        if last:
            self._finish(training, st)

    def _launch(self, lo, hi, st):
        if hi > lo:
            first = ctypes.byref(self.descs, lo * ctypes.sizeof(ConvDesc))
            _lib.check(_lib.lib().pdes_conv_forward(self.ctx, first, hi - lo, st), 'pdes_conv_forward')

    def _finish(self, training, st):
        L = _lib.lib()
        _lib.check(L.pdes_flow_logp(self._logp_acc, self.logdet.data_ptr(), self.n_mix, self.logp.data_ptr(), self.B,
                                    self.nrep, self.rep_stride, st), 'pdes_flow_logp')
        if training:
            _lib.check(L.pdes_bn_update_running(self.bn_table.data_ptr(), self.n_bn, self.max_c, ctypes.c_float(0.1),
                                                self.nrep, self.rep_stride, st), 'pdes_bn_update_running')

    def backward(self, grad_y, grad_logp):
        """parameter gradients are ACCUMULATED into net._gscratch.  grad_y: dL/dy (B, C, H, W) contiguous; grad_logp: dL/dlogp
        (B,) or None"""
        L, st = _lib.lib(), _lib.stream_ptr()
        n = len(self.descs)
        if not hasattr(self, '_reduce_n'):
            self._plan_wgrad_scratch()
        self.descs[self._last].g = grad_y.data_ptr()
        self.descs[self._last].g_ctot, self.descs[self._last].g_coff = grad_y.shape[1], 0
        if grad_logp is None:
            self.glogp.zero_()
        elif grad_logp.data_ptr() != self.glogp.data_ptr():
            self.glogp.copy_(grad_logp)
        self.Dflat.zero_()
        side = side_b = None
        if self.net.wgrad_stream and not torch.cuda.is_current_stream_capturing():
            side = ctypes.c_void_p(self._side_stream().cuda_stream)
            if self.net.wgrad_streams == 2:
                side_b = ctypes.c_void_p(self._side_stream('b').cuda_stream)
        rt = self._reduce_table.data_ptr() if self._reduce_n else None
        _lib.check(L.pdes_backward2(self.ctx, self.descs, n, st, side, side_b, rt, self._reduce_index, None), 'pdes_backward2')
        _lib.check(L.pdes_bn_param_grads(self.bn_table.data_ptr(), self.n_bn, self.max_c, self.nrep, self.rep_stride, st),
                   'pdes_bn_param_grads')
        _lib.check(L.pdes_bn_param_grads(self.aux_table.data_ptr(), self.n_aux, self.max_aux, self.nrep, self.rep_stride, st),
                   'pdes_bn_param_grads (bias / scale)')
        if self.n_mix:
            _lib.check(L.pdes_flow_param_grads(self.flow_table.data_ptr(), self.n_mix, self.glogp.data_ptr(), self.B, self.nrep,
                                               self.rep_stride, st), 'pdes_flow_param_grads')


def _aux_channels(s):
    """channels of the {dbias, dscale} table of a bias/scale op, or of the epilogue folded into a coupling (its 2 n2 shift /
    scale channels)"""
    return 2 * (s.cin // 2) if s.kind in (OP_COUPLING, OP_MIX) else s.cout


def _get_param(net, path):
    mod, _, leaf = path.rpartition('.')
    return getattr(_get(net, mod), leaf)


class _GenerateFn(torch.autograd.Function):
    """generate() as ONE autograd node: (x, eps...) -> (y, log p(y|x)); backward returns the parameter gradients"""

    @staticmethod
    def forward(ctx, x, net, grad_on, n_eps, *rest):
        eps, params = rest[:n_eps], rest[n_eps:]
        eng = net._acquire(x)
        ctx.net, ctx.eng, ctx.n_eps = net, eng, n_eps
        ctx.trained = net.training
        ctx.pver = sum(p._version for p in params)
        with _lib.device_guard(x.device):
            y, logp = eng.forward(x, eps, net.training)
            y, logp = y.clone(), logp.clone()
        ctx.lease = _Lease(eng) if (grad_on and any(p.requires_grad for p in params)) else None
        return y, logp

    @staticmethod
    def backward(ctx, gy, glogp):
        net, eng, lease = ctx.net, ctx.eng, ctx.lease
        if not ctx.trained:
            raise RuntimeError('backward through an eval-mode generate() is not implemented (the reference evaluates under '
                               'torch.no_grad())')
        if lease is None or lease.eng is not eng:
            raise RuntimeError('backward through this generate() a second time: its activations have been released '
                               '(retain_graph is not supported by the HIP flow)')
        if sum(p._version for p in net._params) != ctx.pver:
            lease.release()
            raise RuntimeError('a parameter of the network was modified in place between generate() and backward()')
        dev = eng.dev
        with _lib.device_guard(dev):
            if gy is None:
                gy = torch.zeros_like(eng.X['out'])
            net._pack_weights()
            if eng.n_mix:        # another generate() may have rebuilt the matrices; the live parameters are unchanged
                _lib.check(_lib.lib().pdes_flow_prepare(eng.flow_table.data_ptr(), eng.n_mix, 0 if net.LUdecompose else 1,
                                                        _lib.stream_ptr()), 'pdes_flow_prepare')
            net._gscratch.zero_()
            net._grad_dirty = True
            eng.backward(gy.contiguous(), None if glogp is None else glogp.contiguous())
            fresh = net._gscratch.clone()
        lease.release()
        grads = [fresh[off:off + p.numel()].view(p.shape) for p, off in zip(net._params, net._offsets)]
        return (None, None, None, None) + (None,) * ctx.n_eps + tuple(grads)


class MultiScaleCondGlow(_HipNet):
    """Multiscale conditional Glow (reference glow_msc.py:672-968); see the module docstring"""
    _prefix = ''

    @property
    def _root(self):
        return self

    def __init__(self, img_size, x_channels, y_channels, enc_blocks, flow_blocks, flow_coupling='dense', squeeze_factor=2,
                 LUdecompose=False, train_sampling=True, data_init=False):
        super().__init__()
        if isinstance(img_size, int):
            self.img_size = [img_size, img_size]
        else:
            if len(img_size) != 2:
                raise ValueError('img_size: an int or (height, width)')
            self.img_size = list(img_size)
        enc_blocks, flow_blocks = [int(b) for b in enc_blocks], [int(b) for b in flow_blocks]
        if len(enc_blocks) != len(flow_blocks) or len(flow_blocks) < 2:
            raise ValueError('enc_blocks and flow_blocks must have the same length >= 2 (one conditioning scale per flow level)')
        if flow_coupling != 'dense':
            raise ValueError("flow_coupling: only 'dense' (the reference's default and the one its scripts use) is built")
        if squeeze_factor != 2:
            raise ValueError('squeeze_factor: only 2 is built (every reference script passes 2)')
        if not train_sampling:
            raise ValueError('train_sampling=False (maximum-likelihood training through y -> z) is not built: the y -> z '
                             'direction is inference-only here')
        if x_channels != 1:
            raise ValueError('x_channels must be 1 (the reference\'s own input encoder is only consistent for 1: '
                             'glow_msc.py:34-36 against :495)')
        self.data_init, self.data_initialized = data_init, False
        self.x_channels, self.y_channels = x_channels, y_channels
        self.enc_blocks, self.flow_blocks, self.factor = enc_blocks, flow_blocks, squeeze_factor
        self.LUdecompose, self.train_sampling = bool(LUdecompose), True
        specs, bufs, meta = _plan_glow(y_channels, enc_blocks, flow_blocks, self.LUdecompose)
        self._specs, self._bufs, self._meta = specs, bufs, meta
        self._build(meta)
        self.drop_rate, self._out_act = 0.0, None
        self._flat, self._engines, self._side_streams, self._grad_dirty = None, {}, {}, False
        self.wgrad_stream = os.environ.get('PDES_WGRAD_STREAM', '1') != '0'
        self.wgrad_streams = 1 if os.environ.get('PDES_WGRAD_STREAMS', '2') == '1' else 2

    # -- module tree: the reference's names, creation order and initial values -------------------------------------
    def _build(self, meta, growth=16, init_features=48):
        enc = nn.Sequential()
        c = init_features
        for i, n_layers in enumerate(self.enc_blocks, 1):
            blk = nn.Sequential()
            if i == 1:
                blk.add_module('in_conv', nn.Conv2d(self.x_channels, init_features - 1, kernel_size=3, stride=1, padding=1))
                n_layers -= 1
            for j in range(1, n_layers + 1):
                blk.add_module(f'denselayer{j}', _dense_layer(c, growth))
                c += growth
            enc.add_module(f'dense_block{i}', blk)
            if i < len(self.enc_blocks):
                t = nn.Sequential()
                t.add_module('norm1', nn.BatchNorm2d(c))
                if i > 1:
                    t.add_module('conv1', nn.Conv2d(c, c // 2, kernel_size=1, stride=1, padding=0, bias=False))
                    t.add_module('norm2', nn.BatchNorm2d(c // 2))
                    t.add_module('conv2', nn.Conv2d(c // 2, c // 2, kernel_size=3, stride=2, padding=1, bias=False))
                else:
                    t.add_module('conv1', nn.Conv2d(c, c // 2, kernel_size=3, stride=2, padding=1, bias=False))
                enc.add_module(f'trans_down{i}', t)
                c //= 2
        C, L = meta['C'], meta['L']
        enc.add_module('top_latent', Conv2dZeros(c, 2 * C[L]))
        self.encoder = enc
        # the reference reads the encoder's feature sizes by pushing one random image through it (glow_msc.py:713-714);
        # the sizes are known arithmetically here, the draw keeps torch's RNG stream aligned with the reference's.  (Its
        # side effect there -- the encoder's BatchNorm running statistics start from that image's -- is not reproduced:
        # buffers come from load_state_dict or from training.)
        torch.randn(1, self.x_channels, self.img_size[0], self.img_size[1])
        flow = nn.Sequential()
        conv1x1 = InvertibleConv1x1LU if self.LUdecompose else InvertibleConv1x1
        for i, nl in enumerate(self.flow_blocks, 1):
            cond_c = meta['conds'][i - 1][1]
            Ci = C[i]
            layers = nn.Sequential()
            for j in range(1, nl + 1):
                lay = nn.Module()
                if not (i == 1 and j == 1):
                    lay.norm = ActNorm(Ci)
                    lay.conv1x1 = conv1x1(Ci)
                n2 = Ci // 2
                cin = Ci - n2 + cond_c
                net = nn.Sequential()
                for k in range(1, 4):
                    net.add_module(f'denselayer{k}', _dense_layer(cin + (k - 1) * growth, growth))
                red = nn.Sequential()
                red.add_module('norm1', nn.BatchNorm2d(cin + 3 * growth))
                red.add_module('conv_zero', Conv2dZeros(cin + 3 * growth, 2 * n2))
                net.add_module('reduce', red)
                lay.coupling = nn.Module()
                lay.coupling.coupling_nn = net
                layers.add_module(f'revlayer{j}', lay)
            blk = nn.Module()
            blk.revlayers = layers
            if 1 < i < L:
                blk.split = nn.Module()
                blk.split.latent_encoder = nn.Module()
                blk.split.latent_encoder.conv2d = Conv2dZeros(Ci // 2, Ci)
            flow.add_module(f'revblock{i}', blk)
        self.flow = flow
        if self.data_init:
            for m in self.modules():
                if isinstance(m, ActNorm):
                    m.data_init = True

    def _new_engine(self, key):
        return _GlowEngine(self, *key)

    def _flatten(self, device):
        super()._flatten(device)
        self._fwd_engines = {}                          # the y -> z engines hold pointers into the OLD flat buffer / images
        for m in self.modules():                       # p, sign_s, masks: the kernels read them through device pointers
            for name, buf in m._buffers.items():
                if buf is not None and buf.device != device:
                    m._buffers[name] = buf.to(device)

    @property
    def device(self):
        return next(self.parameters()).device

    # -- the reference's API ---------------------------------------------------------------------------------------
    def _z_shapes(self):
        """shapes of the noise tensors: split latents bottom-up, the top latent last (glow_msc.py:878-896)"""
        C, L = self._meta['C'], self._meta['L']
        hw = list(self.img_size)
        out = []
        for i in range(2, L):
            hw = [v // 2 for v in hw]
            out.append((C[i] // 2, *hw))
        hw = [v // 2 for v in hw]
        out.append((C[L], *hw))
        return out

    def _check_input(self, x):
        _lib.require_cuda(x)
        if x.dim() != 4 or x.shape[1] != self.x_channels:
            raise ValueError(f'expected input (B, {self.x_channels}, H, W); got {tuple(x.shape)}')
        if x.dtype != torch.float32:
            raise RuntimeError('the HIP kernels compute in fp32: pass an fp32 input')
        return x.contiguous()

    def _noise(self, x, eps_list):
        shapes = self._z_shapes_for(x)
        if eps_list is None:
            eps_list = [None] * len(shapes)
        if len(eps_list) != len(shapes):
            raise AssertionError('The specified noise must have the same size as the latent variables')
        out = []
        for e, s in zip(eps_list, shapes):
            if e is None:
                e = torch.randn((x.shape[0],) + s, device=x.device, dtype=torch.float32)
            elif tuple(e.shape) != (x.shape[0],) + s:
                raise ValueError(f'noise of shape {tuple(e.shape)}; the latent is {(x.shape[0],) + s}')
            out.append(e.to(device=x.device, dtype=torch.float32).contiguous())
        return out

    def _z_shapes_for(self, x):
        C, L = self._meta['C'], self._meta['L']
        h, w = x.shape[2], x.shape[3]
        out = [(C[i] // 2, h >> (i - 1), w >> (i - 1)) for i in range(2, L)]
        out.append((C[L], h >> (L - 1), w >> (L - 1)))
        return out

    def generate(self, x, eps_list=None):
        """one sample y ~ p(y|x) per input and log p(y|x) (glow_msc.py:783-829); differentiable wrt the parameters"""
        x = self._check_input(x)
        eps = self._noise(x, eps_list)
        self._engine(x)                                    # flattens the parameters before autograd sees them
        return _GenerateFn.apply(x, self, torch.is_grad_enabled(), len(eps), *eps, *self._params)

    def approx_pred_mean(self, x):
        """every Gaussian replaced by its mean (glow_msc.py:832-838)"""
        return self.generate(x, eps_list=self.create_zero_noise(x.shape[0]))

    def sample(self, x, n_samples, eps_list=None, temperature=None):
        """(n_samples, B, C, H, W) samples of p(y|x); the temperature scales the split latents' noise, not the top
        latent's (glow_msc.py:841-876)"""
        if temperature is None:
            temperature = 0.7
        if eps_list is None:
            eps_list = self.create_fixed_noise(n_samples, batch_size=x.shape[0])
        elif n_samples != eps_list[-1].shape[0] or x.shape[0] != eps_list[-1].shape[1]:
            raise AssertionError('eps_list: (n_samples, B, ...) tensors')
        ys = []
        # the reference runs the conditioning encoder ONCE for all samples (glow_msc.py:860-862); here every sample is a
        # full generate(), so in train() mode only the first one may move the BatchNorm running statistics
        bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)] if (self.training and n_samples > 1) else []
        snap = None
        with torch.no_grad():
            for i in range(n_samples):
                el = [e[i] * temperature for e in eps_list[:-1]] + [eps_list[-1][i]]
                ys.append(self.generate(x, el)[0])
                if i == 0 and bns:
                    snap = [(m.running_mean.clone(), m.running_var.clone(), m.num_batches_tracked.clone()) for m in bns]
            if snap is not None:
                for m, (rm, rv, nb) in zip(bns, snap):
                    m.running_mean.copy_(rm)
                    m.running_var.copy_(rv)
                    m.num_batches_tracked.copy_(nb)
        return torch.stack(ys, 0)

    def predict(self, x_test, n_samples=20, temperature=1.0):
        """predictive mean and variance from samples (glow_msc.py:919-932)"""
        pred = self.sample(x_test, n_samples, temperature=temperature)
        return pred.mean(0), pred.var(0)

    def propagate(self, mc_loader, n_samples=20, temperature=1.0, var_samples=10):
        """uncertainty propagation over a loader of inputs (glow_msc.py:934-968): E[Y] = E_X E[Y|X],
        Var[Y] = E_X Var(Y|X) + Var_X E[Y|X], each estimated `var_samples` times -> (mean and variance of the estimate of
        E[Y], mean and variance of the estimate of Var[Y]), every one (C, H, W)"""
        ey = eyy = None
        for i in range(var_samples):
            print(f'propagating for the {i}-th time...')
            for batch in mc_loader:
                x_mc = batch[0].to(self.device)
                y = self.sample(x_mc, n_samples=n_samples, temperature=temperature)
                if ey is None:
                    ey = torch.zeros((var_samples,) + tuple(y.shape[2:]), device=self.device)
                    eyy = torch.zeros_like(ey)
                ey[i] += y.mean(0).mean(0)
                eyy[i] += y.pow(2).mean(0).mean(0)
        ey /= len(mc_loader)
        eyy /= len(mc_loader)
        vy = eyy - ey ** 2
        return ey.mean(0), ey.var(0), vy.mean(0), vy.var(0)

    def create_fixed_noise(self, n_samples, batch_size=1):
        return [torch.randn(n_samples, batch_size, *s, device=self.device) for s in self._z_shapes()]

    def create_zero_noise(self, batch_size):
        return [torch.zeros(batch_size, *s, device=self.device) for s in self._z_shapes()]

    def init_actnorm(self):
        for m in self.modules():
            if isinstance(m, ActNorm):
                m.data_initialized = True
        self.data_initialized = True

    def reset_parameters(self, verbose=False):
        raise RuntimeError('MultiScaleCondGlow: construct a new model instead (the initial values depend on the numpy and '
                           'torch seeds at construction, as in the reference)')

    # -- y -> z (inference only) ----------------------------------------------------------------------------------
    def _forward_engine(self, x):
        if getattr(self, '_fwd_plan', None) is None:
            index = {(m[4], m[5]): k for k, m in enumerate(self._meta['mix'])}
            self._fwd_plan = _plan_glow_forward(self.y_channels, self.enc_blocks, self.flow_blocks, self.LUdecompose, index)
            self._fwd_engines = {}
        key = (x.device, x.shape[0], x.shape[2], x.shape[3])
        eng = self._fwd_engines.get(key)
        if eng is None:
            eng = self._fwd_engines[key] = _GlowEngine(self, *key[1:], plan=self._fwd_plan)
        return eng

    def forward(self, y, x, return_eps=False):
        """y -> z: (z_top, log p(y|x), [eps per latent] or None) (glow_msc.py:746-780).  Inference only: the outputs
        carry no autograd graph (the reference trains through this direction only with train_sampling=False).
        With data_init=True the first call initialises every ActNorm from the statistics of its input (glow_msc.py:70-83),
        layer by layer."""
        x = self._check_input(x)
        _lib.require_cuda(y)
        if y.dim() != 4 or y.shape[1] != self.y_channels or y.shape[0] != x.shape[0] or y.shape[2:] != x.shape[2:]:
            raise ValueError(f'expected y of shape ({x.shape[0]}, {self.y_channels}, {x.shape[2]}, {x.shape[3]}); got {tuple(y.shape)}')
        y = y.to(torch.float32).contiguous()
        self._engine(x)                                   # parameters flat on this device
        with torch.no_grad(), _lib.device_guard(x.device):
            eng = self._forward_engine(x)
            feed = {'in': x, 'yin': y}
            if self.data_init and not self.data_initialized:
                self._data_init(eng, feed)
            else:
                eng.run(feed, self.training)
            z = eng.X[eng.meta['z']].clone()
            logp = eng.logp.clone()
            eps = [eng.X[eng.meta['eps'][k]].clone() for k in sorted(eng.meta['eps'])] if return_eps else None
        return z, logp, eps

    def _data_init(self, eng, feed):
        """ActNorm._init_parameters (glow_msc.py:70-83) for every ActNorm, in the order the data reaches them: the chain is
        run up to each invertible layer, whose input then gives bias = -mean / std, weight = 1 / std (std unbiased + 1e-6)"""
        lo, first = 0, True
        for i, s in enumerate(eng.specs):
            if s.kind != OP_MIX:
                continue
            eng.run(feed, self.training, lo, i, first=first, last=False)
            lo, first = i, False
            act = eng.X[s.src]
            flat = act.transpose(0, 1).reshape(act.shape[1], -1)
            mean, std = flat.mean(1), flat.std(1) + 1e-6
            an = _get(self, self._meta['mix'][s.x['index']][2])
            an.bias.data.copy_((-(mean / std)).view(-1, 1, 1))
            an.weight.data.copy_((1. / std).view(-1, 1, 1))
        eng.run(feed, self.training, lo, None, first=first, last=True)
        self.init_actnorm()
