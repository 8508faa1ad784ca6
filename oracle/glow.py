"""Oracle (test infrastructure): functional multiscale conditional Glow on PyTorch-CPU ops.

Restates, as pure functions over a reference-format ``state_dict`` (same key names, so one checkpoint drives the
reference, this oracle and the HIP path), what the reference builds in models/glow_msc.py:

  * input encoder  x -> conditioning features at every scale + the top latent's Gaussian   (glow_msc.py:474-546, :27-47)
  * Conv2dZeros    conv3x3 + bias, times exp(3 scale)                                       (glow_msc.py:237-252)
  * ActNorm, invertible 1x1 convolution (plain and LU-parameterised), affine coupling with the dense coupling net
                                                                                            (glow_msc.py:50-96, :99-233, :274-345)
  * squeeze / unsqueeze (quadrant layout, not the interleaved one), split prior              (glow_msc.py:401-432, :435-471, :550-587)
  * generate: z ~ p(z|x) -> y with log p(y|x)  (the TRAINING path of the reverse-KL loss)   (glow_msc.py:783-829)
  * forward:  y -> z with log p(y|x)                                                        (glow_msc.py:746-780)
  * the reverse-KL training loss  beta * loss_pde + E[log p(y|x)] / ln2 / n_pixels          (train_cglow_reverse_kl.py:250-262)

The reference clamps the log-stddev of a split prior IN PLACE on a chunk view (glow_msc.py:438), which PyTorch >= 1.5
refuses under autograd; the out-of-place clamp used here has the same values and the same gradient (zero outside
[-10, log 5]).  The top latent's log-stddev is detached (``.data``, glow_msc.py:524): no gradient reaches it.

Pinned by tests/golden/G18 (small net, every tensor) and G19 (default net, seeded init).
"""
import math

import torch
import torch.nn.functional as F

from . import darcy

LOG2PI = float(math.log(2 * math.pi))
LSD_MIN, LSD_MAX = -10.0, math.log(5.0)


def _bn_relu(sd, p, x, training):
    y = F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'],
                     training=training, momentum=0.1, eps=1e-5)
    if training and (p + '.num_batches_tracked') in sd:
        sd[p + '.num_batches_tracked'] += 1
    return torch.relu(y)


def _dense_layers(sd, prefix, x, first, count, training):
    for j in range(first, first + count):
        p = f'{prefix}.denselayer{j}'
        x = torch.cat([x, F.conv2d(_bn_relu(sd, p + '.norm1', x, training), sd[p + '.conv1.weight'], padding=1)], 1)
    return x


def _count(sd, prefix, pattern):
    n = 0
    while (prefix + pattern.format(n + 1)) in sd:
        n += 1
    return n


def conv_zeros(sd, p, x):
    """Conv2dZeros (glow_msc.py:237-252)"""
    return F.conv2d(x, sd[p + '.conv.weight'], sd[p + '.conv.bias'], padding=1) * torch.exp(sd[p + '.scale'] * 3)


def encoder(sd, x, training):
    """-> ([features per scale], top mean, top log-stddev (detached, clamped)); glow_msc.py:474-546"""
    conds = []
    nb = 1
    while f'encoder.dense_block{nb + 1}.denselayer1.norm1.weight' in sd:
        nb += 1
    for i in range(1, nb + 1):
        blk = f'encoder.dense_block{i}'
        if i == 1:
            x = torch.cat([x, F.conv2d(x, sd[blk + '.in_conv.weight'], sd[blk + '.in_conv.bias'], padding=1)], 1)
        x = _dense_layers(sd, blk, x, 1, _count(sd, blk, '.denselayer{}.norm1.weight'), training)
        conds.append(x)
        if i < nb:
            t = f'encoder.trans_down{i}'
            z = _bn_relu(sd, t + '.norm1', x, training)
            if (t + '.conv2.weight') in sd:
                z = _bn_relu(sd, t + '.norm2', F.conv2d(z, sd[t + '.conv1.weight']), training)
                x = F.conv2d(z, sd[t + '.conv2.weight'], stride=2, padding=1)
            else:
                x = F.conv2d(z, sd[t + '.conv1.weight'], stride=2, padding=1)
    top = conv_zeros(sd, 'encoder.top_latent', x)
    mean, lsd = top.chunk(2, 1)
    return conds, mean, lsd.detach().clamp(LSD_MIN, LSD_MAX)


def gauss_log_prob(x, mean, lsd):
    like = -0.5 * (LOG2PI + lsd * 2. + (x - mean) ** 2 / (lsd * 2.).exp())       # glow_msc.py:442-450
    return like.reshape(x.shape[0], -1).sum(1)


def unsqueeze2(x):
    """Squeeze.reverse, factor 2 (glow_msc.py:422-432): channel 4c + 2i + j becomes QUADRANT (i, j) of channel c"""
    B, C, H, W = x.shape
    return x.reshape(B, C // 4, 2, 2, H, W).transpose(3, 4).reshape(B, C // 4, 2 * H, 2 * W)


def squeeze2(x):
    B, C, H, W = x.shape
    return x.reshape(B, C, 2, H // 2, 2, W // 2).transpose(3, 4).reshape(B, C * 4, H // 2, W // 2)


def conv1x1_weight(sd, p, dtype=None):
    """the matrix of the z -> y direction (train_sampling=True: no inverse on this path) and log|det| per pixel"""
    if (p + '.weight') in sd:                                   # InvertibleConv1x1 (glow_msc.py:99-157)
        W = sd[p + '.weight']
        det = torch.det(W.to(torch.float64)).to(W.dtype)
        return W, det.abs().log()
    l = sd[p + '.l'] * sd[p + '.l_mask'] + sd[p + '.eye']      # InvertibleConv1x1LU (glow_msc.py:161-233)
    u = sd[p + '.u'] * sd[p + '.u_mask'] + torch.diag(sd[p + '.log_s'].exp() * sd[p + '.sign_s'])
    return sd[p + '.p'] @ (l @ u), sd[p + '.log_s'].sum()


def coupling_net(sd, p, x, training):
    """_DenseCoupling (glow_msc.py:274-293)"""
    x = _dense_layers(sd, p, x, 1, _count(sd, p, '.denselayer{}.norm1.weight'), training)
    return conv_zeros(sd, p + '.reduce.conv_zero', _bn_relu(sd, p + '.reduce.norm1', x, training))


def coupling_reverse(sd, p, y, cond, training):
    y1, y2 = y.chunk(2, 1)
    h = coupling_net(sd, p + '.coupling_nn', torch.cat([y1, cond], 1), training)
    shift, scale = h[:, 0::2], torch.sigmoid(h[:, 1::2] + 2.)
    return torch.cat([y1, y2 / scale - shift], 1), scale.log().reshape(y.shape[0], -1).sum(1)


def coupling_forward(sd, p, x, cond, training):
    x1, x2 = x.chunk(2, 1)
    h = coupling_net(sd, p + '.coupling_nn', torch.cat([x1, cond], 1), training)
    shift, scale = h[:, 0::2], torch.sigmoid(h[:, 1::2] + 2.)
    return torch.cat([x1, (x2 + shift) * scale], 1), scale.log().reshape(x.shape[0], -1).sum(1)


def revlayer_reverse(sd, p, y, cond, training):
    """RevLayer.reverse / FirstRevLayer.reverse (glow_msc.py:349-397)"""
    y, logdet = coupling_reverse(sd, p + '.coupling', y, cond, training)
    if (p + '.norm.weight') in sd:
        hw = y.shape[2] * y.shape[3]
        W, ld = conv1x1_weight(sd, p + '.conv1x1')
        y = F.conv2d(y, W.reshape(*W.shape, 1, 1))
        logdet = logdet - ld * hw
        w, b = sd[p + '.norm.weight'], sd[p + '.norm.bias']
        y = (y - b) / w
        logdet = logdet + w.abs().log().sum() * hw
    return y, logdet


def revlayer_forward(sd, p, x, cond, training):
    logdet = 0.
    if (p + '.norm.weight') in sd:
        hw = x.shape[2] * x.shape[3]
        w, b = sd[p + '.norm.weight'], sd[p + '.norm.bias']
        x = w * x + b
        logdet = logdet + w.abs().log().sum() * hw
        W, ld = conv1x1_weight(sd, p + '.conv1x1')
        Wi = torch.inverse(W.double()).to(W.dtype)
        x = F.conv2d(x, Wi.reshape(*Wi.shape, 1, 1))
        logdet = logdet - ld * hw
    x, ld3 = coupling_forward(sd, p + '.coupling', x, cond, training)
    return x, logdet + ld3


def n_flow_blocks(sd):
    return _count(sd, 'flow.revblock', '{}.revlayers.revlayer1.coupling.coupling_nn.denselayer1.norm1.weight')


def latent_shapes(sd, y_channels, imsize):
    """_z_shapes (glow_msc.py:878-896): shapes of the noise tensors, split latents first, the top latent last"""
    nb = n_flow_blocks(sd)
    c, s, out = y_channels, imsize, []
    for _ in range(nb - 2):
        s //= 2
        c = c * 4 // 2
        out.append((c, s, s))
    out.append((c * 4, s // 2, s // 2))
    return out


def generate(sd, x, eps_list, training=True):
    """MultiScaleCondGlow.generate (glow_msc.py:783-829): y (B, 3, H, W) and log p(y|x) (B,).
    eps_list: one noise tensor per latent (latent_shapes order: split priors bottom-up, top latent last)"""
    nb = n_flow_blocks(sd)
    conds, mean, lsd = encoder(sd, x, training)
    z = mean + lsd.exp() * eps_list[-1]
    logp = gauss_log_prob(z, mean, lsd)
    for i in range(nb, 0, -1):
        blk = f'flow.revblock{i}'
        cond = conds[i - 1]
        if 1 < i < nb:                                           # Split.reverse (glow_msc.py:575-587)
            pm, pl = conv_zeros(sd, blk + '.split.latent_encoder.conv2d', z).chunk(2, 1)
            pl = pl.clamp(LSD_MIN, LSD_MAX)
            z2 = pm + pl.exp() * eps_list[i - 2]
            logp = logp + gauss_log_prob(z2, pm, pl)
            z = torch.cat([z, z2], 1)
        nl = _count(sd, blk, '.revlayers.revlayer{}.coupling.coupling_nn.denselayer1.norm1.weight')
        for j in range(nl, 0, -1):
            z, ld = revlayer_reverse(sd, f'{blk}.revlayers.revlayer{j}', z, cond, training)
            logp = logp + ld
        if i > 1:
            z = unsqueeze2(z)
    return z, logp


def forward(sd, y, x, training=False):
    """MultiScaleCondGlow.forward (glow_msc.py:746-780): y -> (z_top, log p(y|x), [eps per latent])"""
    nb = n_flow_blocks(sd)
    conds, mean, lsd = encoder(sd, x, training)
    logp, eps = 0., []
    for i in range(1, nb + 1):
        blk = f'flow.revblock{i}'
        if i > 1:
            y = squeeze2(y)
        nl = _count(sd, blk, '.revlayers.revlayer{}.coupling.coupling_nn.denselayer1.norm1.weight')
        for j in range(1, nl + 1):
            y, ld = revlayer_forward(sd, f'{blk}.revlayers.revlayer{j}', y, conds[i - 1], training)
            logp = logp + ld
        if 1 < i < nb:                                           # Split.forward (glow_msc.py:561-573)
            y, z2 = y.chunk(2, 1)
            pm, pl = conv_zeros(sd, blk + '.split.latent_encoder.conv2d', y).chunk(2, 1)
            pl = pl.clamp(LSD_MIN, LSD_MAX)
            logp = logp + gauss_log_prob(z2, pm, pl)
            eps.append((z2 - pm) / pl.exp())
    logp = logp + gauss_log_prob(y, mean, lsd)
    eps.append((y - mean) / lsd.exp())
    return y, logp, eps


def reverse_kl_loss(sd, x, eps_list, beta, weight_bound, training=True):
    """train_cglow_reverse_kl.py:250-262 -> (loss, loss_pde, neg_entropy, y)"""
    y, logp = generate(sd, x, eps_list, training)
    loss_pde = darcy.mixed_residual_loss(x, y, weight_bound)[0]
    neg_entropy = logp.mean() / math.log(2.) / (y.shape[1] * y.shape[2] * y.shape[3])
    return loss_pde * beta + neg_entropy, loss_pde, neg_entropy, y


def param_keys(sd):
    """keys of the trainable tensors (the reference's named_parameters order = state_dict order minus buffers)"""
    skip = ('running_mean', 'running_var', 'num_batches_tracked', '.p', '.sign_s', '.l_mask', '.u_mask', '.eye')
    return [k for k in sd if not k.endswith(skip)]
