"""Oracle (test infrastructure): functional DenseED / Decoder forward on PyTorch-CPU ops.

Restates the network the reference builds in models/codec.py:
  * dense layer  BN -> ReLU -> conv3x3(C->growth, p1, no bias) -> cat   (codec.py:43-75, :78-86)
  * transition   BN,ReLU,conv1x1(C->C/2), BN,ReLU,[nearest x2,] conv3x3 (s2 when down)  (codec.py:89-160)
  * last decoding BN,ReLU,conv3x3(C->C/2), BN,ReLU,up x2,conv3x3(C/2->C/4), BN,ReLU,conv5x5(C/4->out)
                                                                        (codec.py:163-188)
  * DenseED stage order / channel bookkeeping                           (codec.py:229-293)
  * Decoder (config 5)                                                  (codec.py:321-363)

Weights come from a reference-format ``state_dict`` (same 163 key names), so the same
checkpoint drives the reference, this oracle and the HIP path.  BatchNorm is
``F.batch_norm`` (train mode: batch statistics, biased variance for the normalisation,
unbiased into running_var, momentum 0.1, eps 1e-5).  Pinned by tests/golden/G5,G6,G10.
"""
import torch
import torch.nn.functional as F


def _bn_relu(sd, prefix, x, training):
    rm, rv = sd[prefix + '.running_mean'], sd[prefix + '.running_var']
    y = F.batch_norm(x, rm, rv, sd[prefix + '.weight'], sd[prefix + '.bias'],
                     training=training, momentum=0.1, eps=1e-5)
    if training and (prefix + '.num_batches_tracked') in sd:
        sd[prefix + '.num_batches_tracked'] += 1
    return torch.relu(y)


def _up(x, upsample):
    if upsample == 'nearest':
        return F.interpolate(x, scale_factor=2.0, mode='nearest')
    return F.interpolate(x, scale_factor=2.0, mode='bilinear', align_corners=True)


def _drop(y, masks, training):
    """nn.Dropout2d after a convolution (codec.py:70-71, :111-120, :134-150, :172-173): `masks` is an iterator of
    (B, C) arrays holding 0 or 1/(1-p), consumed in the reference's call order; None = drop_rate 0"""
    if masks is None or not training:
        return y
    m = torch.as_tensor(next(masks), dtype=y.dtype)
    return y * m[:, :, None, None]


def _dense_block(sd, name, x, n_layers, training, masks=None):
    for j in range(1, n_layers + 1):
        p = f'{name}.denselayer{j}'
        z = _bn_relu(sd, p + '.norm1', x, training)
        if (p + '.conv2.weight') in sd:        # bottleneck layer (codec.py:55-62): 1x1 reduction, then norm2 + 3x3
            z = _bn_relu(sd, p + '.norm2', F.conv2d(z, sd[p + '.conv1.weight']), training)
            y = F.conv2d(z, sd[p + '.conv2.weight'], padding=1)
        else:
            y = F.conv2d(z, sd[p + '.conv1.weight'], padding=1)
        x = torch.cat([x, _drop(y, masks, training)], 1)
    return x


def _transition(sd, name, x, down, training, upsample, masks=None):
    z = _bn_relu(sd, name + '.norm1', x, training)
    x = _drop(F.conv2d(z, sd[name + '.conv1.weight']), masks, training)
    z = _bn_relu(sd, name + '.norm2', x, training)
    if down:
        return _drop(F.conv2d(z, sd[name + '.conv2.weight'], stride=2, padding=1), masks, training)
    return _drop(F.conv2d(_up(z, upsample), sd[name + '.conv2.weight'], padding=1), masks, training)


def _last(sd, name, x, training, upsample, masks=None):
    z = _bn_relu(sd, name + '.norm1', x, training)
    x = _drop(F.conv2d(z, sd[name + '.conv1.weight'], padding=1), masks, training)
    z = _bn_relu(sd, name + '.norm2', x, training)
    x = F.conv2d(_up(z, upsample), sd[name + '.conv2.weight'], padding=1)
    z = _bn_relu(sd, name + '.norm3', x, training)
    return F.conv2d(z, sd[name + '.conv3.weight'], padding=2)


def densed_forward(sd, x, blocks, imsize=64, training=True, upsample='nearest', dropout_masks=None):
    """DenseED.forward (codec.py:295-296). `sd` maps 'features.*' keys to tensors; running
    statistics in `sd` are updated in place when training (like nn.BatchNorm2d).  dropout_masks: list of (B, C) channel
    masks (0 or 1/(1-p)) in the reference's Dropout2d call order (drop_rate > 0), None otherwise."""
    masks = iter(dropout_masks) if dropout_masks is not None else None
    f = 'features.'
    enc, dec = blocks[:len(blocks) // 2], blocks[len(blocks) // 2:]
    pad = 3 if imsize % 2 == 0 else 2
    x = F.conv2d(x, sd[f + 'In_conv.weight'], stride=2, padding=pad)
    for i, n in enumerate(enc, 1):
        x = _dense_block(sd, f'{f}EncBlock{i}', x, n, training, masks)
        x = _transition(sd, f'{f}TransDown{i}', x, True, training, upsample, masks)
    for i, n in enumerate(dec, 1):
        x = _dense_block(sd, f'{f}DecBlock{i}', x, n, training, masks)
        if i < len(dec):
            x = _transition(sd, f'{f}TransUp{i}', x, False, training, upsample, masks)
    return _last(sd, f + 'LastTransUp', x, training, upsample, masks)


def decoder_forward(sd, z, blocks, training=True, upsample='nearest'):
    """Decoder.forward (codec.py:359-360)."""
    f = 'features.'
    x = F.conv2d(z, sd[f + 'conv0.weight'], padding=1)
    for i, n in enumerate(blocks, 1):
        x = _dense_block(sd, f'{f}DecBlock{i}', x, n, training)
        if i < len(blocks):
            x = _transition(sd, f'{f}TransUp{i}', x, False, training, upsample)
    return _last(sd, f + 'LastTransUp', x, training, upsample)


def _conv_init(shape):
    """nn.Conv2d default init (kaiming_uniform, a=sqrt(5)); consumes the global RNG like the
    reference's module construction does (one draw of weight.numel() per conv, creation order)."""
    w = torch.empty(shape)
    torch.nn.init.kaiming_uniform_(w, a=5 ** 0.5)
    return w


def _add_bn(sd, prefix, c):
    sd[prefix + '.weight'] = torch.ones(c)
    sd[prefix + '.bias'] = torch.zeros(c)
    sd[prefix + '.running_mean'] = torch.zeros(c)
    sd[prefix + '.running_var'] = torch.ones(c)
    sd[prefix + '.num_batches_tracked'] = torch.tensor(0, dtype=torch.long)


def _init_block(sd, name, c, n, growth):
    for j in range(1, n + 1):
        p = f'{name}.denselayer{j}'
        _add_bn(sd, p + '.norm1', c)
        sd[p + '.conv1.weight'] = _conv_init((growth, c, 3, 3))
        c += growth
    return c


def _init_transition(sd, name, c):
    _add_bn(sd, name + '.norm1', c)
    sd[name + '.conv1.weight'] = _conv_init((c // 2, c, 1, 1))
    _add_bn(sd, name + '.norm2', c // 2)
    sd[name + '.conv2.weight'] = _conv_init((c // 2, c // 2, 3, 3))
    return c // 2


def _init_last(sd, name, c, out_channels):
    _add_bn(sd, name + '.norm1', c)
    sd[name + '.conv1.weight'] = _conv_init((c // 2, c, 3, 3))
    _add_bn(sd, name + '.norm2', c // 2)
    sd[name + '.conv2.weight'] = _conv_init((c // 4, c // 2, 3, 3))
    _add_bn(sd, name + '.norm3', c // 4)
    sd[name + '.conv3.weight'] = _conv_init((out_channels, c // 4, 5, 5))


def densed_init(in_channels, out_channels, blocks, growth_rate=16, init_features=48):
    """Fresh reference-format state_dict in the reference's creation order (codec.py:237-287)."""
    sd, f = {}, 'features.'
    enc, dec = blocks[:len(blocks) // 2], blocks[len(blocks) // 2:]
    sd[f + 'In_conv.weight'] = _conv_init((init_features, in_channels, 7, 7))
    c = init_features
    for i, n in enumerate(enc, 1):
        c = _init_block(sd, f'{f}EncBlock{i}', c, n, growth_rate)
        c = _init_transition(sd, f'{f}TransDown{i}', c)
    for i, n in enumerate(dec, 1):
        c = _init_block(sd, f'{f}DecBlock{i}', c, n, growth_rate)
        if i < len(dec):
            c = _init_transition(sd, f'{f}TransUp{i}', c)
    _init_last(sd, f + 'LastTransUp', c, out_channels)
    return sd


def decoder_init(dim_latent, out_channels, blocks, growth_rate=16, init_features=48):
    sd, f = {}, 'features.'
    sd[f + 'conv0.weight'] = _conv_init((init_features, dim_latent, 3, 3))
    c = init_features
    for i, n in enumerate(blocks, 1):
        c = _init_block(sd, f'{f}DecBlock{i}', c, n, growth_rate)
        if i < len(blocks):
            c = _init_transition(sd, f'{f}TransUp{i}', c)
    _init_last(sd, f + 'LastTransUp', c, out_channels)
    return sd


def param_keys(sd):
    """keys that are nn.Parameters in the reference (what optim.Adam sees), in named_parameters order."""
    return [k for k in sd if not (k.endswith('running_mean') or k.endswith('running_var')
                                  or k.endswith('num_batches_tracked'))]
