#!/usr/bin/env python
"""Generate golden vectors under tests/golden/ by IMPORTING the real reference.

Runs only in the build container, where /root/reference exists (read-only).  The reference's
.py files never travel: what is committed is this script plus small .npz files holding inputs
and the reference's outputs for them.  Inputs come from numpy's default_rng so they can be
re-created without torch's RNG; weights for the default-size net come from torch.manual_seed(1)
in the reference's module-creation order (sha256 of the state_dict is stored so a test can tell
whether the local torch reproduces it).

    python tools/gen_golden.py            # rewrites tests/golden/*.npz
    python tools/gen_golden.py glow       # only G18-G20 (conditional Glow)
    python tools/gen_golden.py round3     # only G21 (any field size; correct=False through the loss functions)
    python tools/gen_golden.py round4     # only W_seeded (seeded initial parameters) and G22 (DenseED at B = 64 / 128 / 256)
    python tools/gen_golden.py round5     # only G23 (BASELINE configs[3]: the default DenseED on channelized fields, B = 32)
    python tools/gen_golden.py round6     # only G24 (G11's outputs, every sample) and G25 (configs[0] end to end through the reference's script)
"""
import hashlib
import io
import contextlib
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = '/root/reference'
sys.path.insert(0, REF)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.append(ROOT)           # pde_surrogate_amd.utils.data (synthetic GRF generator) -- AFTER the reference on the path
OUT = os.path.join(ROOT, 'tests', 'golden')

from models.codec import DenseED, Decoder            # noqa: E402  (reference)
from models import darcy as rdarcy                   # noqa: E402  (reference)
from utils.image_gradient import SobelFilter         # noqa: E402  (reference)
from utils.practices import OneCycleScheduler        # noqa: E402  (reference)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def sd_sha(sd):
    h = hashlib.sha256()
    for k in sd:
        h.update(k.encode())
        h.update(sd[k].detach().cpu().numpy().tobytes())
    return h.hexdigest()


def ref_loss(K, y, sobel, wb, nonlinear=False, b1=0.0, b2=0.0):
    if nonlinear:
        lc = rdarcy.conv_constitutive_constraint_nonlinear(K, y, sobel, b1, b2)
    else:
        lc = rdarcy.conv_constitutive_constraint(K, y, sobel)
    lt = rdarcy.conv_continuity_constraint(y, sobel)
    ld, ln = rdarcy.conv_boundary_condition(y)
    return lc + lt + (ld + ln) * wb, lc, lt, ld, ln


def gen_g13():
    """G13 (its own rng / seeds: regenerating it alone reproduces the round-2 file bit for bit, plus the round-3 keys)"""
    from oracle import codec as ocodec, train as otrain
    sob = SobelFilter(64, correct=True)
    # ---- G13: --upsample bilinear (reference codec.py:33-40, align_corners=True): tiny net with every tensor,
    # default net (B = 4) with output, loss terms, every gradient norm and the tensors next to the upsampling layers
    rng = np.random.default_rng(20190613)
    torch.manual_seed(7)
    net = quiet(DenseED, 1, 3, 16, [1, 1, 1], 4, 8, upsample='bilinear')
    with torch.no_grad():
        for k, v in net.state_dict().items():
            if 'norm' in k and k.endswith('.weight'):
                v.copy_(1 + 0.2 * torch.randn_like(v))
            if k.endswith('.bias'):
                v.copy_(0.1 * torch.randn_like(v))
    sd0 = {k: v.clone().numpy() for k, v in net.state_dict().items()}
    x = np.exp(0.5 * rng.standard_normal((4, 1, 16, 16))).astype(np.float32)
    sob16 = SobelFilter(16, correct=True)
    net.train()
    xt = torch.from_numpy(x)
    yo = net(xt)
    terms = ref_loss(xt, yo, sob16, 10.0)
    terms[0].backward()
    g13 = {'tiny/x': x, 'tiny/y': yo.detach().numpy(), 'tiny/terms': np.array([float(t) for t in terms], np.float64)}
    for k, v in sd0.items():
        g13['tiny/sd0/' + k] = v
    for k, p in net.named_parameters():
        g13['tiny/grad/' + k] = p.grad.numpy()
    torch.manual_seed(1)
    net = quiet(DenseED, 1, 3, 64, [6, 8, 6], 16, 48, upsample='bilinear')
    xb = np.exp(0.5 * rng.standard_normal((4, 1, 64, 64))).astype(np.float32)
    net.train()
    xt = torch.from_numpy(xb)
    yo = net(xt)
    terms = ref_loss(xt, yo, sob, 10.0)
    terms[0].backward()
    g13.update({'x': xb, 'y': yo.detach().numpy(), 'terms': np.array([float(t) for t in terms], np.float64),
                'param_names': np.array([k for k, _ in net.named_parameters()]),
                'grad_norms': np.array([float(p.grad.double().norm()) for _, p in net.named_parameters()])})
    for k, p in net.named_parameters():
        if k.startswith('features.TransUp1.') or (k.startswith('features.LastTransUp.') and 'conv1' not in k):
            g13['grad/' + k] = p.grad.numpy()
        elif 'norm' in k:
            # round 3: EVERY BatchNorm gradient tensor (26 KB in all).  These are the sums with cancellation where one ReLU
            # mask that differs between two fp32 implementations shows: a test can then tell "one element of one tensor
            # moved by one term of its sum" (a flip) from an error spread over the tensor (a bug)
            g13['grad/' + k] = p.grad.numpy()
    # the same pass on the fp64 oracle: where the reference's own fp32 norm is off by ~1e-3 (BatchNorm-bias sums with
    # heavy cancellation) a test may accept the fp64 value instead
    torch.manual_seed(1)
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in ocodec.densed_init(1, 3, [6, 8, 6], 16, 48).items()}
    tr64 = otrain.CpuTrainer(sd64, [6, 8, 6], upsample='bilinear')
    _, l64, _ = tr64.forward_loss(xt.double(), True)
    l64.backward()
    g13['grad_norms_fp64'] = np.array([float(sd64[k].grad.norm()) for k in tr64.keys])
    np.savez_compressed(os.path.join(OUT, 'G13_bilinear.npz'), **g13)



def gen_round2():
    """fixtures added in round 2; every section owns its rng so that G1..G10 stay bit-identical"""
    sob = SobelFilter(64, correct=True)

    # ---- G11: the HEADLINE configuration: default DenseED from manual_seed(1), B = 32 GRF-KLE512 fields (the build's
    # synthetic generator; the fields are stored), reference forward + loss + backward, ALL 82 gradient tensors
    from pde_surrogate_amd.utils.data import grf_kle_fields
    torch.manual_seed(1)
    net = quiet(DenseED, 1, 3, 64, [6, 8, 6], 16, 48)
    xb = grf_kle_fields(32, seed=11, cache_dir='/tmp')
    net.train()
    xt = torch.from_numpy(xb)
    yo = net(xt)
    terms = ref_loss(xt, yo, sob, 10.0)
    terms[0].backward()
    g11 = {'x': xb, 'y0': yo.detach().numpy()[0], 'y_slice': yo.detach().numpy()[:, :, ::8, ::8],
           'terms': np.array([float(t) for t in terms], np.float64),
           'param_names': np.array([k for k, _ in net.named_parameters()])}
    for k, p in net.named_parameters():
        g11['grad/' + k] = p.grad.numpy()
    # rounding floor of the REFERENCE itself: its fp32 gradients against the fp64 oracle (same weights, same input).
    # A few BatchNorm-bias gradients are sums with heavy cancellation and sit at 0.6e-3 .. 1.6e-3 in the reference's
    # own fp32 arithmetic; for those the fp64 gradient is stored too, so a test can tell "differs from the reference
    # by the reference's rounding error" from "wrong".
    from oracle import codec as ocodec, train as otrain
    torch.manual_seed(1)
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in ocodec.densed_init(1, 3, [6, 8, 6], 16, 48).items()}
    tr64 = otrain.CpuTrainer(sd64, [6, 8, 6])
    _, l64, _ = tr64.forward_loss(xt.double(), True)
    l64.backward()
    floor = np.array([float(np.linalg.norm(g11['grad/' + k].astype(np.float64) - sd64[k].grad.numpy())
                            / np.linalg.norm(sd64[k].grad.numpy())) for k in tr64.keys])
    g11['ref_fp32_vs_fp64_floor'] = floor
    for k, f in zip(tr64.keys, floor):
        if f > 3e-4:
            g11['grad64/' + k] = sd64[k].grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'G11_densed_default_b32.npz'), **g11)

    gen_g13()

    # ---- G14: conv_continuity_constraint(use_tb=False) (darcy.py:224) value + gradient; 5x5 Sobel fields
    rng = np.random.default_rng(20190614)
    y = rng.standard_normal((2, 3, 64, 64)).astype(np.float32)
    yt = torch.from_numpy(y).clone().requires_grad_(True)
    lt = rdarcy.conv_continuity_constraint(yt, sob, use_tb=False)
    lt.backward()
    img = rng.standard_normal((2, 1, 64, 64)).astype(np.float32) * 3 + 1
    t = torch.from_numpy(img)
    np.savez_compressed(os.path.join(OUT, 'G14_no_tb_sobel5.npz'), y=y, cont_no_tb=np.array(float(lt)),
                        cont_no_tb_grad=yt.grad.numpy(), img=img,
                        gh5=sob.grad_h(t, filter_size=5).numpy(), gv5=sob.grad_v(t, filter_size=5).numpy())

    # ---- G15: the data-driven harness train_codec_max_likelihood.py:197-211 (same DenseED, F.mse_loss):
    # 3 Adam steps at bs = 8 from manual_seed(1), then the eval-mode test metrics of :166-190
    import torch.nn.functional as F
    rng = np.random.default_rng(20190615)
    torch.manual_seed(1)
    net = quiet(DenseED, 1, 3, 64, [6, 8, 6], 16, 48)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=0.0)
    sched = OneCycleScheduler(lr_max=1e-3, div_factor=2.0, pct_start=0.3)
    data = np.exp(0.5 * rng.standard_normal((16, 1, 64, 64))).astype(np.float32)
    target = rng.standard_normal((16, 3, 64, 64)).astype(np.float32)
    losses, lrs, gn1 = [], [], None
    net.train()
    for step in range(1, 4):
        idx = np.arange(8) + 8 * ((step - 1) % 2)
        inp, tgt = torch.from_numpy(data[idx]), torch.from_numpy(target[idx])
        net.zero_grad()
        loss = F.mse_loss(net(inp), tgt)
        loss.backward()
        if step == 1:
            gn1 = np.array([float(p.grad.double().norm()) for p in net.parameters()])
            g_in = net.features.In_conv.weight.grad.numpy().copy()
            g_last = net.features.LastTransUp.conv3.weight.grad.numpy().copy()
        lr = sched.step(step / 40)
        for g in opt.param_groups:
            g['lr'] = lr
        opt.step()
        losses.append(float(loss))
        lrs.append(lr)
    net.eval()
    with torch.no_grad():
        o, t = net(torch.from_numpy(data)), torch.from_numpy(target)
        mse_eval = float(F.mse_loss(o, t))
        err2 = torch.sum((o - t) ** 2, [-1, -2])
        rel = torch.sqrt(err2 / (t ** 2).sum([-1, -2])).mean(0).numpy()
        yvar = ((target - target.mean(0, keepdims=True)) ** 2).sum(axis=(0, 2, 3))
        r2 = 1 - err2.sum(0).numpy() / yvar
    np.savez_compressed(os.path.join(OUT, 'G15_max_likelihood.npz'), data=data, target=target,
                        losses=np.array(losses), lrs=np.array(lrs), grad_norms_step1=gn1, grad_In_conv_step1=g_in,
                        grad_last_conv3_step1=g_last, mse_eval=np.array(mse_eval), nrmse_eval=rel, r2_eval=r2,
                        y_variation=yvar, y_eval=o.numpy())


def gen_round3():
    """G21: field sizes other than 16 / 32 / 64 and SobelFilter(correct=False) THROUGH the loss functions -- the reference
    takes any square imsize (image_gradient.py:26-47; its docstrings use 65 x 65, darcy.py:165-167) and hands the
    filter's `correct` flag to every gradient (image_gradient.py:72-75, :89-92).  Sobel fields, the four loss terms and
    dL/dy (autograd of the reference) at n = 20, 48, 65, 128; use_tb=False, the nonlinear law and the 5x5 filter with
    its autograd adjoint at n = 65."""
    rng = np.random.default_rng(20190621)
    out = {}
    for n, B in ((20, 2), (48, 2), (65, 2), (128, 1)):
        K = np.exp(0.5 * rng.standard_normal((B, 1, n, n))).astype(np.float32)
        y = rng.standard_normal((B, 3, n, n)).astype(np.float32)
        img = (rng.standard_normal((B, 1, n, n)) * 2 + 0.5).astype(np.float32)
        out[f'K{n}'], out[f'y{n}'], out[f'img{n}'] = K, y, img
        for correct in (True, False):
            if n == 128 and not correct:
                continue
            sfx = '' if correct else '_nocorrect'
            sob = SobelFilter(n, correct=correct)
            it = torch.from_numpy(img)
            out[f'gh{n}{sfx}'] = sob.grad_h(it).numpy()
            out[f'gv{n}{sfx}'] = sob.grad_v(it).numpy()
            yt = torch.from_numpy(y).requires_grad_(True)
            terms = ref_loss(torch.from_numpy(K), yt, sob, 10.0)
            terms[0].backward()
            out[f'terms{n}{sfx}'] = np.array([float(t) for t in terms], np.float64)
            out[f'grad{n}{sfx}'] = yt.grad.numpy()
    n = 65
    K, y, img = (torch.from_numpy(out[f'{k}{n}']) for k in ('K', 'y', 'img'))
    sob = SobelFilter(n, correct=True)
    yt = y.clone().requires_grad_(True)
    terms = ref_loss(K, yt, sob, 10.0, True, 0.1, 0.1)
    terms[0].backward()
    out['terms65_nl'] = np.array([float(t) for t in terms], np.float64)
    out['grad65_nl'] = yt.grad.numpy()
    yt = y.clone().requires_grad_(True)
    lt = rdarcy.conv_continuity_constraint(yt, sob, use_tb=False)
    lt.backward()
    out['cont65_no_tb'] = np.array(float(lt))
    out['grad65_no_tb'] = yt.grad.numpy()
    wh = rng.standard_normal(img.shape).astype(np.float32)
    wv = rng.standard_normal(img.shape).astype(np.float32)
    out['wh65'], out['wv65'] = wh, wv
    for correct in (True, False):
        sfx = '' if correct else '_nocorrect'
        s5 = SobelFilter(n, correct=correct)
        for fs in (3, 5):
            it = img.clone().requires_grad_(True)
            gh, gv = s5.grad_h(it, filter_size=fs), s5.grad_v(it, filter_size=fs)
            ((gh * torch.from_numpy(wh)).sum() + (gv * torch.from_numpy(wv)).sum()).backward()
            if fs == 5:
                out[f'gh65_f5{sfx}'], out[f'gv65_f5{sfx}'] = gh.detach().numpy(), gv.detach().numpy()
            out[f'adj65_f{fs}{sfx}'] = it.grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'G21_any_size.npz'), **out)


def fixed_projection(shape, tag):
    """an RNG-free projection direction for a gradient tensor (tests rebuild it bit for bit): cos of a ramp"""
    n = int(np.prod(shape))
    return np.cos(0.37 * np.arange(n, dtype=np.float64) + 0.11 * tag).reshape(shape)


def gen_round4():
    """round 4.  (1) W_seeded.npz: the seeded INITIAL parameters the G6/G10/G11/G12/G13/G15/G19 fixtures were made
    from (default DenseED from torch.manual_seed(1), Decoder(1,3,[8,6]) from manual_seed(3), the default conditional
    Glow of G19 after its perturbation) -- SURVEY 8(c) G6 fallback: a torch whose RNG stream differs loads these instead
    of skipping.  (2) G22: the default DenseED ABOVE the training batch: train mode at B = 256 and B = 128 (config 3's
    strong-scaled points; output, loss terms, running statistics, gradient norms + projections of all 82 tensors, a
    handful of tensors in full) and eval mode at B = 64 (the reference's default test() batch,
    train_codec_mixed_residual.py:63,166-206) -- the kernel plans are chosen by batch size."""
    from pde_surrogate_amd.utils.data import grf_kle_fields
    sob = SobelFilter(64, correct=True)
    w = {}
    torch.manual_seed(1)
    net = quiet(DenseED, 1, 3, 64, [6, 8, 6], 16, 48)
    g6 = np.load(os.path.join(OUT, 'G6_densed_default.npz'))
    assert sd_sha(net.state_dict()) == str(g6['sha256'])
    for k, v in net.state_dict().items():
        w['densed_seed1/' + k] = v.numpy().copy()
    torch.manual_seed(3)
    dec = Decoder(1, 3, [8, 6])
    assert sd_sha(dec.state_dict()) == str(np.load(os.path.join(OUT, 'G10_decoder.npz'))['sha256'])
    for k, v in dec.state_dict().items():
        w['decoder_seed3/' + k] = v.numpy().copy()
    G = _glow_module()
    torch.manual_seed(1)
    np.random.seed(1)
    gl = quiet(G.MultiScaleCondGlow, 32, 1, 3, [3, 4, 4], [6, 6, 6], LUdecompose=True, squeeze_factor=2)
    g19 = np.load(os.path.join(OUT, 'G19_cglow_default.npz'))
    assert sd_sha({k: v for k, v in gl.named_parameters()}) == str(g19['init_sha256'])
    pn = {k for k, _ in gl.named_parameters()}
    for k, v in gl.state_dict().items():            # buffers only (BatchNorm running statistics, permutations, masks)
        if k not in pn:
            w['cglow_g19_buffers/' + k] = v.detach().numpy().copy()
    _perturb_glow(gl, torch.Generator().manual_seed(13), 0.4)
    assert sd_sha({k: v for k, v in gl.named_parameters()}) == str(g19['param_sha256'])
    for k, v in gl.named_parameters():
        w['cglow_g19/' + k] = v.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, 'W_seeded.npz'), **w)

    # ---- G22
    x = grf_kle_fields(256, seed=22, cache_dir='/tmp')
    g = {'x': x}
    full = ['features.In_conv.weight', 'features.EncBlock1.denselayer3.conv1.weight',
            'features.TransDown1.conv1.weight', 'features.TransDown1.conv2.weight',
            'features.DecBlock1.denselayer5.conv1.weight', 'features.DecBlock1.denselayer5.norm1.bias',
            'features.TransUp1.conv1.weight', 'features.TransUp1.conv2.weight',
            'features.DecBlock2.denselayer6.conv1.weight', 'features.DecBlock2.denselayer2.norm1.weight',
            'features.LastTransUp.conv1.weight', 'features.LastTransUp.conv2.weight',
            'features.LastTransUp.conv3.weight', 'features.LastTransUp.norm3.bias']
    names = [k for k, _ in net.named_parameters()]
    g['param_names'] = np.array(names)
    from oracle import codec as ocodec, train as otrain

    def train_case(B):
        net.train()
        net.zero_grad()
        xt = torch.from_numpy(x[:B])
        yo = net(xt)
        terms = ref_loss(xt, yo, sob, 10.0)
        terms[0].backward()
        t = 'b%d/' % B
        yn = yo.detach().numpy()
        g[t + 'y_first'], g[t + 'y_last'], g[t + 'y_slice'] = yn[0], yn[B - 1], yn[:, :, ::8, ::8]
        g[t + 'terms'] = np.array([float(v) for v in terms], np.float64)
        grads = {k: p.grad.numpy().copy() for k, p in net.named_parameters()}
        g[t + 'grad_norms'] = np.array([float(np.linalg.norm(grads[k].astype(np.float64))) for k in names])
        g[t + 'grad_proj'] = np.array([float((grads[k].astype(np.float64) * fixed_projection(grads[k].shape, i)).sum())
                                       for i, k in enumerate(names)])
        for k in full:
            g[t + 'grad/' + k] = grads[k]
        for k, v in net.state_dict().items():
            if 'running' in k:
                g[t + 'sd/' + k] = v.numpy().copy()
        # the reference's own fp32 rounding floor against the fp64 oracle (same weights, same input), as for G11
        sd64 = {k: (torch.from_numpy(w['densed_seed1/' + k]).double() if w['densed_seed1/' + k].dtype == np.float32
                    else torch.from_numpy(w['densed_seed1/' + k]).clone()) for k in net.state_dict().keys()}
        tr64 = otrain.CpuTrainer(sd64, [6, 8, 6])
        _, l64, _ = tr64.forward_loss(xt.double(), True)
        l64.backward()
        assert list(tr64.keys) == names
        g[t + 'ref_fp32_vs_fp64_floor'] = np.array(
            [float(np.linalg.norm(grads[k].astype(np.float64) - sd64[k].grad.numpy()) / np.linalg.norm(sd64[k].grad.numpy()))
             for k in names])
        g[t + 'grad_norms64'] = np.array([float(sd64[k].grad.norm()) for k in names])
        g[t + 'grad_proj64'] = np.array([float((sd64[k].grad.numpy() * fixed_projection(grads[k].shape, i)).sum())
                                         for i, k in enumerate(names)])
        print('G22 B=%d loss %.6f  floor max %.2e' % (B, float(terms[0]), g[t + 'ref_fp32_vs_fp64_floor'].max()))

    for B in (256, 128):
        train_case(B)
    # eval mode at the default test batch, with the running statistics the two training forwards left behind
    net.eval()
    with torch.no_grad():
        xt = torch.from_numpy(x[:64])
        yo = net(xt)
        terms = ref_loss(xt, yo, sob, 10.0)
    g['e64/y_head'] = yo.numpy()[:4]
    g['e64/y_slice'] = yo.numpy()[:, :, ::4, ::4]
    g['e64/terms'] = np.array([float(v) for v in terms], np.float64)
    train_case(64)          # and the train-mode step at 64 (after the eval case: the entries above stay what they were)
    np.savez_compressed(os.path.join(OUT, 'G22_densed_batches.npz'), **g)
    for f in ('W_seeded.npz', 'G22_densed_batches.npz'):
        print(f, os.path.getsize(os.path.join(OUT, f)))


def gen_round5():
    """round 5.  G23: BASELINE configs[3] -- the default DenseED (seeded initial parameters, W_seeded) on CHANNELIZED
    fields (two-valued, sharp interfaces: the build's synthetic generator, the fields are stored), B = 32, train mode:
    the reference's forward + loss + backward (train_codec_mixed_residual.py:134-143, :224-233).  Output, the five loss
    terms, ALL 82 gradient tensors, the running statistics after the forward, and the reference's own fp32 rounding floor
    against the fp64 oracle per tensor (as G11).  This is the input family SURVEY section 7 flags for E[x^2] - E[x]^2
    cancellation in the first BatchNorms and for ReLU flips."""
    from pde_surrogate_amd.utils.data import channelized_fields
    from oracle import train as otrain
    sob = SobelFilter(64, correct=True)
    torch.manual_seed(1)
    net = quiet(DenseED, 1, 3, 64, [6, 8, 6], 16, 48)
    g6 = np.load(os.path.join(OUT, 'G6_densed_default.npz'))
    assert sd_sha(net.state_dict()) == str(g6['sha256'])
    w0 = {k: v.numpy().copy() for k, v in net.state_dict().items()}
    xb = channelized_fields(32, seed=23)
    assert sorted(np.unique(xb).tolist()) == [1.0, 10.0]
    net.train()
    xt = torch.from_numpy(xb)
    yo = net(xt)
    terms = ref_loss(xt, yo, sob, 10.0)
    terms[0].backward()
    names = [k for k, _ in net.named_parameters()]
    g = {'x': xb, 'y0': yo.detach().numpy()[0], 'y_last': yo.detach().numpy()[31], 'y_slice': yo.detach().numpy()[:, :, ::8, ::8],
         'terms': np.array([float(t) for t in terms], np.float64), 'param_names': np.array(names)}
    grads = {k: p.grad.numpy().copy() for k, p in net.named_parameters()}
    for k in names:
        g['grad/' + k] = grads[k]
    for k, v in net.state_dict().items():
        if 'running' in k:
            g['sd/' + k] = v.numpy().copy()
    sd64 = {k: (torch.from_numpy(w0[k]).double() if w0[k].dtype == np.float32 else torch.from_numpy(w0[k]).clone()) for k in w0}
    tr64 = otrain.CpuTrainer(sd64, [6, 8, 6])
    _, l64, _ = tr64.forward_loss(xt.double(), True)
    l64.backward()
    assert list(tr64.keys) == names
    floor = np.array([float(np.linalg.norm(grads[k].astype(np.float64) - sd64[k].grad.numpy()) / np.linalg.norm(sd64[k].grad.numpy()))
                      for k in names])
    g['ref_fp32_vs_fp64_floor'] = floor
    for k, f in zip(names, floor):
        if f > 3e-4:
            g['grad64/' + k] = sd64[k].grad.numpy()
    print('G23 loss %.6f terms %s floor max %.2e (%s)' % (float(terms[0]), [round(float(t), 5) for t in terms[1:]], floor.max(),
                                                          names[int(floor.argmax())]))
    np.savez_compressed(os.path.join(OUT, 'G23_densed_channelized_b32.npz'), **g)
    print('G23_densed_channelized_b32.npz', os.path.getsize(os.path.join(OUT, 'G23_densed_channelized_b32.npz')))


def gen_round6():
    """round 6.
    G24: per-sample output norms, the last sample and the per-channel means of the reference's output on G11's batch (the
         fixture held sample 0 and a slice): the test asserts 1e-5 on first / last / slice / every sample's norm.
    G25: BASELINE configs[0] END TO END: the reference's own script, unmodified (runpy; h5py replaced by an in-memory module
         holding the arrays -- h5py is not installed here), `--data grf_kle512 --ntrain 512 --batch-size 8 --epochs 2` (+ a
         64-sample test set) on the CPU: the loss of every one of its 128 steps, the permutations its DataLoaders drew, the
         four per-epoch log files, the final first-layer weights.  The dataset is stored quantised (K = uint16 / 256, targets =
         int16 / 1024: exact in fp32, so every machine feeds the same bits)."""
    import runpy
    import shutil
    import tempfile
    import types
    from pde_surrogate_amd.utils.data import grf_kle_fields
    # ---- G24
    g11 = np.load(os.path.join(OUT, 'G11_densed_default_b32.npz'))
    torch.manual_seed(1)
    net = quiet(DenseED, 1, 3, 64, [6, 8, 6], 16, 48)
    net.train()
    yo = net(torch.from_numpy(g11['x'])).detach().numpy()
    assert np.array_equal(yo[0], g11['y0']), 'the reference no longer reproduces G11 bit for bit on this machine'
    np.savez_compressed(os.path.join(OUT, 'G24_densed_default_b32_outputs.npz'), y_last=yo[-1],
                        y_norms=np.sqrt((yo.astype(np.float64) ** 2).sum(axis=(1, 2, 3))),
                        y_channel_means=yo.astype(np.float64).mean(axis=(2, 3)), y_slice4=yo[:, :, ::4, ::4])
    # ---- G25
    ntrain, ntest = 512, 64
    K = grf_kle_fields(ntrain + ntest, seed=25, cache_dir='/tmp')
    kq = np.clip(np.rint(K * 256.0), 1, 65535).astype(np.uint16)
    x_all = kq.astype(np.float32) / 256.0
    rng = np.random.default_rng(2025)
    jj = (np.arange(64) + 0.5) / 64

    def smooth(a):
        for ax in (-1, -2):
            a = (np.roll(a, 1, ax) + 2 * a + np.roll(a, -1, ax)) / 4
        return a
    u = (1.0 - jj)[None, None, None, :] + 0.05 * smooth(smooth(rng.standard_normal((ntest, 1, 64, 64))))
    s1 = x_all[ntrain:] * (1.0 + 0.1 * smooth(rng.standard_normal((ntest, 1, 64, 64))))
    s2 = 0.3 * smooth(smooth(rng.standard_normal((ntest, 1, 64, 64))))
    yq = np.clip(np.rint(np.concatenate([u, s1, s2], 1) * 1024.0), -32768, 32767).astype(np.int16)
    y_test = yq.astype(np.float32) / 1024.0
    files = {'kle512_lhs10000_train.hdf5': {'input': x_all[:ntrain]},
             'kle512_lhs1000_val.hdf5': {'input': x_all[ntrain:], 'output': y_test}}

    class _File(dict):
        def __init__(self, path, mode='r'):
            super().__init__(files[os.path.basename(path)])

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False
    shim = types.ModuleType('h5py')
    shim.File = _File
    sys.modules['h5py'] = shim
    step_losses, perms = [], []
    item0, randperm0 = torch.Tensor.item, torch.randperm

    def item(self):
        v = item0(self)
        if self.grad_fn is not None and self.dim() == 0:        # `loss.item()` of the training loop (:240); test() runs under no_grad
            step_losses.append(v)
        return v

    def randperm(n, *a, **k):
        r = randperm0(n, *a, **k)
        if k.get('generator') is not None:
            perms.append((int(n), r.numpy().copy()))
        return r
    tmp = tempfile.mkdtemp(prefix='g25_')
    argv0, cwd0 = sys.argv, os.getcwd()
    torch.Tensor.item, torch.randperm = item, randperm
    try:
        sys.argv = ['train_codec_mixed_residual.py', '--data', 'grf_kle512', '--ntrain', str(ntrain), '--ntest', str(ntest),
                    '--batch-size', '8', '--test-batch-size', '64', '--epochs', '2', '--exp-dir', tmp, '--data-dir', tmp,
                    '--plot-freq', '1000', '--ckpt-freq', '1000', '--cuda', '0']
        os.chdir(tmp)
        torch.set_num_threads(8)
        ns = runpy.run_path(os.path.join(REF, 'train_codec_mixed_residual.py'), run_name='__main__')
    finally:
        torch.Tensor.item, torch.randperm = item0, randperm0
        sys.argv = argv0
        os.chdir(cwd0)
        del sys.modules['h5py']
    args = ns['args']
    logs = {m: np.loadtxt(os.path.join(args.train_dir, m + '.txt')) for m in ('loss_train', 'loss_test', 'nrmse_test', 'r2_test')}
    model = ns['model']
    assert len(step_losses) == 128, len(step_losses)
    # (a RandomSampler run to exhaustion draws a second, unused permutation from its local generator -- the
    #  `[: num_samples % n]` tail of its __iter__: every other recorded permutation is an epoch's order)
    train_perms = np.stack([p for n, p in perms if n == ntrain][0:4:2])
    test_perms = np.stack([p for n, p in perms if n == ntest][0:4:2])
    assert len([1 for n, _ in perms if n == ntrain]) == 4 and len([1 for n, _ in perms if n == ntest]) >= 4
    np.savez_compressed(os.path.join(OUT, 'G25_config1_cli_run.npz'), k_u16_over_256=kq, y_test_i16_over_1024=yq,
                        argv=np.array(sys.argv if False else ['--data', 'grf_kle512', '--ntrain', '512', '--ntest', '64', '--batch-size', '8',
                                                              '--test-batch-size', '64', '--epochs', '2', '--seed', '1']),
                        step_losses=np.array(step_losses, np.float64), train_perms=train_perms, test_perms=test_perms,
                        final_In_conv=model.features.In_conv.weight.detach().numpy(),
                        final_running_mean=model.features.LastTransUp.norm3.running_mean.numpy(),
                        y_variation=ns['y_test_variation'], **logs)
    print('G25: step losses', step_losses[:4], '...', step_losses[-2:], '| logs', {k: v.tolist() for k, v in logs.items()})
    shutil.rmtree(tmp, ignore_errors=True)


def gen_dropout():
    """G16: --drop-rate > 0 (nn.Dropout2d after the convolution of every dense layer, after both convolutions of a
    transition and after the first convolution of the last decoding: reference codec.py:70-71, :111-120, :134-150,
    :172-173).  The channel masks are drawn here with numpy and INJECTED into the reference through
    torch.nn.functional.dropout2d (what nn.Dropout2d.forward calls), so the fixture pins WHERE the reference applies
    dropout and its 1/(1-p) scaling, independent of any RNG stream: masks (in call order), output, loss terms, all
    gradients, running statistics after the step, and the eval-mode output (dropout inactive)."""
    import torch.nn.functional as F
    rng = np.random.default_rng(20190616)
    p = 0.25
    torch.manual_seed(7)
    net = quiet(DenseED, 1, 3, 16, [2, 2, 2], 4, 8, drop_rate=p)
    with torch.no_grad():
        for k, v in net.state_dict().items():
            if 'norm' in k and k.endswith('.weight'):
                v.copy_(1 + 0.2 * torch.randn_like(v))
            if k.endswith('.bias'):
                v.copy_(0.1 * torch.randn_like(v))
    sd0 = {k: v.clone().numpy() for k, v in net.state_dict().items()}
    x = np.exp(0.5 * rng.standard_normal((4, 1, 16, 16))).astype(np.float32)
    masks = []
    orig = F.dropout2d

    def injected(input, p=0.5, training=True, inplace=False):
        if not training:
            return input
        m = (rng.uniform(size=input.shape[:2]) >= p).astype(np.float32) / (1.0 - p)
        masks.append(m)
        return input * torch.from_numpy(m)[:, :, None, None]
    F.dropout2d = injected
    try:
        sob16 = SobelFilter(16, correct=True)
        net.train()
        xt = torch.from_numpy(x)
        yo = net(xt)
        terms = ref_loss(xt, yo, sob16, 10.0)
        terms[0].backward()
        g = {'x': x, 'p': np.array(p), 'y': yo.detach().numpy(), 'terms': np.array([float(t) for t in terms], np.float64),
             'n_masks': np.array(len(masks))}
        for i, m in enumerate(masks):
            g[f'mask{i}'] = m
        for k, v in sd0.items():
            g['sd0/' + k] = v
        for k, q in net.named_parameters():
            g['grad/' + k] = q.grad.numpy()
        for k, v in net.state_dict().items():
            if 'running' in k:
                g['sd1/' + k] = v.numpy()
        net.eval()
        with torch.no_grad():
            g['y_eval'] = net(xt).numpy()
    finally:
        F.dropout2d = orig
    np.savez_compressed(os.path.join(OUT, 'G16_dropout.npz'), **g)


def gen_bottleneck():
    """G17: DenseED(bottleneck=True, bn_size=2) (codec.py:55-62: dense layers with more than bn_size * growth_rate
    inputs get norm1 / conv1 1x1 -> bn_size * growth_rate, norm2 / conv2 3x3): every tensor of a tiny net"""
    rng = np.random.default_rng(20190617)
    torch.manual_seed(7)
    net = quiet(DenseED, 1, 3, 16, [3, 3, 3], 4, 8, bn_size=2, bottleneck=True)
    with torch.no_grad():
        for k, v in net.state_dict().items():
            if 'norm' in k and k.endswith('.weight'):
                v.copy_(1 + 0.2 * torch.randn_like(v))
            if k.endswith('.bias'):
                v.copy_(0.1 * torch.randn_like(v))
    sd0 = {k: v.clone().numpy() for k, v in net.state_dict().items()}
    x = np.exp(0.5 * rng.standard_normal((4, 1, 16, 16))).astype(np.float32)
    sob16 = SobelFilter(16, correct=True)
    net.train()
    xt = torch.from_numpy(x)
    yo = net(xt)
    terms = ref_loss(xt, yo, sob16, 10.0)
    terms[0].backward()
    g = {'x': x, 'y': yo.detach().numpy(), 'terms': np.array([float(t) for t in terms], np.float64),
         'n_params': np.array(net.model_size[0]), 'n_conv': np.array(net.model_size[1])}
    for k, v in sd0.items():
        g['sd0/' + k] = v
    for k, q in net.named_parameters():
        g['grad/' + k] = q.grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'G17_bottleneck.npz'), **g)


def _glow_module():
    """the reference's models/glow_msc.py, importable under a current PyTorch: GaussianDiag.__init__ (glow_msc.py:435-440)
    clamps a chunk view in place, which autograd has refused since PyTorch 1.5 ("output of a function that returns
    multiple views ... modified inplace"); the out-of-place clamp below has the same values and the same gradient.
    Nothing else of the reference is touched."""
    import math
    import models.glow_msc as G

    def _init(self, mean, log_stddev):
        self.mean = mean
        self.log_stddev = log_stddev.clamp(min=-10., max=math.log(5.))
    G.GaussianDiag.__init__ = _init
    return G


def _perturb_glow(net, gen, amp=1.0):
    """move every parameter off its initial value (the reference initialises the coupling nets' last convolutions, the
    latent encoders and all ActNorms to the identity map, which would leave most gradients exactly zero)"""
    with torch.no_grad():
        for k, p in net.named_parameters():
            r = amp * torch.randn(p.shape, generator=gen)
            if k.endswith('.scale'):
                p.add_(0.1 * r)
            elif 'conv_zero' in k or 'top_latent' in k or 'latent_encoder' in k:
                p.add_((0.02 if k.endswith('weight') else 0.1) * r)
            elif k.endswith('.norm.weight'):         # ActNorm: the z -> y path divides by it
                p.copy_(1 + 0.1 * r)
            elif 'norm' in k and k.endswith('.weight'):
                p.copy_(1 + 0.2 * r)
            elif k.endswith('.bias'):
                p.add_(0.1 * r)
            elif k.endswith('.l') or k.endswith('.u'):
                p.add_(0.05 * r)             # (entries outside the triangular masks never enter the weight)
            elif k.endswith('.log_s'):
                p.add_(0.1 * r)
            elif k.endswith('conv1x1.weight'):
                p.add_(0.05 * r)


def gen_glow():
    """G18 / G19 / G20: multiscale conditional Glow, the TRAINING path of train_cglow_reverse_kl.py:245-262
    (generate -> mixed-residual loss + entropy term -> backward), eval-mode generate, and the y -> z direction"""
    import math
    G = _glow_module()
    rng = np.random.default_rng(20190711)

    def run(net, x, eps, imsize, beta, wb, full):
        sob = SobelFilter(imsize, correct=True)
        xt = torch.from_numpy(x)
        el = [torch.from_numpy(e) for e in eps]
        net.train()
        net.zero_grad()
        y, logp = net.generate(xt, eps_list=el)
        terms = ref_loss(xt, y, sob, wb)
        neg_entropy = logp.mean() / math.log(2.) / y[0].numel()
        loss = terms[0] * beta + neg_entropy
        loss.backward()
        g = {'x': x, 'beta': np.float64(beta), 'weight_bound': np.float64(wb), 'logp': logp.detach().numpy(),
             'terms': np.array([float(loss), float(terms[0]), float(neg_entropy)] + [float(t) for t in terms[1:]], np.float64)}
        for i, e in enumerate(eps):
            g[f'eps{i}'] = e
        return g, y.detach()

    # ---- G18: small net, 16x16, every tensor stored
    torch.manual_seed(7)
    np.random.seed(7)
    cfg = dict(imsize=16, enc_blocks=[2, 1, 1], flow_blocks=[2, 2, 1])
    net = quiet(G.MultiScaleCondGlow, cfg['imsize'], 1, 3, cfg['enc_blocks'], cfg['flow_blocks'], LUdecompose=True,
                squeeze_factor=2)
    _perturb_glow(net, torch.Generator().manual_seed(11))
    sd0 = {k: v.clone().numpy() for k, v in net.state_dict().items()}
    B = 4
    x = np.exp(0.5 * rng.standard_normal((B, 1, 16, 16))).astype(np.float32)
    eps = [rng.standard_normal((B,) + s).astype(np.float32) for s in net._z_shapes()]
    g, y = run(net, x, eps, 16, 150.0, 50.0, True)
    g['y'] = y.numpy()
    g['enc_blocks'], g['flow_blocks'] = np.array(cfg['enc_blocks']), np.array(cfg['flow_blocks'])
    g['n_params'], g['n_layers'] = np.array(net.model_size[0]), np.array(net.model_size[1])
    for k, v in sd0.items():
        g['sd0/' + k] = v
    for k, p in net.named_parameters():
        g['grad/' + k] = p.grad.numpy()
    for k, v in net.state_dict().items():
        if 'running' in k or 'num_batches' in k:
            g['sd1/' + k] = v.numpy()
    net.eval()
    with torch.no_grad():
        ye, lpe = net.generate(torch.from_numpy(x), eps_list=[torch.from_numpy(e) for e in eps])
        g['y_eval'], g['logp_eval'] = ye.numpy(), lpe.numpy()
        z, lpf, el = net.forward(ye, torch.from_numpy(x), return_eps=True)
        g['z_fwd'], g['logp_fwd'] = z.numpy(), lpf.numpy()
        for i, e in enumerate(el):
            g[f'eps_fwd{i}'] = e.numpy()
        samples = net.sample(torch.from_numpy(x[:2]), n_samples=3, eps_list=[torch.from_numpy(
            np.stack([e[:2]] * 3) * (1 + np.arange(3, dtype=np.float32)).reshape(3, 1, 1, 1, 1)) for e in eps], temperature=0.8)
        g['samples'] = samples.numpy()
    np.savez_compressed(os.path.join(OUT, 'G18_cglow_small.npz'), **g)

    # ---- G20: the plain (non-LU) invertible 1x1 convolution, --no-LU-decompose
    torch.manual_seed(8)
    np.random.seed(8)
    net = quiet(G.MultiScaleCondGlow, 16, 1, 3, [1, 1, 1], [2, 1, 1], LUdecompose=False, squeeze_factor=2)
    _perturb_glow(net, torch.Generator().manual_seed(12))
    sd0 = {k: v.clone().numpy() for k, v in net.state_dict().items()}
    eps = [rng.standard_normal((B,) + s).astype(np.float32) for s in net._z_shapes()]
    g, y = run(net, x, eps, 16, 150.0, 50.0, True)
    g['y'] = y.numpy()
    for k, v in sd0.items():
        g['sd0/' + k] = v
    g['param_names'] = np.array([k for k, _ in net.named_parameters()])
    g['grad_norms'] = np.array([float(p.grad.double().norm()) for _, p in net.named_parameters()])
    for k, p in net.named_parameters():
        if 'conv1x1' in k or 'norm.' in k:
            g['grad/' + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'G20_cglow_plain1x1.npz'), **g)

    # ---- G19: the default net of train_cglow_reverse_kl.py (enc [3,4,4], flow [6,6,6], 32x32), weights from
    # torch.manual_seed(1) + np.random.seed(1) in the reference's creation order, then _perturb_glow(seed 13)
    torch.manual_seed(1)
    np.random.seed(1)
    net = quiet(G.MultiScaleCondGlow, 32, 1, 3, [3, 4, 4], [6, 6, 6], LUdecompose=True, squeeze_factor=2)
    init_sha = sd_sha({k: v for k, v in net.named_parameters()})
    init_sha_buffers = sd_sha({k: v for k, v in net.state_dict().items() if 'running' in k})
    _perturb_glow(net, torch.Generator().manual_seed(13), 0.4)
    # the reference's constructor pushes one random image through the encoder in train mode (glow_msc.py:713-714:
    # feature_sizes), which moves the encoder's BatchNorm running statistics; stored so a test can load them
    B = 8
    x = np.exp(0.5 * rng.standard_normal((B, 1, 32, 32))).astype(np.float32)
    eps = [rng.standard_normal((B,) + s).astype(np.float32) for s in net._z_shapes()]
    names = [k for k, _ in net.named_parameters()]
    g, y = run(net, x, eps, 32, 150.0, 50.0, False)
    proj = torch.Generator().manual_seed(14)
    g.update({'init_sha256': np.array(init_sha), 'init_sha256_running': np.array(init_sha_buffers),
              'param_sha256': np.array(sd_sha({k: v for k, v in net.named_parameters()})),
              'y0': y.numpy()[0], 'y_slice': y.numpy()[:, :, ::4, ::4], 'param_names': np.array(names),
              'grad_norms': np.array([float(p.grad.double().norm()) for _, p in net.named_parameters()]),
              'grad_proj': np.array([float((p.grad.double() * torch.randn(p.shape, generator=proj).double()).sum())
                                     for _, p in net.named_parameters()]),
              'n_params': np.array(net.model_size[0]), 'n_layers': np.array(net.model_size[1]),
              'n_state': np.array(len(net.state_dict()))})
    for k in ('encoder.dense_block1.in_conv.weight', 'encoder.top_latent.conv.weight',
              'flow.revblock1.revlayers.revlayer1.coupling.coupling_nn.reduce.conv_zero.conv.weight',
              'flow.revblock2.revlayers.revlayer3.conv1x1.l', 'flow.revblock2.revlayers.revlayer3.conv1x1.log_s',
              'flow.revblock3.revlayers.revlayer6.norm.weight', 'flow.revblock2.split.latent_encoder.conv2d.scale'):
        g['grad/' + k] = dict(net.named_parameters())[k].grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'G19_cglow_default.npz'), **g)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    rng = np.random.default_rng(20190524)

    # ---- G1: Sobel fields, 64x64 and a ragged-ish small 8x8 (edge formulas dominate)
    g1 = {}
    for n in (64, 8):
        img = rng.standard_normal((2, 1, n, n)).astype(np.float32) * 3 + 1
        sob = SobelFilter(n, correct=True)
        sob_nc = SobelFilter(n, correct=False)
        t = torch.from_numpy(img)
        g1[f'img{n}'] = img
        g1[f'gh{n}'] = sob.grad_h(t).numpy()
        g1[f'gv{n}'] = sob.grad_v(t).numpy()
        g1[f'gh{n}_nocorrect'] = sob_nc.grad_h(t).numpy()
        g1[f'gv{n}_nocorrect'] = sob_nc.grad_v(t).numpy()
    np.savez_compressed(os.path.join(OUT, 'G1_sobel.npz'), **g1)

    # ---- G2: linear loss + dL/dy (autograd of the reference), wb=10; G3: nonlinear
    K = np.exp(0.5 * rng.standard_normal((2, 1, 64, 64))).astype(np.float32)
    y = rng.standard_normal((2, 3, 64, 64)).astype(np.float32)
    sob = SobelFilter(64, correct=True)
    out = {'K': K, 'y': y, 'weight_bound': np.float32(10)}
    for tag, nl in (('lin', False), ('nl', True)):
        yt = torch.from_numpy(y).clone().requires_grad_(True)
        terms = ref_loss(torch.from_numpy(K), yt, sob, 10.0, nl, 0.1, 0.1)
        terms[0].backward()
        out[f'{tag}_terms'] = np.array([float(t) for t in terms], np.float64)
        out[f'{tag}_grad'] = yt.grad.numpy()
        # per-term gradients (the drop-in API differentiates each function separately)
        for i, nm in enumerate(('const', 'cont', 'dir', 'neu'), 1):
            yt2 = torch.from_numpy(y).clone().requires_grad_(True)
            t2 = ref_loss(torch.from_numpy(K), yt2, sob, 10.0, nl, 0.1, 0.1)
            t2[i].backward()
            out[f'{tag}_grad_{nm}'] = yt2.grad.numpy()
    out['beta'] = np.array([0.1, 0.1], np.float32)
    np.savez_compressed(os.path.join(OUT, 'G2_G3_loss.npz'), **out)

    # ---- G4: closed form u = 1 - j/W, sigma1 = K, sigma2 = 0
    Kc = np.exp(0.5 * rng.standard_normal((1, 1, 64, 64))).astype(np.float32)
    yc = np.zeros((1, 3, 64, 64), np.float32)
    yc[0, 0] = 1.0 - np.arange(64, dtype=np.float32)[None, :] / 64
    yc[0, 1] = Kc[0, 0]
    terms = ref_loss(torch.from_numpy(Kc), torch.from_numpy(yc), sob, 10.0)
    np.savez_compressed(os.path.join(OUT, 'G4_closed_form.npz'), K=Kc, y=yc,
                        terms=np.array([float(t) for t in terms], np.float64))

    # ---- G5: tiny DenseED, everything stored
    torch.manual_seed(7)
    cfg = dict(blocks=[1, 1, 1], growth_rate=4, init_features=8, imsize=16)
    net = quiet(DenseED, 1, 3, cfg['imsize'], cfg['blocks'], cfg['growth_rate'], cfg['init_features'])
    # make BN affine non-trivial so gamma/beta gradients are exercised
    with torch.no_grad():
        for k, v in net.state_dict().items():
            if k.endswith('norm1.weight') or k.endswith('norm2.weight') or k.endswith('norm3.weight'):
                v.copy_(1 + 0.2 * torch.randn_like(v))
            if k.endswith('.bias'):
                v.copy_(0.1 * torch.randn_like(v))
    sd0 = {k: v.clone().numpy() for k, v in net.state_dict().items()}
    x = np.exp(0.5 * rng.standard_normal((4, 1, 16, 16))).astype(np.float32)
    sob16 = SobelFilter(16, correct=True)
    net.train()
    xt = torch.from_numpy(x)
    yo = net(xt)
    terms = ref_loss(xt, yo, sob16, 10.0)
    terms[0].backward()
    g5 = {'x': x, 'y': yo.detach().numpy(), 'terms': np.array([float(t) for t in terms], np.float64)}
    for k, v in sd0.items():
        g5['sd0/' + k] = v
    for k, p in net.named_parameters():
        g5['grad/' + k] = p.grad.numpy()
    for k, v in net.state_dict().items():
        if 'running' in k or 'num_batches' in k:
            g5['sd1/' + k] = v.numpy()
    net.eval()
    with torch.no_grad():
        g5['y_eval'] = net(xt).numpy()
    np.savez_compressed(os.path.join(OUT, 'G5_densed_tiny.npz'), **g5)

    # ---- G6: default DenseED (blocks [6,8,6], growth 16, init 48), weights from manual_seed(1)
    torch.manual_seed(1)
    net = quiet(DenseED, 1, 3, 64, [6, 8, 6], 16, 48)
    sha = sd_sha(net.state_dict())
    xb = np.exp(0.5 * rng.standard_normal((8, 1, 64, 64))).astype(np.float32)
    net.train()
    xt = torch.from_numpy(xb)
    yo = net(xt)
    terms = ref_loss(xt, yo, sob, 10.0)
    terms[0].backward()
    names = [k for k, _ in net.named_parameters()]
    g6 = {'x': xb, 'sha256': np.array(sha), 'y_slice': yo.detach().numpy()[:, :, ::8, ::8],
          'y0': yo.detach().numpy()[0], 'terms': np.array([float(t) for t in terms], np.float64),
          'param_names': np.array(names),
          'grad_norms': np.array([float(p.grad.double().norm()) for _, p in net.named_parameters()]),
          'grad_In_conv': net.features.In_conv.weight.grad.numpy(),
          'grad_last_conv3': net.features.LastTransUp.conv3.weight.grad.numpy(),
          'grad_enc1_l1_bn_w': net.features.EncBlock1.denselayer1.norm1.weight.grad.numpy(),
          'grad_enc1_l1_bn_b': net.features.EncBlock1.denselayer1.norm1.bias.grad.numpy(),
          'n_params': np.array(net.model_size[0]), 'n_conv': np.array(net.model_size[1]),
          'n_state': np.array(len(net.state_dict()))}
    np.savez_compressed(os.path.join(OUT, 'G6_densed_default.npz'), **g6)

    # ---- G7: 8-step trajectory, bs=8, reference loop body (Adam + one-cycle), explicit batches
    torch.manual_seed(1)
    net = quiet(DenseED, 1, 3, 64, [6, 8, 6], 16, 48)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=0.0)
    sched = OneCycleScheduler(lr_max=1e-3, div_factor=2.0, pct_start=0.3)
    data = np.exp(0.5 * rng.standard_normal((16, 1, 64, 64))).astype(np.float32)
    order = np.array([[0, 1, 2, 3, 4, 5, 6, 7], [8, 9, 10, 11, 12, 13, 14, 15]] * 4)
    total_steps, losses, lrs = 40, [], []
    g12 = {}
    net.train()
    for step, idx in enumerate(order, 1):
        inp = torch.from_numpy(data[idx])
        net.zero_grad()
        outp = net(inp)
        terms = ref_loss(inp, outp, sob, 10.0)
        terms[0].backward()
        # G12 (teacher forcing): the weights this step STARTED from (flat, named_parameters order), its four loss
        # terms and every per-tensor gradient norm -- a test reloads the weights and must reproduce step `step`
        if step > 1:
            g12[f'w{step}'] = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy().copy()
        g12[f'terms{step}'] = np.array([float(t) for t in terms], np.float64)
        g12[f'gnorm{step}'] = np.array([float(p.grad.double().norm()) for p in net.parameters()])
        lr = sched.step(step / total_steps)
        for g in opt.param_groups:
            g['lr'] = lr
        opt.step()
        losses.append(float(terms[0]))
        lrs.append(lr)
    np.savez_compressed(os.path.join(OUT, 'G7_trajectory.npz'), data=data, order=order,
                        total_steps=np.array(total_steps), losses=np.array(losses), lrs=np.array(lrs),
                        final_In_conv=net.features.In_conv.weight.detach().numpy(),
                        final_running_mean=net.features.LastTransUp.norm3.running_mean.numpy())
    g12['param_names'] = np.array([k for k, _ in net.named_parameters()])
    g12['param_numel'] = np.array([p.numel() for p in net.parameters()])
    np.savez_compressed(os.path.join(OUT, 'G12_teacher_forced.npz'), **g12)

    # ---- G8: one-cycle LR values
    pcts = np.linspace(0, 1, 11)
    s = OneCycleScheduler(lr_max=1e-3, div_factor=2.0, pct_start=0.3)
    s25 = OneCycleScheduler(lr_max=5e-4, div_factor=25.0, pct_start=0.3)
    np.savez_compressed(os.path.join(OUT, 'G8_one_cycle.npz'), pcts=pcts,
                        lr=np.array([s.step(p) for p in pcts]), lr25=np.array([s25.step(p) for p in pcts]))

    # ---- G9: metrics as computed in test() (train_codec_mixed_residual.py:180-197, load.py:28-30)
    tgt = rng.standard_normal((6, 3, 16, 16)).astype(np.float32)
    prd = tgt + 0.1 * rng.standard_normal((6, 3, 16, 16)).astype(np.float32)
    yvar = ((tgt - tgt.mean(0, keepdims=True)) ** 2).sum(axis=(0, 2, 3))
    o, t = torch.from_numpy(prd), torch.from_numpy(tgt)
    err2 = torch.sum((o - t) ** 2, [-1, -2])
    rel = torch.sqrt(err2 / (t ** 2).sum([-1, -2])).mean(0).numpy()
    r2 = 1 - err2.sum(0).numpy() / yvar
    np.savez_compressed(os.path.join(OUT, 'G9_metrics.npz'), target=tgt, pred=prd, y_variation=yvar,
                        nrmse=rel, r2=r2)

    # ---- G10: Decoder (config 5) -- shapes, one closure value (nonlinear), grads norms
    torch.manual_seed(3)
    dec = Decoder(1, 3, [8, 6])
    sha_dec = sd_sha(dec.state_dict())
    z = (0.5 * rng.standard_normal((1, 1, 16, 16))).astype(np.float32)
    K1 = np.exp(0.5 * rng.standard_normal((1, 1, 64, 64))).astype(np.float32)
    dec.train()
    yo = dec(torch.from_numpy(z))
    terms = ref_loss(torch.from_numpy(K1), yo, sob, 10.0, True, 0.1, 0.1)
    terms[0].backward()
    np.savez_compressed(os.path.join(OUT, 'G10_decoder.npz'), z=z, K=K1, y=yo.detach().numpy(),
                        sha256=np.array(sha_dec), terms=np.array([float(t) for t in terms], np.float64),
                        param_names=np.array([k for k, _ in dec.named_parameters()]),
                        grad_norms=np.array([float(p.grad.double().norm()) for _, p in dec.named_parameters()]),
                        n_params=np.array(dec.model_size[0]), n_conv=np.array(dec.model_size[1]))
    gen_round2()
    gen_dropout()
    gen_bottleneck()
    gen_glow()
    gen_round3()
    gen_round4()
    gen_round5()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'glow':      # only the conditional-Glow fixtures (G18-G20)
        torch.set_num_threads(8)
        gen_glow()
    elif len(sys.argv) > 1 and sys.argv[1] == 'g13':
        torch.set_num_threads(8)
        gen_g13()
    elif len(sys.argv) > 1 and sys.argv[1] == 'round3':  # only G21 (any field size, correct=False through the loss)
        torch.set_num_threads(8)
        gen_round3()
    elif len(sys.argv) > 1 and sys.argv[1] == 'round4':  # only W_seeded (initial parameters) and G22 (B = 64 / 128 / 256)
        torch.set_num_threads(8)
        gen_round4()
    elif len(sys.argv) > 1 and sys.argv[1] == 'round6':  # only G24 (G11's outputs in full) and G25 (configs[0] through the reference's script)
        torch.set_num_threads(8)
        gen_round6()
    elif len(sys.argv) > 1 and sys.argv[1] == 'round5':  # only G23 (default DenseED on channelized fields, B = 32)
        torch.set_num_threads(8)
        gen_round5()
    else:
        main()
