"""The conditional-Glow oracle (oracle/glow.py) against golden vectors produced by the real reference
(tools/gen_golden.py gen_glow: models/glow_msc.py + train_cglow_reverse_kl.py:245-262 imported from /root/reference).
CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import glow

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _load(name):
    g = np.load(os.path.join(GOLD, name))
    sd = {k[4:]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith('sd0/')}
    eps = [torch.from_numpy(g[f'eps{i}']) for i in range(sum(1 for k in g.files if k.startswith('eps') and k[3:].isdigit()))]
    return g, sd, eps


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _loss_and_grads(g, sd, eps):
    keys = glow.param_keys(sd)
    for k in keys:
        sd[k].requires_grad_(True)
    x = torch.from_numpy(g['x'])
    loss, loss_pde, neg_ent, y = glow.reverse_kl_loss(sd, x, eps, float(g['beta']), float(g['weight_bound']), True)
    loss.backward()
    return keys, loss, loss_pde, neg_ent, y


@pytest.mark.parametrize('name', ['G18_cglow_small.npz', 'G20_cglow_plain1x1.npz'])
def test_generate_loss_and_gradients(name):
    g, sd, eps = _load(name)
    keys, loss, loss_pde, neg_ent, y = _loss_and_grads(g, sd, eps)
    assert _rel(y.detach().numpy(), g['y']) < 2e-5
    np.testing.assert_allclose([float(loss.detach()), float(loss_pde.detach()), float(neg_ent.detach())], g['terms'][:3], rtol=2e-5)
    names = [k[5:] for k in g.files if k.startswith('grad/')]
    assert names
    # (the gradient of in_conv.bias is pure rounding noise: every consumer of those channels starts with a BatchNorm,
    #  which removes a constant shift -- hence the absolute term, relative to the largest gradient of the net)
    gmax = max(float(np.linalg.norm(g['grad/' + k])) for k in names)
    for k in names:
        err = float(np.linalg.norm(sd[k].grad.numpy().astype(np.float64) - g['grad/' + k]))
        assert err < 2e-4 * float(np.linalg.norm(g['grad/' + k])) + 1e-6 * gmax, k
    if 'grad_norms' in g.files:
        assert list(g['param_names']) == keys
        for k, n in zip(keys, g['grad_norms']):
            assert abs(float(sd[k].grad.double().norm()) - n) < 2e-4 * n + 1e-6 * float(g['grad_norms'].max()), k
    else:
        assert sorted(names) == sorted(keys)


def test_running_statistics_eval_generate_forward_direction_and_sampling():
    g, sd, eps = _load('G18_cglow_small.npz')
    x = torch.from_numpy(g['x'])
    with torch.no_grad():
        y, logp = glow.generate(sd, x, eps, training=True)
        np.testing.assert_allclose(logp.numpy(), g['logp'], rtol=2e-5)
        for k in g.files:
            if k.startswith('sd1/'):
                np.testing.assert_allclose(sd[k[4:]].numpy(), g[k], rtol=1e-4, atol=1e-5, err_msg=k)
        ye, lpe = glow.generate(sd, x, eps, training=False)
        assert _rel(ye.numpy(), g['y_eval']) < 2e-5
        np.testing.assert_allclose(lpe.numpy(), g['logp_eval'], rtol=2e-5)
        z, lpf, ef = glow.forward(sd, torch.from_numpy(g['y_eval']), x, training=False)
        assert _rel(z.numpy(), g['z_fwd']) < 1e-4
        np.testing.assert_allclose(lpf.numpy(), g['logp_fwd'], rtol=1e-4)
        for i, e in enumerate(ef):
            assert _rel(e.numpy(), g[f'eps_fwd{i}']) < 1e-3
            assert _rel(e.numpy(), g[f'eps{i}']) < 1e-3          # the flow inverts its own samples
        # MultiScaleCondGlow.sample (glow_msc.py:841-876): temperature scales the split latents' noise, not the top one's
        xs = x[:2]
        for s in range(3):
            el = [eps[0][:2] * (1 + s) * 0.8, eps[1][:2] * (1 + s)]
            ys, _ = glow.generate(sd, xs, el, training=False)
            assert _rel(ys.numpy(), g['samples'][s]) < 2e-5


def test_latent_shapes_and_counts():
    g, sd, eps = _load('G18_cglow_small.npz')
    assert [tuple(e.shape[1:]) for e in eps] == glow.latent_shapes(sd, 3, 16)
    assert sum(sd[k].numel() for k in glow.param_keys(sd)) == int(g['n_params'])


def test_default_net_projections():
    """G19 holds no weights: they come from the reference's constructor under torch.manual_seed(1) / np.random.seed(1);
    the drop-in module's init parity is tested on the GPU side (tests/test_cglow_gpu.py).  Here: shapes only."""
    g = np.load(os.path.join(GOLD, 'G19_cglow_default.npz'))
    assert int(g['n_params']) == 1535549 and len(g['param_names']) == len(g['grad_norms']) == len(g['grad_proj'])
    assert g['eps0'].shape == (8, 6, 16, 16) and g['eps1'].shape == (8, 24, 8, 8)
