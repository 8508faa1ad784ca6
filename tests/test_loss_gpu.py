"""GPU parity tests (run with -m gpu on an MI355X): fused Sobel + Darcy loss HIP kernels, called
through the C ABI, against (1) golden vectors produced by the real reference and (2) the CPU
oracle on seeded inputs.  Tolerances (fp32): Sobel fields atol 1e-4 / rtol 1e-5, loss scalars
rel 1e-5 (north_star), dL/dy rel-L2 1e-5.
"""
import numpy as np
import pytest
import torch

from conftest import golden, rel_l2

pytestmark = pytest.mark.gpu

LOSS_RTOL = 1e-5
GRAD_RL2 = 1e-5


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'gpu tests need an MI355X'
    return torch.device('cuda:0')


def _fields(B, n, seed, dev):
    rng = np.random.default_rng(seed)
    K = np.exp(0.5 * rng.standard_normal((B, 1, n, n))).astype(np.float32)
    y = rng.standard_normal((B, 3, n, n)).astype(np.float32)
    return K, y, torch.from_numpy(K).to(dev), torch.from_numpy(y).to(dev)


def test_g1_sobel_fixture(dev):
    from pde_surrogate_amd.utils.image_gradient import SobelFilter
    g = golden('G1_sobel.npz')
    img = torch.from_numpy(g['img64']).to(dev)
    for correct, sfx in ((True, ''), (False, '_nocorrect')):
        sob = SobelFilter(64, correct=correct, device=dev)
        np.testing.assert_allclose(sob.grad_h(img).cpu().numpy(), g['gh64' + sfx], rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(sob.grad_v(img).cpu().numpy(), g['gv64' + sfx], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize('n', [16, 32, 64, 2, 7, 20, 65, 200])
@pytest.mark.parametrize('B', [1, 5])
@pytest.mark.parametrize('correct', [True, False])
def test_sobel_vs_oracle_and_adjoint(dev, n, B, correct):
    from oracle import darcy as od
    from pde_surrogate_amd.utils.image_gradient import SobelFilter
    rng = np.random.default_rng(n * 10 + B)
    img = (rng.standard_normal((B, 1, n, n)) * 2 + 0.5).astype(np.float32)
    t = torch.from_numpy(img).to(dev).requires_grad_(True)
    sob = SobelFilter(n, correct=correct, device=dev)
    gh, gv = sob.grad_h(t), sob.grad_v(t)
    ref_h = od.sobel_grad_h(torch.from_numpy(img).double(), correct).numpy()
    ref_v = od.sobel_grad_v(torch.from_numpy(img).double(), correct).numpy()
    np.testing.assert_allclose(gh.detach().cpu().numpy(), ref_h, rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(gv.detach().cpu().numpy(), ref_v, rtol=1e-5, atol=1e-4)
    # autograd through the HIP adjoint kernel == autograd through the oracle
    wh = rng.standard_normal(gh.shape).astype(np.float32)
    wv = rng.standard_normal(gv.shape).astype(np.float32)
    ((gh * torch.from_numpy(wh).to(dev)).sum() + (gv * torch.from_numpy(wv).to(dev)).sum()).backward()
    to = torch.from_numpy(img).double().requires_grad_(True)
    ((od.sobel_grad_h(to, correct) * torch.from_numpy(wh).double()).sum()
     + (od.sobel_grad_v(to, correct) * torch.from_numpy(wv).double()).sum()).backward()
    assert rel_l2(t.grad.cpu().numpy(), to.grad.numpy()) < 1e-5


@pytest.mark.parametrize('tag,nl', [('lin', False), ('nl', True)])
def test_g2_g3_fused_loss_fixture(dev, tag, nl):
    from pde_surrogate_amd.models import darcy
    g = golden('G2_G3_loss.npz')
    K = torch.from_numpy(g['K']).to(dev)
    y = torch.from_numpy(g['y']).to(dev).requires_grad_(True)
    b1, b2 = (float(v) for v in g['beta'])
    loss, l_pde, l_dir, l_neu = darcy.darcy_mixed_residual_loss(K, y, 10.0, nl, b1, b2)
    loss.backward()
    ref = g[f'{tag}_terms']
    np.testing.assert_allclose(float(loss), ref[0], rtol=LOSS_RTOL)
    np.testing.assert_allclose(float(l_pde), ref[1] + ref[2], rtol=LOSS_RTOL)
    np.testing.assert_allclose(float(l_dir), ref[3], rtol=LOSS_RTOL)
    np.testing.assert_allclose(float(l_neu), ref[4], rtol=LOSS_RTOL)
    assert rel_l2(y.grad.cpu().numpy(), g[f'{tag}_grad']) < GRAD_RL2


@pytest.mark.parametrize('tag,nl', [('lin', False), ('nl', True)])
def test_g2_g3_dropin_functions_fixture(dev, tag, nl):
    """the reference's call pattern (train_codec_mixed_residual.py:228-233) on the drop-in API"""
    from pde_surrogate_amd.models import darcy
    from pde_surrogate_amd.utils.image_gradient import SobelFilter
    g = golden('G2_G3_loss.npz')
    K = torch.from_numpy(g['K']).to(dev)
    sob = SobelFilter(64, correct=True, device=dev)
    b1, b2 = (float(v) for v in g['beta'])
    ref = g[f'{tag}_terms']

    def terms(y):
        lc = (darcy.conv_constitutive_constraint_nonlinear(K, y, sob, b1, b2) if nl
              else darcy.conv_constitutive_constraint(K, y, sob))
        lt = darcy.conv_continuity_constraint(y, sob)
        ld, ln = darcy.conv_boundary_condition(y)
        return lc, lt, ld, ln

    y = torch.from_numpy(g['y']).to(dev).requires_grad_(True)
    lc, lt, ld, ln = terms(y)
    loss = lc + lt + (ld + ln) * 10.0
    loss.backward()
    np.testing.assert_allclose([float(loss), float(lc), float(lt), float(ld), float(ln)], ref, rtol=LOSS_RTOL)
    assert rel_l2(y.grad.cpu().numpy(), g[f'{tag}_grad']) < GRAD_RL2
    for i, nm in enumerate(('const', 'cont', 'dir', 'neu')):
        y = torch.from_numpy(g['y']).to(dev).requires_grad_(True)
        terms(y)[i].backward()
        assert rel_l2(y.grad.cpu().numpy(), g[f'{tag}_grad_{nm}']) < GRAD_RL2, nm


def test_dropin_loss_functions_share_one_launch_and_never_sync(dev, monkeypatch):
    """the reference's loop body (train_codec_mixed_residual.py:228-233) on the drop-in functions is ONE
    forward-only launch of the fused kernel and ONE backward launch (the three functions return entries of the same
    autograd result), the upstream gradients reach the kernel through device memory -- `loss.backward()` contains no
    device -> host synchronisation (torch's sync-debug mode raises on any) -- and nothing stale is ever served: another
    conductivity, an in-place change of the output, another `use_tb`, or grad mode switched off start a new launch."""
    from pde_surrogate_amd.models import darcy
    from pde_surrogate_amd.utils.image_gradient import SobelFilter
    calls = []
    real = darcy.darcy_loss_launch

    def counting(K, y, weights, want_grad, *a, **k):
        calls.append(('bwd' if want_grad else 'fwd', torch.is_tensor(weights)))
        return real(K, y, weights, want_grad, *a, **k)
    monkeypatch.setattr(darcy, 'darcy_loss_launch', counting)
    Kn, yn, K, y0 = _fields(4, 64, 11, dev)
    sob = SobelFilter(64, correct=True, device=dev)
    yl = y0.clone().requires_grad_(True)
    y = yl * 1.0                      # a network's output is a NON-LEAF: that is where the three functions share one node
    loss_pde = darcy.conv_constitutive_constraint(K, y, sob) + darcy.conv_continuity_constraint(y, sob)
    ld, ln = darcy.conv_boundary_condition(y)
    loss = loss_pde + (ld + ln) * 10.0
    assert calls == [('fwd', False)]
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode('error')
    try:
        loss.backward()
    finally:
        torch.cuda.set_sync_debug_mode('default')
    assert calls == [('fwd', False), ('bwd', True)]
    fused, *_ = darcy.darcy_mixed_residual_loss(K, y0.clone().requires_grad_(True), 10.0)
    ref_terms, ref_grad = real(K, y0, (1.0, 1.0, 10.0, 10.0), True)
    np.testing.assert_allclose(float(loss), float(ref_terms[0]), rtol=1e-6)
    np.testing.assert_allclose(float(fused), float(ref_terms[0]), rtol=1e-6)
    assert rel_l2(yl.grad.cpu().numpy(), ref_grad.cpu().numpy()) < 1e-6
    # for a LEAF output the reference's functions build independent graphs -- `.backward()` on each loss in
    # turn is legal without retain_graph and accumulates; here every call then gets its own node (3 + 3 launches)
    del calls[:]
    y = y0.clone().requires_grad_(True)
    la, lb = darcy.conv_constitutive_constraint(K, y, sob), darcy.conv_continuity_constraint(y, sob)
    ld, ln = darcy.conv_boundary_condition(y)
    la.backward()
    lb.backward()
    ((ld + ln) * 10.0).backward()
    assert [c[0] for c in calls] == ['fwd', 'fwd', 'fwd', 'bwd', 'bwd', 'bwd']
    assert rel_l2(y.grad.cpu().numpy(), ref_grad.cpu().numpy()) < 1e-6
    # the one remembered pair per thread can be dropped by hand (an evaluation loop with grad mode on)
    import gc
    import weakref
    y = yl * 1.0
    lc = darcy.conv_constitutive_constraint(K, y, sob)
    base = weakref.ref(lc._base)
    del lc
    darcy.forget_shared_loss()
    gc.collect()
    assert base() is None
    # a backward pass that reaches only the boundary terms must not be poisoned by a non-finite value in a
    # term it never asked for (zero weight x inf): sigma1 = inf at one pixel, gradient of the boundary loss alone
    del calls[:]
    ybad = y0.clone()
    ybad[0, 1, 20, 20] = float('inf')
    ybl = ybad.clone().requires_grad_(True)
    ld, ln = darcy.conv_boundary_condition(ybl * 1.0)
    (ld + ln).backward()
    _, gb = real(K, y0, (0.0, 0.0, 1.0, 1.0), True)          # the boundary terms do not look at sigma1: the clean field's gradient
    assert torch.isfinite(ybl.grad).all() and rel_l2(ybl.grad.cpu().numpy(), gb.cpu().numpy()) < 1e-6
    # the remembered pair was dropped by its backward; whatever is remembered next must never be served stale
    del calls[:]
    y = y0.clone().requires_grad_(True)
    a = darcy.conv_constitutive_constraint(K, y, sob)
    b = darcy.conv_constitutive_constraint(K * 2.0, y, sob)                  # another conductivity
    assert len(calls) == 2 and float(b) != float(a)
    c = darcy.conv_continuity_constraint(y, sob, use_tb=False)              # another row range
    d = darcy.conv_continuity_constraint(y, sob)
    assert len(calls) == 4 and float(c) != float(d)
    with torch.no_grad():
        e = darcy.conv_continuity_constraint(y, sob)                        # grad mode off: its own (graph-free) result
    assert len(calls) == 5 and not e.requires_grad and float(e) == float(d)
    yv = y0.clone()
    f = darcy.conv_boundary_condition(yv)[0]
    yv.mul_(2.0)                                                            # in-place change of the output
    g = darcy.conv_boundary_condition(yv)[0]
    assert len(calls) == 7 and float(f) != float(g)
    lc_nl = darcy.conv_constitutive_constraint_nonlinear(K, yv, sob, 0.1, 0.1)
    lc_lin = darcy.conv_constitutive_constraint(K, yv, sob)
    assert len(calls) == 9 and float(lc_nl) != float(lc_lin)
    assert float(darcy.conv_continuity_constraint(yv, sob)) == float(real(K, yv, (1, 1, 1, 1), False)[0][2]) and len(calls) == 9


def test_g4_closed_form(dev):
    from pde_surrogate_amd.models import darcy
    g = golden('G4_closed_form.npz')
    K, y = torch.from_numpy(g['K']).to(dev), torch.from_numpy(g['y']).to(dev)
    terms, _ = darcy.darcy_loss_launch(K, y, (1, 1, 10, 10), False)
    t = terms.cpu().numpy()
    assert abs(t[1]) < 1e-9 and t[4] == 0.0
    assert abs(t[3] - (1 / 64) ** 2) < 1e-9
    np.testing.assert_allclose(t[2], g['terms'][2], rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize('n', [48, 65, 66, 96, 100, 128, 129, 130, 131, 200, 256])
def test_band_kernel_fixed_geometry_equals_runtime_plan(dev, n, option):
    """round 6: the instantiations of the row-band kernel with the geometry of the common sizes folded in (default) against the
    same kernel on the run-time plan (PDES_BAND_FIXED=0): the same arithmetic -- the compiler may contract other products, hence
    1e-6 and not bitwise -- forward + backward and forward only, aligned and (multiples of 4) unaligned pointers"""
    from pde_surrogate_amd.models import darcy
    B = 3
    K, y, Kd, yd = _fields(B, n, 7000 + n, dev)
    w = (0.7, 1.3, 9.0, 11.0)
    res = {}
    for fixed in (1, 0):
        option('PDES_BAND_FIXED', fixed)
        t, g = darcy.darcy_loss_launch(Kd, yd, w, True, False, 0.1, 0.2, True, True)
        t2, _ = darcy.darcy_loss_launch(Kd, yd, w, False, False, 0.1, 0.2, True, True)
        res[fixed] = (t.cpu().numpy(), g.cpu().numpy(), t2.cpu().numpy())
    np.testing.assert_allclose(res[1][0], res[0][0], rtol=1e-6)
    np.testing.assert_allclose(res[1][2], res[0][2], rtol=1e-6)
    assert rel_l2(res[1][1], res[0][1]) < 1e-6
    assert np.isfinite(res[1][1]).all()


@pytest.mark.parametrize('n', [16, 32, 64])
@pytest.mark.parametrize('B', [1, 3, 32])
@pytest.mark.parametrize('nl', [False, True])
def test_fused_loss_vs_oracle(dev, n, B, nl):
    from oracle import darcy as od
    from pde_surrogate_amd.models import darcy
    K, y, Kd, yd = _fields(B, n, 1000 + n + B, dev)
    wb = 7.5
    terms, grad = darcy.darcy_loss_launch(Kd, yd, (1, 1, wb, wb), True, nl, 0.3, 0.2)
    ref_terms, ref_grad = od.loss_and_grad_autograd(torch.from_numpy(K).double(), torch.from_numpy(y).double(),
                                                    wb, 0.3, 0.2, nl)
    np.testing.assert_allclose(terms.cpu().numpy(), [float(v) for v in ref_terms], rtol=LOSS_RTOL)
    assert rel_l2(grad.cpu().numpy(), ref_grad.numpy()) < GRAD_RL2
    # forward-only launch (eval path) gives the same scalars
    terms2, g2 = darcy.darcy_loss_launch(Kd, yd, (1, 1, wb, wb), False, nl, 0.3, 0.2)
    assert g2 is None
    np.testing.assert_allclose(terms.cpu().numpy(), terms2.cpu().numpy(), rtol=1e-6)


@pytest.mark.parametrize('n,correct', [(20, True), (20, False), (48, True), (48, False), (65, True), (65, False), (128, True)])
def test_g21_any_field_size_fixture(dev, n, correct):
    """field sizes other than 16 / 32 / 64 and SobelFilter(correct=False) through the drop-in loss functions, against
    the REAL reference (G21: image_gradient.py:26-92 for any imsize, darcy.py:162-233)"""
    from pde_surrogate_amd.models import darcy
    from pde_surrogate_amd.utils.image_gradient import SobelFilter
    g = golden('G21_any_size.npz')
    sfx = '' if correct else '_nocorrect'
    sob = SobelFilter(n, correct=correct, device=dev)
    img = torch.from_numpy(g[f'img{n}']).to(dev)
    np.testing.assert_allclose(sob.grad_h(img).cpu().numpy(), g[f'gh{n}{sfx}'], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(sob.grad_v(img).cpu().numpy(), g[f'gv{n}{sfx}'], rtol=1e-5, atol=1e-4)
    K = torch.from_numpy(g[f'K{n}']).to(dev)
    y = torch.from_numpy(g[f'y{n}']).to(dev).requires_grad_(True)
    lc = darcy.conv_constitutive_constraint(K, y, sob)
    lt = darcy.conv_continuity_constraint(y, sob)
    ld, ln = darcy.conv_boundary_condition(y)
    loss = lc + lt + (ld + ln) * 10.0
    loss.backward()
    ref = g[f'terms{n}{sfx}']
    np.testing.assert_allclose([float(loss), float(lc), float(lt), float(ld), float(ln)], ref, rtol=LOSS_RTOL)
    assert rel_l2(y.grad.cpu().numpy(), g[f'grad{n}{sfx}']) < GRAD_RL2
    # the fused single-launch form gives the same numbers
    terms, grad = darcy.darcy_loss_launch(K, y.detach(), (1, 1, 10, 10), True, correct=correct)
    np.testing.assert_allclose(terms.cpu().numpy(), ref, rtol=LOSS_RTOL)
    assert rel_l2(grad.cpu().numpy(), g[f'grad{n}{sfx}']) < GRAD_RL2


def test_g21_variants_at_65(dev):
    """nonlinear law, use_tb=False, filter_size=5 and the autograd adjoints of both filters for both values of
    `correct`, 65 x 65 (the size the reference's docstrings use), against the real reference"""
    from pde_surrogate_amd.models import darcy
    from pde_surrogate_amd.utils.image_gradient import SobelFilter
    g = golden('G21_any_size.npz')
    K = torch.from_numpy(g['K65']).to(dev)
    sob = SobelFilter(65, correct=True, device=dev)
    y = torch.from_numpy(g['y65']).to(dev).requires_grad_(True)
    loss, l_pde, l_dir, l_neu = darcy.darcy_mixed_residual_loss(K, y, 10.0, True, 0.1, 0.1)
    loss.backward()
    ref = g['terms65_nl']
    np.testing.assert_allclose([float(loss), float(l_pde), float(l_dir), float(l_neu)],
                               [ref[0], ref[1] + ref[2], ref[3], ref[4]], rtol=LOSS_RTOL)
    assert rel_l2(y.grad.cpu().numpy(), g['grad65_nl']) < GRAD_RL2
    y = torch.from_numpy(g['y65']).to(dev).requires_grad_(True)
    lt = darcy.conv_continuity_constraint(y, sob, use_tb=False)
    lt.backward()
    np.testing.assert_allclose(float(lt), float(g['cont65_no_tb']), rtol=LOSS_RTOL)
    assert rel_l2(y.grad.cpu().numpy(), g['grad65_no_tb']) < GRAD_RL2
    wh, wv = torch.from_numpy(g['wh65']).to(dev), torch.from_numpy(g['wv65']).to(dev)
    for correct in (True, False):
        sfx = '' if correct else '_nocorrect'
        s = SobelFilter(65, correct=correct, device=dev)
        for fs in (3, 5):
            img = torch.from_numpy(g['img65']).to(dev).requires_grad_(True)
            gh, gv = s.grad_h(img, filter_size=fs), s.grad_v(img, filter_size=fs)
            if fs == 5:
                np.testing.assert_allclose(gh.detach().cpu().numpy(), g['gh65_f5' + sfx], rtol=1e-5, atol=1e-4)
                np.testing.assert_allclose(gv.detach().cpu().numpy(), g['gv65_f5' + sfx], rtol=1e-5, atol=1e-4)
            ((gh * wh).sum() + (gv * wv).sum()).backward()
            assert rel_l2(img.grad.cpu().numpy(), g[f'adj65_f{fs}{sfx}']) < 1e-5, (correct, fs)


# (64 with correct=True is the specialised kernel, tested above; use_tb=False on a 2-row field is a mean over ZERO rows:
#  nan in the reference too, darcy.py:224)
GENERIC_CASES = [(n, B, f) for n, B in [(2, 3), (3, 2), (5, 4), (33, 3), (64, 2), (96, 2), (130, 2), (300, 1), (20, 300)]
                 for f in (0, 4, 3) if not (n == 64 and f == 0) and not (n == 2 and f & 2)]


@pytest.mark.parametrize('n,B,flags', GENERIC_CASES)
def test_generic_kernel_vs_oracle(dev, n, B, flags):
    """the tiled any-size kernel (csrc/darcy_loss_generic.hip) against the fp64 oracle: smallest fields, one tile,
    several row tiles (n = 96, 130), column tiles too (n = 300), a grid of 300 images; correct=False (flag 4) and
    nonlinear + use_tb=False (flags 3) also route n = 64 to it"""
    from oracle import darcy as od
    from pde_surrogate_amd.models import darcy
    K, y, Kd, yd = _fields(B, n, 1000 + n + flags, dev)
    w = (0.7, 1.3, 9.0, 11.0)
    nl, tb, correct = bool(flags & 1), not (flags & 2), not (flags & 4)
    terms, grad = darcy.darcy_loss_launch(Kd, yd, w, True, nl, 0.1, 0.2, tb, correct)
    rt, rg = od.loss_and_grad_autograd(torch.from_numpy(K).double(), torch.from_numpy(y).double(), 10.0, 0.1, 0.2, nl,
                                       weights=w, correct=correct, use_tb=tb)
    ref = [float(w[0] * rt[1] + w[1] * rt[2] + w[2] * rt[3] + w[3] * rt[4])] + [float(v) for v in rt[1:]]
    np.testing.assert_allclose(terms.cpu().numpy(), ref, rtol=LOSS_RTOL)
    assert rel_l2(grad.cpu().numpy(), rg.numpy()) < GRAD_RL2
    terms2, g2 = darcy.darcy_loss_launch(Kd, yd, w, False, nl, 0.1, 0.2, tb, correct)      # forward only (eval path)
    assert g2 is None
    np.testing.assert_allclose(terms2.cpu().numpy(), terms.cpu().numpy(), rtol=1e-6)


BAND_CASES = [(n, B, f) for n, B in [(8, 5), (9, 3), (10, 3), (11, 3), (17, 4), (20, 7), (48, 3), (63, 2), (65, 5), (66, 2), (67, 2),
                                     (96, 2), (100, 2), (128, 3), (129, 2), (131, 1), (200, 2), (253, 1), (256, 2)]
              for f in (0, 4, 3)]


@pytest.mark.parametrize('n,B,flags', BAND_CASES)
def test_band_kernel_vs_oracle_and_tile_kernel(dev, n, B, flags, monkeypatch):
    """the row-band kernel (csrc/darcy_band.h: 8 <= n <= 256) against the fp64 oracle and against the tile kernel
    (PDES_LOSS_TILED) on the same inputs: every width class (n mod 4), 32 rows per wave down to one, one band and many,
    16-byte and scalar accesses; correct=False (4), nonlinear + use_tb=False (3)"""
    from oracle import darcy as od
    from pde_surrogate_amd.models import darcy
    K, y, Kd, yd = _fields(B, n, 4000 + n + flags, dev)
    w = (0.7, 1.3, 9.0, 11.0)
    nl, tb, correct = bool(flags & 1), not (flags & 2), not (flags & 4)
    monkeypatch.setattr(darcy, 'EXTRA_FLAGS', 16)                   # (64 would otherwise be the specialised kernel)
    terms, grad = darcy.darcy_loss_launch(Kd, yd, w, True, nl, 0.1, 0.2, tb, correct)
    rt, rg = od.loss_and_grad_autograd(torch.from_numpy(K).double(), torch.from_numpy(y).double(), 10.0, 0.1, 0.2, nl,
                                       weights=w, correct=correct, use_tb=tb)
    ref = [float(w[0] * rt[1] + w[1] * rt[2] + w[2] * rt[3] + w[3] * rt[4])] + [float(v) for v in rt[1:]]
    np.testing.assert_allclose(terms.cpu().numpy(), ref, rtol=LOSS_RTOL)
    assert torch.isfinite(grad).all()
    assert rel_l2(grad.cpu().numpy(), rg.numpy()) < GRAD_RL2
    terms2, g2 = darcy.darcy_loss_launch(Kd, yd, w, False, nl, 0.1, 0.2, tb, correct)      # forward only: the same sums
    assert g2 is None
    np.testing.assert_allclose(terms2.cpu().numpy(), terms.cpu().numpy(), rtol=1e-6)      # (another instantiation: other contractions)
    monkeypatch.setattr(darcy, 'EXTRA_FLAGS', 16 | 8)
    terms_t, grad_t = darcy.darcy_loss_launch(Kd, yd, w, True, nl, 0.1, 0.2, tb, correct)
    np.testing.assert_allclose(terms.cpu().numpy(), terms_t.cpu().numpy(), rtol=2e-6)
    assert rel_l2(grad.cpu().numpy(), grad_t.cpu().numpy()) < 2e-6
    # an unaligned base pointer takes the scalar-access path of the same kernel
    if n % 4 == 0:
        monkeypatch.setattr(darcy, 'EXTRA_FLAGS', 16)
        pad = torch.empty(Kd.numel() + 1, device=dev)
        Ku = pad[1:].view_as(Kd).copy_(Kd)
        assert Ku.data_ptr() % 16 == 4
        terms_u, grad_u = darcy.darcy_loss_launch(Ku, yd, w, True, nl, 0.1, 0.2, tb, correct)
        np.testing.assert_allclose(terms_u.cpu().numpy(), terms.cpu().numpy(), rtol=1e-6)      # (the general instantiation)
        assert rel_l2(grad_u.cpu().numpy(), grad.cpu().numpy()) < 1e-6


@pytest.mark.parametrize('n', [16, 32, 64])
def test_band_kernel_equals_the_specialised_kernels(dev, n, monkeypatch):
    """at 16 / 32 / 64 the row-band kernel is the specialised kernel without the compile-time size: same loss terms and
    gradient to rounding (1e-6), linear and nonlinear, B = 32 and one odd batch"""
    from pde_surrogate_amd.models import darcy
    for B in (32, 5):
        for nl in (False, True):
            _, _, Kd, yd = _fields(B, n, 77 + n + B, dev)
            monkeypatch.setattr(darcy, 'EXTRA_FLAGS', 0)
            t0, g0 = darcy.darcy_loss_launch(Kd, yd, (1, 1, 10, 10), True, nl, 0.1, 0.1)
            monkeypatch.setattr(darcy, 'EXTRA_FLAGS', 16)
            t1, g1 = darcy.darcy_loss_launch(Kd, yd, (1, 1, 10, 10), True, nl, 0.1, 0.1)
            np.testing.assert_allclose(t1.cpu().numpy(), t0.cpu().numpy(), rtol=1e-6)
            assert rel_l2(g1.cpu().numpy(), g0.cpu().numpy()) < 1e-6


def test_band_kernel_is_deterministic_and_its_partial_rows_are_its_bands(dev):
    from pde_surrogate_amd import _lib
    from pde_surrogate_amd.models import darcy
    torch.manual_seed(5)
    K = torch.exp(0.5 * torch.randn(6, 1, 65, 65, device=dev))
    y = torch.randn(6, 3, 65, 65, device=dev)
    t0, g0 = darcy.darcy_loss_launch(K, y, (1, 1, 10, 10), True)
    t1, g1 = darcy.darcy_loss_launch(K, y, (1, 1, 10, 10), True)
    assert torch.equal(t0, t1) and torch.equal(g0, g1)
    rows = _lib.loss_partial_rows(6, 65, 65, 0)
    assert rows % 6 == 0 and 2 <= rows // 6 <= 8                          # a few bands per image (the plan of darcy_band.h)
    assert _lib.loss_partial_rows(6, 65, 65, 8) != _lib.loss_partial_rows(6, 65, 65, 0)      # PDES_LOSS_TILED: the tile kernel's rows
    assert _lib.loss_partial_rows(6, 64, 64, 0) == 6 and _lib.loss_partial_rows(6, 64, 64, 16) >= 6


def test_generic_kernel_is_deterministic(dev):
    """the tiled kernel keeps the per-image sums inside one workgroup (fixed-order reductions): same inputs twice are
    bit-identical, loss terms and gradient"""
    from pde_surrogate_amd.models import darcy
    torch.manual_seed(3)
    K = torch.exp(0.5 * torch.randn(4, 1, 96, 96, device=dev))
    y = torch.randn(4, 3, 96, 96, device=dev)
    t0, g0 = darcy.darcy_loss_launch(K, y, (1, 1, 10, 10), True)
    t1, g1 = darcy.darcy_loss_launch(K, y, (1, 1, 10, 10), True)
    assert torch.equal(t0, t1) and torch.equal(g0, g1)


def test_edge_pixels_sharp_interface(dev):
    """channelized-like inputs (config 4): two-valued K, step fields -- exercises every edge formula"""
    from oracle import darcy as od
    from pde_surrogate_amd.models import darcy
    rng = np.random.default_rng(5)
    K = np.where(rng.random((4, 1, 64, 64)) > 0.5, 10.0, 1.0).astype(np.float32)
    y = np.sign(rng.standard_normal((4, 3, 64, 64))).astype(np.float32)
    y[:, :, :3, :] *= 5
    y[:, :, -3:, :] *= -4
    y[:, :, :, :3] *= 3
    y[:, :, :, -3:] *= -2
    terms, grad = darcy.darcy_loss_launch(torch.from_numpy(K).to(dev), torch.from_numpy(y).to(dev),
                                          (1, 1, 10, 10), True)
    rt, rg = od.loss_and_grad_analytic(K, y, 10.0)
    np.testing.assert_allclose(terms.cpu().numpy(), rt, rtol=LOSS_RTOL)
    assert rel_l2(grad.cpu().numpy(), rg) < GRAD_RL2
    # edge rows/cols individually (a wrong boundary stencil hides in a global norm)
    gg = grad.cpu().numpy()
    for sl in (np.s_[:, :, :3, :], np.s_[:, :, -3:, :], np.s_[:, :, :, :3], np.s_[:, :, :, -3:]):
        assert rel_l2(gg[sl], rg[sl]) < 1e-5


def test_full_size_properties(dev):
    """B = 4096 (ntrain of config 2): size-independent properties instead of a CPU re-computation:
    (a) loss == mean over chunks of the chunk losses; (b) gradient of the big batch == chunk
    gradients scaled by chunk/B; (c) sample permutation permutes gradients; (d) oracle on a slice."""
    from oracle import darcy as od
    from pde_surrogate_amd.models import darcy
    B, n, C = 4096, 64, 512
    gen = torch.Generator(device='cpu').manual_seed(11)
    K = torch.exp(0.5 * torch.randn((B, 1, n, n), generator=gen)).to(dev)
    y = torch.randn((B, 3, n, n), generator=gen).to(dev)
    w = (1, 1, 10, 10)
    terms, grad = darcy.darcy_loss_launch(K, y, w, True)
    chunk_terms = []
    for i in range(0, B, C):
        t, g = darcy.darcy_loss_launch(K[i:i + C].contiguous(), y[i:i + C].contiguous(), w, True)
        chunk_terms.append(t.double().cpu().numpy())
        assert rel_l2(grad[i:i + C].cpu().numpy(), g.cpu().numpy() * (C / B)) < 2e-6
    np.testing.assert_allclose(terms.cpu().numpy(), np.mean(chunk_terms, 0), rtol=LOSS_RTOL)
    perm = torch.randperm(B, generator=gen).to(dev)
    t2, g2 = darcy.darcy_loss_launch(K[perm].contiguous(), y[perm].contiguous(), w, True)
    np.testing.assert_allclose(t2.cpu().numpy(), terms.cpu().numpy(), rtol=LOSS_RTOL)
    assert torch.equal(g2, grad[perm])
    rt, rg = od.loss_and_grad_analytic(K[:4].cpu().numpy(), y[:4].cpu().numpy(), 10.0)
    assert rel_l2(grad[:4].cpu().numpy() * (B / 4), rg) < GRAD_RL2


def test_rejects_bad_arguments(dev):
    from pde_surrogate_amd.models import darcy
    from pde_surrogate_amd.utils.image_gradient import SobelFilter
    K, y, Kd, yd = _fields(2, 64, 3, dev)
    with pytest.raises(RuntimeError):      # CPU tensors: no fallback
        darcy.darcy_mixed_residual_loss(torch.from_numpy(K), torch.from_numpy(y))
    with pytest.raises(RuntimeError):      # non-square fields: SobelFilter has one imsize x imsize modifier
        darcy.darcy_loss_launch(Kd[:, :, :48, :].contiguous(), yd[:, :, :48, :].contiguous(), (1, 1, 1, 1), True)
    with pytest.raises(ValueError):        # a filter built for another size
        darcy.conv_constitutive_constraint(Kd, yd, SobelFilter(32, correct=True, device=dev))


def test_g14_continuity_without_top_bottom_rows_and_5x5_sobel(dev):
    """conv_continuity_constraint(use_tb=False) (darcy.py:224) value + gradient, and SobelFilter.grad_h/grad_v with
    filter_size=5 (image_gradient.py:65-67, :82-84) incl. their autograd adjoint against the fp64 oracle"""
    from oracle import darcy as od
    from pde_surrogate_amd.models import darcy
    from pde_surrogate_amd.utils.image_gradient import SobelFilter
    g = golden('G14_no_tb_sobel5.npz')
    sob = SobelFilter(64, correct=True, device=dev)
    y = torch.from_numpy(g['y']).to(dev).requires_grad_(True)
    lt = darcy.conv_continuity_constraint(y, sob, use_tb=False)
    np.testing.assert_allclose(float(lt.detach()), float(g['cont_no_tb']), rtol=LOSS_RTOL)
    lt.backward()
    assert rel_l2(y.grad.cpu().numpy(), g['cont_no_tb_grad']) < 1e-5
    # use_tb=True on the same field still gives the other value (the flag is not sticky)
    assert abs(float(darcy.conv_continuity_constraint(y.detach(), sob)) - float(g['cont_no_tb'])) > 1e-3 * float(g['cont_no_tb'])
    img = torch.from_numpy(g['img']).to(dev)
    np.testing.assert_allclose(sob.grad_h(img, filter_size=5).cpu().numpy(), g['gh5'], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(sob.grad_v(img, filter_size=5).cpu().numpy(), g['gv5'], rtol=1e-5, atol=1e-4)
    for n in (64, 16, 8):                                   # adjoint: <grad(x), w> gradient vs the oracle's autograd
        torch.manual_seed(n)
        xi = torch.randn(2, 1, n, n)
        wh, wv = torch.randn(2, 1, n, n), torch.randn(2, 1, n, n)
        xo = xi.double().requires_grad_(True)
        ((od.sobel_grad_h5(xo) * wh.double()).sum() + (od.sobel_grad_v5(xo) * wv.double()).sum()).backward()
        xd = xi.to(dev).requires_grad_(True)
        s = SobelFilter(n, correct=True, device=dev)
        ((s.grad_h(xd, 5) * wh.to(dev)).sum() + (s.grad_v(xd, 5) * wv.to(dev)).sum()).backward()
        assert rel_l2(xd.grad.cpu().numpy(), xo.grad.numpy()) < 1e-5, n
    with pytest.raises(ValueError):
        sob.grad_h(img, filter_size=7)
