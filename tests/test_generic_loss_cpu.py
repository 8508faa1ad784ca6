"""CPU: the arithmetic and tile geometry of the any-size Sobel / Darcy-loss kernels (csrc/darcy_generic.h, shared by
csrc/darcy_loss_generic.hip) compiled as plain C++ (tests/emu/darcy_generic_emu.cpp) against the oracle -- every halo
case (tiles of 1..n rows / columns, first / last two rows and columns, n = 2 .. 130), correct=True/False, use_tb,
the nonlinear law, the 5x5 filter.  fp32 emulation vs fp64 oracle: the tolerances of the GPU tests apply."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import darcy as od

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F, I, P, LL = ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong
BUDGET = 10240          # floats of LDS the kernel has (GEN_LDSF)
BIG = 1 << 20           # (whole-image / forced-tile cases: the arithmetic, not the fit)


@pytest.fixture(scope='module')
def emu(tmp_path_factory):
    so = str(tmp_path_factory.mktemp('emu') / 'libdarcy_emu.so')
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-ffp-contract=off',
                           os.path.join(ROOT, 'tests', 'emu', 'darcy_generic_emu.cpp'), '-o', so])
    L = ctypes.CDLL(so)
    L.emu_darcy_loss.argtypes = [P, P, P, P, I, I, F, F, F, F, F, F, I, I, I, LL]
    L.emu_sobel.argtypes = [P, P, P, I, I, I, I]
    L.emu_sobel_adjoint.argtypes = [P, P, P, I, I, I, I]
    L.emu_choose_tile.argtypes = [I, LL, P, P]
    L.emu_tile_floats.argtypes = [I, I, I]
    L.emu_tile_floats.restype = LL
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(P)


def _loss(emu, K, y, weights, flags=0, b1=0.0, b2=0.0, tr=0, tc=0, grad=True):
    B, _, n, _ = y.shape
    ncont = B * (n - 2) * n if flags & 2 else B * n * n
    gy = np.full_like(y, np.nan) if grad else None
    part = np.zeros((B * n * n, 4), np.float32)             # (at most one tile per pixel)
    nt = emu.emu_darcy_loss(_p(K), _p(y), _p(gy), _p(part), B, n, 2.0 * weights[0] / (B * n * n), 2.0 * weights[1] / ncont,
                            2.0 * weights[2] / (B * n), 2.0 * weights[3] / (2 * B * n), b1, b2, flags, tr, tc, BIG if tr else BUDGET)
    assert nt > 0
    s = part.astype(np.float64).sum(0)
    terms = np.array([s[0] / (B * n * n), s[1] / ncont, s[2] / (B * n), s[3] / (2 * B * n)])
    return terms, gy, nt


def _fields(B, n, seed):
    rng = np.random.default_rng(seed)
    K = np.exp(0.5 * rng.standard_normal((B, 1, n, n))).astype(np.float32)
    y = rng.standard_normal((B, 3, n, n)).astype(np.float32)
    return K, y


def _oracle(K, y, weights, flags=0, b1=0.0, b2=0.0):
    terms, g = od.loss_and_grad_autograd(torch.from_numpy(K).double(), torch.from_numpy(y).double(), 10.0, b1, b2,
                                         bool(flags & 1), weights=weights, correct=not (flags & 4), use_tb=not (flags & 2))
    return np.array([float(t) for t in terms[1:]]), g.numpy()


@pytest.mark.parametrize('n', [2, 3, 4, 5, 7, 12, 20, 33, 48, 56])
@pytest.mark.parametrize('flags', [0, 4])
def test_loss_whole_image_tiles(emu, n, flags):
    K, y = _fields(2, n, 100 + n)
    w = (1.0, 1.0, 10.0, 10.0)
    terms, gy, nt = _loss(emu, K, y, w, flags, tr=n, tc=n)
    ref_t, ref_g = _oracle(K, y, w, flags)
    np.testing.assert_allclose(terms, ref_t, rtol=1e-5)
    assert rel_l2(gy, ref_g) < 1e-5
    assert nt == 1


@pytest.mark.parametrize('n,tr,tc', [(5, 1, 1), (7, 2, 3), (9, 1, 9), (9, 9, 1), (16, 3, 5), (21, 4, 21), (33, 7, 6),
                                     (40, 13, 40), (65, 33, 65), (65, 16, 20)])
@pytest.mark.parametrize('flags', [0, 4, 2, 1])
def test_loss_tiled_equals_oracle(emu, n, tr, tc, flags):
    """forced small tiles: every combination of halo clipping at the image edges and tile seams"""
    K, y = _fields(2, n, 7 * n + tr)
    w = (0.7, 1.3, 9.0, 11.0)
    terms, gy, nt = _loss(emu, K, y, w, flags, 0.1, 0.2, tr, tc)
    ref_t, ref_g = _oracle(K, y, w, flags, 0.1, 0.2)
    np.testing.assert_allclose(terms, ref_t, rtol=1e-5)
    assert rel_l2(gy, ref_g) < 1e-5
    tce = (tc + 3) // 4 * 4 if n >= 8 else tc            # the strip form (n >= 8) cuts columns on strip boundaries
    assert nt == -(-n // tr) * -(-n // tce)


@pytest.mark.parametrize('n', [17, 64, 65, 96, 128, 130, 300, 1000])
def test_loss_kernel_tile_choice(emu, n):
    """the tile the kernel picks for its 128 KiB of LDS fits, covers the image, and gives the oracle's loss"""
    tr, tc = I(0), I(0)
    assert emu.emu_choose_tile(n, BUDGET, ctypes.byref(tr), ctypes.byref(tc)) == 1
    assert emu.emu_tile_floats(tr.value, tc.value, n) <= BUDGET
    assert 1 <= tr.value <= n and tc.value >= 1
    if n > 130:
        return                                    # geometry only (the fp64 oracle at 300 x 300 is slow)
    K, y = _fields(1, n, n)
    w = (1.0, 1.0, 10.0, 10.0)
    terms, gy, nt = _loss(emu, K, y, w)
    ref_t, ref_g = _oracle(K, y, w)
    np.testing.assert_allclose(terms, ref_t, rtol=1e-5)
    assert rel_l2(gy, ref_g) < 1e-5
    assert nt == -(-n // tr.value) * -(-n // tc.value)
    assert tc.value % 4 == 0 or tc.value >= n


def test_forward_only_leaves_no_gradient(emu):
    K, y = _fields(1, 20, 5)
    t0, _, _ = _loss(emu, K, y, (1, 1, 10, 10), grad=False, tr=6, tc=20)
    t1, _, _ = _loss(emu, K, y, (1, 1, 10, 10), grad=True, tr=6, tc=20)
    np.testing.assert_array_equal(t0, t1)


@pytest.mark.parametrize('n', [2, 3, 4, 6, 17, 65])
@pytest.mark.parametrize('correct', [1, 0])
@pytest.mark.parametrize('five', [0, 1])
def test_sobel_and_adjoint(emu, n, correct, five):
    rng = np.random.default_rng(n + 10 * correct + 100 * five)
    img = (rng.standard_normal((3, 1, n, n)) * 2 + 0.5).astype(np.float32)
    gh, gv = np.empty_like(img), np.empty_like(img)
    emu.emu_sobel(_p(img), _p(gh), _p(gv), 3, n, correct, five)
    fh, fv = (od.sobel_grad_h5, od.sobel_grad_v5) if five else (od.sobel_grad_h, od.sobel_grad_v)
    t = torch.from_numpy(img).double().requires_grad_(True)
    rh, rv = fh(t, bool(correct)), fv(t, bool(correct))
    np.testing.assert_allclose(gh, rh.detach().numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(gv, rv.detach().numpy(), rtol=1e-5, atol=1e-4)
    wh = rng.standard_normal(img.shape).astype(np.float32)
    wv = rng.standard_normal(img.shape).astype(np.float32)
    ((rh * torch.from_numpy(wh).double()).sum() + (rv * torch.from_numpy(wv).double()).sum()).backward()
    out = np.empty_like(img)
    emu.emu_sobel_adjoint(_p(wh), _p(wv), _p(out), 3, n, correct, five)
    assert rel_l2(out, t.grad.numpy()) < 1e-5
    # one-sided calls (grad_h alone / grad_v alone)
    emu.emu_sobel_adjoint(_p(wh), None, _p(out), 3, n, correct, five)
    t2 = torch.from_numpy(img).double().requires_grad_(True)
    (fh(t2, bool(correct)) * torch.from_numpy(wh).double()).sum().backward()
    assert rel_l2(out, t2.grad.numpy()) < 1e-5


# ---- against the REAL reference (tests/golden/G21_any_size.npz, tools/gen_golden.py round3) --------------------------
@pytest.mark.parametrize('n,correct', [(20, True), (20, False), (48, True), (48, False), (65, True), (65, False), (128, True)])
def test_g21_reference_fields_and_loss(emu, n, correct):
    from conftest import golden
    g = golden('G21_any_size.npz')
    sfx = '' if correct else '_nocorrect'
    K, y, img = g[f'K{n}'], g[f'y{n}'], g[f'img{n}']
    B = K.shape[0]
    # oracle (fp64) and emulation (fp32 kernel arithmetic) against the reference's fp32 outputs
    t = torch.from_numpy(img).double()
    np.testing.assert_allclose(od.sobel_grad_h(t, correct).numpy(), g[f'gh{n}{sfx}'], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(od.sobel_grad_v(t, correct).numpy(), g[f'gv{n}{sfx}'], rtol=1e-5, atol=1e-4)
    gh, gv = np.empty_like(img), np.empty_like(img)
    emu.emu_sobel(_p(img), _p(gh), _p(gv), B, n, int(correct), 0)
    np.testing.assert_allclose(gh, g[f'gh{n}{sfx}'], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(gv, g[f'gv{n}{sfx}'], rtol=1e-5, atol=1e-4)
    flags = 0 if correct else 4
    w = (1.0, 1.0, 10.0, 10.0)
    ref = g[f'terms{n}{sfx}']
    ot, og = _oracle(K, y, w, flags)
    np.testing.assert_allclose(ot, ref[1:], rtol=1e-5)
    assert rel_l2(og, g[f'grad{n}{sfx}']) < 1e-5
    terms, gy, _ = _loss(emu, K, y, w, flags)
    np.testing.assert_allclose(terms, ref[1:], rtol=1e-5)
    np.testing.assert_allclose(terms[0] + terms[1] + 10.0 * (terms[2] + terms[3]), ref[0], rtol=1e-5)
    assert rel_l2(gy, g[f'grad{n}{sfx}']) < 1e-5


def test_g21_reference_variants_at_65(emu):
    from conftest import golden
    g = golden('G21_any_size.npz')
    K, y, img = g['K65'], g['y65'], g['img65']
    w = (1.0, 1.0, 10.0, 10.0)
    terms, gy, _ = _loss(emu, K, y, w, 1, 0.1, 0.1)                # nonlinear law
    np.testing.assert_allclose(terms, g['terms65_nl'][1:], rtol=1e-5)
    assert rel_l2(gy, g['grad65_nl']) < 1e-5
    terms, gy, _ = _loss(emu, K, y, (0.0, 1.0, 0.0, 0.0), 2)       # use_tb=False, continuity term alone
    np.testing.assert_allclose(terms[1], float(g['cont65_no_tb']), rtol=1e-5)
    assert rel_l2(gy, g['grad65_no_tb']) < 1e-5
    ot, og = _oracle(K, y, (0.0, 1.0, 0.0, 0.0), 2)
    np.testing.assert_allclose(ot[1], float(g['cont65_no_tb']), rtol=1e-5)
    wh, wv = g['wh65'], g['wv65']
    for correct in (1, 0):
        sfx = '' if correct else '_nocorrect'
        gh, gv, out = np.empty_like(img), np.empty_like(img), np.empty_like(img)
        emu.emu_sobel(_p(img), _p(gh), _p(gv), 2, 65, correct, 1)
        np.testing.assert_allclose(gh, g['gh65_f5' + sfx], rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(gv, g['gv65_f5' + sfx], rtol=1e-5, atol=1e-4)
        for five in (0, 1):
            emu.emu_sobel_adjoint(_p(wh), _p(wv), _p(out), 2, 65, correct, five)
            assert rel_l2(out, g[f'adj65_f{3 + 2 * five}{sfx}']) < 1e-5


# ---------------------------------------------------------------------------------------------------- the row-band kernel
BAND_LDS = 16384 - 64          # floats of LDS the band kernel may use (BAND_LDSF of darcy_loss_generic.hip)


def _band(emu, K, y, weights, flags=0, b1=0.0, b2=0.0, plan=(0, 0, 0), grad=True):
    B, _, n, _ = y.shape
    ncont = B * (n - 2) * n if flags & 2 else B * n * n
    gy = np.full_like(y, np.nan) if grad else None
    part = np.zeros((B * n, 4), np.float32)
    out = (ctypes.c_int * 6)()
    nb = emu.emu_darcy_loss_band(_p(K), _p(y), _p(gy), _p(part), B, n, 2.0 * weights[0] / (B * n * n), 2.0 * weights[1] / ncont,
                                 2.0 * weights[2] / (B * n), 2.0 * weights[3] / (2 * B * n), b1, b2, flags, plan[0], plan[1],
                                 plan[2], BAND_LDS if not plan[0] else 1 << 22, out)
    if nb <= 0:
        return None, None, 0, None
    s = part[:B * nb].astype(np.float64).sum(0)
    terms = np.array([s[0] / (B * n * n), s[1] / ncont, s[2] / (B * n), s[3] / (2 * B * n)])
    return terms, gy, nb, list(out)


@pytest.fixture(scope='module')
def bemu(emu):
    emu.emu_darcy_loss_band.argtypes = [P, P, P, P, I, I, F, F, F, F, F, F, I, I, I, I, LL, P]
    emu.emu_band_plan.argtypes = [I, LL, P]
    return emu


@pytest.mark.parametrize('n', [8, 9, 10, 11, 12, 20, 33, 48, 63, 64, 65, 66, 67, 100, 128, 129, 131, 200, 253, 256])
@pytest.mark.parametrize('flags', [0, 4])
def test_band_kernel_own_plan_vs_oracle(bemu, n, flags):
    """the row-band form (csrc/darcy_band.h) with the plan the kernel itself chooses: every width class (n mod 4, one to
    three partial columns in the last strip), rows that run across wave boundaries (65: 15 rows of 17 strips in four waves)
    and rows that do not, one band and many, correct=True/False"""
    K, y = _fields(2, n, 300 + n)
    w = (1.0, 1.0, 10.0, 10.0)
    terms, gy, nb, plan = _band(bemu, K, y, w, flags)
    assert nb > 0, 'no plan'
    ref_t, ref_g = _oracle(K, y, w, flags)
    np.testing.assert_allclose(terms, ref_t, rtol=1e-5)
    assert np.isfinite(gy).all()
    assert rel_l2(gy, ref_g) < 1e-5
    assert plan[4] <= BAND_LDS
    if n % 4 == 0:              # the general instantiation (what an unaligned pointer selects) at a multiple of 4: same numbers
        t2, g2, _, _ = _band(bemu, K, y, w, flags | 256)
        np.testing.assert_allclose(t2, terms, rtol=1e-6)
        assert rel_l2(g2, gy) < 1e-6


@pytest.mark.parametrize('n,plan', [(8, (1, 1, 1)), (9, (1, 1, 3)), (12, (1, 1, 4)), (17, (1, 1, 2)), (17, (2, 2, 1)), (20, (1, 1, 6)),
                                    (33, (2, 1, 3)), (33, (4, 2, 1)), (65, (4, 2, 3)), (65, (8, 1, 3)), (65, (2, 2, 7)),
                                    (66, (4, 1, 7)), (67, (8, 2, 2)), (128, (8, 2, 5)), (128, (4, 2, 10)), (130, (8, 2, 10)),
                                    (65, (4, 1, 5)), (65, (7, 1, 3)), (130, (7, 2, 6)), (100, (5, 1, 10)), (131, (3, 2, 17)),
                                    (253, (8, 2, 19))])
@pytest.mark.parametrize('flags', [0, 1, 2, 4, 6])
def test_band_kernel_forced_plans_and_flags(bemu, n, plan, flags):
    """forced workgroup shapes / band counts (bands of 3 rows up to the whole image; halo rows at both ends), with the
    nonlinear law, use_tb=False, correct=False and the two together"""
    K, y = _fields(1, n, 500 + n + 7 * flags)
    w = (1.0, 1.0, 10.0, 10.0)
    b1, b2 = (0.1, 0.1) if flags & 1 else (0.0, 0.0)
    terms, gy, nb, _ = _band(bemu, K, y, w, flags, b1, b2, plan)
    assert nb == plan[2], 'the forced plan does not fit'
    ref_t, ref_g = _oracle(K, y, w, flags, b1, b2)
    np.testing.assert_allclose(terms, ref_t, rtol=1e-5)
    assert np.isfinite(gy).all()
    assert rel_l2(gy, ref_g) < 1e-5
    # forward only: the same sums
    t2, _, _, _ = _band(bemu, K, y, w, flags, b1, b2, plan, grad=False)
    np.testing.assert_array_equal(t2, terms)


def test_band_plans_cover_8_to_256(bemu):
    out = (ctypes.c_int * 6)()
    for n in range(8, 257):
        assert bemu.emu_band_plan(n, BAND_LDS, out), n
        waves, npass, nbands, rpp, lds, cap = list(out)
        assert lds <= BAND_LDS and n // nbands >= 3 and (nbands == 1 and cap >= n or -(-n // nbands) + 2 <= cap), (n, list(out))
    assert not bemu.emu_band_plan(7, BAND_LDS, out) and not bemu.emu_band_plan(257, BAND_LDS, out)
