"""CPU tests of oracle/fd_newton.py, the build's own stand-in for the reference's FEniCS validation of config 5
(utils/fenics.py:13-91; parity UNPINNED -- dolfin is unavailable; these are self-consistency checks only)."""
import numpy as np
import torch

from oracle import darcy as od
from oracle.fd_newton import solve_nonlinear_darcy


def _cubic_root(g, K, a1, a2):
    r = np.roots([a2 * K, a1 * np.sqrt(K), 1.0, -g])
    r = r[np.abs(r.imag) < 1e-12].real
    return r[np.argmin(np.abs(r - g))]


def test_uniform_and_layered_permeability_have_closed_forms():
    n = 64
    out, info = solve_nonlinear_darcy(np.full((n, n), 2.5))
    x = np.linspace(0, 1, n)
    np.testing.assert_allclose(out[0], np.broadcast_to(1 - x, (n, n)), atol=1e-12)
    np.testing.assert_allclose(out[1], 2.5, rtol=1e-12)
    np.testing.assert_allclose(out[2], 0.0, atol=1e-12)
    # K = K(y): every row is a 1-D problem with du/dx = -1, so sigma1 solves the cubic with g = K(y)
    Kr = np.exp(0.5 * np.sin(2 * np.pi * np.linspace(0, 1, n)))[:, None] * np.ones((1, n))
    out, info = solve_nonlinear_darcy(Kr, 0.1, 0.1)
    assert info['newton_residuals'][-1] < 1e-11
    np.testing.assert_allclose(out[0], np.broadcast_to(1 - x, (n, n)), atol=1e-10)
    want = np.array([_cubic_root(k, k, 0.1, 0.1) for k in Kr[:, 0]])
    np.testing.assert_allclose(out[1], want[:, None] * np.ones((1, n)), rtol=1e-10)


def test_random_field_converges_conserves_mass_and_has_a_small_mixed_residual():
    from pde_surrogate_amd.utils.data import grf_kle_fields
    K = grf_kle_fields(3, n_kle=128, seed=4, cache_dir='/tmp')[2, 0].astype(np.float64)
    lin, info_l = solve_nonlinear_darcy(K)
    nl, info = solve_nonlinear_darcy(K, 0.1, 0.1)
    assert info['newton_residuals'][-1] < 1e-11 and len(info['newton_residuals']) < 15
    tf = info['throughflow']
    assert np.ptp(tf) < 1e-10 * abs(tf.mean())                     # the same flow crosses every vertical line
    assert abs(tf.mean()) < abs(info_l['throughflow'].mean())      # the nonlinear drag lowers the flow rate
    assert 0.0 <= nl[0].min() and nl[0].max() <= 1.0               # maximum principle
    # the Sobel-based mixed residual (a different discretisation, spacing 1/64 instead of 1/63) of this solution is
    # small against that of a guess that ignores K, for the matching law only
    Kt = torch.from_numpy(K)[None, None]
    loss = lambda f, b: float(od.mixed_residual_loss(Kt, torch.from_numpy(f)[None], 10.0, b, b, b > 0)[0])
    guess = np.stack([np.broadcast_to(1 - np.linspace(0, 1, 64), (64, 64)), np.ones((64, 64)), np.zeros((64, 64))])
    assert loss(nl, 0.1) < 0.02 * loss(guess, 0.1)
    assert loss(lin, 0.0) < 0.02 * loss(guess, 0.0)
    assert loss(nl, 0.1) < loss(lin, 0.1)                          # each solution fits its own law best
