"""Oracle (test infrastructure): an INDEPENDENT numerical solution of the (non)linear Darcy problem of config 5.

    sigma + K grad(u) + a1 sqrt(K) sigma^2 + a2 K sigma^3 = 0   (componentwise powers, as in the reference)
    div(sigma) = 0 in (0,1)^2,   u = 1 on x = 0,  u = 0 on x = 1,  sigma . n = 0 on y = 0 and y = 1

The reference validates `solve_conv_mixed_residual.py --nonlinear` against a FEniCS mixed finite-element solve
(utils/fenics.py:13-91: DRT3 x CG4 on UnitSquareMesh(ngy-1, ngx-1), K interpolated from the vertex values, Newton
abs 1e-8 / rel 1e-6).  dolfin is not installable here and the reference stores none of its outputs, so that comparison
is **parity unpinned**.  This file is a stand-in of the build's own: a vertex-centred finite-volume discretisation on
the same 64 x 64 vertex grid (h = 1/63, K at the vertices, harmonic face means), the constitutive law inverted per
face, Newton on the pressure with an exact sparse Jacobian, everything in fp64.  It shares no code and no stencil with
the Sobel-based loss, so agreement between a decoder trained on the mixed residual and this solution is a
SELF-CONSISTENCY check of the whole config-5 path (loss kernel, decoder, L-BFGS), not parity with FEniCS.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def _flux_of(g, K, a1, a2, iters=60):
    """solve s + a1 sqrt(K) s^2 + a2 K s^3 = g for s (scalar Newton per face, vectorised); returns (s, ds/dg)"""
    s = g.copy()
    sq = np.sqrt(K)
    for _ in range(iters):
        f = s + a1 * sq * s * s + a2 * K * s ** 3 - g
        d = 1.0 + 2.0 * a1 * sq * s + 3.0 * a2 * K * s * s
        step = f / d
        s = s - step
        if np.max(np.abs(step)) < 1e-15 * max(1.0, np.max(np.abs(s))):
            break
    d = 1.0 + 2.0 * a1 * sq * s + 3.0 * a2 * K * s * s
    return s, 1.0 / d


def solve_nonlinear_darcy(K, alpha1=0.0, alpha2=0.0, tol=1e-11, max_newton=50):
    """K: (n, n) permeability at the vertices (rows = y, columns = x).  Returns (out, info) with out (3, n, n) =
    (u, sigma1, sigma2) at the vertices (fluxes averaged from the adjacent faces) and info = Newton history."""
    K = np.asarray(K, np.float64)
    n = K.shape[0]
    assert K.shape == (n, n)
    h = 1.0 / (n - 1)
    idx = np.arange(n * n).reshape(n, n)
    u = np.broadcast_to(1.0 - np.linspace(0.0, 1.0, n)[None, :], (n, n)).copy()
    # faces: horizontal neighbours (x-faces) and vertical neighbours (y-faces); harmonic means of K
    Kx = 2.0 * K[:, :-1] * K[:, 1:] / (K[:, :-1] + K[:, 1:])        # (n, n-1)
    Ky = 2.0 * K[:-1, :] * K[1:, :] / (K[:-1, :] + K[1:, :])        # (n-1, n)
    # control-volume face lengths: boundary rows / columns own half cells
    wy = np.full(n, h); wy[0] = wy[-1] = h / 2                       # length of an x-face in row i
    wx = np.full(n, h); wx[0] = wx[-1] = h / 2                       # length of a y-face in column j
    free = np.ones((n, n), bool)
    free[:, 0] = free[:, -1] = False                                  # Dirichlet columns
    hist = []
    for it in range(max_newton):
        gx = -Kx * (u[:, 1:] - u[:, :-1]) / h
        gy = -Ky * (u[1:, :] - u[:-1, :]) / h
        sx, dsx = _flux_of(gx, Kx, alpha1, alpha2)
        sy, dsy = _flux_of(gy, Ky, alpha1, alpha2)
        # net outflow of every control volume
        R = np.zeros((n, n))
        fx = sx * wy[:, None]
        fy = sy * wx[None, :]
        R[:, :-1] += fx; R[:, 1:] -= fx
        R[:-1, :] += fy; R[1:, :] -= fy
        res = float(np.max(np.abs(R[free])))
        hist.append(res)
        if res < tol:
            break
        # Jacobian dR/du: face flux F = w * s(g), g = -Kf (u_b - u_a) / h  =>  dF/du_a = w ds Kf / h, dF/du_b = -that
        cx = dsx * Kx / h * wy[:, None]
        cy = dsy * Ky / h * wx[None, :]
        a, b = idx[:, :-1].ravel(), idx[:, 1:].ravel()
        c, d = idx[:-1, :].ravel(), idx[1:, :].ravel()
        rows = np.concatenate([a, a, b, b, c, c, d, d])
        cols = np.concatenate([a, b, a, b, c, d, c, d])
        vals = np.concatenate([cx.ravel(), -cx.ravel(), -cx.ravel(), cx.ravel(),
                               cy.ravel(), -cy.ravel(), -cy.ravel(), cy.ravel()])
        J = sp.csr_matrix((vals, (rows, cols)), shape=(n * n, n * n))
        f = np.flatnonzero(free.ravel())
        du = spla.spsolve(J[f][:, f].tocsc(), -R.ravel()[f])
        u.ravel()[f] += du
    # vertex values of the flux: average of the adjacent faces (one-sided at the boundary)
    s1 = np.zeros((n, n)); cnt = np.zeros((n, n))
    s1[:, :-1] += sx; cnt[:, :-1] += 1; s1[:, 1:] += sx; cnt[:, 1:] += 1
    s1 /= cnt
    s2 = np.zeros((n, n)); cnt = np.zeros((n, n))
    s2[:-1, :] += sy; cnt[:-1, :] += 1; s2[1:, :] += sy; cnt[1:, :] += 1
    s2 /= cnt
    s2[0, :] = 0.0; s2[-1, :] = 0.0                                   # no flux through y = 0, 1
    through = (sx * wy[:, None]).sum(0)                               # total flow through every vertical line
    return np.stack([u, s1, s2]), {'newton_residuals': hist, 'throughflow': through}
