"""Oracle (test infrastructure): Sobel gradients + Darcy mixed-residual loss on the CPU.

Restates, with explicit stencils instead of conv2d/matmul calls:
  * SobelFilter.grad_h / grad_v           -- reference utils/image_gradient.py:50-92
    (3x3 kernels :26-33, boundary `modifier` :43-46)
  * conv_constitutive_constraint          -- reference models/darcy.py:162-176
  * conv_constitutive_constraint_nonlinear-- reference models/darcy.py:179-191
  * conv_continuity_constraint            -- reference models/darcy.py:210-224
  * conv_boundary_condition               -- reference models/darcy.py:226-233
  * the loss combination                  -- reference train_codec_mixed_residual.py:228-232

All functions are dtype-parametric torch code (fp32 or fp64) and differentiable, so
``torch.autograd`` supplies dL/dy for checking the HIP kernel's analytic adjoint; a
closed-form adjoint (`loss_and_grad_analytic`) is also given and cross-checked in tests.
Pinned against the real reference by tests/golden/G1..G4 (tools/gen_golden.py).
"""
import numpy as np
import torch


def _rep_rows(p):
    """replicate-pad one row top and bottom (F.pad(mode='replicate') semantics, image_gradient.py:68)."""
    return torch.cat([p[..., :1, :], p, p[..., -1:, :]], dim=-2)


def _rep_cols(p):
    return torch.cat([p[..., :, :1], p, p[..., :, -1:]], dim=-1)


def smooth_v(p):
    """[1,2,1]/4 along H with replicate edges (the Sobel kernel's smoothing factor)."""
    q = _rep_rows(p)
    return (q[..., :-2, :] + 2.0 * q[..., 1:-1, :] + q[..., 2:, :]) / 4.0


def smooth_h(p):
    q = _rep_cols(p)
    return (q[..., :, :-2] + 2.0 * q[..., :, 1:-1] + q[..., :, 2:]) / 4.0


def sobel_grad_h(img, correct=True):
    """d/dx (along W). image_gradient.py:50-75: replicate pad, VSOBEL/8 cross-correlation,
    x image_width, then right-multiply by `modifier` (cols 0 and W-1 <- 4*g0-g1, 4*g[-1]-g[-2])."""
    W = img.shape[-1]
    s = _rep_cols(smooth_v(img))
    g = (s[..., :, 2:] - s[..., :, :-2]) * (W / 2.0)
    if not correct:
        return g
    first = 4.0 * g[..., :, 0:1] - g[..., :, 1:2]
    last = 4.0 * g[..., :, -1:] - g[..., :, -2:-1]
    return torch.cat([first, g[..., :, 1:-1], last], dim=-1)


def sobel_grad_v(img, correct=True):
    """d/dy (along H). image_gradient.py:77-92 (left-multiply by modifier^T)."""
    H = img.shape[-2]
    s = _rep_rows(smooth_h(img))
    g = (s[..., 2:, :] - s[..., :-2, :]) * (H / 2.0)
    if not correct:
        return g
    first = 4.0 * g[..., 0:1, :] - g[..., 1:2, :]
    last = 4.0 * g[..., -1:, :] - g[..., -2:-1, :]
    return torch.cat([first, g[..., 1:-1, :], last], dim=-2)


_SOBEL5 = np.array([[-5, -4, 0, 4, 5], [-8, -10, 0, 10, 8], [-10, -20, 0, 20, 10], [-8, -10, 0, 10, 8],
                    [-5, -4, 0, 4, 5]], np.float64) / 240.0        # image_gradient.py:35-40 (d/dx; transpose = d/dy)


def _sobel5(img, kern, scale, correct, axis):
    """filter_size=5 (image_gradient.py:65-67, :82-84): replicate pad 2, 5x5 cross-correlation, x image size, and
    the SAME 3-point boundary `modifier` as the 3x3 filter."""
    H, W = img.shape[-2:]
    q = img
    for _ in range(2):
        q = _rep_cols(_rep_rows(q))
    g = torch.zeros_like(img)
    for i in range(5):
        for j in range(5):
            if kern[i, j] != 0.0:
                g = g + float(kern[i, j]) * q[..., i:i + H, j:j + W]
    g = g * scale
    if not correct:
        return g
    if axis == 'h':
        first = 4.0 * g[..., :, 0:1] - g[..., :, 1:2]
        last = 4.0 * g[..., :, -1:] - g[..., :, -2:-1]
        return torch.cat([first, g[..., :, 1:-1], last], dim=-1)
    first = 4.0 * g[..., 0:1, :] - g[..., 1:2, :]
    last = 4.0 * g[..., -1:, :] - g[..., -2:-1, :]
    return torch.cat([first, g[..., 1:-1, :], last], dim=-2)


def sobel_grad_h5(img, correct=True):
    return _sobel5(img, _SOBEL5, img.shape[-1], correct, 'h')


def sobel_grad_v5(img, correct=True):
    return _sobel5(img, _SOBEL5.T, img.shape[-2], correct, 'v')


def constitutive(K, y, beta1=0.0, beta2=0.0, nonlinear=False, correct=True):
    """mean[(sigma1 + K du/dx [+nl])^2 + (sigma2 + K du/dy [+nl])^2]; darcy.py:162-176 / :179-191.
    `correct` is the SobelFilter's flag (image_gradient.py:26, :72-75, :89-92)."""
    u, s1, s2 = y[:, 0:1], y[:, 1:2], y[:, 2:3]
    gh, gv = sobel_grad_h(u, correct), sobel_grad_v(u, correct)
    if nonlinear:
        sq = torch.sqrt(K)
        r1 = s1 + beta1 * sq * s1 ** 2 + beta2 * K * s1 ** 3 + K * gh
        r2 = s2 + beta1 * sq * s2 ** 2 + beta2 * K * s2 ** 3 + K * gv
    else:
        r1 = s1 + K * gh
        r2 = s2 + K * gv
    return (r1 ** 2 + r2 ** 2).mean()


def continuity(y, use_tb=True, correct=True):
    """mean[(d sigma1/dx + d sigma2/dy)^2]; darcy.py:210-224."""
    c = sobel_grad_h(y[:, 1:2], correct) + sobel_grad_v(y[:, 2:3], correct)
    if use_tb:
        return (c ** 2).mean()
    return (c ** 2)[:, :, 1:-1, :].mean()


def boundary(y):
    """(dirichlet, neumann); darcy.py:226-233. u=1 on col 0, u=0 on col W-1, sigma2=0 on rows 0,H-1."""
    left, right = y[:, 0, :, 0], y[:, 0, :, -1]
    tb = y[:, 2, [0, -1], :]
    dirichlet = ((left - 1.0) ** 2).mean() + (right ** 2).mean()
    neumann = (tb ** 2).mean()
    return dirichlet, neumann


def mixed_residual_loss(K, y, weight_bound=10.0, beta1=0.0, beta2=0.0, nonlinear=False, correct=True, use_tb=True):
    """train_codec_mixed_residual.py:228-232. Returns (loss, l_const, l_cont, l_dir, l_neu)."""
    lc = constitutive(K, y, beta1, beta2, nonlinear, correct)
    lt = continuity(y, use_tb, correct)
    ld, ln = boundary(y)
    return lc + lt + (ld + ln) * weight_bound, lc, lt, ld, ln


def loss_and_grad_autograd(K, y, weight_bound=10.0, beta1=0.0, beta2=0.0, nonlinear=False,
                           weights=None, correct=True, use_tb=True):
    """dL/dy by autograd. `weights`=(w_const,w_cont,w_dir,w_neu) overrides the default (1,1,wb,wb)."""
    y = y.detach().clone().requires_grad_(True)
    loss, lc, lt, ld, ln = mixed_residual_loss(K, y, weight_bound, beta1, beta2, nonlinear, correct, use_tb)
    if weights is not None:
        loss = weights[0] * lc + weights[1] * lt + weights[2] * ld + weights[3] * ln
    (g,) = torch.autograd.grad(loss, y)
    return [t.detach() for t in (loss, lc, lt, ld, ln)], g


# ---------------------------------------------------------------- matrix form + closed-form adjoint
def sobel_matrices(n, dtype=np.float64):
    """(S, A) with grad_h(U) = n * S U A and grad_v(U) = n * A^T U S (square n x n images).
    S: replicate-edge [1,2,1]/4 smoothing; A = Dc @ modifier (image_gradient.py:43-46)."""
    S = np.zeros((n, n), dtype)
    Dc = np.zeros((n, n), dtype)  # g = s @ Dc  (clamped central difference / 2)
    for i in range(n):
        for d, w in ((-1, 0.25), (0, 0.5), (1, 0.25)):
            S[i, min(max(i + d, 0), n - 1)] += w
        Dc[min(i + 1, n - 1), i] += 0.5
        Dc[max(i - 1, 0), i] -= 0.5
    M = np.eye(n, dtype=dtype)
    M[0:2, 0] = (4, -1)
    M[-2:, -1] = (-1, 4)
    return S, Dc @ M


def loss_and_grad_analytic(K, y, weight_bound=10.0):
    """numpy fp64 closed form of the linear loss and dL/dy (the formulas the HIP kernel implements).
    K: (B,1,n,n), y: (B,3,n,n) arrays."""
    K = np.asarray(K, np.float64)[:, 0]
    y = np.asarray(y, np.float64)
    B, _, n, _ = y.shape
    S, A = sobel_matrices(n)
    gh = lambda U: n * (S @ U @ A)
    gv = lambda U: n * (A.T @ U @ S)
    ghT = lambda G: n * (S @ G @ A.T)
    gvT = lambda G: n * (A @ G @ S)
    u, s1, s2 = y[:, 0], y[:, 1], y[:, 2]
    r1 = s1 + K * gh(u)
    r2 = s2 + K * gv(u)
    c = gh(s1) + gv(s2)
    N = B * n * n
    lc = (r1 ** 2 + r2 ** 2).sum() / N
    lt = (c ** 2).sum() / N
    ld = ((u[:, :, 0] - 1) ** 2).mean() + (u[:, :, -1] ** 2).mean()
    ln = (s2[:, [0, -1], :] ** 2).mean()
    g = np.zeros_like(y)
    g[:, 1] = 2 / N * r1 + ghT(2 / N * c)
    g[:, 2] = 2 / N * r2 + gvT(2 / N * c)
    g[:, 0] = ghT(2 / N * K * r1) + gvT(2 / N * K * r2)
    wb = weight_bound
    g[:, 0, :, 0] += wb * 2 / (B * n) * (u[:, :, 0] - 1)
    g[:, 0, :, -1] += wb * 2 / (B * n) * u[:, :, -1]
    g[:, 2, 0, :] += wb * 2 / (2 * B * n) * s2[:, 0, :]
    g[:, 2, -1, :] += wb * 2 / (2 * B * n) * s2[:, -1, :]
    return (lc + lt + wb * (ld + ln), lc, lt, ld, ln), g
