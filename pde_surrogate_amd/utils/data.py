"""Synthetic permeability fields with the shape/statistics of the reference's datasets.

The reference ships neither data nor a generator (README.md:23 only says K is a Gaussian random
field truncated to N KLE terms, file names `kle512_lhs10000_train.hdf5`).  These are the
build's own generators, used by bench.py, the smoke test and the CLI's `--synthetic` mode:

  grf_kle_fields   log K = sum_i sqrt(lambda_i) phi_i xi_i over the leading `n_kle` eigenpairs of the
                   exponential covariance exp(-|s-s'|_2 / ell) on the 64x64 cell-centred grid of
                   [0,1]^2, xi ~ N(0,1); K = exp(log K).
  channelized_fields  two-valued (sharp-interface) fields from a thresholded anisotropic GRF.
"""
import os

import numpy as np

_KLE_CACHE = {}


def kle_basis(imsize=64, n_kle=512, ell=0.25, cache_dir=None):
    """(n_kle, imsize*imsize) matrix of sqrt(lambda_i) * phi_i (fp64)."""
    key = (imsize, n_kle, ell)
    if key in _KLE_CACHE:
        return _KLE_CACHE[key]
    path = None
    if cache_dir:
        path = os.path.join(cache_dir, f'kle_v2_{imsize}_{n_kle}_{ell}.npy')
        if os.path.exists(path):
            _KLE_CACHE[key] = np.load(path)
            return _KLE_CACHE[key]
    g = (np.arange(imsize) + 0.5) / imsize
    xx, yy = np.meshgrid(g, g, indexing='xy')
    pts = np.stack([xx.ravel(), yy.ravel()], 1)
    d = np.sqrt(((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1))
    cov = np.exp(-d / ell)
    # The covariance has the square's symmetries: many eigenvalues come in PAIRS, and LAPACK's basis of such a plane (like
    # every eigenvector's sign) changes with the BLAS thread count.  Round 6: the fields -- and with them every number a
    # test or bench run prints -- depended on which process wrote the cache file first, one with a capped or an uncapped
    # pool.  The decomposition therefore runs on ONE thread (7 s for 512 of 4,096 pairs, once per box: cached), and the
    # component of largest magnitude of every eigenvector is made positive.
    try:
        import scipy.linalg                      # (loaded BEFORE the limiter is built: it only sees libraries already mapped)
    except Exception:  # pragma: no cover
        pass
    try:
        from threadpoolctl import threadpool_limits
        one_thread = threadpool_limits(limits=1)
    except Exception:  # pragma: no cover
        import contextlib
        one_thread = contextlib.nullcontext()
    with one_thread:
        try:
            from scipy.linalg import eigh
            n = cov.shape[0]
            lam, phi = eigh(cov, subset_by_index=[n - n_kle, n - 1])
        except Exception:  # pragma: no cover
            lam, phi = np.linalg.eigh(cov)
            lam, phi = lam[-n_kle:], phi[:, -n_kle:]
    lam, phi = lam[::-1], phi[:, ::-1]
    pivot = np.abs(phi).argmax(axis=0)
    phi = phi * np.sign(phi[pivot, np.arange(phi.shape[1])])[None, :]
    basis = (phi * np.sqrt(np.maximum(lam, 0.0))[None, :]).T.copy()
    _KLE_CACHE[key] = basis
    if path:
        try:                                     # atomic publish: concurrent ranks may race on the cache file
            tmp = f'{path}.{os.getpid()}.tmp.npy'
            np.save(tmp, basis)
            os.replace(tmp, path)
        except OSError:
            pass
    return basis


def grf_kle_fields(n, imsize=64, n_kle=512, ell=0.25, seed=20190524, cache_dir=None):
    """(n, 1, imsize, imsize) fp32 permeability fields K = exp(GRF truncated to n_kle KLE terms)."""
    basis = kle_basis(imsize, n_kle, ell, cache_dir)
    rng = np.random.default_rng(seed)
    xi = rng.standard_normal((n, n_kle))
    logk = xi @ basis
    return np.exp(logk).astype(np.float32).reshape(n, 1, imsize, imsize)


def channelized_fields(n, imsize=64, seed=20190525, k_low=1.0, k_high=10.0):
    """(n, 1, imsize, imsize) fp32 two-valued fields with sharp channel-like interfaces."""
    rng = np.random.default_rng(seed)
    fy = np.fft.fftfreq(imsize)[:, None]
    fx = np.fft.fftfreq(imsize)[None, :]
    out = np.empty((n, 1, imsize, imsize), np.float32)
    for i in range(n):
        th = rng.uniform(-0.5, 0.5)
        u = fx * np.cos(th) + fy * np.sin(th)
        v = -fx * np.sin(th) + fy * np.cos(th)
        spec = np.exp(-((u / 0.02) ** 2 + (v / 0.12) ** 2))          # long along one axis
        noise = np.fft.fft2(rng.standard_normal((imsize, imsize)))
        f = np.real(np.fft.ifft2(noise * spec))
        out[i, 0] = np.where(f > np.quantile(f, 0.6), k_high, k_low)
    return out
