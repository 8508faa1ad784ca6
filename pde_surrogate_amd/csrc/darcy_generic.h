// Per-pixel arithmetic of the Sobel gradients and of the Darcy mixed-residual loss for ANY square field size, written
// once for the device kernels of darcy_loss_generic.hip and (compiled as plain C++) for the host emulation the CPU
// tests run against the oracle (tests/emu/darcy_generic_emu.cpp).
//
// Reference (file:line relative to the reference repository):
//   utils/image_gradient.py:26-47   kernels and the boundary `modifier` (any imsize)
//   utils/image_gradient.py:50-75   grad_h: replicate pad, cross-correlation, x image_width, [@ modifier if correct]
//   utils/image_gradient.py:77-92   grad_v: the transpose, [modifier^T @ ...]
//   models/darcy.py:162-233         the three loss functions on (B, 3, H, W) fields
//
// The arithmetic follows the reference's ORDER: the raw (uncorrected) response first, then the modifier --
//   (g @ modifier)[:, 0] = 4 g[:, 0] - g[:, 1],   (g @ modifier)[:, n-1] = 4 g[:, n-1] - g[:, n-2]
// -- which holds for every n >= 2 (the closed one-sided forms used by the 16/32/64 kernels need n >= 3 columns).
// Adjoint of the modifier: (G @ modifier^T)[:, j] = m_jj G[:, j] - [j == 1] G[:, 0] - [j == n-2] G[:, n-1], m_jj = 4 for
// j in {0, n-1}, else 1.
#pragma once

#ifndef PDES_HD
#ifdef __HIPCC__
#define PDES_HD __host__ __device__ __forceinline__
#else
#define PDES_HD inline
#endif
#endif

namespace pdes {

struct LossParams {
  float a_const;   // w_const * 2 / (B n n)
  float a_cont;    // w_cont  * 2 / (B n n)
  float b_dir;     // w_dir   * 2 / (B n)
  float b_neu;     // w_neu   * 2 / (2 B n)
  float beta1, beta2;
  int nt;          // 1: streaming (non-temporal) global loads / stores
};

namespace gen {

constexpr int kNonlinear = 1, kNoTB = 2, kUncorrected = 4;    // = PDES_LOSS_* of include/pdes_hip.h

PDES_HD int clampi(int v, int n) { return v < 0 ? 0 : (v > n - 1 ? n - 1 : v); }
PDES_HD int imin(int a, int b) { return a < b ? a : b; }
PDES_HD int imax(int a, int b) { return a > b ? a : b; }

// A window [r_lo, ...) x [c_lo, ...) of an n x n plane (global memory: the whole plane, r_lo = c_lo = 0, stride = n;
// LDS: a tile with its halo).  Image coordinates in, clamping is always against the IMAGE size n.
struct Plane {
  const float* p;
  int r_lo, c_lo, stride, n;
  PDES_HD float at(int r, int c) const { return p[(r - r_lo) * stride + (c - c_lo)]; }
};

// ---- filter_size = 3 -------------------------------------------------------------------------------------------------
// raw response x n: [1,2,1]/4 across the derivative, clamped central difference along it (replicate padding)
template <bool HORIZ>
PDES_HD float sobel_raw(const Plane& P, int r, int c) {
  const int n = P.n;
  const int ru = clampi(r - 1, n), rd = clampi(r + 1, n), cl = clampi(c - 1, n), cr = clampi(c + 1, n);
  float lo, hi;
  if (HORIZ) {
    lo = 0.25f * P.at(ru, cl) + 0.5f * P.at(r, cl) + 0.25f * P.at(rd, cl);
    hi = 0.25f * P.at(ru, cr) + 0.5f * P.at(r, cr) + 0.25f * P.at(rd, cr);
  } else {
    lo = 0.25f * P.at(ru, cl) + 0.5f * P.at(ru, c) + 0.25f * P.at(ru, cr);
    hi = 0.25f * P.at(rd, cl) + 0.5f * P.at(rd, c) + 0.25f * P.at(rd, cr);
  }
  return 0.5f * (float)n * (hi - lo);
}

template <bool HORIZ>
PDES_HD float sobel_grad(const Plane& P, int r, int c, bool correct) {
  float g = sobel_raw<HORIZ>(P, r, c);
  if (correct) {
    const int n = P.n, k = HORIZ ? c : r;
    if (k == 0) g = 4.f * g - (HORIZ ? sobel_raw<HORIZ>(P, r, 1) : sobel_raw<HORIZ>(P, 1, c));
    else if (k == n - 1) g = 4.f * g - (HORIZ ? sobel_raw<HORIZ>(P, r, n - 2) : sobel_raw<HORIZ>(P, n - 2, c));
  }
  return g;
}

// (G @ modifier^T) along the axis, at (r, c)
template <bool HORIZ>
PDES_HD float modifier_adj(const Plane& G, int r, int c, bool correct) {
  float v = G.at(r, c);
  if (correct) {
    const int n = G.n, j = HORIZ ? c : r;
    if (j == 0 || j == n - 1) v *= 4.f;
    if (j == 1) v -= HORIZ ? G.at(r, 0) : G.at(0, c);
    if (j == n - 2) v -= HORIZ ? G.at(r, n - 1) : G.at(n - 1, c);
  }
  return v;
}

// adjoint of [clamped central difference, then modifier] along the axis, at (r, c):
// 0.5 ([k >= 1] G'[k-1] + [k == n-1] G'[n-1] - [k <= n-2] G'[k+1] - [k == 0] G'[0])
template <bool HORIZ>
PDES_HD float diff_adj(const Plane& G, int r, int c, bool correct) {
  const int n = G.n, k = HORIZ ? c : r;
  float a = 0.f;
  if (k >= 1) a += HORIZ ? modifier_adj<HORIZ>(G, r, c - 1, correct) : modifier_adj<HORIZ>(G, r - 1, c, correct);
  if (k == n - 1) a += modifier_adj<HORIZ>(G, r, c, correct);
  if (k <= n - 2) a -= HORIZ ? modifier_adj<HORIZ>(G, r, c + 1, correct) : modifier_adj<HORIZ>(G, r + 1, c, correct);
  if (k == 0) a -= modifier_adj<HORIZ>(G, r, c, correct);
  return 0.5f * a;
}

// grad_h^T(G) (HORIZ) / grad_v^T(G) at (r, c): the smoothing matrix is symmetric, so its adjoint is the same clamped
// [1,2,1]/4 across the derivative axis
template <bool HORIZ>
PDES_HD float sobel_adj(const Plane& G, int r, int c, bool correct) {
  const int n = G.n;
  float s;
  if (HORIZ) {
    const int ru = clampi(r - 1, n), rd = clampi(r + 1, n);
    s = 0.25f * diff_adj<HORIZ>(G, ru, c, correct) + 0.5f * diff_adj<HORIZ>(G, r, c, correct) +
        0.25f * diff_adj<HORIZ>(G, rd, c, correct);
  } else {
    const int cl = clampi(c - 1, n), cr = clampi(c + 1, n);
    s = 0.25f * diff_adj<HORIZ>(G, r, cl, correct) + 0.5f * diff_adj<HORIZ>(G, r, c, correct) +
        0.25f * diff_adj<HORIZ>(G, r, cr, correct);
  }
  return (float)n * s;
}

// ---- filter_size = 5 (image_gradient.py:35-41: NOT separable; replicate pad 2; the same modifier) --------------------
PDES_HD float sobel5_w(int i, int j) {     // d/dx kernel [i][j]; d/dy = its transpose
  const float col[5] = {-1.f, -1.f, 0.f, 1.f, 1.f};
  const float mag[5][3] = {{5.f, 4.f, 0.f}, {8.f, 10.f, 0.f}, {10.f, 20.f, 0.f}, {8.f, 10.f, 0.f}, {5.f, 4.f, 0.f}};
  const int d = j < 2 ? j : (j > 2 ? 4 - j : 2);
  return col[j] * mag[i][d] * (1.f / 240.f);
}

template <bool HORIZ>
PDES_HD float sobel5_raw(const Plane& P, int r, int c) {
  const int n = P.n;
  float a = 0.f;
  for (int i = 0; i < 5; ++i)
    for (int j = 0; j < 5; ++j) {
      const float w = HORIZ ? sobel5_w(i, j) : sobel5_w(j, i);
      a += w * P.at(clampi(r + i - 2, n), clampi(c + j - 2, n));
    }
  return a * (float)n;
}

template <bool HORIZ>
PDES_HD float sobel5_grad(const Plane& P, int r, int c, bool correct) {
  float g = sobel5_raw<HORIZ>(P, r, c);
  if (correct) {
    const int n = P.n, k = HORIZ ? c : r;
    if (k == 0) g = 4.f * g - (HORIZ ? sobel5_raw<HORIZ>(P, r, 1) : sobel5_raw<HORIZ>(P, 1, c));
    else if (k == n - 1) g = 4.f * g - (HORIZ ? sobel5_raw<HORIZ>(P, r, n - 2) : sobel5_raw<HORIZ>(P, n - 2, c));
  }
  return g;
}

// outputs o in [0, n) whose tap t (offset t - 2) reads the clamped source index s: [lo, hi] (empty when lo > hi)
PDES_HD void taps5(int s, int t, int n, int& lo, int& hi) {
  lo = s - (t - 2);
  hi = lo;                                     // interior: exactly one output per tap
  if (s == 0) { lo = 0; hi = -(t - 2); }       // o + t - 2 <= 0
  if (s == n - 1) { lo = n - 1 - (t - 2); hi = n - 1; }   // o + t - 2 >= n - 1
  if (lo < 0) lo = 0;
  if (hi > n - 1) hi = n - 1;
}

// img_bar(r, c) = grad_h5^T(Gh) + grad_v5^T(Gv) (either plane may be absent: p == nullptr)
PDES_HD float sobel5_adj(const Plane& Gh, const Plane& Gv, int r, int c, bool correct) {
  const int n = Gh.p ? Gh.n : Gv.n;
  float acc = 0.f;
  for (int i = 0; i < 5; ++i) {
    int olo, ohi;
    taps5(r, i, n, olo, ohi);
    for (int orow = olo; orow <= ohi; ++orow)
      for (int j = 0; j < 5; ++j) {
        int clo, chi;
        taps5(c, j, n, clo, chi);
        for (int ocol = clo; ocol <= chi; ++ocol) {
          if (Gh.p) acc += sobel5_w(i, j) * modifier_adj<true>(Gh, orow, ocol, correct);
          if (Gv.p) acc += sobel5_w(j, i) * modifier_adj<false>(Gv, orow, ocol, correct);
        }
      }
  }
  return acc * (float)n;
}

// ---- the loss at one pixel -----------------------------------------------------------------------------------------
struct PixelTerms {
  float s_const, s_cont, s_dir, s_neu;   // this pixel's contributions to the four sums
  float src_p1, src_p2, src_cc;          // adjoint sources: a_const K r1, a_const K r2, a_cont c
  float d_u, d_s1, d_s2;                 // direct (stencil-free) part of dL/dy
};

PDES_HD PixelTerms loss_pixel(const Plane& U, const Plane& S1, const Plane& S2, float K, int r, int c,
                              const LossParams& p, int flags) {
  const int n = U.n;
  const bool correct = !(flags & kUncorrected);
  const float u = U.at(r, c), x1 = S1.at(r, c), x2 = S2.at(r, c);
  const float ghu = sobel_grad<true>(U, r, c, correct), gvu = sobel_grad<false>(U, r, c, correct);
  const float gh1 = sobel_grad<true>(S1, r, c, correct), gv2 = sobel_grad<false>(S2, r, c, correct);
  float r1 = x1 + K * ghu, r2 = x2 + K * gvu, q1 = 1.f, q2 = 1.f;
  if (flags & kNonlinear) {               // darcy.py:179-191
#ifdef __HIP_DEVICE_COMPILE__
    const float sq = sqrtf(K);
#else
    const float sq = __builtin_sqrtf(K);
#endif
    r1 += p.beta1 * sq * x1 * x1 + p.beta2 * K * x1 * x1 * x1;
    r2 += p.beta1 * sq * x2 * x2 + p.beta2 * K * x2 * x2 * x2;
    q1 += 2.f * p.beta1 * sq * x1 + 3.f * p.beta2 * K * x1 * x1;
    q2 += 2.f * p.beta1 * sq * x2 + 3.f * p.beta2 * K * x2 * x2;
  }
  const bool tb = (r == 0) || (r == n - 1);
  const float cc = ((flags & kNoTB) && tb) ? 0.f : gh1 + gv2;     // darcy.py:224 (use_tb=False)
  PixelTerms o;
  o.s_const = r1 * r1 + r2 * r2;
  o.s_cont = cc * cc;
  o.s_neu = tb ? x2 * x2 : 0.f;
  o.s_dir = 0.f;
  o.d_u = 0.f;
  if (c == 0) { const float e = u - 1.f; o.s_dir += e * e; o.d_u += p.b_dir * e; }
  if (c == n - 1) { o.s_dir += u * u; o.d_u += p.b_dir * u; }
  o.d_s1 = p.a_const * r1 * q1;
  o.d_s2 = p.a_const * r2 * q2 + (tb ? p.b_neu * x2 : 0.f);
  o.src_p1 = p.a_const * K * r1;
  o.src_p2 = p.a_const * K * r2;
  o.src_cc = p.a_cont * cc;
  return o;
}

// ---- tile geometry of the fused kernel -------------------------------------------------------------------------------
// One workgroup walks the tiles of one image.  A tile OWNS [r0, r1) x [c0, c1); the adjoint stencils at its pixels read
// the sources within +-2 of them, and the sources' forward stencils read the fields within +-1 (+-2 where the
// modifier couples the two outermost rows / columns): sources on the own region +-2, fields on it +-3, clipped.
struct TileGeo {
  int r0, r1, c0, c1;          // own
  int sr0, sr1, sc0, sc1;      // sources
  int ir0, ir1, ic0, ic1;      // fields
};

PDES_HD TileGeo tile_geo(int n, int tr, int tc, int ti, int tj) {
  TileGeo g;
  g.r0 = ti * tr; g.r1 = imin(g.r0 + tr, n);
  g.c0 = tj * tc; g.c1 = imin(g.c0 + tc, n);
  g.sr0 = imax(g.r0 - 2, 0); g.sr1 = imin(g.r1 + 2, n);
  g.sc0 = imax(g.c0 - 2, 0); g.sc1 = imin(g.c1 + 2, n);
  g.ir0 = imax(g.r0 - 3, 0); g.ir1 = imin(g.r1 + 3, n);
  g.ic0 = imax(g.c0 - 3, 0); g.ic1 = imin(g.c1 + 3, n);
  return g;
}

// LDS floats one tile needs: 3 field planes + 3 source planes + 3 direct planes
PDES_HD long long tile_floats(int tr, int tc, int n) {
  const long long ih = imin(tr + 6, n), iw = imin(tc + 6, n), sh = imin(tr + 4, n), sw = imin(tc + 4, n);
  return 3 * (ih * iw + sh * sw + (long long)imin(tr, n) * imin(tc, n));
}

// tile size for an n x n image within `budget` floats of LDS: columns split only beyond 256, rows as tall as fits,
// both balanced over the image.  Returns false when nothing fits (never for budget >= 32768 floats).
inline bool choose_tile(int n, long long budget, int& tr, int& tc) {
  const int ntc = (n + 255) / 256;
  tc = (n + ntc - 1) / ntc;
  int best = 0;
  for (int t = 1; t <= n; ++t) {
    if (tile_floats(t, tc, n) > budget) break;
    best = t;
  }
  if (!best) return false;
  const int ntr = (n + best - 1) / best;
  tr = (n + ntr - 1) / ntr;
  return true;
}

}  // namespace gen
}  // namespace pdes
