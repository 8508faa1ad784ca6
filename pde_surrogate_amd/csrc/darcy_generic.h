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
  const float* wdev;   // pdes_darcy_loss_dw: the four term weights live in DEVICE memory -- the fields above were computed
                       // for weights of 1 and the backward kernels multiply them by wdev[0..3] (NULL: weights by value)
};

#ifdef __HIPCC__
// the per-term weights of pdes_darcy_loss_dw: four uniform (scalar) loads at the head of a backward kernel
__device__ __forceinline__ LossParams loss_params_weighted(LossParams p) {
  if (p.wdev) {
    p.a_const *= p.wdev[0];
    p.a_cont *= p.wdev[1];
    p.b_dir *= p.wdev[2];
    p.b_neu *= p.wdev[3];
  }
  return p;
}
#endif

// w * x; SAFE: an exactly zero weight SKIPS the term -- a backward pass that reaches only some of the four loss terms
// (conv_boundary_condition alone, say) launches the kernel with zero weights for the others, and 0 * inf / 0 * NaN of a
// residual that term never looked at must not poison the gradient of the terms it did.  `w` is uniform.  The
// select costs the specialised kernel 1.3 % (2.3 % nonlinear) at B = 16,384, so only the instantiations behind
// pdes_darcy_loss_dw (weights in device memory: the autograd path) carry it; the any-size kernels use the plain product.
#ifdef __HIPCC__
template <bool SAFE>
__device__ __forceinline__ float wmul(float w, float x) { return (SAFE && w == 0.f) ? 0.f : w * x; }
#endif

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

PDES_HD float sqrt_f(float v) {
#ifdef __HIP_DEVICE_COMPILE__
  return sqrtf(v);
#else
  return __builtin_sqrtf(v);
#endif
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
    const float sq = sqrt_f(K);
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

// ---- 16-byte accesses of four consecutive floats (LDS planes / image rows that are laid out 16-byte aligned) ----------
PDES_HD void ld4(const float* q, float* o) {
#ifdef __HIP_DEVICE_COMPILE__
  const float4 t = *reinterpret_cast<const float4*>(q);
  o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
#else
  o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; o[3] = q[3];
#endif
}
PDES_HD void st4(float* q, const float* v) {
#ifdef __HIP_DEVICE_COMPILE__
  *reinterpret_cast<float4*>(q) = make_float4(v[0], v[1], v[2], v[3]);
#else
  q[0] = v[0]; q[1] = v[1]; q[2] = v[2]; q[3] = v[3];
#endif
}
// ---- tile geometry of the per-pixel form (n < 8) ------------------------------------------------------------------------
// One workgroup processes one tile of one image.  A tile OWNS [r0, r1) x [c0, c1); the adjoint stencils at its pixels read
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

// LDS layout of a tile: three planes each of fields / adjoint sources / direct terms.  A plane's column origin is rounded
// DOWN to a multiple of 4 image columns and its row stride UP to a multiple of 4 floats (+ 4: load6 reads one past a strip),
// so that every strip starting at an image column that is a multiple of 4 is 16-byte aligned.
struct TileLayout {
  int lf, wf, nf;              // fields: first LDS column (image coordinate), row stride, floats per plane
  int ls, ws, ns;              // sources
  int ld, wd, nd;              // direct terms
};
PDES_HD int pad_width(int c_lo, int c_hi, int& origin) {
  origin = c_lo & ~3;
  return (((c_hi - origin) + 3) & ~3) + 4;
}
PDES_HD TileLayout tile_layout(const TileGeo& g) {
  TileLayout t;
  t.wf = pad_width(g.ic0, g.ic1, t.lf); t.nf = (g.ir1 - g.ir0) * t.wf;
  t.ws = pad_width(g.sc0, g.sc1, t.ls); t.ns = (g.sr1 - g.sr0) * t.ws;
  t.wd = pad_width(g.c0, g.c1, t.ld);   t.nd = (g.r1 - g.r0) * t.wd;
  return t;
}
// upper bound of the LDS floats of any tile of this size (+ 4 floats of slack in front: load6 reads one before a strip)
PDES_HD long long tile_floats(int tr, int tc, int n) {
  const long long ih = imin(tr + 6, n), sh = imin(tr + 4, n), oh = imin(tr, n);
  const long long w = ((imin(tc + 6, n) + 3 + 3) & ~3) + 4;           // origin rounding (<= 3) + padding, any of the three
  return 3 * (ih + sh + oh) * w + 4;
}

// tile size for an n x n image within `budget` floats of LDS: whole rows if at least 8 of them fit (fewer for smaller
// images), otherwise the columns are cut too (multiples of 4 wide, so that strips stay aligned); both balanced over the
// image.  Returns false when nothing fits.
inline bool choose_tile(int n, long long budget, int& tr, int& tc) {
  for (int ntc = 1; ntc <= n; ++ntc) {
    tc = (((n + ntc - 1) / ntc) + 3) & ~3;
    if (tc > n) tc = n;
    int best = 0;
    for (int t = 1; t <= n; ++t) {
      if (tile_floats(t, tc, n) > budget) break;
      best = t;
    }
    if (best >= imin(8, n) || (best >= 1 && tc <= 16)) {
      const int ntr = (n + best - 1) / best;
      tr = (n + ntr - 1) / ntr;
      return true;
    }
  }
  return false;
}

// ---- one tile, start to finish: the per-pixel form (tiny images, n < 8) ---------------------------------------------------------------------------------------
// The SAME code runs as a HIP workgroup (Exec: tid / nthreads of the block, barrier = __syncthreads) and as a serial
// loop on the host (tests/emu: one "thread", barrier a no-op).  Phases: fields -> LDS | sources + direct terms + sums |
// adjoint + stores.  sums[4] are this thread's contributions {const, cont, dir, neu}.
template <bool BWD, class Exec>
PDES_HD void process_tile_pixelwise(const float* Kb, const float* yb, float* gb, int n, const TileGeo& g, const LossParams& p, int flags,
                          float* lds, Exec& ex, float* sums) {
  const size_t nn = (size_t)n * n;
  const bool correct = !(flags & kUncorrected);
  const TileLayout L = tile_layout(g);
  float* F = lds + 4;                               // (4 floats of slack: a strip at the plane's first column reads q[-1])
  float* S = F + 3 * L.nf;
  float* D = S + 3 * L.ns;
  // ---- phase A: the three fields of the tile (+-3)
  {
    const int iw = g.ic1 - g.ic0, ni = (g.ir1 - g.ir0) * iw;
    for (int i = ex.tid; i < ni; i += ex.nthreads) {
      const int rr = i / iw, cc = i - rr * iw;
      const size_t o = (size_t)(g.ir0 + rr) * n + (g.ic0 + cc);
      const int l = rr * L.wf + (g.ic0 + cc - L.lf);
      F[l] = yb[o];
      F[L.nf + l] = yb[nn + o];
      F[2 * L.nf + l] = yb[2 * nn + o];
    }
  }
  ex.barrier();
  const Plane U{F, g.ir0, L.lf, L.wf, n}, X1{F + L.nf, g.ir0, L.lf, L.wf, n}, X2{F + 2 * L.nf, g.ir0, L.lf, L.wf, n};
  // ---- phase B: per source pixel the residuals; per OWN pixel the sums and the direct part of the gradient
  {
    const int q0 = g.sc0 >> 2, nq = ((g.sc1 + 3) >> 2) - q0, nslots = (g.sr1 - g.sr0) * nq;
    for (int i = ex.tid; i < nslots; i += ex.nthreads) {
      const int rr = i / nq, r = g.sr0 + rr, c0 = 4 * (q0 + i - rr * nq);
      const bool own_row = r >= g.r0 && r < g.r1;
      for (int j = 0; j < 4; ++j) {
        const int c = c0 + j;
        if (c < g.sc0 || c >= g.sc1) continue;
        const PixelTerms t = loss_pixel(U, X1, X2, Kb[(size_t)r * n + c], r, c, p, flags);
        if (own_row && c >= g.c0 && c < g.c1) {
          sums[0] += t.s_const; sums[1] += t.s_cont; sums[2] += t.s_dir; sums[3] += t.s_neu;
          if (BWD) {
            const int ldd = (r - g.r0) * L.wd + (c - L.ld);
            D[ldd] = t.d_u; D[L.nd + ldd] = t.d_s1; D[2 * L.nd + ldd] = t.d_s2;
          }
        }
        if (BWD) {
          const int ls = rr * L.ws + (c - L.ls);
          S[ls] = t.src_p1; S[L.ns + ls] = t.src_p2; S[2 * L.ns + ls] = t.src_cc;
        }
      }
    }
  }
  if (!BWD) return;
  ex.barrier();
  // ---- phase C: dL/dy = direct part + adjoint stencils of the sources
  const Plane G1{S, g.sr0, L.ls, L.ws, n}, G2{S + L.ns, g.sr0, L.ls, L.ws, n}, GC{S + 2 * L.ns, g.sr0, L.ls, L.ws, n};
  {
    const int q0 = g.c0 >> 2, nq = ((g.c1 + 3) >> 2) - q0, nslots = (g.r1 - g.r0) * nq;
    for (int i = ex.tid; i < nslots; i += ex.nthreads) {
      const int rr = i / nq, r = g.r0 + rr, c0 = 4 * (q0 + i - rr * nq);
      for (int j = 0; j < 4; ++j) {
        const int c = c0 + j;
        if (c < g.c0 || c >= g.c1) continue;
        const int ldd = rr * L.wd + (c - L.ld);
        const size_t o = (size_t)r * n + c;
        gb[o] = D[ldd] + sobel_adj<true>(G1, r, c, correct) + sobel_adj<false>(G2, r, c, correct);
        gb[nn + o] = D[L.nd + ldd] + sobel_adj<true>(GC, r, c, correct);
        gb[2 * nn + o] = D[2 * L.nd + ldd] + sobel_adj<false>(GC, r, c, correct);
      }
    }
  }
}

// ---- one tile, start to finish: the strip form (n >= 8) ----------------------------------------------------------------
// Every pixel -- border or not -- goes through the SAME branch-free code: along an axis of length n the corrected
// (or uncorrected) clamped central difference, its adjoint, and the replicate-edge smoothing are 5- resp. 3-point
// stencils whose coefficients depend on the index only (tables below, derived from image_gradient.py:26-47 as in the
// header comment: rows 0-2 and n-3..n-1 of the adjoint carry the modifier's couplings), and the LDS planes are ZEROED
// before they are filled, so a stencil may touch cells outside the image with a zero coefficient.  A thread owns a 1 x 4
// strip (16-byte LDS accesses, columns c0-2 .. c0+5 of a row in registers); waves do not diverge on border strips.
struct Ax5 { float c[5]; };                 // weights of x[k-2 .. k+2]
struct Ax3 { float c[3]; };                 // weights of x[k-1 .. k+1]
struct Row8 { float v[8]; };                // columns c0-2 .. c0+5

PDES_HD Ax5 axis_diff(int k, int n, bool correct) {
  Ax5 t = {{0.f, -0.5f, 0.f, 0.5f, 0.f}};
  if (k == 0) { if (correct) t = {{0.f, 0.f, -1.5f, 2.f, -0.5f}}; else t = {{0.f, 0.f, -0.5f, 0.5f, 0.f}}; }
  else if (k == n - 1) { if (correct) t = {{0.5f, -2.f, 1.5f, 0.f, 0.f}}; else t = {{0.f, -0.5f, 0.5f, 0.f, 0.f}}; }
  return t;
}
PDES_HD Ax5 axis_diff_adj(int k, int n, bool correct) {      // n >= 6: the cases are distinct
  Ax5 t = {{0.f, 0.5f, 0.f, -0.5f, 0.f}};
  if (correct) {
    if (k == 0) t = {{0.f, 0.f, -1.5f, -0.5f, 0.f}};
    else if (k == 1) t = {{0.f, 2.f, 0.f, -0.5f, 0.f}};
    else if (k == 2) t = {{-0.5f, 0.5f, 0.f, -0.5f, 0.f}};
    else if (k == n - 3) t = {{0.f, 0.5f, 0.f, -0.5f, 0.5f}};
    else if (k == n - 2) t = {{0.f, 0.5f, 0.f, -2.f, 0.f}};
    else if (k == n - 1) t = {{0.f, 0.5f, 1.5f, 0.f, 0.f}};
  } else {
    if (k == 0) t = {{0.f, 0.f, -0.5f, -0.5f, 0.f}};
    else if (k == n - 1) t = {{0.f, 0.5f, 0.5f, 0.f, 0.f}};
  }
  return t;
}
PDES_HD Ax3 axis_smooth(int k, int n) {
  Ax3 t = {{0.25f, 0.5f, 0.25f}};
  if (k == 0) t = {{0.f, 0.75f, 0.25f}};
  else if (k == n - 1) t = {{0.25f, 0.75f, 0.f}};
  return t;
}

// a plane of a strip tile: rows [r_lo, r_hi), row stride w, LDS column 0 = image column c_org (a multiple of 4, possibly
// negative); every cell of the allocation is zero or a field / source value
struct SPlane {
  const float* p;
  int r_lo, r_hi, c_org, w;
};
PDES_HD Row8 load8(const SPlane& P, int r, int c0) {       // r clamped into the plane (callers pass a zero weight then)
  const int rr = r < P.r_lo ? P.r_lo : (r >= P.r_hi ? P.r_hi - 1 : r);
  const float* q = P.p + (rr - P.r_lo) * P.w + (c0 - P.c_org);
  Row8 o;
  o.v[0] = q[-2]; o.v[1] = q[-1];
  ld4(q, &o.v[2]);
  o.v[6] = q[4]; o.v[7] = q[5];
  return o;
}
// n * sum_q colt[j][q] * (sum_p rows[p] * P(r+p-1, c0+j+q-2)): grad_h with (axis_diff, axis_smooth), grad_h^T with
// (axis_diff_adj, axis_smooth)
PDES_HD void strip_h(const SPlane& P, int r, int c0, const Ax3& rows, const Ax5* colt, float fn, float* out) {
  const Row8 a = load8(P, r - 1, c0), b = load8(P, r, c0), d = load8(P, r + 1, c0);
  float sv[8];
  for (int k = 0; k < 8; ++k) sv[k] = rows.c[0] * a.v[k] + rows.c[1] * b.v[k] + rows.c[2] * d.v[k];
  for (int j = 0; j < 4; ++j)
    out[j] = fn * (colt[j].c[0] * sv[j] + colt[j].c[1] * sv[j + 1] + colt[j].c[2] * sv[j + 2] + colt[j].c[3] * sv[j + 3] +
                   colt[j].c[4] * sv[j + 4]);
}
// n * sum_p cols[j][p] * (sum_q rowt[q] * P(r+q-2, c0+j+p-1)): grad_v with (axis_diff, axis_smooth), grad_v^T with the adjoint table
PDES_HD void strip_v(const SPlane& P, int r, int c0, const Ax5& rowt, const Ax3* cols, float fn, float* out) {
  float dv[8];
  {
    const Row8 a = load8(P, r - 1, c0), d = load8(P, r + 1, c0);
    for (int k = 0; k < 8; ++k) dv[k] = rowt.c[1] * a.v[k] + rowt.c[3] * d.v[k];
  }
  if (rowt.c[2] != 0.f) { const Row8 b = load8(P, r, c0); for (int k = 0; k < 8; ++k) dv[k] += rowt.c[2] * b.v[k]; }
  if (rowt.c[0] != 0.f) { const Row8 b = load8(P, r - 2, c0); for (int k = 0; k < 8; ++k) dv[k] += rowt.c[0] * b.v[k]; }
  if (rowt.c[4] != 0.f) { const Row8 b = load8(P, r + 2, c0); for (int k = 0; k < 8; ++k) dv[k] += rowt.c[4] * b.v[k]; }
  for (int j = 0; j < 4; ++j)
    out[j] = fn * (cols[j].c[0] * dv[j + 1] + cols[j].c[1] * dv[j + 2] + cols[j].c[2] * dv[j + 3]);
}

struct StripGeo {
  int r0, r1, c0, c1;          // own pixels; c0 a multiple of 4
  int sr0, sr1, sc0, sc1;      // source strips: rows, columns [sc0, sc1) in whole strips (sc1 may exceed n: zero cells)
  int fr0, fr1;                // field rows
  int fo, wf, so, ws, wd;      // plane column origins (multiples of 4) and row strides
  int nf, ns, nd;              // floats per plane
};
PDES_HD StripGeo strip_geo(int n, int tr, int tc, int ti, int tj) {
  StripGeo g;
  const int na = (n + 3) & ~3;
  g.r0 = ti * tr; g.r1 = imin(g.r0 + tr, n);
  g.c0 = tj * tc; g.c1 = imin(g.c0 + tc, n);
  const int c1a = (g.c1 + 3) & ~3;
  g.sr0 = imax(g.r0 - 2, 0); g.sr1 = imin(g.r1 + 2, n);
  g.sc0 = imax(g.c0 - 4, 0); g.sc1 = imin(c1a + 4, na);
  g.fr0 = imax(g.sr0 - 1, 0); g.fr1 = imin(g.sr1 + 1, n);
  g.fo = g.sc0 - 4; g.wf = g.sc1 + 4 - g.fo;
  g.so = g.sc0 - 4; g.ws = g.sc1 + 4 - g.so;
  g.wd = c1a - g.c0;
  g.nf = (g.fr1 - g.fr0) * g.wf; g.ns = (g.sr1 - g.sr0) * g.ws; g.nd = (g.r1 - g.r0) * g.wd;
  return g;
}
PDES_HD long long strip_tile_floats(int tr, int tc, int n) {       // upper bound over the tiles of an image
  const long long w = imin(tc, (n + 3) & ~3) + 16;
  return 3 * ((long long)imin(tr + 6, n) * w + (long long)imin(tr + 4, n) * w + (long long)imin(tr, n) * (w - 12));
}
inline bool choose_strip_tile(int n, long long budget, int& tr, int& tc) {
  for (int ntc = 1; ntc <= n; ++ntc) {
    tc = (((n + ntc - 1) / ntc) + 3) & ~3;
    int best = 0;
    for (int t = 1; t <= n; ++t) {
      if (strip_tile_floats(t, tc, n) > budget) break;
      best = t;
    }
    if (best >= 8 || (best >= 1 && tc <= 16)) {
      const int ntr = (n + best - 1) / best;
      tr = (n + ntr - 1) / ntr;
      return true;
    }
  }
  return false;
}

struct StripCtx {
  const float* Kb;
  float* gb;
  int n, flags;
  bool correct, vec;
  float fn;
  size_t nn;
  StripGeo g;
  LossParams p;
  SPlane U, X1, X2, G1, G2, GC;
  float* F;
  float* S;
  float* D;
};

// the residuals of the four pixels of a strip from their gradients: sums, adjoint sources, direct terms (phase B, both paths)
// the conductivities of a strip (global memory: issued ahead of the field staging for a thread's first strips)
PDES_HD void load_k(const StripCtx& x, int r, int c0, float* kk) {
  const float* kp = x.Kb + (size_t)r * x.n + c0;
  if (x.vec) ld4(kp, kk);
  else for (int j = 0; j < 4; ++j) kk[j] = (c0 + j < x.n) ? kp[j] : 0.f;
}

template <bool BWD>
PDES_HD void strip_residuals(const StripCtx& x, int r, int c0, const float* kk, const float* ghu, const float* gvu, const float* gh1,
                             const float* gv2, float* sums) {
  const StripGeo& g = x.g;
  const LossParams& p = x.p;
  const int n = x.n;
  float u4[4], v14[4], v24[4];
  const int fl = (r - g.fr0) * g.wf + (c0 - g.fo);
  ld4(x.F + fl, u4); ld4(x.F + g.nf + fl, v14); ld4(x.F + 2 * g.nf + fl, v24);
  const bool tb = (r == 0) || (r == n - 1);
  const bool own_row = r >= g.r0 && r < g.r1;
  float sp1[4], sp2[4], scc[4], d1[4], d2[4], du[4];
  for (int j = 0; j < 4; ++j) {
    const int c = c0 + j;
    const bool in = c < n;
    const float K = kk[j], v1 = v14[j], v2 = v24[j];
    float r1 = v1 + K * ghu[j], r2 = v2 + K * gvu[j], q1 = 1.f, q2 = 1.f;
    if (x.flags & kNonlinear) {               // darcy.py:179-191
      const float sq = sqrt_f(K);
      r1 += p.beta1 * sq * v1 * v1 + p.beta2 * K * v1 * v1 * v1;
      r2 += p.beta1 * sq * v2 * v2 + p.beta2 * K * v2 * v2 * v2;
      q1 += 2.f * p.beta1 * sq * v1 + 3.f * p.beta2 * K * v1 * v1;
      q2 += 2.f * p.beta1 * sq * v2 + 3.f * p.beta2 * K * v2 * v2;
    }
    const float cc = ((x.flags & kNoTB) && tb) ? 0.f : gh1[j] + gv2[j];
    sp1[j] = in ? p.a_const * K * r1 : 0.f;
    sp2[j] = in ? p.a_const * K * r2 : 0.f;
    scc[j] = in ? p.a_cont * cc : 0.f;
    d1[j] = p.a_const * r1 * q1;
    d2[j] = p.a_const * r2 * q2 + (tb ? p.b_neu * v2 : 0.f);
    float e = 0.f, dd = 0.f;
    if (c == 0) { e = u4[j] - 1.f; dd = p.b_dir * e; }
    if (c == n - 1) { e = u4[j]; dd = p.b_dir * e; }
    du[j] = dd;
    if (own_row && in && c >= g.c0 && c < g.c1) {
      sums[0] += r1 * r1 + r2 * r2;
      sums[1] += cc * cc;
      sums[2] += e * e;
      if (tb) sums[3] += v2 * v2;
    }
  }
  if (BWD) {
    const int ls = (r - g.sr0) * g.ws + (c0 - g.so);
    st4(x.S + ls, sp1); st4(x.S + g.ns + ls, sp2); st4(x.S + 2 * g.ns + ls, scc);
    if (own_row && c0 >= g.c0 && c0 < g.c1) {
      const int ldd = (r - g.r0) * g.wd + (c0 - g.c0);
      st4(x.D + ldd, du); st4(x.D + g.nd + ldd, d1); st4(x.D + 2 * g.nd + ldd, d2);
    }
  }
}

template <bool BWD>
PDES_HD void phase_b_strip(const StripCtx& x, int r, int c0, const float* kk, float* sums) {
  float ghu[4], gvu[4], gh1[4], gv2[4];
  Ax5 colf[4];
  Ax3 cols[4];
  for (int j = 0; j < 4; ++j) { colf[j] = axis_diff(c0 + j, x.n, x.correct); cols[j] = axis_smooth(c0 + j, x.n); }
  const Ax5 rowf = axis_diff(r, x.n, x.correct);
  const Ax3 rows = axis_smooth(r, x.n);
  strip_h(x.U, r, c0, rows, colf, x.fn, ghu);
  strip_v(x.U, r, c0, rowf, cols, x.fn, gvu);
  strip_h(x.X1, r, c0, rows, colf, x.fn, gh1);
  strip_v(x.X2, r, c0, rowf, cols, x.fn, gv2);
  strip_residuals<BWD>(x, r, c0, kk, ghu, gvu, gh1, gv2, sums);
}

PDES_HD void phase_c_strip(const StripCtx& x, int r, int c0) {
  const StripGeo& g = x.g;
  float a1[4], a2[4], ac1[4], ac2[4], du[4], d1[4], d2[4];
  Ax5 cola[4];
  Ax3 cols[4];
  for (int j = 0; j < 4; ++j) { cola[j] = axis_diff_adj(c0 + j, x.n, x.correct); cols[j] = axis_smooth(c0 + j, x.n); }
  const Ax5 rowa = axis_diff_adj(r, x.n, x.correct);
  const Ax3 rows = axis_smooth(r, x.n);
  strip_h(x.G1, r, c0, rows, cola, x.fn, a1);
  strip_v(x.G2, r, c0, rowa, cols, x.fn, a2);
  strip_h(x.GC, r, c0, rows, cola, x.fn, ac1);
  strip_v(x.GC, r, c0, rowa, cols, x.fn, ac2);
  const int ldd = (r - g.r0) * g.wd + (c0 - g.c0);
  ld4(x.D + ldd, du); ld4(x.D + g.nd + ldd, d1); ld4(x.D + 2 * g.nd + ldd, d2);
  for (int j = 0; j < 4; ++j) { du[j] += a1[j] + a2[j]; d1[j] += ac1[j]; d2[j] += ac2[j]; }
  float* o = x.gb + (size_t)r * x.n + c0;
  if (x.vec) { st4(o, du); st4(o + x.nn, d1); st4(o + 2 * x.nn, d2); }
  else for (int j = 0; j < 4; ++j) if (c0 + j < x.n) { o[j] = du[j]; o[x.nn + j] = d1[j]; o[2 * x.nn + j] = d2[j]; }
}

template <bool BWD, class Exec>
PDES_HD void process_tile_strips(const float* Kb, const float* yb, float* gb, int n, const StripGeo& g, const LossParams& p,
                                 int flags, float* lds, Exec& ex, float* sums) {
  StripCtx x;
  x.Kb = Kb; x.gb = gb; x.n = n; x.flags = flags; x.g = g; x.p = p;
  x.nn = (size_t)n * n;
  x.correct = !(flags & kUncorrected);
  x.vec = (n & 3) == 0;
  x.fn = (float)n;
  x.F = lds; x.S = x.F + 3 * g.nf; x.D = x.S + 3 * g.ns;
  // ---- zero the planes (stencils may touch cells outside the image / the tile with a zero weight)
  {
    const int tot = 3 * (g.nf + g.ns + g.nd);
    const float zz[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 4 * ex.tid; i < tot; i += 4 * ex.nthreads) st4(lds + i, zz);
  }
  ex.barrier();
  // the conductivities of this thread's first strip: their latency hides behind the field staging
  const int sw = (g.sc1 - g.sc0) >> 2, nslots_b = (g.sr1 - g.sr0) * sw;
  float kk[4] = {0.f, 0.f, 0.f, 0.f};
  if (ex.tid < nslots_b) { const int rr = ex.tid / sw; load_k(x, g.sr0 + rr, g.sc0 + 4 * (ex.tid - rr * sw), kk); }
  // ---- phase A: the three fields on the source strips +- one strip, rows fr0 .. fr1; loads issued in batches (one
  // round trip to memory per batch, not per element)
  {
    const int rows = g.fr1 - g.fr0;
    if (x.vec) {
      const int ca = imax(g.sc0 - 4, 0), cb = imin(g.sc1 + 4, n), qw = (cb - ca) >> 2, per = rows * qw, ni = 3 * per;
      for (int base = ex.tid; base < ni; base += 4 * ex.nthreads) {
        float v[4][4];
        int dst[4];
        for (int u = 0; u < 4; ++u) {
          const int i = base + u * ex.nthreads;
          dst[u] = -1;
          if (i < ni) {
            const int pl = i / per, rem = i - pl * per, rr = rem / qw, cc = ca + 4 * (rem - rr * qw);
            ld4(yb + pl * x.nn + (size_t)(g.fr0 + rr) * n + cc, v[u]);
            dst[u] = pl * g.nf + rr * g.wf + (cc - g.fo);
          }
        }
        for (int u = 0; u < 4; ++u) if (dst[u] >= 0) st4(x.F + dst[u], v[u]);
      }
    } else {
      const int ca = imax(g.sc0 - 2, 0), cb = imin(g.sc1 + 2, n), iw = cb - ca, per = rows * iw, ni = 3 * per;
      for (int base = ex.tid; base < ni; base += 8 * ex.nthreads) {
        float v[8];
        int dst[8];
        for (int u = 0; u < 8; ++u) {
          const int i = base + u * ex.nthreads;
          dst[u] = -1;
          if (i < ni) {
            const int pl = i / per, rem = i - pl * per, rr = rem / iw, cc = ca + (rem - rr * iw);
            v[u] = yb[pl * x.nn + (size_t)(g.fr0 + rr) * n + cc];
            dst[u] = pl * g.nf + rr * g.wf + (cc - g.fo);
          }
        }
        for (int u = 0; u < 8; ++u) if (dst[u] >= 0) x.F[dst[u]] = v[u];
      }
    }
  }
  ex.barrier();
  x.U = SPlane{x.F, g.fr0, g.fr1, g.fo, g.wf};
  x.X1 = SPlane{x.F + g.nf, g.fr0, g.fr1, g.fo, g.wf};
  x.X2 = SPlane{x.F + 2 * g.nf, g.fr0, g.fr1, g.fo, g.wf};
  // ---- phase B: the residuals on the source strips
  for (int i = ex.tid; i < nslots_b; i += ex.nthreads) {
    const int rr = i / sw, r = g.sr0 + rr, c0 = g.sc0 + 4 * (i - rr * sw);
    if (i != ex.tid) load_k(x, r, c0, kk);
    phase_b_strip<BWD>(x, r, c0, kk, sums);
  }
  if (!BWD) return;
  ex.barrier();
  // ---- phase C: dL/dy on the own strips
  x.G1 = SPlane{x.S, g.sr0, g.sr1, g.so, g.ws};
  x.G2 = SPlane{x.S + g.ns, g.sr0, g.sr1, g.so, g.ws};
  x.GC = SPlane{x.S + 2 * g.ns, g.sr0, g.sr1, g.so, g.ws};
  const int ow = g.wd >> 2, nslots_c = (g.r1 - g.r0) * ow;
  for (int i = ex.tid; i < nslots_c; i += ex.nthreads) {
    const int rr = i / ow;
    phase_c_strip(x, g.r0 + rr, g.c0 + 4 * (i - rr * ow));
  }
}

}  // namespace gen
}  // namespace pdes
