// The fused Darcy mixed-residual loss for any square field of 8 .. 256 pixels as ROW BANDS: the structure of the
// 16 / 32 / 64 kernels of darcy_loss.hip (whole rows of 1 x 4 strips in consecutive lanes, vertical neighbours by 16-byte
// LDS reads, horizontal neighbours from the adjacent lane, the three LDS planes reused for the adjoint sources) without
// their compile-time size.  Host/device code: darcy_loss_generic.hip runs it as a workgroup (neighbour lanes by DPP wave
// shifts), tests/emu/darcy_generic_emu.cpp as a lane-by-lane emulation (neighbour lanes from arrays) that the CPU tests
// compare with the oracle before any GPU run.
//
// Reference (file:line relative to the reference repository): utils/image_gradient.py:26-92 (SobelFilter for any imsize,
// correct=True/False), models/darcy.py:162-233 (the loss functions; their docstrings use 65 x 65 fields).
//
// Geometry.  An image row is SPR = ceil(n / 4) strips; the last strip holds jl + 1 = ((n - 1) & 3) + 1 real columns.  The
// strips of a workgroup pass are packed DENSELY into its 64 * waves lanes: slot s = 64 wave + lane holds strip s % SPR of
// the pass's row s / SPR, RPW = 64 waves / SPR whole rows per pass (the slots behind RPW * SPR idle: 65 x 65 fills 255 of 256
// lanes, 130 x 130 495 of 512; a wave-by-wave packing of whole rows would leave 13 resp. 31 of every 64 idle).  A strip's
// left / right neighbour is the adjacent lane; where a row runs across a wave boundary (64 not a multiple of SPR: `seam`)
// lane 63 and the next wave's lane 0 hand each other their edge columns through a few LDS words.  A workgroup owns a BAND
// of rows [r0, r1) of one image: it computes
// the residuals ("sources") on [r0 - 1, r1 + 1) in `npass` passes from the fields on [r0 - 2, r1 + 2) (three LDS planes of
// row stride 4 * SPR floats), keeps the direct terms of its own rows in registers, overwrites the planes with the
// sources and applies the adjoint stencils on its own rows.  Rows 0 .. 2 and n-3 .. n-1 couple through the reference's
// boundary `modifier` (one-sided differences): bands are at least 3 rows, so those sit in the first / last band together.
//
// Global memory.  A strip is four consecutive floats of an image row: 16-byte accesses when n is a multiple of 4 behind
// aligned pointers (width class J = 4), otherwise the same four-dword accesses at DWORD alignment (global_load_dwordx4 needs
// no more on this part), the row's last strip read as the row's last four floats and shifted (never a byte outside the
// row), its jl + 1 real columns stored one by one.
//
// Arithmetic: S = replicate-edge [1,2,1]/4 smoother, A = clamped central difference (x modifier when `correct`):
// grad_h = n S_rows(.) A_cols, grad_v = n A_rows(.) S_cols, adjoints with A^T.  The column operators are written as the
// interior stencil on (replicate- resp. zero-) extended strips plus border terms that are per-lane CONSTANTS
// (LaneConst): waves never diverge on border strips.
#pragma once

#include "darcy_generic.h"

namespace pdes {
namespace band {

using gen::imax;
using gen::imin;
using gen::kNoTB;
using gen::kNonlinear;
using gen::kUncorrected;
using gen::ld4;
using gen::sqrt_f;
using gen::st4;

constexpr int kMinN = 8, kMaxN = 256;
constexpr int kRowTab = 12;      // dwords of the row table per field row (RowTab below)

struct V4 { float v[4]; };

// ---- the plan of a field size ---------------------------------------------------------------------------------------
struct Plan {
  int n, spr, jl, rpw, w;      // strips per row, last real column of the last strip, rows per workgroup pass, LDS row stride
  int waves, npass, cap;       // workgroup: waves, passes, source rows it can hold = rpw * npass
  int nbands, rows_f;          // bands per image; field rows of the tallest band
  int own_base, own_rem;       // band b owns own_base (+1 for b < own_rem) rows: first row b * own_base + min(b, own_rem)
  int inv_spr;                 // slot / spr == (slot * inv_spr) >> 16 for slot < 512
  int seam;                    // rows run across wave boundaries (64 % spr != 0 and more than one wave)
  long long lds_floats;        // 3 * rows_f * w + the row table (kRowTab floats per field row)
};

PDES_HD int band_lo(const Plan& p, int band) { return band * p.own_base + imin(band, p.own_rem); }

// fills the fields that follow from (n, waves, npass, nbands); false: the shape cannot hold the bands
inline bool make_plan(int n, int waves, int npass, int nbands, Plan& p) {
  const int spr = (n + 3) >> 2;
  p.n = n; p.spr = spr; p.jl = (n - 1) & 3; p.w = 4 * spr;
  p.waves = waves; p.npass = npass; p.rpw = 64 * waves / spr; p.cap = p.rpw * npass;
  p.nbands = nbands;
  if (nbands < 1 || p.rpw < 1) return false;
  const int own_max = (n + nbands - 1) / nbands;
  if (nbands == 1 ? p.cap < n : (own_max + 2 > p.cap || n / nbands < 3)) return false;
  p.own_base = n / nbands; p.own_rem = n % nbands;
  p.inv_spr = 65536 / spr + 1;
  p.seam = (waves > 1 && 64 % spr != 0) ? 1 : 0;
  p.rows_f = imin(own_max + 4, n);
  p.lds_floats = 3ll * p.rows_f * p.w + (long long)kRowTab * p.rows_f;
  return true;
}

// The same plan as a compile-time constant (round 6: instantiations of the kernel with the geometry folded in for the common
// sizes -- every n, spr, w, rpw, plane offset and band boundary becomes an immediate).  The launcher uses such an instantiation
// only when plan_equal() says that choose_plan picked exactly this plan at run time.
constexpr Plan fixed_plan(int n, int waves, int npass, int nbands) {
  Plan p{};
  const int spr = (n + 3) >> 2;
  p.n = n; p.spr = spr; p.jl = (n - 1) & 3; p.w = 4 * spr;
  p.waves = waves; p.npass = npass; p.rpw = 64 * waves / spr; p.cap = p.rpw * npass;
  p.nbands = nbands;
  const int own_max = (n + nbands - 1) / nbands;
  p.own_base = n / nbands; p.own_rem = n % nbands;
  p.inv_spr = 65536 / spr + 1;
  p.seam = (waves > 1 && 64 % spr != 0) ? 1 : 0;
  p.rows_f = own_max + 4 < n ? own_max + 4 : n;
  p.lds_floats = 3ll * p.rows_f * p.w + (long long)kRowTab * p.rows_f;
  return p;
}
inline bool plan_equal(const Plan& a, const Plan& b) {
  return a.n == b.n && a.spr == b.spr && a.jl == b.jl && a.rpw == b.rpw && a.w == b.w && a.waves == b.waves &&
         a.npass == b.npass && a.cap == b.cap && a.nbands == b.nbands && a.rows_f == b.rows_f && a.own_base == b.own_base &&
         a.own_rem == b.own_rem && a.inv_spr == b.inv_spr && a.seam == b.seam && a.lds_floats == b.lds_floats;
}

// waves in {1, 2, 4, 8} (workgroups of 3, 5, 6, 7 waves load the four SIMDs of a CU unevenly: measured 10-45 % slower at the
// same slot use), passes in {1, 2}: the most strip slots doing own work.  Ties (measured, EXPERIMENTS.md round 5): two
// passes before one (half the bands, half the halo rows: 5-12 %), then 4 waves, 2, 8, 1 (2-4 %).  False: size not served.
PDES_HD int plan_rank(int waves, int npass) {
  const int wr = waves == 4 ? 0 : (waves == 2 ? 1 : (waves == 8 ? 2 : 3));
  return (2 - npass) * 4 + wr;       // smaller is better
}
inline bool choose_plan(int n, long long lds_floats_max, Plan& best) {
  if (n < kMinN || n > kMaxN) return false;
  const int spr = (n + 3) >> 2;
  bool found = false;
  double best_eff = -1.0;
  for (int waves = 1; waves <= 8; waves *= 2)
    for (int npass = 1; npass <= 2; ++npass) {
      const int cap = (64 * waves / spr) * npass;
      int nbands = 1;
      if (cap < n) {
        if (cap < 5) continue;
        nbands = (n + (cap - 2) - 1) / (cap - 2);
      }
      Plan p;
      if (!make_plan(n, waves, npass, nbands, p) || p.lds_floats > lds_floats_max) continue;
      const double eff = (double)n * spr / ((double)nbands * npass * 64 * waves);
      const bool tie = found && eff > best_eff - 1e-9 && eff < best_eff + 1e-9;
      const bool tie_wins = tie && plan_rank(waves, npass) < plan_rank(best.waves, best.npass);
      if (!found || eff >= best_eff + 1e-9 || tie_wins) {
        best = p; best_eff = eff; found = true;
      }
    }
  return found;
}

struct BandGeo {
  int r0, r1;       // own rows
  int sr0, sr1;     // source rows
  int fr0, fr1;     // field rows
};
PDES_HD BandGeo band_geo(const Plan& p, int band) {
  BandGeo g;
  g.r0 = band_lo(p, band); g.r1 = band_lo(p, band + 1);
  g.sr0 = imax(g.r0 - 1, 0); g.sr1 = imin(g.r1 + 1, p.n);
  g.fr0 = imax(g.sr0 - 1, 0); g.fr1 = imin(g.sr1 + 1, p.n);
  return g;
}

// ---- rows: clamped neighbours and the coefficients of the vertical difference / its adjoint ---------------------------
struct RowGeom {
  int up, dn, farF, farA;
  float f_own, f_up, f_dn, f_far;   // forward:  f_own x[r] + f_up x[up] + f_dn x[dn] + f_far x[farF]
  float a_own, a_up, a_dn, a_far;   // adjoint:  a_own g[r] + a_up g[up] + a_dn g[dn] + a_far g[farA]
};
PDES_HD RowGeom row_geom(int r, int n, bool correct) {
  RowGeom g;
  g.up = r > 0 ? r - 1 : 0;
  g.dn = r < n - 1 ? r + 1 : n - 1;
  g.farF = r; g.farA = r;
  g.f_own = 0.f; g.f_up = -0.5f; g.f_dn = 0.5f; g.f_far = 0.f;
  g.a_own = 0.f; g.a_up = 0.5f; g.a_dn = -0.5f; g.a_far = 0.f;
  if (r == 0) {
    if (correct) { g.f_own = -1.5f; g.f_up = 0.f; g.f_dn = 2.f; g.f_far = -0.5f; g.farF = 2; g.a_own = -1.5f; }
    else { g.f_own = -0.5f; g.f_up = 0.f; g.a_own = -0.5f; }
    g.a_up = 0.f;
  } else if (r == n - 1) {
    if (correct) { g.f_own = 1.5f; g.f_up = -2.f; g.f_dn = 0.f; g.f_far = 0.5f; g.farF = n - 3; g.a_own = 1.5f; }
    else { g.f_own = 0.5f; g.f_dn = 0.f; g.a_own = 0.5f; }
    g.a_dn = 0.f;
  } else if (correct) {
    if (r == 1) g.a_up = 2.f;
    if (r == n - 2) g.a_dn = -2.f;
    if (r == 2) { g.a_far = -0.5f; g.farA = 0; }
    if (r == n - 3) { g.a_far = 0.5f; g.farA = n - 1; }      // (n >= 8: the four cases are distinct rows)
  }
  return g;
}

// ---- what a lane knows about its strip column (constant over the passes) ---------------------------------------------
struct LaneConst {
  int cs, lrow;            // strip of the row; row of the workgroup's pass (slot / spr)
  bool active;             // slot < rpw * spr
  bool first, last;
  bool valid[4];           // column < n
  float cl[4], cr[4];      // adjoint of the column difference: border terms cl[j] g[0] + cr[j] g[n-1]
};
// the border terms of the column difference's adjoint (needed by phase C only: the kernel fills them in behind phase B)
PDES_HD void lane_const_adj(const Plan& p, LaneConst& c, bool correct) {
  for (int j = 0; j < 4; ++j) {
    const int col = 4 * c.cs + j;
    c.cl[j] = 0.f; c.cr[j] = 0.f;
    if (correct) {
      if (col == 0) c.cl[j] = -1.5f;
      if (col == 1) c.cl[j] = 1.5f;
      if (col == 2) c.cl[j] = -0.5f;
      if (col == p.n - 1) c.cr[j] = 1.5f;
      if (col == p.n - 2) c.cr[j] = -1.5f;
      if (col == p.n - 3) c.cr[j] = 0.5f;
    } else {
      if (col == 0) c.cl[j] = -0.5f;
      if (col == p.n - 1) c.cr[j] = 0.5f;
    }
  }
}
PDES_HD LaneConst lane_const(const Plan& p, int slot, bool correct, bool with_adj = true) {       // slot = 64 wave + lane
  LaneConst c;
  c.active = slot < p.rpw * p.spr;
  c.lrow = (slot * p.inv_spr) >> 16;
  c.cs = slot - c.lrow * p.spr;
  c.first = c.cs == 0;
  c.last = c.cs == p.spr - 1;
  for (int j = 0; j < 4; ++j) {
    c.valid[j] = 4 * c.cs + j < p.n;
    c.cl[j] = 0.f; c.cr[j] = 0.f;
  }
  if (with_adj) lane_const_adj(p, c, correct);
  return c;
}

// ---- strips -----------------------------------------------------------------------------------------------------------
// An LDS plane of a band: rows [fr0, fr1) of the image, row stride w; strip (r, cs) = 4 floats at (r - fr0) * w + 4 cs.
struct BPlane {
  const float* p;
  int fr0, fr1, w;
};
PDES_HD V4 ldoff(const BPlane& P, int off) {
  V4 o;
  ld4(P.p + off, o.v);
  return o;
}
// The row table of a band, built once per workgroup (one thread per field row) behind the planes: what row_geom says about
// a row, with the neighbour rows as plane offsets -- a strip reads 2 x 16 bytes per phase instead of re-deriving ~50
// selects.  Entry of field row r (fr0 <= r < fr1) at dword kRowTab * (r - fr0):
//   ints   [0..3]  (up - fr0) w, (dn - fr0) w, (farF - fr0) w, (farA - fr0) w        (rows clamped into the plane)
//   floats [4..7]  f_own, f_up, f_dn, f_far        [8..11]  a_own, a_up, a_dn, a_far
union Dword { float f; int i; };
PDES_HD void rowtab_build(float* tab, int r, int n, bool correct, int fr0, int fr1, int w) {
  const RowGeom g = row_geom(r, n, correct);
  float* t = tab + kRowTab * (r - fr0);
  const int rows[4] = {g.up, g.dn, g.farF, g.farA};
  for (int k = 0; k < 4; ++k) {
    const int rr = rows[k] < fr0 ? fr0 : (rows[k] >= fr1 ? fr1 - 1 : rows[k]);
    Dword d;
    d.i = (rr - fr0) * w;
    t[k] = d.f;
  }
  t[4] = g.f_own; t[5] = g.f_up; t[6] = g.f_dn; t[7] = g.f_far;
  t[8] = g.a_own; t[9] = g.a_up; t[10] = g.a_dn; t[11] = g.a_far;
}
struct RowTab {
  int o_own, o_up, o_dn, o_far;     // float offsets of the strip (row, cs) and of its neighbour rows in a plane
  float k_own, k_up, k_dn, k_far;   // the vertical difference (forward or adjoint)
};
template <bool ADJ>
PDES_HD RowTab rowtab_read(const float* tab, int r, int cs, int fr0, int w) {
  const float* t = tab + kRowTab * (r - fr0);
  float a[4], b[4];
  ld4(t, a);
  ld4(t + (ADJ ? 8 : 4), b);
  Dword d0, d1, d2;
  d0.f = a[0]; d1.f = a[1]; d2.f = ADJ ? a[3] : a[2];
  RowTab o;
  o.o_own = (r - fr0) * w + 4 * cs;
  o.o_up = d0.i + 4 * cs; o.o_dn = d1.i + 4 * cs; o.o_far = d2.i + 4 * cs;
  o.k_own = b[0]; o.k_up = b[1]; o.k_dn = b[2]; o.k_far = b[3];
  return o;
}
PDES_HD V4 vsmooth3(const V4& up, const V4& own, const V4& dn) {
  V4 o;
  for (int i = 0; i < 4; ++i) o.v[i] = 0.25f * up.v[i] + 0.5f * own.v[i] + 0.25f * dn.v[i];
  return o;
}
PDES_HD V4 comb4(float a, const V4& x, float b, const V4& y, float c, const V4& z, float d, const V4& w) {
  V4 o;
  for (int i = 0; i < 4; ++i) o.v[i] = a * x.v[i] + b * y.v[i] + c * z.v[i] + d * w.v[i];
  return o;
}
// x.v[j] for a uniform runtime j without indexing the register array
PDES_HD float pick(const V4& x, int j) { return j == 0 ? x.v[0] : (j == 1 ? x.v[1] : (j == 2 ? x.v[2] : x.v[3])); }
// the columns behind the image's last one: replicate it (forward operators, smoothing) / zero (adjoint of the difference)
// The functions below are instantiated per WIDTH CLASS J: J = 0 .. 3 is jl, the last real column of the last strip (J = 3: a
// multiple of 4 behind pointers that are not 16-byte aligned); J = 4 is the aligned multiple of 4 (jl = 3, every access 16
// bytes).  With jl a compile-time constant the tail handling is a handful of selects on the lane's `last` flag.
template <int J> struct WidthClass { static constexpr int jl = J == 4 ? 3 : J; static constexpr bool aligned = J == 4; };

template <int J>
PDES_HD void tail_replicate(V4& x, const LaneConst& c) {
  constexpr int jl = WidthClass<J>::jl;
  for (int j = 1; j < 4; ++j) if (j > jl) x.v[j] = c.last ? x.v[jl] : x.v[j];
}
template <int J>
PDES_HD void tail_zero(V4& x, const LaneConst& c) {
  constexpr int jl = WidthClass<J>::jl;
  for (int j = 1; j < 4; ++j) if (j > jl) x.v[j] = c.last ? 0.f : x.v[j];
}

// what a strip takes from its neighbour lanes
struct Halo {
  float l, l2;     // the left neighbour's v[3], v[2]
  float r, rjl;    // the right neighbour's v[0], v[jl]
};

// scale * [1,2,1]/4 along the row, replicate edges.  x: tail-replicated.
PDES_HD V4 hsmooth(const V4& x, const Halo& h, const LaneConst& c, float scale) {
  const float l = c.first ? x.v[0] : h.l, r = c.last ? x.v[3] : h.r;
  V4 o;
  o.v[0] = scale * (0.25f * l + 0.5f * x.v[0] + 0.25f * x.v[1]);
  o.v[1] = scale * (0.25f * x.v[0] + 0.5f * x.v[1] + 0.25f * x.v[2]);
  o.v[2] = scale * (0.25f * x.v[1] + 0.5f * x.v[2] + 0.25f * x.v[3]);
  o.v[3] = scale * (0.25f * x.v[2] + 0.5f * x.v[3] + 0.25f * r);
  return o;
}
// scale * (x A) along the row: clamped central difference; `correct`: one-sided second-order differences in the first /
// last column (image_gradient.py:43-46).  x: tail-replicated.
template <int J>
PDES_HD V4 hdiff(const V4& x, const Halo& h, const LaneConst& c, bool correct, float scale) {
  constexpr int jl = WidthClass<J>::jl;
  const float l = c.first ? x.v[0] : h.l, r = c.last ? x.v[3] : h.r;
  V4 o;
  o.v[0] = 0.5f * (x.v[1] - l);
  o.v[1] = 0.5f * (x.v[2] - x.v[0]);
  o.v[2] = 0.5f * (x.v[3] - x.v[1]);
  o.v[3] = 0.5f * (r - x.v[2]);
  if (correct) {
    const float e0 = 0.5f * (-3.f * x.v[0] + 4.f * x.v[1] - x.v[2]);
    o.v[0] = c.first ? e0 : o.v[0];
    const float m1 = jl >= 1 ? x.v[jl >= 1 ? jl - 1 : 0] : h.l;
    const float m2 = jl >= 2 ? x.v[jl >= 2 ? jl - 2 : 0] : (jl == 1 ? h.l : h.l2);
    const float e = 0.5f * (3.f * x.v[jl] - 4.f * m1 + m2);
    o.v[jl] = c.last ? e : o.v[jl];
  }
  for (int i = 0; i < 4; ++i) o.v[i] *= scale;
  return o;
}
// scale * (g A^T) along the row.  g: tail-zeroed.
template <int J>
PDES_HD V4 hdiff_adj(const V4& g, const Halo& h, const LaneConst& c, float scale) {
  constexpr int jl = WidthClass<J>::jl;
  const float l = c.first ? 0.f : h.l, r = c.last ? 0.f : h.r;
  const float g0 = g.v[0], gl = jl >= 2 ? g.v[jl] : (c.last ? g.v[jl] : h.rjl);      // (jl >= 2: n-1, n-2, n-3 are all in the last strip)
  V4 o;
  o.v[0] = 0.5f * (l - g.v[1]) + c.cl[0] * g0 + c.cr[0] * gl;
  o.v[1] = 0.5f * (g.v[0] - g.v[2]) + c.cl[1] * g0 + c.cr[1] * gl;
  o.v[2] = 0.5f * (g.v[1] - g.v[3]) + c.cl[2] * g0 + c.cr[2] * gl;
  o.v[3] = 0.5f * (g.v[2] - r) + c.cl[3] * g0 + c.cr[3] * gl;
  for (int i = 0; i < 4; ++i) o.v[i] *= scale;
  return o;
}

// ---- phase B of one strip, in two halves around the neighbour exchange ---------------------------------------------------
struct FwdVert {          // vertical combinations (tail-replicated) + the strip's own values
  V4 us, ud, as, bd;      // S_rows u, A_rows u, S_rows sigma1, A_rows sigma2
  V4 u, s1, s2;
};
template <int J>
PDES_HD FwdVert fwd_vert(const BPlane& U, const BPlane& X1, const BPlane& X2, const RowTab& t, const LaneConst& c) {
  FwdVert o;
  o.u = ldoff(U, t.o_own); o.s1 = ldoff(X1, t.o_own); o.s2 = ldoff(X2, t.o_own);      // (the three planes share their geometry)
  const V4 u_up = ldoff(U, t.o_up), u_dn = ldoff(U, t.o_dn), u_far = ldoff(U, t.o_far);
  const V4 a_up = ldoff(X1, t.o_up), a_dn = ldoff(X1, t.o_dn);
  const V4 b_up = ldoff(X2, t.o_up), b_dn = ldoff(X2, t.o_dn), b_far = ldoff(X2, t.o_far);
  o.us = vsmooth3(u_up, o.u, u_dn);
  o.ud = comb4(t.k_own, o.u, t.k_up, u_up, t.k_dn, u_dn, t.k_far, u_far);
  o.as = vsmooth3(a_up, o.s1, a_dn);
  o.bd = comb4(t.k_own, o.s2, t.k_up, b_up, t.k_dn, b_dn, t.k_far, b_far);
  tail_replicate<J>(o.us, c); tail_replicate<J>(o.ud, c);
  tail_replicate<J>(o.as, c); tail_replicate<J>(o.bd, c);
  return o;
}

struct StripOut {
  V4 d1, d2;              // direct part of dL/dsigma1, dL/dsigma2 (kept by the lane until phase C)
  float du;               // direct part of dL/du: the Dirichlet term of the strip's border column (first / last strip)
  V4 p1, p2, cc;          // adjoint sources a_const K r1, a_const K r2, a_cont c (zero outside the image)
};
// `own`: the strip's row belongs to the band (its pixels enter the sums)
template <int J>
PDES_HD StripOut fwd_finish(const FwdVert& f, const Halo& hus, const Halo& hud, const Halo& has, const Halo& hbd, const V4& K,
                            int r, int n, const LaneConst& c, const LossParams& p, int flags, float fn, bool own,
                            float* sums) {
  constexpr int jl = WidthClass<J>::jl;
  const bool correct = !(flags & kUncorrected);
  const V4 ghu = hdiff<J>(f.us, hus, c, correct, fn);
  const V4 gvu = hsmooth(f.ud, hud, c, fn);
  const V4 gh1 = hdiff<J>(f.as, has, c, correct, fn);
  const V4 gv2 = hsmooth(f.bd, hbd, c, fn);
  const bool tb = (r == 0) || (r == n - 1);
  StripOut o;
  o.du = 0.f;
  for (int j = 0; j < 4; ++j) {
    const float k = K.v[j], x1 = f.s1.v[j], x2 = f.s2.v[j];
    float r1 = x1 + k * ghu.v[j], r2 = x2 + k * gvu.v[j], q1 = 1.f, q2 = 1.f;
    if (flags & kNonlinear) {                 // darcy.py:179-191
      const float sq = sqrt_f(k);
      r1 += p.beta1 * sq * x1 * x1 + p.beta2 * k * x1 * x1 * x1;
      r2 += p.beta1 * sq * x2 * x2 + p.beta2 * k * x2 * x2 * x2;
      q1 += 2.f * p.beta1 * sq * x1 + 3.f * p.beta2 * k * x1 * x1;
      q2 += 2.f * p.beta1 * sq * x2 + 3.f * p.beta2 * k * x2 * x2;
    }
    const float cc = ((flags & kNoTB) && tb) ? 0.f : gh1.v[j] + gv2.v[j];     // darcy.py:224
    const bool in = j <= jl || c.valid[j];         // (columns up to jl exist in every strip)
    o.p1.v[j] = in ? p.a_const * k * r1 : 0.f;
    o.p2.v[j] = in ? p.a_const * k * r2 : 0.f;
    o.cc.v[j] = in ? p.a_cont * cc : 0.f;
    o.d1.v[j] = p.a_const * r1 * q1;
    o.d2.v[j] = p.a_const * r2 * q2 + (tb ? p.b_neu * x2 : 0.f);
    if (own && in) {
      sums[0] += r1 * r1 + r2 * r2;
      sums[1] += cc * cc;
      if (tb) sums[3] += x2 * x2;
    }
  }
  // Dirichlet columns (darcy.py:226-233): u = 1 on the left, u = 0 on the right
  {                                           // (a strip is never first and last: n >= 8)
    const float e = c.first ? f.u.v[0] - 1.f : f.u.v[jl];
    const bool edge = c.first || c.last;
    o.du = edge ? p.b_dir * e : 0.f;
    sums[2] += (own && edge) ? e * e : 0.f;
  }
  return o;
}

// ---- phase C of one strip -----------------------------------------------------------------------------------------------
struct AdjVert {
  V4 p1s, p2d, ccs, ccd;  // S_rows p1, A^T_rows p2, S_rows cc, A^T_rows cc
};
template <int J>
PDES_HD AdjVert adj_vert(const BPlane& G1, const BPlane& G2, const BPlane& GC, const RowTab& t, const LaneConst& c) {
  AdjVert o;
  const V4 p1 = ldoff(G1, t.o_own), p1_up = ldoff(G1, t.o_up), p1_dn = ldoff(G1, t.o_dn);
  const V4 p2 = ldoff(G2, t.o_own), p2_up = ldoff(G2, t.o_up), p2_dn = ldoff(G2, t.o_dn), p2_far = ldoff(G2, t.o_far);
  const V4 cc = ldoff(GC, t.o_own), cc_up = ldoff(GC, t.o_up), cc_dn = ldoff(GC, t.o_dn), cc_far = ldoff(GC, t.o_far);
  o.p1s = vsmooth3(p1_up, p1, p1_dn);
  o.p2d = comb4(t.k_own, p2, t.k_up, p2_up, t.k_dn, p2_dn, t.k_far, p2_far);
  o.ccs = vsmooth3(cc_up, cc, cc_dn);
  o.ccd = comb4(t.k_own, cc, t.k_up, cc_up, t.k_dn, cc_dn, t.k_far, cc_far);
  tail_zero<J>(o.p1s, c); tail_zero<J>(o.ccs, c);                           // -> the column difference's adjoint
  tail_replicate<J>(o.p2d, c); tail_replicate<J>(o.ccd, c);                 // -> the (symmetric) column smoothing
  return o;
}
// dL/du, dL/dsigma1, dL/dsigma2 of the strip
template <int J>
PDES_HD void adj_finish(const AdjVert& a, const Halo& hp1, const Halo& hp2, const Halo& hcs, const Halo& hcd, const StripOut& s,
                        const LaneConst& c, float fn, V4& du, V4& d1, V4& d2) {
  constexpr int jl = WidthClass<J>::jl;
  const V4 ghT_p1 = hdiff_adj<J>(a.p1s, hp1, c, fn);
  const V4 gvT_p2 = hsmooth(a.p2d, hp2, c, fn);
  const V4 ghT_c = hdiff_adj<J>(a.ccs, hcs, c, fn);
  const V4 gvT_c = hsmooth(a.ccd, hcd, c, fn);
  for (int j = 0; j < 4; ++j) {
    du.v[j] = ghT_p1.v[j] + gvT_p2.v[j];
    d1.v[j] = s.d1.v[j] + ghT_c.v[j];
    d2.v[j] = s.d2.v[j] + gvT_c.v[j];
  }
  du.v[0] += c.first ? s.du : 0.f;
  du.v[jl] += c.last ? s.du : 0.f;
}

// (pass, slot) -> source row of the band (may be >= sr1: an idle slot)
PDES_HD int slot_row(const Plan& p, const BandGeo& g, int pass, const LaneConst& c) {
  return g.sr0 + pass * p.rpw + c.lrow;
}

// the last strip of a row arrives as the row's LAST FOUR floats (columns n-4 .. n-1): its jl + 1 real columns are the
// vector's last ones -> shifted to the front, the tail zeroed
template <int J>
PDES_HD V4 last_strip_shift(const V4& x, const LaneConst& c) {
  constexpr int jl = WidthClass<J>::jl;
  V4 o;
  for (int j = 0; j < 4; ++j) {
    const int src = j + 3 - jl < 3 ? j + 3 - jl : 3;
    o.v[j] = c.last ? (j <= jl ? x.v[src] : 0.f) : x.v[j];
  }
  return o;
}
// float offset of the strip's four-dword access in its image row
template <int J>
PDES_HD int strip_col(const Plan& p, const LaneConst& c) {
  return (WidthClass<J>::jl < 3 && c.last) ? p.n - 4 : 4 * c.cs;
}

}  // namespace band
}  // namespace pdes
