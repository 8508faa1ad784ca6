// Test infrastructure (CPU): the per-pixel arithmetic and the tile geometry of csrc/darcy_loss_generic.hip compiled as
// plain C++ (csrc/darcy_generic.h is host/device code): process_tile -- the whole procedure of one workgroup, phases,
// strip fast path and per-pixel border path included -- runs here with ONE serial "thread" per tile and a poisoned buffer
// standing in for the LDS.  tests/test_generic_loss_cpu.py builds this with g++ and compares it with the oracle
// for field sizes / tile sizes that exercise every halo case, so the kernels' arithmetic is checked before any GPU run.
// It is NOT a product path: nothing under pde_surrogate_amd/ loads it.
#include <vector>
#include <cstddef>
#include "../../pde_surrogate_amd/csrc/darcy_generic.h"
#include "../../pde_surrogate_amd/csrc/darcy_band.h"

using namespace pdes;
using namespace pdes::gen;

extern "C" {

struct SerialExec {                     // one "thread" runs every slot; the phases are sequential already
  int tid = 0, nthreads = 1;
  void barrier() const {}
};

// tr/tc <= 0: the kernel's own choice for `budget` floats of LDS.  partials: (B * tiles, 4).  Returns the number of
// tiles per image (0: no fit).
int emu_darcy_loss(const float* Kp, const float* yp, float* gyp, float* partials, int B, int n, float a_const,
                   float a_cont, float b_dir, float b_neu, float beta1, float beta2, int flags, int tr, int tc,
                   long long budget) {
  const bool strips = n >= 8;                                  // = STRIP_MIN_N of darcy_loss_generic.hip
  if (tr <= 0 || tc <= 0) {
    if (strips ? !choose_strip_tile(n, budget, tr, tc) : !choose_tile(n, budget, tr, tc)) return 0;
  }
  if (strips) tc = (tc + 3) & ~3;                              // tile columns start on strip boundaries
  const long long need = strips ? strip_tile_floats(tr, tc, n) : tile_floats(tr, tc, n);
  if (need > budget) return 0;
  LossParams p{a_const, a_cont, b_dir, b_neu, beta1, beta2, 0};
  const size_t nn = (size_t)n * n;
  const int ntr = (n + tr - 1) / tr, ntc = (n + tc - 1) / tc;
  for (int b = 0; b < B; ++b)
    for (int tile = 0; tile < ntr * ntc; ++tile) {
      std::vector<float> lds((size_t)need + 8, -12345.f);     // poison: a read outside what was written / zeroed shows
      float sums[4] = {0, 0, 0, 0};
      SerialExec ex;
      const float* Kb = Kp + b * nn;
      const float* yb = yp + b * 3 * nn;
      float* gb = gyp ? gyp + b * 3 * nn : nullptr;
      if (strips) {
        const StripGeo g = strip_geo(n, tr, tc, tile / ntc, tile % ntc);
        if (3ll * (g.nf + g.ns + g.nd) > need) return 0;       // the bound the tile choice relies on
        if (gb) process_tile_strips<true>(Kb, yb, gb, n, g, p, flags, lds.data(), ex, sums);
        else process_tile_strips<false>(Kb, yb, nullptr, n, g, p, flags, lds.data(), ex, sums);
      } else {
        const TileGeo g = tile_geo(n, tr, tc, tile / ntc, tile % ntc);
        if (gb) process_tile_pixelwise<true>(Kb, yb, gb, n, g, p, flags, lds.data(), ex, sums);
        else process_tile_pixelwise<false>(Kb, yb, nullptr, n, g, p, flags, lds.data(), ex, sums);
      }
      for (int k = 0; k < 4; ++k) partials[((size_t)b * ntr * ntc + tile) * 4 + k] = sums[k];
    }
  return ntr * ntc;
}

void emu_sobel(const float* img, float* gh, float* gv, int nimg, int n, int correct, int five) {
  for (int b = 0; b < nimg; ++b) {
    const size_t base = (size_t)b * n * n;
    const Plane P{img + base, 0, 0, n, n};
    for (int r = 0; r < n; ++r)
      for (int c = 0; c < n; ++c) {
        const size_t o = base + (size_t)r * n + c;
        if (gh) gh[o] = five ? sobel5_grad<true>(P, r, c, correct != 0) : sobel_grad<true>(P, r, c, correct != 0);
        if (gv) gv[o] = five ? sobel5_grad<false>(P, r, c, correct != 0) : sobel_grad<false>(P, r, c, correct != 0);
      }
  }
}

void emu_sobel_adjoint(const float* ghb, const float* gvb, float* out, int nimg, int n, int correct, int five) {
  for (int b = 0; b < nimg; ++b) {
    const size_t base = (size_t)b * n * n;
    const Plane Gh{ghb ? ghb + base : nullptr, 0, 0, n, n}, Gv{gvb ? gvb + base : nullptr, 0, 0, n, n};
    for (int r = 0; r < n; ++r)
      for (int c = 0; c < n; ++c) {
        float a = 0.f;
        if (five) a = sobel5_adj(Gh, Gv, r, c, correct != 0);
        else {
          if (ghb) a += sobel_adj<true>(Gh, r, c, correct != 0);
          if (gvb) a += sobel_adj<false>(Gv, r, c, correct != 0);
        }
        out[base + (size_t)r * n + c] = a;
      }
  }
}

// ---- the row-band kernel (csrc/darcy_band.h), lane by lane ------------------------------------------------------------------
// One workgroup = an array of 64 * waves slots; a strip's neighbour is the adjacent slot (inside a wave a DPP shift, across
// a wave boundary the seam words of darcy_loss_generic.hip; the first / last slot of the workgroup receive 0).  Every slot
// -- idle ones included -- runs the arithmetic, as on the GPU.  Global memory is touched as the kernel touches it: four
// floats per strip, a row's last strip as the row's last four floats, shifted.
// waves/npass <= 0: the kernel's own plan for `lds_floats_max` floats of LDS.  partials: (B * bands, 4).  Returns the bands
// per image (0: no plan).
}  // extern "C"

static band::Halo halo_from(const std::vector<band::V4>& x, int slot, int jl) {       // jl = WidthClass<J>::jl
  band::Halo h;
  const int last = (int)x.size() - 1;
  h.l = slot > 0 ? x[slot - 1].v[3] : 0.f;
  h.l2 = slot > 0 ? x[slot - 1].v[2] : 0.f;
  h.r = slot < last ? x[slot + 1].v[0] : 0.f;
  h.rjl = slot < last ? band::pick(x[slot + 1], jl) : 0.f;
  return h;
}

template <int J>
static band::V4 strip_load(const float* rowp, const band::Plan& pl, const band::LaneConst& c) {
  band::V4 x;
  ld4(rowp + band::strip_col<J>(pl, c), x.v);
  return band::WidthClass<J>::jl < 3 ? band::last_strip_shift<J>(x, c) : x;
}

template <int J>
static int band_emulation(const float* Kp, const float* yp, float* gyp, float* partials, int B, const band::Plan& pl,
                          const LossParams& p, int flags) {
  using namespace band;
  const int n = pl.n;
  const bool correct = !(flags & kUncorrected);
  const size_t nn = (size_t)n * n;
  const float fn = (float)n;
  const int jl = WidthClass<J>::jl, plane = pl.rows_f * pl.w, nslot = 64 * pl.waves;
  for (int b = 0; b < B; ++b)
    for (int bi = 0; bi < pl.nbands; ++bi) {
      const BandGeo g = band_geo(pl, bi);
      std::vector<float> lds((size_t)3 * pl.rows_f * pl.w, -12345.f);       // poison
      const float* Kb = Kp + b * nn;
      const float* yb = yp + b * 3 * nn;
      float* gb = gyp ? gyp + b * 3 * nn : nullptr;
      // stage the fields, strip by strip (rows fr0 .. fr1)
      for (int q = 0; q < 3; ++q)
        for (int r = g.fr0; r < g.fr1; ++r)
          for (int cs = 0; cs < pl.spr; ++cs) {
            LaneConst c;
            c.cs = cs; c.last = cs == pl.spr - 1;
            const V4 x = strip_load<J>(yb + q * nn + (size_t)r * n, pl, c);
            st4(lds.data() + (size_t)q * plane + (r - g.fr0) * pl.w + 4 * cs, x.v);
          }
      const BPlane U{lds.data(), g.fr0, g.fr1, pl.w}, X1{lds.data() + plane, g.fr0, g.fr1, pl.w},
          X2{lds.data() + 2 * plane, g.fr0, g.fr1, pl.w};
      std::vector<float> tab((size_t)kRowTab * pl.rows_f, -999.f);           // (behind the planes in the kernel's LDS)
      for (int r = g.fr0; r < g.fr1; ++r) rowtab_build(tab.data(), r, n, correct, g.fr0, g.fr1, pl.w);
      std::vector<StripOut> keep((size_t)pl.npass * nslot);
      std::vector<float> lane_sums((size_t)nslot * 4, 0.f);
      // ---- phase B
      for (int pass = 0; pass < pl.npass; ++pass) {
        std::vector<FwdVert> f(nslot);
        std::vector<V4> us(nslot), ud(nslot), as(nslot), bd(nslot);
        std::vector<int> rc(nslot);
        for (int slot = 0; slot < nslot; ++slot) {
          const LaneConst c = lane_const(pl, slot, correct);
          const int r = slot_row(pl, g, pass, c);
          rc[slot] = r < g.sr1 ? r : g.sr1 - 1;
          f[slot] = fwd_vert<J>(U, X1, X2, rowtab_read<false>(tab.data(), rc[slot], c.cs, g.fr0, pl.w), c);
          us[slot] = f[slot].us; ud[slot] = f[slot].ud; as[slot] = f[slot].as; bd[slot] = f[slot].bd;
        }
        for (int slot = 0; slot < nslot; ++slot) {
          const LaneConst c = lane_const(pl, slot, correct);
          const int r = slot_row(pl, g, pass, c);
          const bool ok = c.active && r < g.sr1, own = ok && r >= g.r0 && r < g.r1;
          V4 K;
          for (int j = 0; j < 4; ++j) K.v[j] = 0.f;
          if (ok) K = strip_load<J>(Kb + (size_t)r * n, pl, c);
          keep[(size_t)pass * nslot + slot] =
              fwd_finish<J>(f[slot], halo_from(us, slot, jl), halo_from(ud, slot, jl), halo_from(as, slot, jl),
                            halo_from(bd, slot, jl), K, rc[slot], n, c, p, flags, fn, own, &lane_sums[(size_t)slot * 4]);
        }
      }
      for (int k = 0; k < 4; ++k) {
        float t = 0.f;
        for (int wave = 0; wave < pl.waves; ++wave) {
          float wsum = 0.f;
          for (int lane = 0; lane < 64; ++lane) wsum += lane_sums[((size_t)wave * 64 + lane) * 4 + k];
          t += wsum;
        }
        partials[((size_t)b * pl.nbands + bi) * 4 + k] = t;
      }
      if (!gb) continue;
      // ---- the planes become the sources (after the barrier), rows sr0 .. sr1
      std::fill(lds.begin(), lds.end(), -54321.f);
      for (int pass = 0; pass < pl.npass; ++pass)
        for (int slot = 0; slot < nslot; ++slot) {
          const LaneConst c = lane_const(pl, slot, correct);
          const int r = slot_row(pl, g, pass, c);
          if (!(c.active && r < g.sr1)) continue;
          const StripOut& s = keep[(size_t)pass * nslot + slot];
          float* q = lds.data() + (r - g.fr0) * pl.w + 4 * c.cs;
          st4(q, s.p1.v); st4(q + plane, s.p2.v); st4(q + 2 * plane, s.cc.v);
        }
      const BPlane G1{lds.data(), g.fr0, g.fr1, pl.w}, G2{lds.data() + plane, g.fr0, g.fr1, pl.w},
          GC{lds.data() + 2 * plane, g.fr0, g.fr1, pl.w};
      // ---- phase C: own strips leave as four floats, a row's last strip column by column
      for (int pass = 0; pass < pl.npass; ++pass) {
        std::vector<AdjVert> a(nslot);
        std::vector<V4> p1s(nslot), p2d(nslot), ccs(nslot), ccd(nslot);
        for (int slot = 0; slot < nslot; ++slot) {
          const LaneConst c = lane_const(pl, slot, correct);
          const int r = slot_row(pl, g, pass, c);
          const int rc = r < g.r0 ? g.r0 : (r < g.r1 ? r : g.r1 - 1);          // idle / halo slots: any own row
          a[slot] = adj_vert<J>(G1, G2, GC, rowtab_read<true>(tab.data(), rc, c.cs, g.fr0, pl.w), c);
          p1s[slot] = a[slot].p1s; p2d[slot] = a[slot].p2d; ccs[slot] = a[slot].ccs; ccd[slot] = a[slot].ccd;
        }
        for (int slot = 0; slot < nslot; ++slot) {
          const LaneConst c = lane_const(pl, slot, correct);
          const int r = slot_row(pl, g, pass, c);
          V4 out[3];
          adj_finish<J>(a[slot], halo_from(p1s, slot, jl), halo_from(p2d, slot, jl), halo_from(ccs, slot, jl),
                        halo_from(ccd, slot, jl), keep[(size_t)pass * nslot + slot], c, fn, out[0], out[1], out[2]);
          if (!(c.active && r >= g.r0 && r < g.r1)) continue;
          for (int q = 0; q < 3; ++q) {
            float* o = gb + q * nn + (size_t)r * n + 4 * c.cs;
            if (jl == 3 || !c.last) st4(o, out[q].v);
            else for (int j = 0; j <= jl; ++j) o[j] = out[q].v[j];
          }
        }
      }
    }
  return pl.nbands;
}

extern "C" {

int emu_darcy_loss_band(const float* Kp, const float* yp, float* gyp, float* partials, int B, int n, float a_const,
                        float a_cont, float b_dir, float b_neu, float beta1, float beta2, int flags, int waves, int npass,
                        int nbands, long long lds_floats_max, int* plan_out) {
  using namespace band;
  Plan pl;
  if (waves > 0) {                       // a forced plan (tests walk band heights the chooser would not pick)
    if (n < kMinN || n > kMaxN || !make_plan(n, waves, npass, nbands, pl)) return 0;
  } else if (!choose_plan(n, lds_floats_max, pl)) return 0;
  if (plan_out) { plan_out[0] = pl.waves; plan_out[1] = pl.npass; plan_out[2] = pl.nbands; plan_out[3] = pl.rpw; plan_out[4] = (int)pl.lds_floats; }
  LossParams p{a_const, a_cont, b_dir, b_neu, beta1, beta2, 0};
  // the kernel runs its J = 4 instantiation when n is a multiple of 4 and the pointers are aligned; flags bit 256 (emulation
  // only) selects J = 3 there, as an unaligned pointer does
  if ((n & 3) == 0 && !(flags & 256)) return band_emulation<4>(Kp, yp, gyp, partials, B, pl, p, flags);
  flags &= 255;
  switch (pl.jl) {
    case 3: return band_emulation<3>(Kp, yp, gyp, partials, B, pl, p, flags);
    case 2: return band_emulation<2>(Kp, yp, gyp, partials, B, pl, p, flags);
    case 1: return band_emulation<1>(Kp, yp, gyp, partials, B, pl, p, flags);
    default: return band_emulation<0>(Kp, yp, gyp, partials, B, pl, p, flags);
  }
}

int emu_band_plan(int n, long long lds_floats_max, int* out) {
  band::Plan pl;
  if (!band::choose_plan(n, lds_floats_max, pl)) return 0;
  out[0] = pl.waves; out[1] = pl.npass; out[2] = pl.nbands; out[3] = pl.rpw; out[4] = (int)pl.lds_floats; out[5] = pl.cap;
  return 1;
}

int emu_choose_tile(int n, long long budget, int* tr, int* tc) {
  return (n >= 8 ? choose_strip_tile(n, budget, *tr, *tc) : choose_tile(n, budget, *tr, *tc)) ? 1 : 0;
}
long long emu_tile_floats(int tr, int tc, int n) { return n >= 8 ? strip_tile_floats(tr, tc, n) : tile_floats(tr, tc, n); }

}  // extern "C"
