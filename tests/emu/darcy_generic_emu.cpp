// Test infrastructure (CPU): the per-pixel arithmetic and the tile geometry of csrc/darcy_loss_generic.hip compiled as
// plain C++ (csrc/darcy_generic.h is host/device code): process_tile -- the whole procedure of one workgroup, phases,
// strip fast path and per-pixel border path included -- runs here with ONE serial "thread" per tile and a poisoned buffer
// standing in for the LDS.  tests/test_generic_loss_cpu.py builds this with g++ and compares it with the oracle
// for field sizes / tile sizes that exercise every halo case, so the kernels' arithmetic is checked before any GPU run.
// It is NOT a product path: nothing under pde_surrogate_amd/ loads it.
#include <vector>
#include <cstddef>
#include "../../pde_surrogate_amd/csrc/darcy_generic.h"

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

int emu_choose_tile(int n, long long budget, int* tr, int* tc) {
  return (n >= 8 ? choose_strip_tile(n, budget, *tr, *tc) : choose_tile(n, budget, *tr, *tc)) ? 1 : 0;
}
long long emu_tile_floats(int tr, int tc, int n) { return n >= 8 ? strip_tile_floats(tr, tc, n) : tile_floats(tr, tc, n); }

}  // extern "C"
