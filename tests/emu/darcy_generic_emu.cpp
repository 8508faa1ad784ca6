// Test infrastructure (CPU): the per-pixel arithmetic and the tile geometry of csrc/darcy_loss_generic.hip compiled as
// plain C++ (csrc/darcy_generic.h is host/device code) and driven by serial loops that follow the kernel's phases --
// fields -> [barrier] -> sources + direct terms + sums -> [barrier] -> adjoint -- tile by tile, with separate buffers
// standing in for the LDS planes.  tests/test_generic_loss_cpu.py builds this with g++ and compares it with the oracle
// for field sizes / tile sizes that exercise every halo case, so the kernels' arithmetic is checked before any GPU run.
// It is NOT a product path: nothing under pde_surrogate_amd/ loads it.
#include <vector>
#include <cstddef>
#include "../../pde_surrogate_amd/csrc/darcy_generic.h"

using namespace pdes;
using namespace pdes::gen;

extern "C" {

// tr/tc <= 0: the kernel's own choice for `budget` floats of LDS.  Returns the number of tiles per image (0: no fit).
int emu_darcy_loss(const float* Kp, const float* yp, float* gyp, float* partials, int B, int n, float a_const,
                   float a_cont, float b_dir, float b_neu, float beta1, float beta2, int flags, int tr, int tc,
                   long long budget) {
  if (tr <= 0 || tc <= 0) {
    if (!choose_tile(n, budget, tr, tc)) return 0;
  }
  if (tile_floats(tr, tc, n) > budget) return 0;
  LossParams p{a_const, a_cont, b_dir, b_neu, beta1, beta2, 0};
  const bool correct = !(flags & kUncorrected);
  const size_t nn = (size_t)n * n;
  const int ntr = (n + tr - 1) / tr, ntc = (n + tc - 1) / tc;
  for (int b = 0; b < B; ++b) {
    const float* Kb = Kp + b * nn;
    const float* yb = yp + b * 3 * nn;
    float* gb = gyp ? gyp + b * 3 * nn : nullptr;
    float sums[4] = {0, 0, 0, 0};
    for (int ti = 0; ti < ntr; ++ti)
      for (int tj = 0; tj < ntc; ++tj) {
        const TileGeo g = tile_geo(n, tr, tc, ti, tj);
        const int ih = g.ir1 - g.ir0, iw = g.ic1 - g.ic0, sh = g.sr1 - g.sr0, sw = g.sc1 - g.sc0;
        const int oh = g.r1 - g.r0, ow = g.c1 - g.c0;
        const int ni = ih * iw, ns = sh * sw, no = oh * ow;
        std::vector<float> lds(3 * (size_t)(ni + ns + no), -12345.f);     // poison: a read outside what was written shows
        float* F = lds.data();
        float* S = F + 3 * ni;
        float* D = S + 3 * ns;
        for (int i = 0; i < ni; ++i) {
          const int rr = i / iw, cc = i - rr * iw;
          const size_t o = (size_t)(g.ir0 + rr) * n + (g.ic0 + cc);
          F[i] = yb[o]; F[ni + i] = yb[nn + o]; F[2 * ni + i] = yb[2 * nn + o];
        }
        const Plane U{F, g.ir0, g.ic0, iw, n}, X1{F + ni, g.ir0, g.ic0, iw, n}, X2{F + 2 * ni, g.ir0, g.ic0, iw, n};
        for (int i = 0; i < ns; ++i) {
          const int rr = i / sw, cc = i - rr * sw;
          const int r = g.sr0 + rr, c = g.sc0 + cc;
          const PixelTerms t = loss_pixel(U, X1, X2, Kb[(size_t)r * n + c], r, c, p, flags);
          if (r >= g.r0 && r < g.r1 && c >= g.c0 && c < g.c1) {
            sums[0] += t.s_const; sums[1] += t.s_cont; sums[2] += t.s_dir; sums[3] += t.s_neu;
            const int k = (r - g.r0) * ow + (c - g.c0);
            D[k] = t.d_u; D[no + k] = t.d_s1; D[2 * no + k] = t.d_s2;
          }
          S[i] = t.src_p1; S[ns + i] = t.src_p2; S[2 * ns + i] = t.src_cc;
        }
        if (gb) {
          const Plane G1{S, g.sr0, g.sc0, sw, n}, G2{S + ns, g.sr0, g.sc0, sw, n}, GC{S + 2 * ns, g.sr0, g.sc0, sw, n};
          for (int i = 0; i < no; ++i) {
            const int rr = i / ow, cc = i - rr * ow;
            const int r = g.r0 + rr, c = g.c0 + cc;
            const size_t o = (size_t)r * n + c;
            gb[o] = D[i] + sobel_adj<true>(G1, r, c, correct) + sobel_adj<false>(G2, r, c, correct);
            gb[nn + o] = D[no + i] + sobel_adj<true>(GC, r, c, correct);
            gb[2 * nn + o] = D[2 * no + i] + sobel_adj<false>(GC, r, c, correct);
          }
        }
      }
    for (int k = 0; k < 4; ++k) partials[b * 4 + k] = sums[k];
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

int emu_choose_tile(int n, long long budget, int* tr, int* tc) { return choose_tile(n, budget, *tr, *tc) ? 1 : 0; }
long long emu_tile_floats(int tr, int tc, int n) { return tile_floats(tr, tc, n); }

}  // extern "C"
