// Sobel gradients and the fused Darcy mixed-residual loss for ANY square field size (and for
// SobelFilter(correct=False)): the kernels behind pdes_darcy_loss / pdes_sobel_grad* / pdes_sobel5_* whenever the
// 16 / 32 / 64 specialisations of darcy_loss.hip do not apply.
//
// Reference: utils/image_gradient.py:24-92 (SobelFilter(imsize) for any imsize, correct=True/False, filter_size 3/5),
// models/darcy.py:162-233 (its docstrings use 65 x 65 fields), train_codec_mixed_residual.py:52 (--imsize).
//
// Fused loss: ONE workgroup (1024 threads) per image walks the image in tiles; per tile the three fields (own region
// +-3), the three adjoint sources (own region +-2) and the direct part of dL/dy (own region) live in 128 KiB of LDS, so
// every field value is read from HBM once per tile that touches it (the +-3 halo of neighbouring tiles of the same
// image comes out of L2), the gradient is written once, and the per-image loss sums stay in the workgroup: partials
// keep the (B, 4) contract of the specialised kernel and its fixed-order reductions.  Accesses are scalar (rows of an
// odd width are not 16-byte aligned) but coalesced along the row.  HBM-bound like the specialised kernel; the
// per-pixel arithmetic is shared with the CPU emulation the tests check against the oracle (darcy_generic.h).
//
// Stand-alone Sobel / adjoint kernels (3x3 and 5x5): one thread per pixel straight from global memory (the 9..50
// taps of a pixel hit L1/L2); these are the autograd-facing SobelFilter.grad_h / grad_v of fields the fused loss does
// not cover, not a hot path.
#include "pdes_common.h"
#include "darcy_generic.h"

namespace pdes {

using namespace gen;

constexpr int GEN_NT = 1024;
constexpr int GEN_LDSF = 32768;          // floats of tile planes (128 KiB)

template <bool BWD>
__global__ __launch_bounds__(GEN_NT) void darcy_loss_generic_kernel(const float* __restrict__ Kp,
                                                                    const float* __restrict__ yp,
                                                                    float* __restrict__ gyp,
                                                                    float* __restrict__ partials, LossParams p, int n,
                                                                    int tr, int tc, int flags) {
  __shared__ float lds[GEN_LDSF];
  __shared__ float red[(GEN_NT / 64) * 4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const size_t nn = (size_t)n * n;
  const float* Kb = Kp + (size_t)b * nn;
  const float* yb = yp + (size_t)b * 3 * nn;
  float* gb = BWD ? gyp + (size_t)b * 3 * nn : nullptr;
  const bool correct = !(flags & kUncorrected);
  const int ntr = cdiv(n, tr), ntc = cdiv(n, tc);
  float s_const = 0.f, s_cont = 0.f, s_dir = 0.f, s_neu = 0.f;

  for (int ti = 0; ti < ntr; ++ti)
    for (int tj = 0; tj < ntc; ++tj) {
      const TileGeo g = tile_geo(n, tr, tc, ti, tj);
      const int ih = g.ir1 - g.ir0, iw = g.ic1 - g.ic0, sh = g.sr1 - g.sr0, sw = g.sc1 - g.sc0;
      const int oh = g.r1 - g.r0, ow = g.c1 - g.c0;
      const int ni = ih * iw, ns = sh * sw, no = oh * ow;
      float* F = lds;                  // fields u, sigma1, sigma2
      float* S = F + 3 * ni;           // adjoint sources a_const K r1, a_const K r2, a_cont c
      float* D = S + 3 * ns;           // direct part of dL/du, dL/dsigma1, dL/dsigma2
      for (int i = tid; i < ni; i += GEN_NT) {
        const int rr = i / iw, cc = i - rr * iw;
        const size_t o = (size_t)(g.ir0 + rr) * n + (g.ic0 + cc);
        F[i] = yb[o];
        F[ni + i] = yb[nn + o];
        F[2 * ni + i] = yb[2 * nn + o];
      }
      __syncthreads();
      const Plane U{F, g.ir0, g.ic0, iw, n}, X1{F + ni, g.ir0, g.ic0, iw, n}, X2{F + 2 * ni, g.ir0, g.ic0, iw, n};
      for (int i = tid; i < ns; i += GEN_NT) {
        const int rr = i / sw, cc = i - rr * sw;
        const int r = g.sr0 + rr, c = g.sc0 + cc;
        const PixelTerms t = loss_pixel(U, X1, X2, Kb[(size_t)r * n + c], r, c, p, flags);
        if (r >= g.r0 && r < g.r1 && c >= g.c0 && c < g.c1) {
          s_const += t.s_const; s_cont += t.s_cont; s_dir += t.s_dir; s_neu += t.s_neu;
          if (BWD) {
            const int k = (r - g.r0) * ow + (c - g.c0);
            D[k] = t.d_u; D[no + k] = t.d_s1; D[2 * no + k] = t.d_s2;
          }
        }
        if (BWD) { S[i] = t.src_p1; S[ns + i] = t.src_p2; S[2 * ns + i] = t.src_cc; }
      }
      __syncthreads();
      if (BWD) {
        const Plane G1{S, g.sr0, g.sc0, sw, n}, G2{S + ns, g.sr0, g.sc0, sw, n}, GC{S + 2 * ns, g.sr0, g.sc0, sw, n};
        for (int i = tid; i < no; i += GEN_NT) {
          const int rr = i / ow, cc = i - rr * ow;
          const int r = g.r0 + rr, c = g.c0 + cc;
          const float du = D[i] + sobel_adj<true>(G1, r, c, correct) + sobel_adj<false>(G2, r, c, correct);
          const float d1 = D[no + i] + sobel_adj<true>(GC, r, c, correct);
          const float d2 = D[2 * no + i] + sobel_adj<false>(GC, r, c, correct);
          const size_t o = (size_t)r * n + c;
          gb[o] = du; gb[nn + o] = d1; gb[2 * nn + o] = d2;
        }
        __syncthreads();               // the next tile's loads overwrite the planes
      }
    }

  const float t0 = wave_sum(s_const), t1 = wave_sum(s_cont), t2 = wave_sum(s_dir), t3 = wave_sum(s_neu);
  if ((tid & 63) == 0) {
    const int w = tid >> 6;
    red[w * 4 + 0] = t0; red[w * 4 + 1] = t1; red[w * 4 + 2] = t2; red[w * 4 + 3] = t3;
  }
  __syncthreads();
  if (tid < 4) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < GEN_NT / 64; ++w) t += red[w * 4 + tid];     // fixed order: deterministic
    partials[(size_t)b * 4 + tid] = t;
  }
}

// ---- stand-alone gradients and adjoints, one thread per pixel -------------------------------------------------------
template <bool FIVE>
__global__ __launch_bounds__(256) void sobel_generic_kernel(const float* __restrict__ img, float* __restrict__ gh,
                                                            float* __restrict__ gv, int n, int correct) {
  const int px = blockIdx.x * 256 + threadIdx.x;
  if (px >= n * n) return;
  const size_t base = (size_t)blockIdx.y * n * n;
  const int r = px / n, c = px - r * n;
  const Plane P{img + base, 0, 0, n, n};
  if (gh) gh[base + px] = FIVE ? sobel5_grad<true>(P, r, c, correct != 0) : sobel_grad<true>(P, r, c, correct != 0);
  if (gv) gv[base + px] = FIVE ? sobel5_grad<false>(P, r, c, correct != 0) : sobel_grad<false>(P, r, c, correct != 0);
}

template <bool FIVE>
__global__ __launch_bounds__(256) void sobel_adjoint_generic_kernel(const float* __restrict__ ghb,
                                                                    const float* __restrict__ gvb,
                                                                    float* __restrict__ out, int n, int correct) {
  const int px = blockIdx.x * 256 + threadIdx.x;
  if (px >= n * n) return;
  const size_t base = (size_t)blockIdx.y * n * n;
  const int r = px / n, c = px - r * n;
  const Plane Gh{ghb ? ghb + base : nullptr, 0, 0, n, n}, Gv{gvb ? gvb + base : nullptr, 0, 0, n, n};
  float a;
  if (FIVE) {
    a = sobel5_adj(Gh, Gv, r, c, correct != 0);
  } else {
    a = 0.f;
    if (ghb) a += sobel_adj<true>(Gh, r, c, correct != 0);
    if (gvb) a += sobel_adj<false>(Gv, r, c, correct != 0);
  }
  out[base + px] = a;
}

// ---- launchers (darcy_loss.hip's entry points validate the arguments) -----------------------------------------------
int launch_loss_generic(const float* K, const float* y, float* gy, float* partials, int B, int n, LossParams p,
                        int flags, hipStream_t st) {
  int tr = 0, tc = 0;
  if (n < 2 || !choose_tile(n, GEN_LDSF, tr, tc)) return PDES_ENOSUP;
  if (gy) hipLaunchKernelGGL(darcy_loss_generic_kernel<true>, dim3(B), dim3(GEN_NT), 0, st, K, y, gy, partials, p, n, tr, tc, flags);
  else hipLaunchKernelGGL(darcy_loss_generic_kernel<false>, dim3(B), dim3(GEN_NT), 0, st, K, y, gy, partials, p, n, tr, tc, flags);
  return PDES_OK;
}

int launch_sobel_generic(const float* img, float* gh, float* gv, int nimg, int n, int correct, int five, hipStream_t st) {
  if (n < 2 || (long long)n * n > (1ll << 30)) return PDES_ENOSUP;
  const dim3 grid(cdiv(n * n, 256), nimg), block(256);
  if (five) hipLaunchKernelGGL(sobel_generic_kernel<true>, grid, block, 0, st, img, gh, gv, n, correct);
  else hipLaunchKernelGGL(sobel_generic_kernel<false>, grid, block, 0, st, img, gh, gv, n, correct);
  return PDES_OK;
}

int launch_sobel_adjoint_generic(const float* ghb, const float* gvb, float* out, int nimg, int n, int correct, int five,
                                 hipStream_t st) {
  if (n < 2 || (long long)n * n > (1ll << 30)) return PDES_ENOSUP;
  const dim3 grid(cdiv(n * n, 256), nimg), block(256);
  if (five) hipLaunchKernelGGL(sobel_adjoint_generic_kernel<true>, grid, block, 0, st, ghb, gvb, out, n, correct);
  else hipLaunchKernelGGL(sobel_adjoint_generic_kernel<false>, grid, block, 0, st, ghb, gvb, out, n, correct);
  return PDES_OK;
}

}  // namespace pdes
