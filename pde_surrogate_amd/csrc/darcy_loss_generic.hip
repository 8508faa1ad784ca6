// Sobel gradients and the fused Darcy mixed-residual loss for ANY square field size (and for
// SobelFilter(correct=False)): the kernels behind pdes_darcy_loss / pdes_sobel_grad* / pdes_sobel5_* whenever the
// 16 / 32 / 64 specialisations of darcy_loss.hip do not apply.
//
// Reference: utils/image_gradient.py:24-92 (SobelFilter(imsize) for any imsize, correct=True/False, filter_size 3/5),
// models/darcy.py:162-233 (its docstrings use 65 x 65 fields), train_codec_mixed_residual.py:52 (--imsize).
//
// Fused loss: one workgroup (256 threads) per TILE of an image (rows -- for wide images also columns -- split so that a tile's planes fit 40 KiB of LDS:
// the three fields on the own region +-3, the three adjoint sources +-2, the direct part of dL/dy); every field value is
// read from HBM once per tile that touches it (the halo of neighbouring tiles comes out of L2 / Infinity Cache), the
// gradient is written once.  Away from the image border a thread handles a 1 x 4 strip with 16-byte LDS accesses and
// branch-free stencils (the adjoints there are minus the forward operators); the two outermost rows / columns take the
// per-pixel path that follows the reference's order of operations.  Per-(image, tile) partial sums, reduced in a fixed
// order (pdes_darcy_loss_partial_rows tells the caller how many rows).  The whole tile procedure is ONE function shared
// with the CPU emulation the tests check against the oracle (darcy_generic.h: process_tile).
//
// Stand-alone Sobel / adjoint kernels (3x3 and 5x5): one thread per pixel straight from global memory (the 9..50
// taps of a pixel hit L1/L2); these are the autograd-facing SobelFilter.grad_h / grad_v of fields the fused loss does
// not cover, not a hot path.
#include "pdes_common.h"
#include "darcy_generic.h"

namespace pdes {

using namespace gen;

// 256 threads and 40 KiB of tile planes per workgroup: four workgroups per CU, whose load / stencil / store phases
// overlap (one 1024-thread workgroup with 128 KiB per CU ran its phases back to back: 0.10 of 8 TB/s)
constexpr int GEN_NT = 256;
constexpr int GEN_LDSF = 10240;

struct BlockExec {
  int tid, nthreads;
  __device__ __forceinline__ void barrier() const { __syncthreads(); }
};

// grid = (tiles of an image, images); partials: one row {const, cont, dir, neu} per (image, tile), reduced in a fixed
// order by darcy_loss_finalize / the end-of-step launch (deterministic)
template <bool BWD>
__global__ __launch_bounds__(GEN_NT) void darcy_loss_generic_kernel(const float* __restrict__ Kp,
                                                                    const float* __restrict__ yp,
                                                                    float* __restrict__ gyp,
                                                                    float* __restrict__ partials, LossParams p, int n,
                                                                    int tr, int tc, int ntc, int flags) {
  __shared__ __attribute__((aligned(16))) float lds[GEN_LDSF];
  __shared__ float red[(GEN_NT / 64) * 4];
  const int tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const size_t nn = (size_t)n * n;
  const TileGeo g = tile_geo(n, tr, tc, tile / ntc, tile % ntc);
  float sums[4] = {0.f, 0.f, 0.f, 0.f};
  BlockExec ex{tid, GEN_NT};
  process_tile_pixelwise<BWD>(Kp + (size_t)b * nn, yp + (size_t)b * 3 * nn, BWD ? gyp + (size_t)b * 3 * nn : nullptr, n, g, p,
                              flags, lds, ex, sums);
  const float t0 = wave_sum(sums[0]), t1 = wave_sum(sums[1]), t2 = wave_sum(sums[2]), t3 = wave_sum(sums[3]);
  if ((tid & 63) == 0) {
    const int w = tid >> 6;
    red[w * 4 + 0] = t0; red[w * 4 + 1] = t1; red[w * 4 + 2] = t2; red[w * 4 + 3] = t3;
  }
  __syncthreads();
  if (tid < 4) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < GEN_NT / 64; ++w) t += red[w * 4 + tid];     // fixed order: deterministic
    partials[((size_t)b * gridDim.x + tile) * 4 + tid] = t;
  }
}

// the strip form (n >= 8): every pixel through the same branch-free code (darcy_generic.h: process_tile_strips)
template <bool BWD>
__global__ __launch_bounds__(GEN_NT) void darcy_loss_strips_kernel(const float* __restrict__ Kp, const float* __restrict__ yp,
                                                                   float* __restrict__ gyp, float* __restrict__ partials,
                                                                   LossParams p, int n, int tr, int tc, int ntc, int flags) {
  __shared__ __attribute__((aligned(16))) float lds[GEN_LDSF];
  __shared__ float red[(GEN_NT / 64) * 4];
  const int tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const size_t nn = (size_t)n * n;
  const StripGeo g = strip_geo(n, tr, tc, tile / ntc, tile % ntc);
  float sums[4] = {0.f, 0.f, 0.f, 0.f};
  BlockExec ex{tid, GEN_NT};
  process_tile_strips<BWD>(Kp + (size_t)b * nn, yp + (size_t)b * 3 * nn, BWD ? gyp + (size_t)b * 3 * nn : nullptr, n, g, p, flags,
                           lds, ex, sums);
  const float t0 = wave_sum(sums[0]), t1 = wave_sum(sums[1]), t2 = wave_sum(sums[2]), t3 = wave_sum(sums[3]);
  if ((tid & 63) == 0) {
    const int w = tid >> 6;
    red[w * 4 + 0] = t0; red[w * 4 + 1] = t1; red[w * 4 + 2] = t2; red[w * 4 + 3] = t3;
  }
  __syncthreads();
  if (tid < 4) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < GEN_NT / 64; ++w) t += red[w * 4 + tid];
    partials[((size_t)b * gridDim.x + tile) * 4 + tid] = t;
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
constexpr int STRIP_MIN_N = 8;        // below: the per-pixel kernel (the adjoint tables of the strip form need n >= 6)

int loss_generic_tiles(int n) {       // tiles per image (<= 0: size not supported)
  int tr = 0, tc = 0;
  if (n < 2) return 0;
  if (n >= STRIP_MIN_N) {
    if (!choose_strip_tile(n, GEN_LDSF, tr, tc)) return 0;
  } else if (!choose_tile(n, GEN_LDSF, tr, tc)) return 0;
  return cdiv(n, tr) * cdiv(n, tc);
}

int launch_loss_generic(const float* K, const float* y, float* gy, float* partials, int B, int n, LossParams p,
                        int flags, hipStream_t st) {
  int tr = 0, tc = 0;
  if (n < 2) return PDES_ENOSUP;
  const bool strips = n >= STRIP_MIN_N;
  if (strips ? !choose_strip_tile(n, GEN_LDSF, tr, tc) : !choose_tile(n, GEN_LDSF, tr, tc)) return PDES_ENOSUP;
  const int ntc = cdiv(n, tc);
  const dim3 grid(cdiv(n, tr) * ntc, B), block(GEN_NT);
  if (strips) {
    if (gy) hipLaunchKernelGGL(darcy_loss_strips_kernel<true>, grid, block, 0, st, K, y, gy, partials, p, n, tr, tc, ntc, flags);
    else hipLaunchKernelGGL(darcy_loss_strips_kernel<false>, grid, block, 0, st, K, y, gy, partials, p, n, tr, tc, ntc, flags);
  } else {
    if (gy) hipLaunchKernelGGL(darcy_loss_generic_kernel<true>, grid, block, 0, st, K, y, gy, partials, p, n, tr, tc, ntc, flags);
    else hipLaunchKernelGGL(darcy_loss_generic_kernel<false>, grid, block, 0, st, K, y, gy, partials, p, n, tr, tc, ntc, flags);
  }
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
