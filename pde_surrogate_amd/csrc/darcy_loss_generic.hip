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
#include <cstdio>
#include <cstdlib>
#include "pdes_common.h"
#include "darcy_generic.h"
#include "darcy_band.h"
#include "pdes_options.h"
#include "../../include/pdes_hip.h"

namespace pdes {

using namespace gen;
using gen::imin;

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
                                                                    float* __restrict__ partials, LossParams p_in, int n,
                                                                    int tr, int tc, int ntc, int flags) {
  const LossParams p = BWD ? loss_params_weighted(p_in) : p_in;
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
                                                                   LossParams p_in, int n, int tr, int tc, int ntc, int flags) {
  const LossParams p = BWD ? loss_params_weighted(p_in) : p_in;
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


// ---- the row-band kernel (8 <= n <= 256): darcy_band.h --------------------------------------------------------------------
// grid = (bands of an image, images), block = 64 * plan.waves, dynamic LDS = 3 planes of plan.rows_f rows + the row table.
// Neighbour strips are neighbour lanes: DPP wave shifts; lane 0 / 63 receive 0 there, and -- SEAM: rows run across wave
// boundaries -- take the neighbour wave's edge columns from `seam` instead (written in front of a barrier of their own).
constexpr long long BAND_LDSF = 16384 - 64;          // floats of dynamic LDS a band workgroup may use (64 KiB - reductions)

__device__ __forceinline__ float wave_shr1(float v) {      // lane i <- lane i - 1
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_shl1(float v) {      // lane i <- lane i + 1
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}
// seam[sq ..]: what the neighbour WAVE published for this lane and quantity (lane 0: its left neighbour's v[3], v[2]; lane 63:
// its right neighbour's v[0], v[jl]); sq < 0: no neighbour wave on that side (or not an edge lane)
template <int J, bool SEAM>
__device__ __forceinline__ band::Halo halo_of(const band::V4& x, int lane, const float* seam, int sq) {       // (what a caller does not use is never computed)
  band::Halo h;
  h.l = wave_shr1(x.v[3]);
  h.l2 = wave_shr1(x.v[2]);
  h.r = wave_shl1(x.v[0]);
  h.rjl = wave_shl1(x.v[band::WidthClass<J>::jl]);
  if (SEAM) {
    float2 e = make_float2(0.f, 0.f);
    if (sq >= 0) e = *reinterpret_cast<const float2*>(seam + sq);
    h.l = lane == 0 ? e.x : h.l; h.l2 = lane == 0 ? e.y : h.l2;
    h.r = lane == 63 ? e.x : h.r; h.rjl = lane == 63 ? e.y : h.rjl;
  }
  return h;
}
// lane 0 / lane 63 of a wave publish the edge columns of four quantities: sb = 16 floats of the (pass, wave):
// [0..7] lane 0's (v[0], v[jl]) x 4, [8..15] lane 63's (v[3], v[2]) x 4
template <int J>
__device__ __forceinline__ void seam_publish(float* sb, int lane, const band::V4& a, const band::V4& b, const band::V4& c,
                                             const band::V4& d) {
  constexpr int jl = band::WidthClass<J>::jl;
  if (lane == 0 || lane == 63) {
    const bool hi = lane == 63;
    float e[8];
    e[0] = hi ? a.v[3] : a.v[0]; e[1] = hi ? a.v[2] : a.v[jl];
    e[2] = hi ? b.v[3] : b.v[0]; e[3] = hi ? b.v[2] : b.v[jl];
    e[4] = hi ? c.v[3] : c.v[0]; e[5] = hi ? c.v[2] : c.v[jl];
    e[6] = hi ? d.v[3] : d.v[0]; e[7] = hi ? d.v[2] : d.v[jl];
    float* q = sb + (hi ? 8 : 0);
    gen::st4(q, e); gen::st4(q + 4, e + 4);
  }
}
// behind the barrier: lane 0 reads what lane 63 of the wave before published, lane 63 what lane 0 of the wave behind did
__device__ __forceinline__ int seam_source(int pass, int lane, int wave, int waves) {
  if ((lane == 0 && wave > 0) || (lane == 63 && wave + 1 < waves))
    return pass * 8 * 16 + (lane == 0 ? (wave - 1) * 16 + 8 : (wave + 1) * 16);
  return -64;
}

// four consecutive floats of global memory: 16-byte aligned (A) or dword aligned (global_load / store_dwordx4 take both)
typedef float vec4_a16 __attribute__((ext_vector_type(4)));
typedef float vec4_a4 __attribute__((ext_vector_type(4), aligned(4)));
template <bool A>
__device__ __forceinline__ band::V4 gld4(const float* q, int nt) {
  band::V4 o;
  if (A) {
    const vec4_a16 t = nt ? __builtin_nontemporal_load(reinterpret_cast<const vec4_a16*>(q)) : *reinterpret_cast<const vec4_a16*>(q);
    o.v[0] = t[0]; o.v[1] = t[1]; o.v[2] = t[2]; o.v[3] = t[3];
  } else {
    const vec4_a4 t = nt ? __builtin_nontemporal_load(reinterpret_cast<const vec4_a4*>(q)) : *reinterpret_cast<const vec4_a4*>(q);
    o.v[0] = t[0]; o.v[1] = t[1]; o.v[2] = t[2]; o.v[3] = t[3];
  }
  return o;
}
template <bool A>
__device__ __forceinline__ void gst4(float* q, const band::V4& x, int nt) {
  if (A) {
    const vec4_a16 t = {x.v[0], x.v[1], x.v[2], x.v[3]};
    if (nt) __builtin_nontemporal_store(t, reinterpret_cast<vec4_a16*>(q)); else *reinterpret_cast<vec4_a16*>(q) = t;
  } else {
    const vec4_a4 t = {x.v[0], x.v[1], x.v[2], x.v[3]};
    if (nt) __builtin_nontemporal_store(t, reinterpret_cast<vec4_a4*>(q)); else *reinterpret_cast<vec4_a4*>(q) = t;
  }
}

// J = 4: n is a multiple of 4 and every pointer 16-byte aligned -> 16-byte global accesses, no tail code.  J = 0 .. 3
// (= the last real column of the last strip): the same accesses at dword alignment; a row's last strip is read as the
// row's last four floats and shifted, and stored column by column.
template <bool BWD, int NPASS, int J, bool SEAM>
__global__ __launch_bounds__(512, 4) void darcy_loss_band_kernel(const float* __restrict__ Kp, const float* __restrict__ yp,
                                                              float* __restrict__ gyp, float* __restrict__ partials,
                                                              LossParams p_in, band::Plan pl, int flags) {
#include "darcy_band_kernel_body.inc"
}

// Round 6: the same body with the geometry of ONE field size and the default loss flags as compile-time
// constants.  ALIGNED = the aligned width class (n a multiple of 4 behind 16-byte aligned pointers: checked by the launcher).
// OCC: waves per SIMD the register budget is sized for (4 = 128 registers; the two-pass instantiations WITH seams need 3 = 168 to
// stay out of scratch, which costs this kernel more than the fourth wave brings).
template <bool BWD, int N, int WAVES, int NPASS, int NBANDS, bool ALIGNED, int OCC>
__global__ __launch_bounds__(64 * WAVES, OCC) void darcy_loss_band_fixed_kernel(const float* __restrict__ Kp,
                                                                            const float* __restrict__ yp,
                                                                            float* __restrict__ gyp,
                                                                            float* __restrict__ partials, LossParams p_in) {
  constexpr band::Plan pl = band::fixed_plan(N, WAVES, NPASS, NBANDS);
  constexpr int J = ALIGNED ? 4 : ((N - 1) & 3);
  constexpr bool SEAM = pl.seam != 0;
  constexpr int flags = 0;
  static_assert(!ALIGNED || (N & 3) == 0, "the aligned width class needs a multiple of 4");
#include "darcy_band_kernel_body.inc"
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

// the row-band kernel serves 8 <= n <= 256 unless the caller asks for the tile kernel (PDES_LOSS_TILED: cross-checks)
static bool band_plan(int n, int flags, band::Plan& pl) {
  if (flags & PDES_LOSS_TILED) return false;
#ifdef PDES_BAND_PLAN_ENV
  if (const char* e = getenv("PDES_BAND_PLAN")) {          // plan-sweep builds (tools/sweep_band_plan.sh): "waves,npass,nbands"
    int wv = 0, np = 0, nb = 0;
    if (sscanf(e, "%d,%d,%d", &wv, &np, &nb) == 3 && n >= band::kMinN && n <= band::kMaxN && band::make_plan(n, wv, np, nb, pl) &&
        pl.lds_floats <= BAND_LDSF)
      return true;
  }
#endif
  return band::choose_plan(n, BAND_LDSF, pl);
}

int loss_generic_tiles(int n, int flags) {       // tiles (bands) per image (<= 0: size not supported)
  int tr = 0, tc = 0;
  if (n < 2) return 0;
  band::Plan pl;
  if (band_plan(n, flags, pl)) return pl.nbands;
  if (n >= STRIP_MIN_N) {
    if (!choose_strip_tile(n, GEN_LDSF, tr, tc)) return 0;
  } else if (!choose_tile(n, GEN_LDSF, tr, tc)) return 0;
  return cdiv(n, tr) * cdiv(n, tc);
}

int launch_loss_generic(const float* K, const float* y, float* gy, float* partials, int B, int n, LossParams p,
                        int flags, hipStream_t st) {
  int tr = 0, tc = 0;
  if (n < 2) return PDES_ENOSUP;
  band::Plan pl;
  if (band_plan(n, flags, pl)) {
    const bool a16 = (n & 3) == 0 && aligned16(K) && aligned16(y) && (!gy || aligned16(gy));
    const dim3 grid(pl.nbands, B), block(64 * pl.waves);
    const size_t shmem = (size_t)pl.lds_floats * sizeof(float);
    // the common sizes with the reference's default loss (flags 0): geometry folded into the instantiation (round 6)
    if ((flags & (gen::kNonlinear | gen::kNoTB | gen::kUncorrected)) == 0 && opt().band_fixed) {       // (PDES_BAND_FIXED=0: the run-time plan, for cross-checks)
#define PDES_BAND_FIXED(N_, W_, NP_, NB_, A_, OCC_)                                                                      \
      if (n == N_ && a16 == A_ && band::plan_equal(pl, band::fixed_plan(N_, W_, NP_, NB_))) {                                \
        if (gy) hipLaunchKernelGGL((darcy_loss_band_fixed_kernel<true, N_, W_, NP_, NB_, A_, OCC_>), grid, block, shmem, st, K, y, gy, partials, p); \
        else hipLaunchKernelGGL((darcy_loss_band_fixed_kernel<false, N_, W_, NP_, NB_, A_, 4>), grid, block, shmem, st, K, y, gy, partials, p);   \
        return PDES_OK;                                                                                                     \
      }
      PDES_BAND_FIXED(48, 2, 2, 3, true, 3)
      PDES_BAND_FIXED(65, 4, 1, 5, false, 4)
      PDES_BAND_FIXED(66, 4, 2, 3, false, 4)
      PDES_BAND_FIXED(96, 4, 2, 6, true, 3)
      PDES_BAND_FIXED(128, 4, 2, 10, true, 4)
      PDES_BAND_FIXED(256, 8, 2, 19, true, 4)
      PDES_BAND_FIXED(100, 4, 2, 6, true, 3)
      PDES_BAND_FIXED(129, 8, 2, 5, false, 4)
      PDES_BAND_FIXED(130, 8, 2, 5, false, 4)
      PDES_BAND_FIXED(131, 8, 2, 5, false, 4)
      PDES_BAND_FIXED(200, 8, 2, 12, true, 4)
      PDES_BAND_FIXED(100, 4, 2, 6, true, 3)
      PDES_BAND_FIXED(129, 8, 2, 5, false, 4)
      PDES_BAND_FIXED(130, 8, 2, 5, false, 4)
      PDES_BAND_FIXED(131, 8, 2, 5, false, 4)
      PDES_BAND_FIXED(200, 8, 2, 12, true, 4)
#undef PDES_BAND_FIXED
    }
#define PDES_BAND_LAUNCH(BWD_, NP_, J_, S_) \
    hipLaunchKernelGGL((darcy_loss_band_kernel<BWD_, NP_, J_, S_>), grid, block, shmem, st, K, y, gy, partials, p, pl, flags)
#define PDES_BAND_LAUNCH_S(BWD_, NP_, J_) \
    do { if (pl.seam) PDES_BAND_LAUNCH(BWD_, NP_, J_, true); else PDES_BAND_LAUNCH(BWD_, NP_, J_, false); } while (0)
#define PDES_BAND_LAUNCH_J(J_)                                                                             \
    do {                                                                                                   \
      if (pl.npass == 1) { if (gy) PDES_BAND_LAUNCH_S(true, 1, J_); else PDES_BAND_LAUNCH_S(false, 1, J_); }     \
      else { if (gy) PDES_BAND_LAUNCH_S(true, 2, J_); else PDES_BAND_LAUNCH_S(false, 2, J_); }                   \
    } while (0)
    switch (a16 ? 4 : pl.jl) {
      case 4: PDES_BAND_LAUNCH_J(4); break;
      case 3: PDES_BAND_LAUNCH_J(3); break;
      case 2: PDES_BAND_LAUNCH_J(2); break;
      case 1: PDES_BAND_LAUNCH_J(1); break;
      default: PDES_BAND_LAUNCH_J(0); break;
    }
#undef PDES_BAND_LAUNCH_S
#undef PDES_BAND_LAUNCH_J
#undef PDES_BAND_LAUNCH
    return PDES_OK;
  }
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
