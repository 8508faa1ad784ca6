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
#include "darcy_band.h"
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
// grid = (bands of an image, images), block = 64 * plan.waves, dynamic LDS = 3 planes of plan.rows_f rows.  Neighbour
// strips are neighbour lanes: DPP wave shifts (lane 0 / 63 receive 0 and never use it: a row starts at a first strip).
constexpr long long BAND_LDSF = 16384 - 64;          // floats of dynamic LDS a band workgroup may use (64 KiB - reductions)

__device__ __forceinline__ float wave_shr1(float v) {      // lane i <- lane i - 1
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_shl1(float v) {      // lane i <- lane i + 1
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}
template <int J>
__device__ __forceinline__ band::Halo halo_of(const band::V4& x) {       // (what a caller does not use is never computed)
  band::Halo h;
  h.l = wave_shr1(x.v[3]);
  h.l2 = wave_shr1(x.v[2]);
  h.r = wave_shl1(x.v[0]);
  h.rjl = wave_shl1(x.v[band::WidthClass<J>::jl]);
  return h;
}

// J = 4: n is a multiple of 4 and every pointer 16-byte aligned -> 16-byte global accesses, no tail code.  J = 0 .. 3
// (= the last real column of the last strip): scalar global accesses over CONTIGUOUS ranges (a band's rows are one range of
// the plane): the fields and the conductivities enter through the LDS planes, the gradient leaves through them.
template <bool BWD, int NPASS, int J>
__global__ __launch_bounds__(512, 4) void darcy_loss_band_kernel(const float* __restrict__ Kp, const float* __restrict__ yp,
                                                              float* __restrict__ gyp, float* __restrict__ partials,
                                                              LossParams p_in, band::Plan pl, int flags) {
  using namespace band;
  const LossParams p = BWD ? loss_params_weighted(p_in) : p_in;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float red[8 * 4];
  const int bi = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
  constexpr bool A = J == 4;
  const int n = pl.n, w = pl.w;
  const size_t nn = (size_t)n * n;
  const float fn = (float)n;
  const bool correct = !(flags & kUncorrected);
  const BandGeo g = band_geo(pl, bi);
  const LaneConst c = lane_const(pl, lane, correct);
  const float* Kb = Kp + (size_t)b * nn;
  const float* yb = yp + (size_t)b * 3 * nn;
  float* gb = BWD ? gyp + (size_t)b * 3 * nn : nullptr;
  const int plane = pl.rows_f * w, rows_f = g.fr1 - g.fr0;
  float* tab = lds + pl.planes * plane;                      // the row table (darcy_band.h), read after the first barrier
  if (tid < rows_f) rowtab_build(tab, g.fr0 + tid, n, correct, g.fr0, g.fr1, w);

  int row[NPASS];
#pragma unroll
  for (int k = 0; k < NPASS; ++k) row[k] = slot_row(pl, g, k, wave, c);
  V4 kk[NPASS];
  if (A) {
    // the conductivities of this lane's strips: requested first, used after the staging
#pragma unroll
    for (int k = 0; k < NPASS; ++k) {
      const bool ok = c.active && row[k] < g.sr1;
      const float4* kp = reinterpret_cast<const float4*>(Kb + (size_t)row[k] * n + 4 * c.cs);
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) t = p.nt ? nt_load4(kp) : *kp;
      kk[k].v[0] = t.x; kk[k].v[1] = t.y; kk[k].v[2] = t.z; kk[k].v[3] = t.w;
    }
    // the three fields on rows fr0 .. fr1 (w == n: a plane is the row range itself).  The loads of ALL planes are issued
    // before the first LDS store: bytes in flight are what the staging phase runs on
    const int per = rows_f * pl.spr;
#pragma unroll 1
    for (int base = tid; base < per; base += 2 * nthreads) {
      float4 v[3][2];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const float4* src = reinterpret_cast<const float4*>(yb + q * nn + (size_t)g.fr0 * n);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int i = base + u * nthreads;
          if (i < per) v[q][u] = p.nt ? nt_load4(src + i) : src[i];
        }
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        float4* dst = reinterpret_cast<float4*>(lds + q * plane);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int i = base + u * nthreads;
          if (i < per) dst[i] = v[q][u];
        }
      }
    }
  } else {
    const int per = rows_f * n;
    const float inv = 1.0f / (float)n;
    const bool kplane = pl.planes == 4;
#pragma unroll 1
    for (int base = tid; base < per; base += 4 * nthreads) {
      float v[4][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float* src = (q < 3 ? yb + q * nn : Kb) + (size_t)g.fr0 * n;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = base + u * nthreads;
          if (i < per && (q < 3 || kplane)) v[q][u] = p.nt ? __builtin_nontemporal_load(src + i) : src[i];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = base + u * nthreads;
        if (i < per) {
          const int rr = (int)(((float)i + 0.5f) * inv);          // i / n (exact: i < 2^16, n <= 256)
          const int o = rr * w + (i - rr * n);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (q < 3 || kplane) lds[q * plane + o] = v[q][u];
        }
      }
    }
  }
  __syncthreads();
  if (!A) {
#pragma unroll
    for (int k = 0; k < NPASS; ++k) {
      const int rc = row[k] < g.sr1 ? row[k] : g.sr1 - 1;
      if (pl.planes == 4) ld4(lds + 3 * plane + (rc - g.fr0) * w + 4 * c.cs, kk[k].v);
      else {                              // (a multiple of 4 behind an unaligned pointer, no room for a fourth plane)
        const float* kp = Kb + (size_t)rc * n + 4 * c.cs;
#pragma unroll
        for (int j = 0; j < 4; ++j) kk[k].v[j] = kp[j];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) kk[k].v[j] = c.valid[j] ? kk[k].v[j] : 0.f;        // (tail columns of the plane were never written)
    }
  }

  const BPlane U{lds, g.fr0, g.fr1, w}, X1{lds + plane, g.fr0, g.fr1, w}, X2{lds + 2 * plane, g.fr0, g.fr1, w};
  float sums[4] = {0.f, 0.f, 0.f, 0.f};
  StripOut so[NPASS];
#ifdef PDES_TUNE
  if (flags & 4096) {                  // timing experiment: no arithmetic (the staged values pass through)
#pragma unroll
    for (int k = 0; k < NPASS; ++k) {
      const int rc = row[k] < g.sr1 ? row[k] : g.sr1 - 1;
      const int o = (rc - g.fr0) * w + 4 * c.cs;
      so[k].p1 = ldoff(U, o); so[k].p2 = ldoff(X1, o); so[k].cc = ldoff(X2, o); so[k].d1 = kk[k]; so[k].d2 = kk[k]; so[k].du = 0.f;
    }
  } else
#endif
#pragma unroll
  for (int k = 0; k < NPASS; ++k) {
    const int r = row[k], rc = r < g.sr1 ? r : g.sr1 - 1;
    const bool ok = c.active && r < g.sr1, own = ok && r >= g.r0 && r < g.r1;
    const FwdVert f = fwd_vert<J>(U, X1, X2, rowtab_read<false>(tab, rc, c.cs, g.fr0, w), c);
    so[k] = fwd_finish<J>(f, halo_of<J>(f.us), halo_of<J>(f.ud), halo_of<J>(f.as), halo_of<J>(f.bd), kk[k], rc, n, c, p, flags, fn,
                          own, sums);
  }
  {
    const float t0 = wave_sum(sums[0]), t1 = wave_sum(sums[1]), t2 = wave_sum(sums[2]), t3 = wave_sum(sums[3]);
    if (lane == 0) { red[wave * 4 + 0] = t0; red[wave * 4 + 1] = t1; red[wave * 4 + 2] = t2; red[wave * 4 + 3] = t3; }
  }
  __syncthreads();                     // also: every read of the field planes is done
  if (tid < 4) {
    float t = 0.f;
    for (int wv = 0; wv < pl.waves; ++wv) t += red[wv * 4 + tid];           // fixed order: deterministic
    partials[((size_t)b * pl.nbands + bi) * 4 + tid] = t;
  }
  if (!BWD) return;
#pragma unroll
  for (int k = 0; k < NPASS; ++k) {
    if (c.active && row[k] < g.sr1) {
      float* q = lds + (row[k] - g.fr0) * w + 4 * c.cs;
      st4(q, so[k].p1.v); st4(q + plane, so[k].p2.v); st4(q + 2 * plane, so[k].cc.v);
    }
  }
  __syncthreads();
  const BPlane G1{lds, g.fr0, g.fr1, w}, G2{lds + plane, g.fr0, g.fr1, w}, GC{lds + 2 * plane, g.fr0, g.fr1, w};
  constexpr int NKEEP = A ? 1 : NPASS;          // (A stores each pass at once)
  V4 du[NKEEP], d1[NKEEP], d2[NKEEP];
#pragma unroll
  for (int k = 0; k < NPASS; ++k) {
    const int r = row[k], rc = r < g.r0 ? g.r0 : (r < g.r1 ? r : g.r1 - 1);
    const bool own = c.active && r >= g.r0 && r < g.r1;
    constexpr int kk_ = 0;
    const int ko = A ? kk_ : k;
#ifdef PDES_TUNE
    if (flags & 4096) {
      const int o = (rc - g.fr0) * w + 4 * c.cs;
      du[ko] = ldoff(G1, o); d1[ko] = ldoff(G2, o); d2[ko] = ldoff(GC, o);
    } else
#endif
    {
      const AdjVert a = adj_vert<J>(G1, G2, GC, rowtab_read<true>(tab, rc, c.cs, g.fr0, w), c);
      adj_finish<J>(a, halo_of<J>(a.p1s), halo_of<J>(a.p2d), halo_of<J>(a.ccs), halo_of<J>(a.ccd), so[k], c, fn, du[ko], d1[ko], d2[ko]);
    }
    if (A && own) {
      float* o = gb + (size_t)r * n + 4 * c.cs;
      if (p.nt) {
        nt_store4(reinterpret_cast<float4*>(o), make_float4(du[0].v[0], du[0].v[1], du[0].v[2], du[0].v[3]));
        nt_store4(reinterpret_cast<float4*>(o + nn), make_float4(d1[0].v[0], d1[0].v[1], d1[0].v[2], d1[0].v[3]));
        nt_store4(reinterpret_cast<float4*>(o + 2 * nn), make_float4(d2[0].v[0], d2[0].v[1], d2[0].v[2], d2[0].v[3]));
      } else {
        st4(o, du[0].v); st4(o + nn, d1[0].v); st4(o + 2 * nn, d2[0].v);
      }
    }
  }
  if (A) return;
#ifdef PDES_TUNE
  if (flags & 8192) return;
#endif
  // the general path: own strips -> the planes (every source has been read) -> one contiguous range per plane
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NPASS; ++k) {
    if (c.active && row[k] >= g.r0 && row[k] < g.r1) {
      float* q = lds + (row[k] - g.fr0) * w + 4 * c.cs;
      const int ko = A ? 0 : k;
      st4(q, du[ko].v); st4(q + plane, d1[ko].v); st4(q + 2 * plane, d2[ko].v);
    }
  }
  __syncthreads();
  {
    const int per = (g.r1 - g.r0) * n;
    const float inv = 1.0f / (float)n;
#pragma unroll 1
    for (int q = 0; q < 3; ++q) {
      float* dst = gb + q * nn + (size_t)g.r0 * n;
      const float* src = lds + q * plane + (g.r0 - g.fr0) * w;
#pragma unroll 2
      for (int i = tid; i < per; i += nthreads) {
        const int rr = (int)(((float)i + 0.5f) * inv);
        const float v = src[rr * w + (i - rr * n)];
        if (p.nt) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
      }
    }
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

// the row-band kernel serves 8 <= n <= 256 unless the caller asks for the tile kernel (PDES_LOSS_TILED: cross-checks)
static bool band_plan(int n, int flags, band::Plan& pl) {
  return !(flags & PDES_LOSS_TILED) && band::choose_plan(n, BAND_LDSF, pl);
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
#define PDES_BAND_LAUNCH(BWD_, NP_, J_) \
    hipLaunchKernelGGL((darcy_loss_band_kernel<BWD_, NP_, J_>), grid, block, shmem, st, K, y, gy, partials, p, pl, flags)
#define PDES_BAND_LAUNCH_J(J_)                                                                             \
    do {                                                                                                   \
      if (pl.npass == 1) { if (gy) PDES_BAND_LAUNCH(true, 1, J_); else PDES_BAND_LAUNCH(false, 1, J_); }     \
      else { if (gy) PDES_BAND_LAUNCH(true, 2, J_); else PDES_BAND_LAUNCH(false, 2, J_); }                   \
    } while (0)
    switch (a16 ? 4 : pl.jl) {
      case 4: PDES_BAND_LAUNCH_J(4); break;
      case 3: PDES_BAND_LAUNCH_J(3); break;
      case 2: PDES_BAND_LAUNCH_J(2); break;
      case 1: PDES_BAND_LAUNCH_J(1); break;
      default: PDES_BAND_LAUNCH_J(0); break;
    }
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
