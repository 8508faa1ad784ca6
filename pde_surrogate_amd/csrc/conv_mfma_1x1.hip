// 1x1 convolutions (the channel-halving GEMMs of the transition layers, models/codec.py:105-111 and :128-134 of the
// reference) and their data gradients on v_mfma_f32_16x16x4_f32 WITHOUT an LDS tile: a 1x1 convolution has no
// halo and uses every staged activation for exactly one tap, so staging through LDS (conv_mfma.hip) costs more
// than the matrix work.  Here the A operand goes global -> registers -> BatchNorm+ReLU -> MFMA:
//   * a wave owns 32 consecutive pixels of one image plane (two M-tiles of 16) and NTW output-channel tiles; lane
//     (i = lane & 15, kq = lane >> 4) loads pixels i and 16 + i of channel 4*kstep + kq (16 lanes read 64
//     contiguous bytes per channel and M-tile);
//   * the accumulator of a lane (rows 4*(lane>>4) + r) is 4 consecutive pixels of one channel per M-tile: one
//     float4 store (forward) / float4 read-modify-write of T (data gradient) each, 64 contiguous bytes per channel
//     and instruction;
//   * the weights come straight from the packed MFMA image (pack_mfma_item) -- 4 bytes per lane per (K-step, N-tile),
//     prefetched two stages ahead with the activations (three register sets, straight-line loop, no conditional
//     loads: see conv_mfma.hip for why);
//   * K is split over the KSPLIT waves of a workgroup that share a pixel group (the maps are small: 8192..32768
//     pixels per minibatch, so N- and K-splits are what fills 1024 SIMDs); the partial sums meet in LDS and each
//     wave finishes every KSPLIT-th N-tile (stores, BatchNorm statistics / BatchNorm-backward epilogue);
//   * a workgroup is 8 waves (4 when 8 would leave CUs idle) = NW / KSPLIT pixel groups; their per-channel statistics are combined in LDS before
//     the fp64 atomics (measured: the atomics of one wave per 32 pixels cost 4 of the layer's 20 us).
#include <stdlib.h>
#include "pdes_common.h"
#include "pdes_options.h"
#include "../../include/pdes_hip.h"

namespace pdes {

typedef float v4f __attribute__((ext_vector_type(4)));

enum { P1_FWD = 0, P1_BWD = 1 };

struct BnP { float mean, invstd, gamma, beta; };
__device__ __forceinline__ BnP bn_coef_p(const pdes_conv_desc& d, int c, bool publish = false) {
  BnP o;
  if (d.eval_mode) {
    o.mean = d.run_mean[c];
    o.invstd = (float)(1.0 / sqrt((double)d.run_var[c] + (double)d.eps));
  } else {
    const MeanInv mi = batch_mean_invstd(d.coef, d.x_stats, d.rep_stride, (double)d.B * d.Hin * d.Win, d.eps, c, publish);
    o.mean = mi.mean;
    o.invstd = mi.invstd;
  }
  o.gamma = d.gamma[c];
  o.beta = d.beta[c];
  return o;
}

// grid: (ceil(pixel groups / (NW / KSPLIT)), ceil(N-tiles / NTW)), 64 NW threads; dynamic LDS: [kC] float4 coefficients
// (forward) + [NW waves][NTW][2 M-tiles][64 lanes] float4 partial sums + [NW waves][NOWN][16 channels] float4 statistics
template <int NTW, int KSPLIT, int NW, int MODE>
__global__ __launch_bounds__(64 * NW, 2) void conv1x1_mfma_kernel(pdes_conv_desc d, const float* __restrict__ wm,
                                                              int nt_total, int groups) {
  constexpr int GP = NW / KSPLIT;                      // pixel groups per workgroup
  constexpr int NOWN = (NTW + KSPLIT - 1) / KSPLIT;    // N-tiles a wave finishes
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_p1[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pg = wave / KSPLIT, kslice = wave % KSPLIT;
  const int HW = d.Hin * d.Win;
  const int gpi = HW >> 5;                              // pixel groups per image plane
  const bool gvalid = blockIdx.x * GP + pg < groups;    // the last workgroup may hold idle pixel groups
  const int G = min((int)blockIdx.x * GP + pg, groups - 1);
  const int b = G / gpi, pix0 = (G % gpi) << 5;
  const int nt_base = blockIdx.y * NTW;
  const int ntp = (nt_total + 7) & ~7;                  // N-tiles of the weight image
  const int kC = MODE == P1_FWD ? d.Cin : d.Cout;
  const int ksteps = (kC + 3) >> 2, per = (ksteps + KSPLIT - 1) / KSPLIT;
  const int kbeg = kslice * per, kend = min(kbeg + per, ksteps);
  const int nst = (per + 1) >> 1;                       // stages of two K-steps (the same for every wave)
  const int ksf = ((kC + 15) >> 4) * 4;                 // K-steps of the weight image
  const float* kbase = MODE == P1_FWD ? d.x + (size_t)b * d.x_ctot * HW
                                      : d.g + ((size_t)b * d.g_ctot + d.g_coff) * HW;
  float4* cf4 = reinterpret_cast<float4*>(smem_p1);
  v4f* red = reinterpret_cast<v4f*>(smem_p1 + (MODE == P1_FWD ? 16 * (size_t)kC : 0));
  float4* sstat = reinterpret_cast<float4*>(red + NW * NTW * 2 * 64);

  if (MODE == P1_FWD) {
    for (int c = tid; c < kC; c += 64 * NW) {
      const BnP k = bn_coef_p(d, c, (blockIdx.x | blockIdx.y | blockIdx.z) == 0);
      cf4[c] = make_float4(k.mean, k.gamma * k.invstd, k.beta, 0.f);
    }
  }
  // data gradient: the BatchNorm coefficients of the epilogue are fetched before the matrix loop
  BnP kepi[NOWN];
  if (MODE == P1_BWD) {
#pragma unroll
    for (int j = 0; j < NOWN; ++j)
      kepi[j] = bn_coef_p(d, min((nt_base + kslice + j * KSPLIT) * 16 + (lane & 15), d.Cin - 1));
  }

  struct Stage { float x[2][2]; float w[2][NTW]; };
  const int kq = lane >> 4;
  const float* xlane = kbase + pix0 + (lane & 15);
  auto issue = [&](int st, Stage& s) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ks = kbeg + 2 * st + j;
      const float* xp = xlane + (size_t)min(4 * ks + kq, kC - 1) * HW;
      s.x[j][0] = xp[0];
      s.x[j][1] = xp[16];
      const float* wp = wm + ((size_t)min(ks, ksf - 1) * ntp) * 64 + lane;
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) s.w[j][nt] = wp[(size_t)min(nt_base + nt, ntp - 1) * 64];
    }
  };

  v4f acc[2][NTW];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[q][nt] = (v4f){0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int st, const Stage& s) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ks = kbeg + 2 * st + j, c = 4 * ks + kq;
      const bool ok = ks < kend && c < kC;
      float a0 = s.x[j][0], a1 = s.x[j][1];
      if (MODE == P1_FWD) {
        const float4 k = cf4[min(c, kC - 1)];
        a0 = fmaxf(0.f, (a0 - k.x) * k.y + k.z);
        a1 = fmaxf(0.f, (a1 - k.x) * k.y + k.z);
      }
      a0 = ok ? a0 : 0.f;
      a1 = ok ? a1 : 0.f;
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, s.w[j][nt], acc[0][nt], 0, 0, 0);
        acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, s.w[j][nt], acc[1][nt], 0, 0, 0);
      }
    }
  };

  // no barrier separates the stages, so the machine scheduler would sink a stage's loads down to their first use
  // (and with them the whole prefetch): pin every issue / compute group with a scheduling barrier
#define PDES_P1_ISSUE(st_, s_) do { issue(st_, s_); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PDES_P1_COMPUTE(st_, s_) do { compute(st_, s_); __builtin_amdgcn_sched_barrier(0); } while (0)
  Stage s0, s1, s2;
  PDES_P1_ISSUE(0, s0);
  PDES_P1_ISSUE(1, s1);
  __syncthreads();                    // coefficients visible
  {
    int st = 0;
    for (; st + 2 < nst; st += 3) {
      PDES_P1_ISSUE(st + 2, s2); PDES_P1_COMPUTE(st, s0);
      PDES_P1_ISSUE(st + 3, s0); PDES_P1_COMPUTE(st + 1, s1);
      PDES_P1_ISSUE(st + 4, s1); PDES_P1_COMPUTE(st + 2, s2);
    }
    if (st < nst) {
      PDES_P1_COMPUTE(st, s0);
      if (st + 1 < nst) PDES_P1_COMPUTE(st + 1, s1);
    }
  }
#undef PDES_P1_ISSUE
#undef PDES_P1_COMPUTE

  // ---- the K-split partial sums meet in LDS; wave (pg, kslice) finishes N-tiles kslice, kslice + KSPLIT, ...
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int q = 0; q < 2; ++q) red[((wave * NTW + nt) * 2 + q) * 64 + lane] = acc[q][nt];
  __syncthreads();

  const int pp = pix0 + 4 * kq;        // acc[q][r] of this lane = pixel pp + 16 q + r
#pragma unroll
  for (int j = 0; j < NOWN; ++j) {
    const int nt = kslice + j * KSPLIT;
    float4 stat = make_float4(0.f, 0.f, 0.f, 0.f);     // forward: (sum, sum of squares); data gradient: (dg, db, st, sx)
    if (nt < NTW && nt_base + nt < nt_total && gvalid) {
      v4f v[2] = {(v4f){0.f, 0.f, 0.f, 0.f}, (v4f){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int w = 0; w < KSPLIT; ++w)
#pragma unroll
        for (int q = 0; q < 2; ++q) v[q] += red[(((pg * KSPLIT + w) * NTW + nt) * 2 + q) * 64 + lane];
      const int ch = (nt_base + nt) * 16 + (lane & 15);
      if (MODE == P1_FWD) {
        if (ch < d.Cout) {
          float* ob = d.out + ((size_t)b * d.out_ctot + d.out_coff + ch) * HW + pp;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            *reinterpret_cast<float4*>(ob + 16 * q) = make_float4(v[q][0], v[q][1], v[q][2], v[q][3]);
#pragma unroll
            for (int r = 0; r < 4; ++r) { stat.x += v[q][r]; stat.y += v[q][r] * v[q][r]; }
          }
        }
      } else if (ch < d.Cin) {
        const BnP k = kepi[j];
        const float scale = k.gamma * k.invstd;
        const bool fin = ch >= d.final_c0 && ch < d.final_c1;
        const size_t idx = ((size_t)b * d.x_ctot + ch) * HW + pp;
        float4 xq[2], tq[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          xq[q] = *reinterpret_cast<const float4*>(d.x + idx + 16 * q);
          tq[q] = d.t_accumulate ? *reinterpret_cast<const float4*>(d.t_in + idx + 16 * q)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float xs[4] = {xq[q].x, xq[q].y, xq[q].z, xq[q].w};
          float ts[4] = {tq[q].x, tq[q].y, tq[q].z, tq[q].w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float y = (xs[r] - k.mean) * scale + k.beta;
            const float xh = (xs[r] - k.mean) * k.invstd;
            const float dyv = (y > 0.f) ? v[q][r] : 0.f;
            stat.y += dyv; stat.x += dyv * xh;
            ts[r] += k.gamma * dyv;
            if (fin) { stat.z += ts[r]; stat.w += ts[r] * xh; }
          }
          *reinterpret_cast<float4*>(d.t_in + idx + 16 * q) = make_float4(ts[0], ts[1], ts[2], ts[3]);
        }
      }
    }
    // the four lanes of a channel, then (below) the pixel groups of the workgroup
    stat.x += __shfl_xor(stat.x, 16, 64); stat.x += __shfl_xor(stat.x, 32, 64);
    stat.y += __shfl_xor(stat.y, 16, 64); stat.y += __shfl_xor(stat.y, 32, 64);
    if (MODE == P1_BWD) {
      stat.z += __shfl_xor(stat.z, 16, 64); stat.z += __shfl_xor(stat.z, 32, 64);
      stat.w += __shfl_xor(stat.w, 16, 64); stat.w += __shfl_xor(stat.w, 32, 64);
    }
    if (lane < 16) sstat[(wave * NOWN + j) * 16 + lane] = stat;
  }
  __syncthreads();
  if (pg == 0 && lane < 16 && (MODE == P1_BWD || d.out_stats)) {
    const long long ro = (long long)rep_of_block(d.nrep) * d.rep_stride;
#pragma unroll
    for (int j = 0; j < NOWN; ++j) {
      const int nt = kslice + j * KSPLIT, ch = (nt_base + nt) * 16 + lane;
      if (nt >= NTW || nt_base + nt >= nt_total || ch >= (MODE == P1_FWD ? d.Cout : d.Cin)) continue;
      double tx = 0.0, ty = 0.0, tz = 0.0, tw = 0.0;   // 32-pixel fp32 partial sums meet in fp64
#pragma unroll
      for (int g = 0; g < GP; ++g) {
        const float4 u = sstat[((g * KSPLIT + kslice) * NOWN + j) * 16 + lane];
        tx += (double)u.x; ty += (double)u.y; tz += (double)u.z; tw += (double)u.w;
      }
      if (MODE == P1_FWD) {
        atomicAdd(&d.out_stats[ro + 2 * (d.out_coff + ch)], tx);
        atomicAdd(&d.out_stats[ro + 2 * (d.out_coff + ch) + 1], ty);
      } else {
        atomicAdd(&d.bn_grad[ro + 2 * ch], tx);
        atomicAdd(&d.bn_grad[ro + 2 * ch + 1], ty);
        if (ch >= d.final_c0 && ch < d.final_c1) {
          atomicAdd(&d.t_stats[ro + 2 * ch], tz);
          atomicAdd(&d.t_stats[ro + 2 * ch + 1], tw);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of a 1x1 convolution: dW[co][ci] = sum over pixels of g[co][p] * relu(bn(x[ci][p])) -- a GEMM with
// K = pixels, whose operands are both pixel-contiguous in NCHW: lane (row/col = lane & 15, kq = lane >> 4) loads ONE
// float4 = pixels p + 4 kq .. + 3 of its channel per operand tile, and component t of the float4s is K-step t (the
// K index (kq, t) <-> pixel p + 4 kq + t is the same permutation on both sides).  No LDS tile, no barrier in the
// loop.  A workgroup = KSW waves that split the pixels of ONE image for MTW x NTW output tiles (all output-channel
// tiles x a group of input-channel tiles); their accumulators meet in LDS and go to the split-K partial buffer of
// split blockIdx.x (the plan of conv_mfma_wgrad.hip: spi splits per image), reduced later with every other layer.
template <int MTW, int NTW, int KSW>
__global__ __launch_bounds__(64 * KSW, 2) void conv1x1_wgrad_kernel(pdes_conv_desc d, float* __restrict__ part, int spi) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_w1[];
  const int tid = threadIdx.x, lane = tid & 63, i16 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / spi, nt0 = blockIdx.y * NTW;        // spi splits (workgroups) per image
  const int HW = d.Hin * d.Win;
  const int npx = HW / (spi * KSW), nst = npx >> 4;    // pixels and 16-pixel stages of this wave
  float4* cf4 = reinterpret_cast<float4*>(smem_w1);   // [NTW * 16] BatchNorm coefficients of the B-operand channels
  v4f* red = reinterpret_cast<v4f*>(smem_w1 + 16 * NTW * 16);

  if (tid < NTW * 16) {
    const BnP k = bn_coef_p(d, min(nt0 * 16 + tid, d.Cin - 1));
    cf4[tid] = make_float4(k.mean, k.gamma * k.invstd, k.beta, 0.f);
  }
  const float* ga[MTW];
  const float* xb[NTW];
  const size_t poff = (size_t)((blockIdx.x % spi) * KSW + wave) * npx + 4 * kq;
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
    ga[mt] = d.g + ((size_t)b * d.g_ctot + d.g_coff + min(mt * 16 + i16, d.Cout - 1)) * HW + poff;
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
    xb[nt] = d.x + ((size_t)b * d.x_ctot + min((nt0 + nt) * 16 + i16, d.Cin - 1)) * HW + poff;

  struct Stage { float4 a[MTW]; float4 x[NTW]; };
  auto issue = [&](int st, Stage& s) __attribute__((always_inline)) {
    const int o = min(st, nst - 1) << 4;
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) s.a[mt] = *reinterpret_cast<const float4*>(ga[mt] + o);
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) s.x[nt] = *reinterpret_cast<const float4*>(xb[nt] + o);
  };
  v4f acc[MTW][NTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = (v4f){0.f, 0.f, 0.f, 0.f};
  float4 kc[NTW];
  auto compute = [&](const Stage& s) __attribute__((always_inline)) {
    float bv[NTW][4];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const float xs[4] = {s.x[nt].x, s.x[nt].y, s.x[nt].z, s.x[nt].w};
#pragma unroll
      for (int t = 0; t < 4; ++t) bv[nt][t] = fmaxf(0.f, (xs[t] - kc[nt].x) * kc[nt].y + kc[nt].z);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) {
        const float av = t == 0 ? s.a[mt].x : (t == 1 ? s.a[mt].y : (t == 2 ? s.a[mt].z : s.a[mt].w));
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[nt][t], acc[mt][nt], 0, 0, 0);
      }
  };

#define PDES_W1_ISSUE(st_, s_) do { issue(st_, s_); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PDES_W1_COMPUTE(s_) do { compute(s_); __builtin_amdgcn_sched_barrier(0); } while (0)
  Stage s0, s1, s2;
  PDES_W1_ISSUE(0, s0);
  PDES_W1_ISSUE(1, s1);
  __syncthreads();                    // coefficients visible
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) kc[nt] = cf4[nt * 16 + i16];
  {
    int st = 0;
    for (; st + 2 < nst; st += 3) {
      PDES_W1_ISSUE(st + 2, s2); PDES_W1_COMPUTE(s0);
      PDES_W1_ISSUE(st + 3, s0); PDES_W1_COMPUTE(s1);
      PDES_W1_ISSUE(st + 4, s1); PDES_W1_COMPUTE(s2);
    }
    if (st < nst) {
      PDES_W1_COMPUTE(s0);
      if (st + 1 < nst) PDES_W1_COMPUTE(s1);
    }
  }
#undef PDES_W1_ISSUE
#undef PDES_W1_COMPUTE

  // ---- the pixel-split partial sums meet in LDS; wave w finishes tiles w, w + KSW, ...
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) red[((wave * MTW + mt) * NTW + nt) * 64 + lane] = acc[mt][nt];
  __syncthreads();
  float* pb = part + (size_t)blockIdx.x * d.Cout * d.Cin;
  for (int tt = wave; tt < MTW * NTW; tt += KSW) {
    v4f v = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < KSW; ++w) v += red[(w * MTW * NTW + tt) * 64 + lane];
    const int mt = tt / NTW, nt = tt % NTW;
    const int ci = (nt0 + nt) * 16 + i16, co0 = mt * 16 + 4 * kq;
    if (ci < d.Cin) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (co0 + r < d.Cout) pb[(size_t)(co0 + r) * d.Cin + ci] = v[r];
    }
  }
}

// ------------------------------------------------------------------------------- host dispatch
// PDES_MFMA_1X1: bit mask of the register-operand 1x1 kernels: 1 forward, 2 data gradient, 4 weight gradient (default 7;
// 0: conv_mfma.hip / conv_mfma_wgrad.hip serve the 1x1 layers)
static bool p1_enabled(bool bwd) { return (opt().mfma_1x1 & (bwd ? 2 : 1)) != 0; }

static bool p1_shape_ok(const pdes_conv_desc& d, bool bwd) {
  if (d.ksize != 1 || d.stride != 1 || d.pad != 0 || d.upsample || !d.has_bn || d.nrep != PDES_NREP) return false;
  if (d.Hin != d.Hout || d.Win != d.Wout) return false;
  const int kC = bwd ? d.Cout : d.Cin, nC = bwd ? d.Cin : d.Cout;
  if (kC < 32 || nC < 33 || kC > 1024) return false;          // at least three N-tiles; coefficient table <= 16 KB
  return (d.Hin * d.Win) % 32 == 0;
}

template <int MODE>
static int launch_p1(const pdes_conv_desc& d, const float* wm, hipStream_t st) {
  const bool bwd = MODE == P1_BWD;
  const int kC = bwd ? d.Cout : d.Cin, nC = bwd ? d.Cin : d.Cout;
  const int nt_total = (nC + 15) / 16;
  const int ntw = (nt_total <= 5 || (nt_total > 7 && nt_total <= 10)) ? 5 : 7;
  const int nz = (nt_total + ntw - 1) / ntw;
  const long long groups = (long long)d.B * (d.Hin * d.Win / 32);
  // K-split: enough waves for 1024 SIMDs, but at least four K-steps per wave
  int ksplit = (groups * nz * 2 >= 1024 || kC < 64) ? 2 : 4;
  // 8 waves per workgroup (more pixel groups share one set of statistics atomics) unless that leaves CUs idle
  const int nw = ((groups + 8 / ksplit - 1) / (8 / ksplit)) * nz >= 256 ? 8 : 4;
  const int gp = nw / ksplit, nown = (ntw + ksplit - 1) / ksplit;
  dim3 grid((unsigned)((groups + gp - 1) / gp), nz), block(64 * nw);
  const size_t lds = (bwd ? 0 : 16 * (size_t)kC) + (size_t)nw * ntw * 2 * 64 * 16 + (size_t)nw * nown * 16 * 16;
#define PDES_P1_LAUNCH(NTW_, KS_)                                                                              \
  do {                                                                                                         \
    if (nw == 8) hipLaunchKernelGGL((conv1x1_mfma_kernel<NTW_, KS_, 8, MODE>), grid, block, lds, st, d, wm,    \
                                    nt_total, (int)groups);                                                    \
    else hipLaunchKernelGGL((conv1x1_mfma_kernel<NTW_, KS_, 4, MODE>), grid, block, lds, st, d, wm, nt_total,  \
                            (int)groups);                                                                      \
  } while (0)
  if (ntw == 5) { if (ksplit == 2) PDES_P1_LAUNCH(5, 2); else PDES_P1_LAUNCH(5, 4); }
  else { if (ksplit == 2) PDES_P1_LAUNCH(7, 2); else PDES_P1_LAUNCH(7, 4); }
#undef PDES_P1_LAUNCH
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

int conv_forward_1x1(const pdes_conv_desc& d, hipStream_t st, bool dry) {
  if (!p1_enabled(false) || !d.wm_fwd || !p1_shape_ok(d, false)) return PDES_ENOSUP;
  if (dry) return PDES_OK;
  return launch_p1<P1_FWD>(d, d.wm_fwd, st);
}

// dry = true: only report whether this implementation would take the descriptor
int conv_backward_data_1x1(const pdes_conv_desc& d, hipStream_t st, bool dry) {
  if (!p1_enabled(true) || !d.wm_bwd || !p1_shape_ok(d, true) || d.eval_mode || d.g_fused) return PDES_ENOSUP;
  if (dry) return PDES_OK;
  return launch_p1<P1_BWD>(d, d.wm_bwd, st);
}


// weight gradient into the split-K partial buffer d.ws, spi splits per image (the plan of conv_mfma_wgrad.hip);
// PDES_ENOSUP leaves the layer to the generic kernel
int conv_backward_weight_1x1(const pdes_conv_desc& d, int spi, hipStream_t st) {
  if (!(opt().mfma_1x1 & 4)) return PDES_ENOSUP;
  if (!p1_shape_ok(d, false) || !d.ws || d.eval_mode || d.g_fused || spi < 1) return PDES_ENOSUP;
  const int mtiles = (d.Cout + 15) / 16, ntiles = (d.Cin + 15) / 16, HW = d.Hin * d.Win;
  if ((long long)d.B * spi * d.Cout * d.Cin * 4 > d.ws_bytes) return PDES_ENOSUP;
  if (mtiles == 5 && spi == 4 && HW % (4 * 4 * 16) == 0) {
    dim3 grid(d.B * spi, (ntiles + 2) / 3), block(256);
    const size_t lds = 16 * 3 * 16 + (size_t)4 * 5 * 3 * 64 * 16;
    hipLaunchKernelGGL((conv1x1_wgrad_kernel<5, 3, 4>), grid, block, lds, st, d, d.ws, spi);
  } else if (mtiles == 5 && spi == 1 && HW % 128 == 0) {
    dim3 grid(d.B, (ntiles + 2) / 3), block(512);
    const size_t lds = 16 * 3 * 16 + (size_t)8 * 5 * 3 * 64 * 16;
    hipLaunchKernelGGL((conv1x1_wgrad_kernel<5, 3, 8>), grid, block, lds, st, d, d.ws, spi);
  } else if (mtiles == 7 && spi == 1 && HW % 64 == 0) {
    dim3 grid(d.B, (ntiles + 1) / 2), block(256);
    const size_t lds = 16 * 2 * 16 + (size_t)4 * 7 * 2 * 64 * 16;
    hipLaunchKernelGGL((conv1x1_wgrad_kernel<7, 2, 4>), grid, block, lds, st, d, d.ws, spi);
  } else {
    return PDES_ENOSUP;
  }
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

}  // namespace pdes
