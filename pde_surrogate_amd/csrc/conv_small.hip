// 3x3 convolutions on 8x8 feature maps (the coarsest level of the conditional Glow, models/glow_msc.py: 24 coupling-net
// convolutions + the encoder's last dense block per training step) on v_mfma_f32_16x16x4_f32 -- forward, data gradient
// and weight gradient.  conv_mfma.hip tiles a map in rows of 16 pixels, so an 8-wide map fell to the VALU kernels
// (339 / 68 / 70 us per layer at batch 32: 87 % of the conditional Glow's step).  Here a whole 8x8 image is ONE
// workgroup's tile:
//   * forward / data gradient: the K-side planes of an image (BatchNorm+ReLU'd input channels, or the raw output
//     gradient) sit in LDS with a zero halo, [channel][10 x 10] at an ODD channel stride; wave w owns the M-tile of
//     rows 2w, 2w+1 (16 pixels), the B operand streams from the packed weight image (pack_mfma_item: 4 B per lane
//     per (k-step, tap, N-tile), coalesced), one MFMA per (k-step of 4 channels, tap, N-tile);
//     the accumulator of a lane is 4 consecutive pixels of one row and one channel: float4 stores, the statistics
//     (forward) / ReLU mask, gamma, T, dgamma, dbeta, finished-channel sums (data gradient) are reduced over the
//     workgroup before one fp64 atomic per channel;
//   * weight gradient: M = 16 input channels, N = output channels, K = pixels (16 k-steps per image and tap);
//     a workgroup = (input-channel tile, slice of the batch), each wave one image at a time out of its own LDS
//     region, the four waves' sums are added in a fixed order and written as one split-K partial (reduced by
//     pdes_wgrad_reduce_all with every other layer's): deterministic.
#include <hip/hip_ext.h>
#include "pdes_common.h"
#include "pdes_options.h"
#include "../../include/pdes_hip.h"

namespace pdes {

hipEvent_t take_dgrad_stop_event();       // conv_mfma.hip: the completion signal pdes_backward2 wants on the next data gradient

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int SM_PITCH = 10;     // 8 columns + halo
constexpr int SM_CS = 101;       // plane stride: 10 x 10 + 1, odd (channel-major operand reads spread over the banks)
constexpr int SM_GS = 65;        // stride of a halo-free 64-pixel plane

struct BnS { float mean, invstd, gamma, beta; };
__device__ __forceinline__ BnS bn_coef_s(const pdes_conv_desc& d, int c) {
  BnS o;
  if (d.eval_mode) {
    o.mean = d.run_mean[c];
    o.invstd = (float)(1.0 / sqrt((double)d.run_var[c] + (double)d.eps));
  } else {
    const double n = (double)d.B * d.Hin * d.Win;
    const double m = rep_sum(d.x_stats, 2 * c, d.nrep, d.rep_stride) / n;
    double var = rep_sum(d.x_stats, 2 * c + 1, d.nrep, d.rep_stride) / n - m * m;
    var = var < 0.0 ? 0.0 : var;
    o.mean = (float)m;
    o.invstd = (float)(1.0 / sqrt(var + (double)d.eps));
  }
  o.gamma = d.gamma[c];
  o.beta = d.beta[c];
  return o;
}

// (a convolution without a BatchNorm in front -- Conv2dZeros on the encoder's features, glow_msc.py:519 -- takes the same
//  kernels: the planes are staged as they are, the data gradient is the plain accumulation)
bool conv_small_applies(const pdes_conv_desc& d) {
  return opt().mfma_small && !opt().conv_direct && d.ksize == 3 && d.stride == 1 && d.pad == 1 && !d.upsample &&
         d.Hin == 8 && d.Win == 8 && d.Hout == 8 && d.Wout == 8 && d.Cin <= 352 && d.Cout <= 352 && d.nrep == PDES_NREP &&
         (!d.g_fused || (d.Cout <= 16 && d.fin_xstats && d.fin_tstats && d.out && d.g_ctot == d.out_ctot && d.g_coff == d.out_coff &&
                         !d.g_add));
}

// GF (pdes_conv_desc.g_fused, <= 16 output channels): `g` still holds the accumulator T of the layer's output buffer; the
// data- and weight-gradient kernels apply the BatchNorm-backward finalize  invstd (T - mean T - xhat mean(T xhat))  while they
// stage the gradient planes (raw activation = `out`): {mean, invstd, mean T, mean T xhat} of the 16 channels, from the
// replicated fp64 tables -- one load per thread and round, 16-lane shuffle reductions (flow_copy_bwd_kernel's scheme)
__device__ __forceinline__ void fin_table_small(const pdes_conv_desc& d, float4* fc, double (*sums)[4], int tid, int nthreads) {
  for (int e = tid; e < 16 * 64; e += nthreads) {
    const int c = e >> 6, q = (e >> 4) & 3, r = e & 15;
    const int ch = d.g_coff + min(c, d.Cout - 1);
    double v = r < PDES_NREP ? (q < 2 ? d.fin_xstats : d.fin_tstats)[(long long)r * d.rep_stride + 2 * ch + (q & 1)] : 0.0;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 16);
    if (r == 0) sums[c][q] = v;
  }
  __syncthreads();
  if (tid < 16) {
    const double inv_n = 1.0 / ((double)d.B * 64);
    const double m = sums[tid][0] * inv_n;
    double var = sums[tid][1] * inv_n - m * m;
    var = var < 0.0 ? 0.0 : var;
    fc[tid] = make_float4((float)m, (float)(1.0 / sqrt(var + (double)d.eps)), (float)(sums[tid][2] * inv_n),
                          (float)(sums[tid][3] * inv_n));
  }
}
__device__ __forceinline__ float fin_apply(const float4 k, float t, float x) { return k.y * (t - k.z - (x - k.x) * k.y * k.w); }

// forward only: also the stride-2 transition onto the 8x8 level (16x16 input planes)
static bool small_fwd_s2_applies(const pdes_conv_desc& d) {
  return opt().mfma_small && !opt().conv_direct && d.ksize == 3 && d.stride == 2 && d.pad == 1 && !d.upsample && d.has_bn &&
         d.Hin == 16 && d.Win == 16 && d.Hout == 8 && d.Wout == 8 && d.Cin <= 112 && d.nrep == PDES_NREP;
}

// ------------------------------------------------------------------------------------------------------- forward
// grid (B, ceil(N-tiles / NT)), 256 threads.  dynamic LDS: coefficients [Cin] float4 + planes [Cin][SM_CS] + red [4][NT][16][2]
// S = stride: the input planes are (8 S) x (8 S) with a halo column / row on each side
// KG = K-split wave groups: group g (waves 4g .. 4g+3) multiplies every KG-th k-step; the groups' sums meet in LDS
// (one workgroup per image is all the parallelism an 8x8 map offers: the serial chain of a wave is what a layer costs)
template <int NT, int S, int KG>
__global__ __launch_bounds__(256 * KG) void conv_small_fwd_kernel(pdes_conv_desc d, const float* __restrict__ wm, int nt_total) {
  extern __shared__ __attribute__((aligned(16))) float sm_small[];
  constexpr int WI = 8 * S, PITCH = WI + 2, CS = (PITCH * PITCH) | 1, HWI = WI * WI, NTH = 256 * KG;
  const int Cin = d.Cin, tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, kgrp = tid >> 8, b = blockIdx.x;
  float4* cf = reinterpret_cast<float4*>(sm_small);
  float* pl = sm_small + 4 * Cin;
  float* red = pl + Cin * CS;
  const int ntp = (nt_total + 7) & ~7, nt_base = blockIdx.y * NT;
  const int ksf = ((Cin + 15) >> 4) * 4;
  // the weight fragments of this wave group's FIRST k-step: nothing in front of the matrix loop depends on them, so their
  // round trip runs under the staging of the planes; inside the loop the next k-step's nine fragments are requested
  // before the current one's MFMAs (a k-step is 9 MFMAs = 0.12 us of matrix time against ~1 us of L2 round trip: without
  // the prefetch a layer was its 13-22 round trips in a row)
  float w[9][NT], wn[9][NT];
  {
    const float* wp = wm + ((size_t)min(kgrp, ksf - 1) * 9 * ntp + nt_base) * 64 + lane;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int n = 0; n < NT; ++n) w[t][n] = wp[((size_t)t * ntp + n) * 64];
  }
  if (d.has_bn)
    for (int c = tid; c < Cin; c += NTH) {
      const BnS k = bn_coef_s(d, c);
      cf[c] = make_float4(k.mean, k.gamma * k.invstd, k.beta, 0.f);
    }
  // the halo of every plane (+ the pad word): the interior is written by the staging loop below, no barrier in between
  constexpr int NHALO = CS - HWI;
  for (int i = tid; i < Cin * NHALO; i += NTH) {
    const int c = i / NHALO, h = i % NHALO;
    // h: 0 .. PITCH-1 top row, PITCH .. 2 PITCH-1 bottom row, then the two side columns of the WI inner rows, then the pad
    int off;
    if (h < PITCH) off = h;
    else if (h < 2 * PITCH) off = (PITCH - 1) * PITCH + (h - PITCH);
    else if (h < 2 * PITCH + 2 * WI) { const int r = (h - 2 * PITCH) >> 1; off = (r + 1) * PITCH + ((h & 1) ? PITCH - 1 : 0); }
    else off = PITCH * PITCH + (h - 2 * PITCH - 2 * WI);
    pl[c * CS + off] = 0.f;
  }
  if (d.has_bn) __syncthreads();               // the coefficients
  const float* xb = d.x + (size_t)b * d.x_ctot * HWI;
  for (int e = tid; e < Cin * (HWI / 4); e += NTH) {
    const int c = e / (HWI / 4), p = (e % (HWI / 4)) * 4;
    const float4 v = *reinterpret_cast<const float4*>(xb + (size_t)c * HWI + p);
    float z[4] = {v.x, v.y, v.z, v.w};
    if (d.has_bn) {
      const float4 k = cf[c];
#pragma unroll
      for (int j = 0; j < 4; ++j) z[j] = fmaxf(0.f, (z[j] - k.x) * k.y + k.z);
    }
    float* q = pl + c * CS + (p / WI + 1) * PITCH + (p % WI) + 1;      // (WI is a multiple of 4: the four pixels share a row)
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = z[j];
  }
  __syncthreads();
  const int i = lane & 15, kq = lane >> 4;
  // output pixel (oy, ox) reads input (S oy + ky - 1, S ox + kx - 1): with the halo, plane[(S oy + ky) PITCH + S ox + kx]
  const int aoff = (S * (2 * wave + (i >> 3)) + 1) * PITCH + S * (i & 7) + 1;
  constexpr int SM_PITCH = PITCH, SM_CS = CS;                    // (shadow the 8x8 constants below)
  v4f acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) acc[n] = (v4f){0.f, 0.f, 0.f, 0.f};
  for (int ks = kgrp; ks < ksf; ks += KG) {
    const float* ap = pl + min(4 * ks + kq, Cin - 1) * SM_CS + aoff;      // (channels past Cin meet zero weights)
    {
      const float* wp = wm + ((size_t)min(ks + KG, ksf - 1) * 9 * ntp + nt_base) * 64 + lane;
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) wn[t][n] = wp[((size_t)t * ntp + n) * 64];
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float a = ap[(t / 3 - 1) * SM_PITCH + (t % 3 - 1)];
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w[t][n], acc[n], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int n = 0; n < NT; ++n) w[t][n] = wn[t][n];
  }
  if (KG > 1) {                      // the other groups' partial sums: through LDS (the planes are no longer needed)
    __syncthreads();
    v4f* xch = reinterpret_cast<v4f*>(pl);
    if (kgrp > 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t) xch[((kgrp - 1) * NT + t) * 256 + (tid & 255)] = acc[t];
    }
    __syncthreads();
    if (kgrp > 0) return;
#pragma unroll
    for (int q = 0; q < KG - 1; ++q)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] += xch[(q * NT + t) * 256 + tid];
    __syncthreads();
  }
  // lane (n = lane & 15, g = lane >> 4): channel n of each N-tile, pixels 4g .. 4g+3 of the wave's two rows
  const int g = lane >> 4, n = lane & 15;
  const int opix = (2 * wave + (g >> 1)) * 8 + 4 * (g & 1);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int co = (nt_base + t) * 16 + n;
    float s = (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]);
    float q = (acc[t][0] * acc[t][0] + acc[t][1] * acc[t][1]) + (acc[t][2] * acc[t][2] + acc[t][3] * acc[t][3]);
    if (co < d.Cout)
      *reinterpret_cast<float4*>(d.out + ((size_t)b * d.out_ctot + d.out_coff + co) * 64 + opix) =
          make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
    s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
    q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
    if (lane < 16) { red[((wave * NT + t) * 16 + n) * 2] = s; red[((wave * NT + t) * 16 + n) * 2 + 1] = q; }
  }
  if (!d.out_stats) return;
  __syncthreads();
  if (tid < NT * 32) {
    const int t = tid >> 5, nn = (tid >> 1) & 15, w = tid & 1;
    const int co = (nt_base + t) * 16 + nn;
    if (co < d.Cout) {
      double v = 0.0;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) v += (double)red[((wv * NT + t) * 16 + nn) * 2 + w];
      atomicAdd(&d.out_stats[(long long)rep_of_block(d.nrep) * d.rep_stride + 2 * (d.out_coff + co) + w], v);
    }
  }
}

// ------------------------------------------------------------------------------------------------- data gradient
// grid (B, ceil(input-channel tiles / NT)).  dynamic LDS: planes of g [Cout][SM_CS] + coefficients [NT * 16] float4 +
// red [4][NT][16][4]
template <int NT, bool GF>
__global__ __launch_bounds__(256) void conv_small_bwd_kernel(pdes_conv_desc d, const float* __restrict__ wm, int nt_total) {
  extern __shared__ __attribute__((aligned(16))) float sm_small[];
  __shared__ float4 fin_c[GF ? 16 : 1];
  __shared__ double fin_s[GF ? 16 : 1][4];
  const int Cout = d.Cout, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
  const int nt_base = blockIdx.y * NT;
  float* pl = sm_small;
  float4* cf = reinterpret_cast<float4*>(pl + ((Cout * SM_CS + 3) & ~3));
  float* red = reinterpret_cast<float*>(cf + NT * 16);
  const int ntp = (nt_total + 7) & ~7;
  const int ksb = ((Cout + 15) >> 4) * 4;
  // (the first k-step's weight fragments are requested before anything else, the next k-step's before the current one's
  //  MFMAs: see the forward kernel)
  float w[9][NT], wn[9][NT];
  {
    const float* wp = wm + (size_t)nt_base * 64 + lane;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int n = 0; n < NT; ++n) w[t][n] = wp[((size_t)t * ntp + n) * 64];
  }
  if (d.has_bn)
    for (int c = tid; c < NT * 16; c += 256) {
      const BnS k = bn_coef_s(d, min(nt_base * 16 + c, d.Cin - 1));
      cf[c] = make_float4(k.mean, k.invstd, k.gamma, k.beta);
    }
  for (int i = tid; i < Cout * SM_CS; i += 256) pl[i] = 0.f;
  if (GF) fin_table_small(d, fin_c, fin_s, tid, 256);
  __syncthreads();
  const float* gb = d.g + ((size_t)b * d.g_ctot + d.g_coff) * 64;
  const float* ob = GF ? d.out + ((size_t)b * d.out_ctot + d.out_coff) * 64 : nullptr;
  for (int e = tid; e < Cout * 64; e += 256) {
    const int c = e >> 6, p = e & 63;
    float z = gb[e];
    if (GF) z = fin_apply(fin_c[c], z, ob[e]);
    pl[c * SM_CS + ((p >> 3) + 1) * SM_PITCH + (p & 7) + 1] = z;
  }
  __syncthreads();
  const int i = lane & 15, kq = lane >> 4;
  const int aoff = (2 * wave + (i >> 3) + 1) * SM_PITCH + (i & 7) + 1;
  // the epilogue's operands (raw activation, accumulator T) do not depend on the matrix loop: fetch them first
  const int opix = (2 * wave + (lane >> 5)) * 8 + 4 * ((lane >> 4) & 1);
  float4 xpre[NT], tpre[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int ci = min((nt_base + t) * 16 + (lane & 15), d.Cin - 1);
    const size_t idx = ((size_t)b * d.x_ctot + ci) * 64 + opix;
    xpre[t] = *reinterpret_cast<const float4*>(d.x + idx);
    tpre[t] = d.t_accumulate ? *reinterpret_cast<const float4*>(d.t_in + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  v4f acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) acc[n] = (v4f){0.f, 0.f, 0.f, 0.f};
  for (int ks = 0; ks < ksb; ++ks) {
    const float* ap = pl + min(4 * ks + kq, Cout - 1) * SM_CS + aoff;
    {
      const float* wp = wm + ((size_t)min(ks + 1, ksb - 1) * 9 * ntp + nt_base) * 64 + lane;
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) wn[t][n] = wp[((size_t)t * ntp + n) * 64];
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float a = ap[(t / 3 - 1) * SM_PITCH + (t % 3 - 1)];
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w[t][n], acc[n], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int n = 0; n < NT; ++n) w[t][n] = wn[t][n];
  }
  // epilogue: ReLU mask, gamma, T (+)=, dgamma / dbeta, sums of the channels whose T is complete
  const int n = lane & 15;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int ci = (nt_base + t) * 16 + n;
    float dg = 0.f, db = 0.f, st = 0.f, sx = 0.f;
    if (ci < d.Cin && !d.has_bn) {             // plain gradient of an input that is read as it is
      const size_t idx = ((size_t)b * d.x_ctot + ci) * 64 + opix;
      *reinterpret_cast<float4*>(d.t_in + idx) = make_float4(tpre[t].x + acc[t][0], tpre[t].y + acc[t][1],
                                                             tpre[t].z + acc[t][2], tpre[t].w + acc[t][3]);
    } else if (ci < d.Cin) {
      const float4 k = cf[t * 16 + n];
      const size_t idx = ((size_t)b * d.x_ctot + ci) * 64 + opix;
      const float4 xv = xpre[t];
      float4* tp = reinterpret_cast<float4*>(d.t_in + idx);
      float4 tv = tpre[t];
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
      float to[4] = {tv.x, tv.y, tv.z, tv.w};
      const bool fin = ci >= d.final_c0 && ci < d.final_c1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float y = (xs[r] - k.x) * (k.z * k.y) + k.w;        // the forward's expression
        const float xh = (xs[r] - k.x) * k.y;
        const float dyv = y > 0.f ? acc[t][r] : 0.f;
        db += dyv;
        dg += dyv * xh;
        to[r] += k.z * dyv;
        if (fin) { st += to[r]; sx += to[r] * xh; }
      }
      *tp = make_float4(to[0], to[1], to[2], to[3]);
    }
    dg += __shfl_xor(dg, 16, 64); dg += __shfl_xor(dg, 32, 64);
    db += __shfl_xor(db, 16, 64); db += __shfl_xor(db, 32, 64);
    st += __shfl_xor(st, 16, 64); st += __shfl_xor(st, 32, 64);
    sx += __shfl_xor(sx, 16, 64); sx += __shfl_xor(sx, 32, 64);
    if (lane < 16) {
      float* r = red + ((wave * NT + t) * 16 + n) * 4;
      r[0] = dg; r[1] = db; r[2] = st; r[3] = sx;
    }
  }
  if (!d.has_bn) return;
  __syncthreads();
  for (int e = tid; e < NT * 64; e += 256) {
    const int t = e >> 6, nn = (e >> 2) & 15, q = e & 3;
    const int ci = (nt_base + t) * 16 + nn;
    if (ci >= d.Cin) continue;
    double v = 0.0;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) v += (double)red[((wv * NT + t) * 16 + nn) * 4 + q];
    const long long rep = (long long)rep_of_block(d.nrep) * d.rep_stride;
    if (q < 2) atomicAdd(&d.bn_grad[rep + 2 * ci + q], v);
    else if (ci >= d.final_c0 && ci < d.final_c1) atomicAdd(&d.t_stats[rep + 2 * ci + (q - 2)], v);
  }
}

// ------------------------------------------------------------------------------------------------- weight gradient
// grid (ceil(Cin / 16), nsplit), 256 threads.  dynamic LDS: per wave z planes [16][SM_CS] + g planes [NTC * 16][SM_GS];
// shared: coefficients [16] float4, sums [NTC * 16][16][9]
template <int NTC, bool GF = false>
__global__ __launch_bounds__(256) void conv_small_wgrad_kernel(pdes_conv_desc d, float* __restrict__ part, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) float sm_small[];
  __shared__ float4 fin_c[GF ? 16 : 1];
  __shared__ double fin_s[GF ? 16 : 1][4];
  constexpr int WREG = 16 * SM_CS + NTC * 16 * SM_GS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ci0 = blockIdx.x * 16, split = blockIdx.y;
  const int b0 = (int)(((long long)d.B * split) / nsplit), b1 = (int)(((long long)d.B * (split + 1)) / nsplit);
  float4* cf = reinterpret_cast<float4*>(sm_small);
  float* sums = sm_small + 64;                          // [NTC*16 co][16 ci][9]
  float* zpl = sums + NTC * 16 * 16 * 9 + wave * WREG;
  float* gpl = zpl + 16 * SM_CS;
  if (tid < 16 && d.has_bn) {
    const BnS k = bn_coef_s(d, min(ci0 + tid, d.Cin - 1));
    cf[tid] = make_float4(k.mean, k.gamma * k.invstd, k.beta, 0.f);
  }
  for (int e = tid; e < NTC * 16 * 16 * 9; e += 256) sums[e] = 0.f;
  for (int e = lane; e < 16 * SM_CS; e += 64) zpl[e] = 0.f;
  if (GF) fin_table_small(d, fin_c, fin_s, tid, 256);
  __syncthreads();
  v4f acc[9][NTC];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int n = 0; n < NTC; ++n) acc[t][n] = (v4f){0.f, 0.f, 0.f, 0.f};
  const int i = lane & 15, kq = lane >> 4;
  const int niter = (b1 - b0 + 3) >> 2;
  for (int it = 0; it < niter; ++it) {
    const int b = b0 + it * 4 + wave;
    const bool valid = b < b1;
    const float* xb = d.x + ((size_t)(valid ? b : b0) * d.x_ctot + ci0) * 64;
    const float* gb = d.g + ((size_t)(valid ? b : b0) * d.g_ctot + d.g_coff) * 64;
    for (int e = lane; e < 16 * 64; e += 64) {
      const int cl = e >> 6, p = e & 63;
      float z = 0.f;
      if (valid && ci0 + cl < d.Cin) {
        z = xb[e];
        if (d.has_bn) {
          const float4 k = cf[cl];
          z = fmaxf(0.f, (z - k.x) * k.y + k.z);
        }
      }
      zpl[cl * SM_CS + ((p >> 3) + 1) * SM_PITCH + (p & 7) + 1] = z;
    }
    const float* ob = GF ? d.out + ((size_t)(valid ? b : b0) * d.out_ctot + d.out_coff) * 64 : nullptr;
    for (int e = lane; e < NTC * 16 * 64; e += 64) {
      const int co = e >> 6, p = e & 63;
      float gvv = 0.f;
      if (valid && co < d.Cout) {
        gvv = gb[e];
        if (GF) gvv = fin_apply(fin_c[co], gvv, ob[e]);
      }
      gpl[co * SM_GS + p] = gvv;
    }
    __syncthreads();
#pragma unroll 2
    for (int ks = 0; ks < 16; ++ks) {
      const int p = 4 * ks + kq;
      const float* ap = zpl + i * SM_CS + (p >> 3) * SM_PITCH + (p & 7);       // tap (ky, kx) adds ky * pitch + kx
      float bv[NTC];
#pragma unroll
      for (int n = 0; n < NTC; ++n) bv[n] = gpl[(n * 16 + i) * SM_GS + p];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float a = ap[(t / 3) * SM_PITCH + (t % 3)];
#pragma unroll
        for (int n = 0; n < NTC; ++n) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[n], acc[t][n], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // lane (co = lane & 15, rows ci = 4 (lane >> 4) + r): the four waves add in a fixed order
  const int g = lane >> 4, n = lane & 15;
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) sums[((nt * 16 + n) * 16 + 4 * g + r) * 9 + t] += acc[t][nt][r];
    }
    __syncthreads();
  }
  float* out = part + (size_t)split * d.Cout * d.Cin * 9;
  for (int e = tid; e < NTC * 16 * 16 * 9; e += 256) {
    const int t = e % 9, cl = (e / 9) & 15, co = e / 144;
    if (co < d.Cout && ci0 + cl < d.Cin) out[((size_t)co * d.Cin + ci0 + cl) * 9 + t] = sums[e];
  }
}

__global__ __launch_bounds__(64) void small_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int n, int nsplit) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < nsplit; ++k) s += part[(size_t)k * n + i];
  dw[i] += s;
}

int wgrad_small_splits(const pdes_conv_desc& d) { return (d.B + 3) / 4; }       // four images (one per wave) per workgroup

int conv_forward_small(const pdes_conv_desc& d, hipStream_t st, bool dry) {       // dry: capability query only
  const bool s2 = small_fwd_s2_applies(d);
  if ((!conv_small_applies(d) && !s2) || !d.wm_fwd) return PDES_ENOSUP;
  if (dry) return PDES_OK;
  if (!d.x || !d.out) return PDES_EINVAL;
  if (d.has_bn && (!d.gamma || !d.beta || (d.eval_mode ? (!d.run_mean || !d.run_var) : !d.x_stats))) return PDES_EINVAL;
  if (!aligned16(d.x) || !aligned16(d.out)) return PDES_EALIGN;
  const int nt_total = (d.Cout + 15) / 16;
  const int cs = s2 ? ((18 * 18) | 1) : SM_CS;
  // one N-tile per workgroup: the staging of a tile is duplicated, but twice the CUs work on the layer
  const size_t lds = (size_t)(4 * d.Cin + d.Cin * cs + 4 * 16 * 2) * sizeof(float);
  if (s2) hipLaunchKernelGGL((conv_small_fwd_kernel<1, 2, 1>), dim3(d.B, nt_total), dim3(256), lds, st, d, d.wm_fwd, nt_total);
  // K-split wave groups (PDES_MFMA_SMALL=2: none, =3: at most two): stand-alone 40.8 -> 23.0 us (two groups) for the
  // 176 -> 24 layer, 24.0 -> 14.6 us for 84 -> 16
  else if (d.Cin >= (opt().mfma_small == 4 ? 128 : 96) && opt().mfma_small != 2 && opt().mfma_small != 3)
    hipLaunchKernelGGL((conv_small_fwd_kernel<1, 1, 4>), dim3(d.B, nt_total), dim3(1024), lds, st, d, d.wm_fwd, nt_total);
  else if (d.Cin >= (opt().mfma_small == 4 ? 64 : 32) && opt().mfma_small != 2)
    hipLaunchKernelGGL((conv_small_fwd_kernel<1, 1, 2>), dim3(d.B, nt_total), dim3(512), lds, st, d, d.wm_fwd, nt_total);
  else hipLaunchKernelGGL((conv_small_fwd_kernel<1, 1, 1>), dim3(d.B, nt_total), dim3(256), lds, st, d, d.wm_fwd, nt_total);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

int conv_backward_data_small(const pdes_conv_desc& d, hipStream_t st, bool dry) {
  if (!conv_small_applies(d) || !d.wm_bwd) return PDES_ENOSUP;
  if (dry) return PDES_OK;
  if (!d.g || !d.x || !d.t_in || d.eval_mode) return PDES_EINVAL;
  if (d.has_bn && (!d.bn_grad || !d.t_stats || !d.x_stats)) return PDES_EINVAL;
  if (!aligned16(d.x) || !aligned16(d.t_in)) return PDES_EALIGN;
  const int nt_total = (d.Cin + 15) / 16;
  constexpr int NT = 4;
  const size_t lds = (size_t)(((d.Cout * SM_CS + 3) & ~3) + NT * 16 * 4 + 4 * NT * 16 * 4) * sizeof(float);
  // (the fork of the next layer's weight gradient rides on this launch's completion signal when pdes_backward2 asks for it)
  hipEvent_t se = take_dgrad_stop_event();
  const dim3 grid(d.B, cdiv(nt_total, NT));
  if (d.g_fused && se)
    hipExtLaunchKernelGGL((conv_small_bwd_kernel<NT, true>), grid, dim3(256), lds, st, nullptr, se, 0, d, d.wm_bwd, nt_total);
  else if (d.g_fused) hipLaunchKernelGGL((conv_small_bwd_kernel<NT, true>), grid, dim3(256), lds, st, d, d.wm_bwd, nt_total);
  else if (se)
    hipExtLaunchKernelGGL((conv_small_bwd_kernel<NT, false>), grid, dim3(256), lds, st, nullptr, se, 0, d, d.wm_bwd, nt_total);
  else hipLaunchKernelGGL((conv_small_bwd_kernel<NT, false>), grid, dim3(256), lds, st, d, d.wm_bwd, nt_total);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

int conv_backward_weight_small(const pdes_conv_desc& d, hipStream_t st) {
  if (!conv_small_applies(d) || d.Cout > 48 || !d.ws) return PDES_ENOSUP;
  const int nsplit = wgrad_small_splits(d);
  const long long per = (long long)d.Cout * d.Cin * 9;
  if (nsplit * per * 4 > d.ws_bytes) return PDES_ENOSUP;
  if (!d.g || !d.x || !d.dw || (d.has_bn && !d.x_stats)) return PDES_EINVAL;
  const int ntc = (d.Cout + 15) / 16;
  const size_t lds = (size_t)(64 + ntc * 16 * 16 * 9 + 4 * (16 * SM_CS + ntc * 16 * SM_GS)) * sizeof(float);
  dim3 grid(cdiv(d.Cin, 16), nsplit);
  if (d.g_fused) hipLaunchKernelGGL((conv_small_wgrad_kernel<1, true>), grid, dim3(256), lds, st, d, d.ws, nsplit);      // (<= 16 output channels)
  else if (ntc == 1) hipLaunchKernelGGL(conv_small_wgrad_kernel<1>, grid, dim3(256), lds, st, d, d.ws, nsplit);
  else if (ntc == 2) hipLaunchKernelGGL(conv_small_wgrad_kernel<2>, grid, dim3(256), lds, st, d, d.ws, nsplit);
  else hipLaunchKernelGGL(conv_small_wgrad_kernel<3>, grid, dim3(256), lds, st, d, d.ws, nsplit);
  PDES_LAUNCH_CHECK();
  if (!d.ws_defer) {
    hipLaunchKernelGGL(small_reduce_kernel, dim3(cdiv((int)per, 64)), dim3(64), 0, st, d.ws, d.dw, (int)per, nsplit);
    PDES_LAUNCH_CHECK();
  }
  return PDES_OK;
}

bool wgrad_small_applies(const pdes_conv_desc& d) { return conv_small_applies(d) && d.Cout <= 48; }
// would conv_backward_weight_small take `d` as it is (scratch included)?
bool wgrad_small_ready(const pdes_conv_desc& d) {
  if (!wgrad_small_applies(d) || !d.ws) return false;
  return (long long)wgrad_small_splits(d) * d.Cout * d.Cin * 9 * 4 <= d.ws_bytes;
}

}  // namespace pdes
