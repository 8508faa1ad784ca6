// Generic direct convolutions with fused BatchNorm+ReLU(+nearest x2) operand transform for
// DenseED / Decoder (reference models/codec.py:43-188), gfx950 VALU path.
//
// These kernels cover EVERY convolution shape of the network (k = 1,3,5,7; stride 1,2; fused
// nearest upsampling; any Cin/Cout) and are the reference HIP path the MFMA kernels
// (conv_mfma.hip) are tested against on the GPU; the dispatcher in conv_dispatch.hip prefers the
// MFMA kernels where they apply.
//
// Mapping: one thread = one output pixel x COT output channels kept in registers.  Weights are
// read through the scalar cache (s_load_dwordx16 from the packed (Cin, k*k, cout_pad) copy: the
// address is wave-uniform), so the inner loop is 1 vector load + COT v_fmac per (ci, tap).
// Activations stay NCHW; lanes of a wave are consecutive pixels of one channel plane -> coalesced.
// BatchNorm coefficients (mean, gamma*invstd, beta) are derived per workgroup from the fp64
// {sum, sum^2} the producer accumulated, so no separate BN kernel or normalised copy exists.
#include "pdes_common.h"
#include "../../include/pdes_hip.h"

namespace pdes {

typedef const float __attribute__((address_space(4)))* kfloatp;   // scalar-cache (constant) loads

// BN coefficients of channel c of the input buffer.
struct BnC { float mean, invstd, gamma, beta; };

__device__ __forceinline__ BnC bn_coef(const pdes_conv_desc& d, int c) {
  BnC o;
  if (!d.has_bn) { o.mean = 0.f; o.invstd = 1.f; o.gamma = 1.f; o.beta = 0.f; return o; }
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

// ------------------------------------------------------------------------------------------ fwd
// grid: (ceil(Hout*Wout/256), ceil(Cout/COT), B), block 256.  dyn LDS: 3*Cin floats + 4*COT*2 doubles
template <int COT>
__global__ __launch_bounds__(256) void conv_fwd_direct(pdes_conv_desc d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* red = reinterpret_cast<double*>(smem_raw);                 // [4 waves][COT][2]
  float* cf = reinterpret_cast<float*>(smem_raw + 4 * COT * 2 * sizeof(double));  // [Cin][3]
  const int tid = threadIdx.x;
  for (int c = tid; c < d.Cin; c += 256) {
    const BnC k = bn_coef(d, c);
    cf[3 * c + 0] = k.mean;
    cf[3 * c + 1] = k.gamma * k.invstd;
    cf[3 * c + 2] = k.beta;
  }
  __syncthreads();

  const int HWo = d.Hout * d.Wout, HWi = d.Hin * d.Win;
  const int p = blockIdx.x * 256 + tid;
  const int co0 = blockIdx.y * COT;
  const int b = blockIdx.z;
  const bool active = p < HWo;
  const int oy = active ? p / d.Wout : 0, ox = active ? p % d.Wout : 0;
  const int Hc = d.upsample ? 2 * d.Hin : d.Hin, Wc = d.upsample ? 2 * d.Win : d.Win;
  const int k = d.ksize, KK = k * k;
  const float* xb = d.x + (size_t)b * d.x_ctot * HWi;
  float acc[COT];
#pragma unroll
  for (int j = 0; j < COT; ++j) acc[j] = 0.f;

  for (int ci = 0; ci < d.Cin; ++ci) {
    const float mean = cf[3 * ci], scale = cf[3 * ci + 1], beta = cf[3 * ci + 2];
    const float* xc = xb + (size_t)ci * HWi;
    const kfloatp wrow = (kfloatp)(d.w_fwd + (size_t)ci * KK * d.cout_pad + co0);
    for (int ky = 0; ky < k; ++ky) {
      const int cy = oy * d.stride + ky - d.pad;
      const bool vy = active && cy >= 0 && cy < Hc;
      const int sy = d.upsample ? (cy >> 1) : cy;
      for (int kx = 0; kx < k; ++kx) {
        const int cx = ox * d.stride + kx - d.pad;
        const bool v = vy && cx >= 0 && cx < Wc;
        const int sx = d.upsample ? (cx >> 1) : cx;
        float z = 0.f;
        if (v) {
          const float x = xc[sy * d.Win + sx];
          z = d.has_bn ? fmaxf(0.f, (x - mean) * scale + beta) : x;
        }
        const kfloatp wp = wrow + (ky * k + kx) * d.cout_pad;
#pragma unroll
        for (int j = 0; j < COT; ++j) acc[j] += z * wp[j];
      }
    }
  }
  float* ob = d.out + ((size_t)b * d.out_ctot + d.out_coff + co0) * HWo;
  if (active) {
#pragma unroll
    for (int j = 0; j < COT; ++j)
      if (co0 + j < d.Cout) ob[(size_t)j * HWo + p] = acc[j];
  }
  if (d.out_stats) {
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int j = 0; j < COT; ++j) {
      const float v = active ? acc[j] : 0.f;
      const float s = wave_sum(v), q = wave_sum(v * v);
      if (lane == 0) { red[(wave * COT + j) * 2] = (double)s; red[(wave * COT + j) * 2 + 1] = (double)q; }
    }
    __syncthreads();
    if (tid < COT * 2) {
      const int j = tid >> 1, w = tid & 1;
      if (co0 + j < d.Cout) {
        double t = 0.0;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) t += red[(wv * COT + j) * 2 + w];
        atomicAdd(&d.out_stats[(long long)rep_of_block(d.nrep) * d.rep_stride + 2 * (d.out_coff + co0 + j) + w], t);
      }
    }
  }
}

// ------------------------------------------------------------------------------------ bwd data
// one thread = one INPUT pixel (low-res pixel when upsample) x CIT input channels.
// grid: (ceil(Hin*Win/256), ceil(Cin/CIT), B).  dyn LDS: 4 waves*CIT*4 doubles + 4*Cin floats
template <int CIT>
__global__ __launch_bounds__(256) void conv_bwd_data_direct(pdes_conv_desc d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* red = reinterpret_cast<double*>(smem_raw);                 // [4][CIT][4]
  float* cf = reinterpret_cast<float*>(smem_raw + 4 * CIT * 4 * sizeof(double));  // [Cin][4]
  const int tid = threadIdx.x;
  const int ci0 = blockIdx.y * CIT;
  for (int c = tid; c < CIT; c += 256) {
    if (ci0 + c < d.Cin) {
      const BnC k = bn_coef(d, ci0 + c);
      cf[4 * c + 0] = k.mean; cf[4 * c + 1] = k.invstd; cf[4 * c + 2] = k.gamma; cf[4 * c + 3] = k.beta;
    }
  }
  __syncthreads();

  const int HWo = d.Hout * d.Wout, HWi = d.Hin * d.Win;
  const int p = blockIdx.x * 256 + tid;
  const int b = blockIdx.z;
  const bool active = p < HWi;
  const int iy = active ? p / d.Win : 0, ix = active ? p % d.Win : 0;
  const int k = d.ksize, KK = k * k;
  const int nsub = d.upsample ? 2 : 1;
  const float* gb = d.g + ((size_t)b * d.g_ctot + d.g_coff) * HWo;
  float acc[CIT];
#pragma unroll
  for (int j = 0; j < CIT; ++j) acc[j] = 0.f;

  for (int co = 0; co < d.Cout; ++co) {
    const float* gc = gb + (size_t)co * HWo;
    const kfloatp wrow = (kfloatp)(d.w_bwd + (size_t)co * KK * d.cin_pad + ci0);
    for (int dy = 0; dy < nsub; ++dy) {
      const int cy = d.upsample ? 2 * iy + dy : iy;          // conv-input coordinate
      for (int ky = 0; ky < k; ++ky) {
        const int ty = cy + d.pad - ky;
        if (ty < 0) continue;                                 // wave-divergent but cheap
        const int oy = ty / d.stride;
        const bool vy = active && (ty - oy * d.stride == 0) && oy < d.Hout;
        for (int dx = 0; dx < nsub; ++dx) {
          const int cx = d.upsample ? 2 * ix + dx : ix;
          for (int kx = 0; kx < k; ++kx) {
            const int tx = cx + d.pad - kx;
            const int ox = tx / d.stride;
            const bool v = vy && tx >= 0 && (tx - ox * d.stride == 0) && ox < d.Wout;
            const float g = v ? gc[oy * d.Wout + ox] : 0.f;
            const kfloatp wp = wrow + (ky * k + kx) * d.cin_pad;
#pragma unroll
            for (int j = 0; j < CIT; ++j) acc[j] += g * wp[j];
          }
        }
      }
    }
  }

  // epilogue: ReLU mask, gamma, T (+)=, dgamma/dbeta, finished-channel sums
  const float* xb = d.x + (size_t)b * d.x_ctot * HWi;
  float* tb = d.t_in + (size_t)b * d.x_ctot * HWi;
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int j = 0; j < CIT; ++j) {
    const int ci = ci0 + j;
    float dg = 0.f, db = 0.f, st = 0.f, sx = 0.f;
    if (ci < d.Cin && active) {
      const float mean = cf[4 * j], invstd = cf[4 * j + 1], gamma = cf[4 * j + 2], beta = cf[4 * j + 3];
      const size_t idx = (size_t)ci * HWi + p;
      if (!d.has_bn) {
        // a convolution that reads its input as is (glow_msc.py Conv2dZeros on a latent / on the encoder's features):
        // t_in is the plain gradient of the input, no mask, no BatchNorm sums
        tb[idx] = d.t_accumulate ? tb[idx] + acc[j] : acc[j];
      } else {
        const float x = xb[idx];
        const float y = (x - mean) * (gamma * invstd) + beta;     // same expression as the forward
        const float xh = (x - mean) * invstd;
        const float dyv = (y > 0.f) ? acc[j] : 0.f;
        db = dyv;
        dg = dyv * xh;
        float t = gamma * dyv;
        if (d.t_accumulate) t += tb[idx];
        tb[idx] = t;
        if (ci >= d.final_c0 && ci < d.final_c1) { st = t; sx = t * xh; }
      }
    }
    if (!d.has_bn) continue;
    const float r0 = wave_sum(dg), r1 = wave_sum(db), r2 = wave_sum(st), r3 = wave_sum(sx);
    if (lane == 0) {
      double* r = &red[(wave * CIT + j) * 4];
      r[0] = r0; r[1] = r1; r[2] = r2; r[3] = r3;
    }
  }
  __syncthreads();
  if (d.has_bn && tid < CIT * 4) {
    const int j = tid >> 2, q = tid & 3, ci = ci0 + j;
    if (ci < d.Cin) {
      double t = 0.0;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) t += red[(wv * CIT + j) * 4 + q];
      if (q < 2) atomicAdd(&d.bn_grad[(long long)rep_of_block(d.nrep) * d.rep_stride + 2 * ci + q], t);
      else if (ci >= d.final_c0 && ci < d.final_c1) atomicAdd(&d.t_stats[(long long)rep_of_block(d.nrep) * d.rep_stride + 2 * ci + (q - 2)], t);
    }
  }
}

// ---------------------------------------------------------------------------------- bwd weight
// one workgroup = one input channel x COT output channels x a slice of the batch; every thread
// accumulates its pixels' COT x KS*KS products, then wave/LDS reduce and fp32 atomicAdd into dw.
// grid: (Cin, ceil(Cout/COT), nsplit), block 256.
template <int KS, int COT>
__global__ __launch_bounds__(256) void conv_bwd_weight_direct(pdes_conv_desc d, int nsplit) {
  constexpr int KK = KS * KS;
  __shared__ float red[4][COT * KK];
  const int tid = threadIdx.x;
  const int ci = blockIdx.x, co0 = blockIdx.y * COT;
  const int b0 = (int)(((long long)d.B * blockIdx.z) / nsplit), b1 = (int)(((long long)d.B * (blockIdx.z + 1)) / nsplit);
  const BnC bc = bn_coef(d, ci);
  const float mean = bc.mean, scale = bc.gamma * bc.invstd, beta = bc.beta;
  const int HWo = d.Hout * d.Wout, HWi = d.Hin * d.Win;
  const int Hc = d.upsample ? 2 * d.Hin : d.Hin, Wc = d.upsample ? 2 * d.Win : d.Win;
  float acc[COT][KK];
#pragma unroll
  for (int j = 0; j < COT; ++j)
#pragma unroll
    for (int t = 0; t < KK; ++t) acc[j][t] = 0.f;

  for (int b = b0; b < b1; ++b) {
    const float* xc = d.x + ((size_t)b * d.x_ctot + ci) * HWi;
    const float* gb = d.g + ((size_t)b * d.g_ctot + d.g_coff + co0) * HWo;
    for (int p = tid; p < HWo; p += 256) {
      const int oy = p / d.Wout, ox = p % d.Wout;
      float z[KK];
#pragma unroll
      for (int ky = 0; ky < KS; ++ky) {
        const int cy = oy * d.stride + ky - d.pad;
        const bool vy = cy >= 0 && cy < Hc;
        const int sy = d.upsample ? (cy >> 1) : cy;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          const int cx = ox * d.stride + kx - d.pad;
          const bool v = vy && cx >= 0 && cx < Wc;
          const int sx = d.upsample ? (cx >> 1) : cx;
          float zz = 0.f;
          if (v) {
            const float x = xc[sy * d.Win + sx];
            zz = d.has_bn ? fmaxf(0.f, (x - mean) * scale + beta) : x;
          }
          z[ky * KS + kx] = zz;
        }
      }
#pragma unroll
      for (int j = 0; j < COT; ++j) {
        const float g = (co0 + j < d.Cout) ? gb[(size_t)j * HWo + p] : 0.f;
#pragma unroll
        for (int t = 0; t < KK; ++t) acc[j][t] += g * z[t];
      }
    }
  }
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int j = 0; j < COT; ++j)
#pragma unroll
    for (int t = 0; t < KK; ++t) {
      const float s = wave_sum(acc[j][t]);
      if (lane == 0) red[wave][j * KK + t] = s;
    }
  __syncthreads();
  for (int i = tid; i < COT * KK; i += 256) {
    const int j = i / KK, t = i % KK;
    if (co0 + j < d.Cout) {
      const float s = red[0][i] + red[1][i] + red[2][i] + red[3][i];
      atomicAdd(&d.dw[((size_t)(co0 + j) * d.Cin + ci) * KK + t], s);
    }
  }
}


// ------------------------------------------------------------------------- fwd, first convolution
// 7x7 stride-2 convolution of a 1-channel field (reference models/codec.py:236-238, no BatchNorm in front):
// out[co][oy][ox] = sum_{ky,kx} x[2 oy + ky - 3][2 ox + kx - 3] w[co][ky][kx].  One workgroup = 4 output rows
// of one sample, all output channels: the 13 zero-bordered input rows and the transposed weights [tap][co] sit
// in LDS; a thread owns 4 consecutive output pixels x 8 channels (32 accumulators), reads per kernel row the 14
// input values they touch as aligned float4 and the 8 channel weights of a tap as two float4 broadcasts.
// grid (Hout/4, B), block = 32 pixel groups x Cout/8 channel groups (<= 256 threads).
//
// PAIR (round 5): one workgroup = TWO groups of 4 output rows x HALF the channels (grid (Hout/8, 2, B): the same number of
// workgroups, threads and accumulators).  A half wave is what a (row group, channel group) was before -- 32 pixel groups,
// the same shuffle tree -- so every fp32 partial sum of the statistics is bit-identical to the unpaired form; the two row
// groups of a channel (the two halves of a wave) are added in fp64, exactly, in front of ONE atomic instead of two:
// 12.3 k instead of 24.5 k simultaneous fp64 atomics at the head of every step (15.2 -> 12.2 us stand-alone).  (Summing the
// eight rows in fp32 instead moved the first BatchNorms' statistics in their last bits and with them which marginal ReLUs
// flip in the small-batch fixtures: EXPERIMENTS.md round 5.)
template <bool PAIR>
__global__ __launch_bounds__(256) void conv_fwd_first7(pdes_conv_desc d) {
  constexpr int K = 7, RB = PAIR ? 8 : 4, XR = 2 * RB + K - 2;    // 13 (21) input rows
  extern __shared__ __attribute__((aligned(16))) float smq[];
  const int LW = ((d.Win + 2 * 3 + 2 + 3) / 4) * 4;                // bordered row, 16-B pitch (72 for 64 columns)
  const int CW = PAIR ? d.Cout / 2 : d.Cout, c_lo = PAIR ? (int)blockIdx.y * CW : 0;      // this workgroup's channels
  float* xs = smq;                                                 // [XR][LW]
  float* wt = smq + XR * LW;                                       // [49][CW]
  const int tid = threadIdx.x, b = PAIR ? blockIdx.z : blockIdx.y, oy0 = blockIdx.x * RB;
  const int nthr = (PAIR ? 64 : 32) * (CW / 8);
  const float* xin = d.x + (size_t)b * d.x_ctot * d.Hin * d.Win;
  // staging in two unrolled batches (all global loads of a batch are in flight together)
  {
    constexpr int NX = PAIR ? 6 : 4;                  // XR * LW <= 21 * 72 <= 6 * 256  (13 * 76 < 4 * 256)
    float v[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int i = tid + 256 * k, r = i / LW, q = i % LW;
      const int yy = 2 * oy0 - 3 + r, xx = q - 3;
      const bool ok = i < XR * LW && yy >= 0 && yy < d.Hin && xx >= 0 && xx < d.Win;
      v[k] = ok ? xin[yy * d.Win + xx] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < NX; ++k) { const int i = tid + 256 * k; if (i < XR * LW) xs[i] = v[k]; }
    float w[13];
    const float* wsrc = d.w + (size_t)c_lo * 49;      // (Cout, 1, 7, 7): a channel range is one contiguous block
#pragma unroll
    for (int k = 0; k < 13; ++k) {                    // 49 * CW <= 49 * 64 < 13 * 256; coalesced reads
      const int i = tid + 256 * k;
      w[k] = i < 49 * CW ? wsrc[i] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 13; ++k) { const int i = tid + 256 * k; if (i < 49 * CW) wt[(i % 49) * CW + i / 49] = w[k]; }
  }
  __syncthreads();
  if (tid >= nthr) return;
  // PAIR: lanes 0-31 of a wave = row group 0, lanes 32-63 = row group 1 of the same channel group
  const int pg = tid & 31, cg = PAIR ? tid >> 6 : tid >> 5, rg = PAIR ? (tid >> 5) & 1 : 0;
  const int row = 4 * rg + (pg >> 3), ox0 = 4 * (pg & 7);
  float acc[4][8];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[u][j] = 0.f;
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    const float* xr = xs + (2 * row + ky) * LW + 2 * ox0;
    float xv[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 t = *reinterpret_cast<const float4*>(xr + 4 * j);
      xv[4 * j] = t.x; xv[4 * j + 1] = t.y; xv[4 * j + 2] = t.z; xv[4 * j + 3] = t.w;
    }
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const float* wp = wt + (ky * K + kx) * CW + 8 * cg;
      const float4 w0 = *reinterpret_cast<const float4*>(wp), w1 = *reinterpret_cast<const float4*>(wp + 4);
      const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[u][j] += xv[2 * u + kx] * wv[j];
    }
  }
  const int HWo = d.Hout * d.Wout, c0 = c_lo + 8 * cg;
  float* ob = d.out + ((size_t)b * d.out_ctot + d.out_coff + c0) * HWo + (size_t)(oy0 + row) * d.Wout + ox0;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    *reinterpret_cast<float4*>(ob + (size_t)j * HWo) = make_float4(acc[0][j], acc[1][j], acc[2][j], acc[3][j]);
  if (d.out_stats) {
    double* os = d.out_stats + (long long)rep_of_block(d.nrep) * d.rep_stride;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float s = (acc[0][j] + acc[1][j]) + (acc[2][j] + acc[3][j]);
      float q = (acc[0][j] * acc[0][j] + acc[1][j] * acc[1][j]) + (acc[2][j] * acc[2][j] + acc[3][j] * acc[3][j]);
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) { s += __shfl_xor(s, o, 32); q += __shfl_xor(q, o, 32); }
      double sd = (double)s, qd = (double)q;
      if (PAIR) { sd += __shfl_xor(sd, 32, 64); qd += __shfl_xor(qd, 32, 64); }      // the other row group: exact in fp64
      if (pg == 0 && rg == 0) {
        atomicAdd(&os[2 * (d.out_coff + c0 + j)], sd);
        atomicAdd(&os[2 * (d.out_coff + c0 + j) + 1], qd);
      }
    }
  }
}

// ---------------------------------------------------------------- bwd weight, first convolution
// The first convolution (no BatchNorm in front, 1..4 input channels, e.g. 7x7 stride 2 on the
// permeability field) has a tiny weight tensor but a 49-tap reduction over every output pixel.
// One workgroup = one sample x COT output channels: the zero-bordered input plane(s) and the COT gradient
// planes sit in LDS.  A thread owns (channel, kernel row ky, row group) and keeps the k taps of that kernel row
// in registers: per 4 output pixels it reads one float4 of the gradient and the 4*stride + k - 1 input values
// they touch (aligned float4 LDS reads), i.e. ~5 LDS reads per 28 FMAs for the 7x7 stride-2 first layer.
// The row groups of one (channel, ky) are adjacent lanes and are combined with two shuffles.
// part != nullptr: this image's contribution is STORED to part[b][(co, ci, ky, kx)] (every element has exactly one
// writer) for the fixed-order split-K reduce -- deterministic; nullptr: fp32 atomics straight into dw.
// GF (pdes_conv_desc.g_fused): `g` still holds the accumulator T of this layer's output buffer; the BatchNorm-backward
// finalize  dL/dx = invstd (T - mean(T) - xhat mean(T xhat))  is applied while the gradient planes are staged (raw
// activation = `out`, statistics as in bn_bwd_finalize_kernel) -- the first layer has no data gradient, so nothing else
// reads its finalized gradient and the finalize launch in front of this kernel disappears from the end of the chain.
template <int COT, int K, int S, bool GF>
__global__ __launch_bounds__(256) void conv_bwd_weight_first(pdes_conv_desc d, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float smf[];
  __shared__ float4 fin_c[COT];                          // GF: {mean, invstd, mean T, mean T xhat} of the workgroup's channels
  constexpr int KK = K * K;
  constexpr int RG = 4;                                 // row groups per (channel, ky)
  constexpr int NX = 4 * S + K - 1;                     // input values touched by 4 consecutive outputs
  constexpr int NX4 = (NX + 3) / 4;
  const int HWo = d.Hout * d.Wout, HWi = d.Hin * d.Win;
  const int LH = d.Hin + 2 * d.pad + S, LW = ((d.Win + 2 * d.pad + S + 4 + 3) / 4) * 4;   // bordered plane, 16-B rows
  float* xs = smf;                                   // [Cin][LH][LW]
  float* gs = smf + d.Cin * LH * LW;                 // [COT][HWo]
  const int tid = threadIdx.x, b = blockIdx.x, co0 = blockIdx.y * COT;
  // staging: global loads are issued in batches of 4 (x) / 8 (g) float4 per thread before any LDS write, so the
  // workgroup pays the memory latency once per batch instead of once per element
  for (int i = tid; i < d.Cin * LH * LW / 4; i += 256) reinterpret_cast<float4*>(xs)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (GF) {
    // 4 sums x PDES_NREP (<= 16) replicas per channel: one load per lane, 16-lane shuffle reduction (flow_copy_bwd_kernel)
    static_assert(COT * 64 <= 512, "two rounds of 256 threads");
    __shared__ double fin_s[COT][4];
    for (int e = tid; e < COT * 64; e += 256) {
      const int c = e >> 6, q = (e >> 4) & 3, r = e & 15;
      const int ch = d.g_coff + min(co0 + c, d.Cout - 1);
      double v = r < PDES_NREP ? (q < 2 ? d.fin_xstats : d.fin_tstats)[(long long)r * d.rep_stride + 2 * ch + (q & 1)] : 0.0;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 16);
      if (r == 0) fin_s[c][q] = v;
    }
    __syncthreads();
    if (tid < COT) {
      const double inv_n = 1.0 / ((double)d.B * HWo);
      const double m = fin_s[tid][0] * inv_n;
      double var = fin_s[tid][1] * inv_n - m * m;
      var = var < 0.0 ? 0.0 : var;
      fin_c[tid] = make_float4((float)m, (float)(1.0 / sqrt(var + (double)d.eps)), (float)(fin_s[tid][2] * inv_n),
                               (float)(fin_s[tid][3] * inv_n));
    }
  }
  __syncthreads();
  const int W4 = d.Win / 4, nx4 = d.Cin * d.Hin * W4;            // Win % 4 == 0 (checked on the host)
  for (int i0 = 0; i0 < nx4; i0 += 4 * 256) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = min(i0 + tid + 256 * u, nx4 - 1);
      const int cc = i / (d.Hin * W4), yy = (i / W4) % d.Hin, x4 = i % W4;
      v[u] = *reinterpret_cast<const float4*>(d.x + ((size_t)b * d.x_ctot + cc) * HWi + yy * d.Win + 4 * x4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + tid + 256 * u;
      if (i < nx4) {
        const int cc = i / (d.Hin * W4), yy = (i / W4) % d.Hin, x4 = i % W4;
        float* dst = xs + (cc * LH + yy + d.pad) * LW + d.pad + 4 * x4;
        dst[0] = v[u].x; dst[1] = v[u].y; dst[2] = v[u].z; dst[3] = v[u].w;
      }
    }
  }
  const int ng4 = COT * HWo / 4;
  for (int i0 = 0; i0 < ng4; i0 += 8 * 256) {
    float4 v[8], xa[GF ? 8 : 1];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = min(i0 + tid + 256 * u, ng4 - 1);
      const int cc = min((4 * i) / HWo, d.Cout - 1 - co0);
      const size_t off = ((size_t)b * d.g_ctot + d.g_coff + co0 + cc) * HWo + (4 * i) % HWo;
      v[u] = *reinterpret_cast<const float4*>(d.g + off);
      if (GF) xa[u] = *reinterpret_cast<const float4*>(d.out + off);      // (g_ctot == out_ctot, g_coff == out_coff: checked)
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + tid + 256 * u;
      if (i < ng4) {
        const bool ok = co0 + (4 * i) / HWo < d.Cout;
        float4 t = v[u];
        if (GF) {
          const float4 k = fin_c[min((4 * i) / HWo, COT - 1)];
          const float4 x = xa[u];
          t.x = k.y * (t.x - k.z - (x.x - k.x) * k.y * k.w);
          t.y = k.y * (t.y - k.z - (x.y - k.x) * k.y * k.w);
          t.z = k.y * (t.z - k.z - (x.z - k.x) * k.y * k.w);
          t.w = k.y * (t.w - k.z - (x.w - k.x) * k.y * k.w);
        }
        *reinterpret_cast<float4*>(gs + 4 * i) = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  __syncthreads();
  const int c = tid / (K * RG), ky = (tid / RG) % K, grp = tid % RG;
  const bool live = c < COT;
  const int rows = d.Hout / RG;                         // output rows per group (Hout % 4 == 0 checked on the host)
  for (int ci = 0; ci < d.Cin; ++ci) {
    float a[K];
#pragma unroll
    for (int kx = 0; kx < K; ++kx) a[kx] = 0.f;
    if (live) {
      const float* gp = gs + c * HWo;
      for (int oy = grp * rows; oy < (grp + 1) * rows; ++oy) {
        const float* xr = xs + (ci * LH + oy * S + ky) * LW;
        const float* gr = gp + oy * d.Wout;
        for (int ox = 0; ox < d.Wout; ox += 4) {
          const float4 gv = *reinterpret_cast<const float4*>(gr + ox);
          float xv[4 * NX4];
#pragma unroll
          for (int j = 0; j < NX4; ++j) {
            const float4 t = *reinterpret_cast<const float4*>(xr + ox * S + 4 * j);
            xv[4 * j] = t.x; xv[4 * j + 1] = t.y; xv[4 * j + 2] = t.z; xv[4 * j + 3] = t.w;
          }
          const float g4[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int kx = 0; kx < K; ++kx) a[kx] += g4[u] * xv[u * S + kx];
        }
      }
    }
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      float v = a[kx];
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      if (live && grp == 0 && co0 + c < d.Cout) {
        const size_t idx = ((size_t)(co0 + c) * d.Cin + ci) * KK + ky * K + kx;
        if (part) part[(size_t)b * d.Cout * d.Cin * KK + idx] = v;
        else atomicAdd(&d.dw[idx], v);
      }
    }
  }
}

// ------------------------------------------------------------------------------- host dispatch
// the first convolution's weight gradient through per-image partials + the fixed-order reduce (one split per image)
bool first_layer_shape(const pdes_conv_desc& d) {
  if (!(!d.has_bn && !d.upsample && d.Cin <= 4 && d.ksize == 7 && d.stride == 2 && d.Wout % 4 == 0 && d.Hout % 4 == 0 &&
        d.Win % 4 == 0)) return false;
  const int LH = d.Hin + 2 * d.pad + 2, LW = ((d.Win + 2 * d.pad + 2 + 4 + 3) / 4) * 4;
  return ((size_t)d.Cin * LH * LW + (size_t)8 * d.Hout * d.Wout) * sizeof(float) <= 150 * 1024;
}
bool first_layer_partials(const pdes_conv_desc& d) {
  return first_layer_shape(d) && d.ws && d.ws_defer && (long long)d.B * d.Cout * d.Cin * 49 * 4 <= d.ws_bytes;
}

static int validate(const pdes_conv_desc& d, int mode) {
  if (d.nrep != PDES_NREP) return PDES_EINVAL;
  if (d.B <= 0 || d.Cin <= 0 || d.Cout <= 0 || d.Hin <= 0 || d.Win <= 0 || d.Hout <= 0 || d.Wout <= 0) return PDES_EINVAL;
  if (!d.x) return PDES_EINVAL;
  if (!(d.ksize == 1 || d.ksize == 3 || d.ksize == 5 || d.ksize == 7)) return PDES_ENOSUP;
  if (d.stride < 1 || d.stride > 2 || (d.upsample && d.stride != 1)) return PDES_ENOSUP;
  if (d.has_bn && (!d.gamma || !d.beta || (d.eval_mode ? (!d.run_mean || !d.run_var) : !d.x_stats))) return PDES_EINVAL;
  const int Hc = d.upsample ? 2 * d.Hin : d.Hin, Wc = d.upsample ? 2 * d.Win : d.Win;
  if ((Hc + 2 * d.pad - d.ksize) / d.stride + 1 != d.Hout || (Wc + 2 * d.pad - d.ksize) / d.stride + 1 != d.Wout) return PDES_EINVAL;
  if (d.cout_pad % 16 || d.cin_pad % 16 || d.cout_pad < d.Cout || d.cin_pad < d.Cin) return PDES_EINVAL;
  if (mode == 0 && (!d.out || !d.w_fwd)) return PDES_EINVAL;
  if (mode == 1 && (!d.g || !d.dw)) return PDES_EINVAL;
  if (mode == 2 && (!d.g || !d.w_bwd || !d.t_in || d.eval_mode)) return PDES_EINVAL;
  if (mode == 2 && d.has_bn && (!d.bn_grad || !d.t_stats)) return PDES_EINVAL;
  return PDES_OK;
}

// the 7x7 first convolution reads the live weight tensor; every other VALU forward the packed image w_fwd
bool conv_forward_direct_first7(const pdes_conv_desc& d) {
  return !d.has_bn && !d.upsample && d.Cin == 1 && d.ksize == 7 && d.stride == 2 && d.pad == 3 && d.w && d.Wout == 32 &&
         d.Win == 64 && d.Hout % 4 == 0 && d.Hin == 2 * d.Hout && d.Cout % 8 == 0 && d.Cout <= 64;
}

int conv_forward_direct(const pdes_conv_desc& d, hipStream_t st) {
  const int rc = validate(d, 0);
  if (rc) return rc;
  if (conv_forward_direct_first7(d)) {
    const int LW = ((d.Win + 2 * 3 + 2 + 3) / 4) * 4;
    if (d.Hout % 8 == 0 && d.Cout % 16 == 0) {          // two row groups x half the channels per workgroup: half the atomics
      const size_t lds = ((size_t)21 * LW + (size_t)49 * (d.Cout / 2)) * sizeof(float);
      hipLaunchKernelGGL(conv_fwd_first7<true>, dim3(d.Hout / 8, 2, d.B), dim3(256), lds, st, d);
    } else {
      const size_t lds = ((size_t)13 * LW + (size_t)49 * d.Cout) * sizeof(float);
      hipLaunchKernelGGL(conv_fwd_first7<false>, dim3(d.Hout / 4, d.B), dim3(256), lds, st, d);
    }
    PDES_LAUNCH_CHECK();
    return PDES_OK;
  }
  const int HWo = d.Hout * d.Wout;
  // fewer channels per thread when that is needed to fill the chip (>= 2 waves per SIMD)
  const long long px_blocks = (long long)cdiv(HWo, 256) * d.B;
  int cot = 16;
  if (d.Cout <= 4) cot = 4;
  else if (px_blocks * cdiv(d.Cout, 16) < 512 && d.Cout >= 8) cot = 8;
  dim3 grid(cdiv(HWo, 256), cdiv(d.Cout, cot), d.B), block(256);
  const size_t lds = 4 * cot * 2 * sizeof(double) + (size_t)3 * d.Cin * sizeof(float);
  if (cot == 16) hipLaunchKernelGGL(conv_fwd_direct<16>, grid, block, lds, st, d);
  else if (cot == 8) hipLaunchKernelGGL(conv_fwd_direct<8>, grid, block, lds, st, d);
  else hipLaunchKernelGGL(conv_fwd_direct<4>, grid, block, lds, st, d);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

int conv_backward_data_direct(const pdes_conv_desc& d, hipStream_t st) {
  const int rc = validate(d, 2);
  if (rc) return rc;
  const int HWi = d.Hin * d.Win;
  const long long px_blocks = (long long)cdiv(HWi, 256) * d.B;
  int cit = 16;
  if (px_blocks * cdiv(d.Cin, 16) < 512) cit = 8;
  dim3 grid(cdiv(HWi, 256), cdiv(d.Cin, cit), d.B), block(256);
  const size_t lds = 4 * cit * 4 * sizeof(double) + (size_t)4 * cit * sizeof(float);
  if (cit == 16) hipLaunchKernelGGL(conv_bwd_data_direct<16>, grid, block, lds, st, d);
  else hipLaunchKernelGGL(conv_bwd_data_direct<8>, grid, block, lds, st, d);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

// does the first-convolution weight-gradient kernel take `d` (the one VALU kernel that can finalize on load)?
bool conv_backward_weight_first_applies(const pdes_conv_desc& d) {
  if (!(!d.has_bn && !d.upsample && d.Cin <= 4 && d.ksize == 7 && d.stride == 2 && d.Wout % 4 == 0 && d.Hout % 4 == 0 &&
        d.Win % 4 == 0))
    return false;
  const int LH = d.Hin + 2 * d.pad + 2, LW = ((d.Win + 2 * d.pad + 2 + 4 + 3) / 4) * 4;
  return ((size_t)d.Cin * LH * LW + (size_t)8 * d.Hout * d.Wout) * sizeof(float) <= 150 * 1024;
}

int conv_backward_weight_direct(const pdes_conv_desc& d, hipStream_t st) {
  const int rc = validate(d, 1);
  if (rc) return rc;
  if (!d.has_bn && !d.upsample && d.Cin <= 4 && d.ksize == 7 && d.stride == 2 && d.Wout % 4 == 0 && d.Hout % 4 == 0 &&
      d.Win % 4 == 0) {
    // first convolution (7x7, stride 2): LDS-resident planes, register-tiled kernel rows (conv_bwd_weight_first)
    const int LH = d.Hin + 2 * d.pad + 2, LW = ((d.Win + 2 * d.pad + 2 + 4 + 3) / 4) * 4;
    const size_t lds = ((size_t)d.Cin * LH * LW + (size_t)8 * d.Hout * d.Wout) * sizeof(float);
    if (lds <= 150 * 1024) {
      float* part = first_layer_partials(d) ? d.ws : nullptr;
      if (d.g_fused) {
        if (!d.fin_xstats || !d.fin_tstats || !d.out || d.g_ctot != d.out_ctot || d.g_coff != d.out_coff || d.g_add ||
            d.nrep != PDES_NREP)
          return PDES_EINVAL;
        hipLaunchKernelGGL((conv_bwd_weight_first<8, 7, 2, true>), dim3(d.B, cdiv(d.Cout, 8)), dim3(256), lds, st, d, part);
      } else {
        hipLaunchKernelGGL((conv_bwd_weight_first<8, 7, 2, false>), dim3(d.B, cdiv(d.Cout, 8)), dim3(256), lds, st, d, part);
      }
      PDES_LAUNCH_CHECK();
      return PDES_OK;
    }
  }
  if (d.g_fused) return PDES_ENOSUP;                     // (only the first-convolution kernel above finalizes on load)
  int cot;
  switch (d.ksize) { case 1: cot = 16; break; case 3: cot = 8; break; case 5: cot = 4; break; default: cot = 2; }
  const int ncog = cdiv(d.Cout, cot);
  int nsplit = cdiv(1024, d.Cin * ncog);
  nsplit = nsplit < 1 ? 1 : (nsplit > d.B ? d.B : nsplit);
  dim3 grid(d.Cin, ncog, nsplit), block(256);
  switch (d.ksize) {
    case 1: hipLaunchKernelGGL((conv_bwd_weight_direct<1, 16>), grid, block, 0, st, d, nsplit); break;
    case 3: hipLaunchKernelGGL((conv_bwd_weight_direct<3, 8>), grid, block, 0, st, d, nsplit); break;
    case 5: hipLaunchKernelGGL((conv_bwd_weight_direct<5, 4>), grid, block, 0, st, d, nsplit); break;
    default: hipLaunchKernelGGL((conv_bwd_weight_direct<7, 2>), grid, block, 0, st, d, nsplit); break;
  }
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

}  // namespace pdes
