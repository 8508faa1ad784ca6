// Flow operators of the multiscale conditional Glow (reference models/glow_msc.py) as descriptors of the same chain
// that runs the DenseED convolutions (include/pdes_hip.h: PDES_OP_COPY .. PDES_OP_GAUSS), z -> y direction with its
// backward pass (the TRAINING path of train_cglow_reverse_kl.py:245-262) and the y -> z direction without one.
//
// Every operator is HBM/L2-bandwidth or latency bound (3..48 channels at 32x32 .. 8x8): one thread per pixel (or per
// float4 of a channel plane), lanes of a wave on consecutive pixels, per-channel / per-sample reductions by wave
// shuffles -> LDS -> one fp64 atomic per workgroup into a replica of the caller's accumulator arena (like the BatchNorm
// statistics: order dependence < 1e-16 relative).  The matrix of an invertible 1x1 convolution (C <= 48) sits in LDS.
#include <math.h>
#include "pdes_common.h"
#include "../../include/pdes_hip.h"

namespace pdes {

#define PDES_LOG2PI 1.8378770664093453f
#define PDES_LSD_MIN (-10.f)
#define PDES_LSD_MAX 1.6094379124341003f       // log 5

static bool flow_common_ok(const pdes_conv_desc& d) {
  return d.ksize == 0 && d.B > 0 && d.Cin > 0 && d.Cout > 0 && d.Hin > 0 && d.Win > 0 && d.nrep == PDES_NREP;
}

// ------------------------------------------------------------------------------------------------ COPY
// grid (ceil(HW / 1024), C, B), 256 threads x float4
// channels [0, c1) come from x, the rest from x2 (torch.cat((y1, cond), 1) in one launch)
__global__ __launch_bounds__(256) void flow_copy_kernel(const float* __restrict__ x, int x_ctot, int c1,
                                                        const float* __restrict__ x2, int x2_ctot, float* __restrict__ out,
                                                        int out_ctot, int out_coff, int HW, double* __restrict__ stats,
                                                        int nrep, long long rs) {
  __shared__ double red[4][2];
  const int c = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const float4* src = reinterpret_cast<const float4*>(c < c1 ? x + ((size_t)b * x_ctot + c) * HW
                                                             : x2 + ((size_t)b * x2_ctot + (c - c1)) * HW);
  float4* dst = reinterpret_cast<float4*>(out + ((size_t)b * out_ctot + out_coff + c) * HW);
  const int i = blockIdx.x * 256 + tid;
  float s = 0.f, q = 0.f;
  if (i < HW / 4) {
    const float4 v = src[i];
    dst[i] = v;
    s = (v.x + v.y) + (v.z + v.w);
    q = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (!stats) return;
  const float ws = wave_sum(s), wq = wave_sum(q);
  if ((tid & 63) == 0) { red[tid >> 6][0] = ws; red[tid >> 6][1] = wq; }
  __syncthreads();
  if (tid < 2) {
    const double t = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    atomicAdd(&stats[(long long)rep_of_block(nrep) * rs + 2 * (out_coff + c) + tid], t);
  }
}

// FUSED (pdes_conv_desc.g_fused): `g` still holds the accumulator T of the copied channels; the BatchNorm-backward
// finalize  dL/dx = invstd (T - mean(T) - xhat mean(T xhat))  is applied here, on the way to the sources' gradients
// (`act` = the copied activation, the statistics as in bn_bwd_finalize_kernel), instead of a separate pass over T
template <bool FUSED>
__global__ __launch_bounds__(256) void flow_copy_bwd_kernel(const float* __restrict__ g, int g_ctot, int g_coff,
                                                            float* __restrict__ t, int t_ctot, int c1, float* __restrict__ t2,
                                                            int t2_ctot, int HW, int accumulate, const float* __restrict__ act,
                                                            const double* __restrict__ x_stats, const double* __restrict__ t_stats,
                                                            int B, float eps, long long rs) {
  const int c = blockIdx.y, b = blockIdx.z;
  __shared__ float sc[4];
  __shared__ double sums[4];
  if (FUSED) {
    if (threadIdx.x < 64) {          // 4 quantities x PDES_NREP (<= 16) replicas: one load per lane, 16-lane shuffle reduction
      const int q = threadIdx.x >> 4, r = threadIdx.x & 15;
      double v = r < PDES_NREP ? (q < 2 ? x_stats : t_stats)[(long long)r * rs + 2 * (g_coff + c) + (q & 1)] : 0.0;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 16);
      if (r == 0) sums[q] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const double inv_n = 1.0 / ((double)B * HW);
      const double m = sums[0] * inv_n;
      double var = sums[1] * inv_n - m * m;
      var = var < 0.0 ? 0.0 : var;
      sc[0] = (float)m;
      sc[1] = (float)(1.0 / sqrt(var + (double)eps));
      sc[2] = (float)(sums[2] * inv_n);
      sc[3] = (float)(sums[3] * inv_n);
    }
    __syncthreads();
  }
  const float4* src = reinterpret_cast<const float4*>(g + ((size_t)b * g_ctot + g_coff + c) * HW);
  float4* dst;
  if (c < c1) {
    if (!t) return;
    dst = reinterpret_cast<float4*>(t + ((size_t)b * t_ctot + c) * HW);
  } else {
    if (!t2) return;
    dst = reinterpret_cast<float4*>(t2 + ((size_t)b * t2_ctot + (c - c1)) * HW);
    accumulate = 1;                       // the second source's gradient always collects several consumers
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= HW / 4) return;
  float4 v = src[i];
  if (FUSED) {
    const float4 xv = reinterpret_cast<const float4*>(act + ((size_t)b * g_ctot + g_coff + c) * HW)[i];
    const float mean = sc[0], invstd = sc[1], m1 = sc[2], m2 = sc[3];
    v.x = invstd * (v.x - m1 - (xv.x - mean) * invstd * m2);
    v.y = invstd * (v.y - m1 - (xv.y - mean) * invstd * m2);
    v.z = invstd * (v.z - m1 - (xv.z - mean) * invstd * m2);
    v.w = invstd * (v.w - m1 - (xv.w - mean) * invstd * m2);
  }
  if (accumulate) {
    const float4 o = dst[i];
    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
  }
  dst[i] = v;
}

static bool copy_ok(const pdes_conv_desc& d) {
  if (!(flow_common_ok(d) && d.upsample == PDES_OP_COPY && d.Hin == d.Hout && d.Win == d.Wout && (d.Hin * d.Win) % 4 == 0 &&
        d.x && d.out))
    return false;
  if (d.Cout == d.Cin) return true;
  return d.Cout > d.Cin && d.x2 && d.Cout - d.Cin <= d.x2_ctot && aligned16(d.x2);
}

int flow_copy_forward(const pdes_conv_desc& d, hipStream_t st) {
  if (!copy_ok(d)) return PDES_EINVAL;
  if (!aligned16(d.x) || !aligned16(d.out)) return PDES_EALIGN;
  const int HW = d.Hin * d.Win;
  hipLaunchKernelGGL(flow_copy_kernel, dim3(cdiv(HW / 4, 256), d.Cout, d.B), dim3(256), 0, st, d.x, d.x_ctot, d.Cin, d.x2,
                     d.x2_ctot, d.out, d.out_ctot, d.out_coff, HW, d.out_stats, d.nrep, d.rep_stride);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

int flow_copy_backward(const pdes_conv_desc& d, hipStream_t st) {
  if (!copy_ok(d)) return PDES_EINVAL;
  if (!d.t_in && !d.t2) return PDES_OK;              // the sources are input data
  if (!d.g || !aligned16(d.g) || (d.t_in && !aligned16(d.t_in)) || (d.t2 && !aligned16(d.t2))) return PDES_EINVAL;
  const int HW = d.Hin * d.Win;
  if (d.g_fused) {
    if (!d.fin_xstats || !d.fin_tstats || d.g_ctot != d.out_ctot || d.g_coff != d.out_coff || d.g_add) return PDES_EINVAL;
    hipLaunchKernelGGL(flow_copy_bwd_kernel<true>, dim3(cdiv(HW / 4, 256), d.Cout, d.B), dim3(256), 0, st, d.g, d.g_ctot,
                       d.g_coff, d.t_in, d.x_ctot, d.Cin, d.t2, d.x2_ctot, HW, d.t_accumulate, (const float*)d.out, d.fin_xstats,
                       d.fin_tstats, d.B, d.eps, d.rep_stride);
  } else {
    hipLaunchKernelGGL(flow_copy_bwd_kernel<false>, dim3(cdiv(HW / 4, 256), d.Cout, d.B), dim3(256), 0, st, d.g, d.g_ctot,
                       d.g_coff, d.t_in, d.x_ctot, d.Cin, d.t2, d.x2_ctot, HW, d.t_accumulate, (const float*)nullptr,
                       (const double*)nullptr, (const double*)nullptr, d.B, d.eps, d.rep_stride);
  }
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

// ------------------------------------------------------------------------------------------ BIAS_SCALE
// MODE 0: forward, in place on `buf` (+ statistics); MODE 1: backward, in place on `buf` = g, `y` = the forward output
template <int MODE>
__global__ __launch_bounds__(256) void flow_bias_scale_kernel(float* __restrict__ buf, int ctot, int coff,
                                                              const float* __restrict__ y, int y_ctot, int y_coff, int HW,
                                                              const float* __restrict__ bias, const float* __restrict__ scale,
                                                              double* __restrict__ acc, int nrep, long long rs) {
  __shared__ double red[4][2];
  const int c = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const float e = scale ? expf(scale[c] * 3.f) : 1.f;
  const float bs = bias[c];
  float4* p = reinterpret_cast<float4*>(buf + ((size_t)b * ctot + coff + c) * HW);
  const int i = blockIdx.x * 256 + tid;
  float s = 0.f, q = 0.f;
  if (i < HW / 4) {
    float4 v = p[i];
    if (MODE == 0) {
      v.x = (v.x + bs) * e; v.y = (v.y + bs) * e; v.z = (v.z + bs) * e; v.w = (v.w + bs) * e;
      s = (v.x + v.y) + (v.z + v.w);
      q = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    } else {
      const float4 o = reinterpret_cast<const float4*>(y + ((size_t)b * y_ctot + y_coff + c) * HW)[i];
      q = 3.f * ((v.x * o.x + v.y * o.y) + (v.z * o.z + v.w * o.w));     // dscale: d/ds (u exp(3 s)) = 3 out
      v.x *= e; v.y *= e; v.z *= e; v.w *= e;
      s = (v.x + v.y) + (v.z + v.w);                                     // dbias
    }
    p[i] = v;
  }
  if (!acc) return;
  const float ws = wave_sum(s), wq = wave_sum(q);
  if ((tid & 63) == 0) { red[tid >> 6][0] = ws; red[tid >> 6][1] = wq; }
  __syncthreads();
  if (tid < 2) {
    const double t = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    const int slot = MODE == 0 ? 2 * (coff + c) + tid : 2 * c + tid;     // statistics of the buffer / {dbias, dscale} of the op
    atomicAdd(&acc[(long long)rep_of_block(nrep) * rs + slot], t);
  }
}

static bool bias_ok(const pdes_conv_desc& d) {
  return flow_common_ok(d) && d.upsample == PDES_OP_BIAS_SCALE && d.Cin == d.Cout && d.Hin == d.Hout && d.Win == d.Wout &&
         (d.Hin * d.Win) % 4 == 0 && d.out && d.p0;
}

int flow_bias_scale_forward(const pdes_conv_desc& d, hipStream_t st) {
  if (!bias_ok(d)) return PDES_EINVAL;
  if (!aligned16(d.out)) return PDES_EALIGN;
  const int HW = d.Hin * d.Win;
  hipLaunchKernelGGL(flow_bias_scale_kernel<0>, dim3(cdiv(HW / 4, 256), d.Cout, d.B), dim3(256), 0, st, d.out, d.out_ctot,
                     d.out_coff, (const float*)nullptr, 0, 0, HW, d.p0, d.p1, d.out_stats, d.nrep, d.rep_stride);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

int flow_bias_scale_backward(const pdes_conv_desc& d, hipStream_t st) {
  if (!bias_ok(d) || !d.g || !d.acc) return PDES_EINVAL;
  if (!aligned16(d.g)) return PDES_EALIGN;
  const int HW = d.Hin * d.Win;
  hipLaunchKernelGGL(flow_bias_scale_kernel<1>, dim3(cdiv(HW / 4, 256), d.Cout, d.B), dim3(256), 0, st,
                     const_cast<float*>(d.g), d.g_ctot, d.g_coff, (const float*)d.out, d.out_ctot, d.out_coff, HW, d.p0, d.p1,
                     d.acc, d.nrep, d.rep_stride);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

// -------------------------------------------------------------------------------------------- COUPLING
__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

// block-wide sum of `v` (256 threads) added to acc[b] of one replica
__device__ __forceinline__ void block_add_logp(float v, double* acc, int b, int nrep, long long rs, double* red) {
  const float ws = wave_sum(v);
  const int tid = threadIdx.x;
  if ((tid & 63) == 0) red[tid >> 6] = (double)ws;
  __syncthreads();
  if (tid == 0) atomicAdd(&acc[(long long)rep_of_block(nrep) * rs + b], (red[0] + red[1]) + (red[2] + red[3]));
}

// The coupling network's last layer is a Conv2dZeros: h = (conv + bias) * exp(3 scale) (glow_msc.py:519-531).  With
// d.gamma = bias and d.beta = scale (NULL: none) the coupling applies that epilogue itself -- `x2` holds the RAW convolution
// output, is rewritten in place with h (the backward pass reads it), and no PDES_OP_BIAS_SCALE launch stands between
// the convolution and the coupling; the backward pass turns dL/dh into dL/d(conv) and accumulates {dbias, dscale} into
// d.bn_grad (slot 2 c + {0, 1} of a replica: the layout of flow_bias_scale_kernel<1>).
// Both kernels walk the channels of ONE pixel per thread: the loads of FLOW_KB channels are issued together, then their
// arithmetic and stores (a load - compute - store loop per channel was one memory round trip per channel in a row, the
// stores keeping the compiler from moving the next channel's loads up: 9-14 us per launch on the 12-channel levels).
constexpr int FLOW_KB = 8;
// grid (ceil(HW / 256), B)
__global__ __launch_bounds__(256) void flow_coupling_kernel(pdes_conv_desc d) {
  __shared__ double red[4];
  const int HW = d.Hin * d.Win, b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  const int C = d.Cin, n2 = C / 2, n1 = C - n2;
  const bool act = p < HW;
  const float* x = d.x + (size_t)b * d.x_ctot * HW + p;
  float* h = const_cast<float*>(d.x2) + (size_t)b * d.x2_ctot * HW + p;
  float* o = d.out + ((size_t)b * d.out_ctot + d.out_coff) * HW + p;
  const bool fwd = (d.flags & PDES_FLOW_FORWARD) != 0;
  float ld = 0.f;
  if (act) {
    for (int c0 = 0; c0 < n1; c0 += FLOW_KB) {
      float v[FLOW_KB];
#pragma unroll
      for (int j = 0; j < FLOW_KB; ++j) v[j] = x[(size_t)min(c0 + j, n1 - 1) * HW];
#pragma unroll
      for (int j = 0; j < FLOW_KB; ++j)
        if (c0 + j < n1) o[(size_t)(c0 + j) * HW] = v[j];
    }
    for (int k0 = 0; k0 < n2; k0 += FLOW_KB) {
      float hs[FLOW_KB], hr[FLOW_KB], xv[FLOW_KB];
#pragma unroll
      for (int j = 0; j < FLOW_KB; ++j) {
        const int k = min(k0 + j, n2 - 1);
        hs[j] = h[(size_t)(2 * k) * HW];
        hr[j] = h[(size_t)(2 * k + 1) * HW];
        xv[j] = x[(size_t)(n1 + k) * HW];
      }
#pragma unroll
      for (int j = 0; j < FLOW_KB; ++j) {
        const int k = k0 + j;
        if (k < n2) {
          float shift = hs[j], raw = hr[j];
          if (d.gamma) {
            shift = (shift + d.gamma[2 * k]) * (d.beta ? expf(d.beta[2 * k] * 3.f) : 1.f);
            raw = (raw + d.gamma[2 * k + 1]) * (d.beta ? expf(d.beta[2 * k + 1] * 3.f) : 1.f);
            h[(size_t)(2 * k) * HW] = shift;
            h[(size_t)(2 * k + 1) * HW] = raw;
          }
          const float sg = sigmoidf_(raw + 2.f);
          o[(size_t)(n1 + k) * HW] = fwd ? (xv[j] + shift) * sg : xv[j] / sg - shift;
          ld += logf(sg);
        }
      }
    }
  }
  if (d.acc) block_add_logp(ld, d.acc, b, d.nrep, d.rep_stride, red);
}

__global__ __launch_bounds__(256) void flow_coupling_bwd_kernel(pdes_conv_desc d) {
  __shared__ float part[2 * 24 * 2][4];          // [h channel][dbias, dscale][wave]   (coupling_ok: C <= 48)
  const int HW = d.Hin * d.Win, b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x, tid = threadIdx.x;
  const bool act = p < HW, fold = d.gamma != nullptr;
  if (!act && !fold) return;
  const int C = d.Cin, n2 = C / 2, n1 = C - n2;
  const float cst = d.p1 ? d.p1[b] : 0.f;
  const float* x = d.x + (size_t)b * d.x_ctot * HW + p;
  const float* h = d.x2 + (size_t)b * d.x2_ctot * HW + p;
  const float* g = d.g + ((size_t)b * d.g_ctot + d.g_coff) * HW + p;
  float* tx = d.t_in + (size_t)b * d.x_ctot * HW + p;
  float* th = d.t2 + (size_t)b * d.x2_ctot * HW + p;
  const bool accu = d.t_accumulate != 0;
  if (act)
    for (int c0 = 0; c0 < n1; c0 += FLOW_KB) {
      float v[FLOW_KB];
#pragma unroll
      for (int j = 0; j < FLOW_KB; ++j) {
        const int c = min(c0 + j, n1 - 1);
        v[j] = g[(size_t)c * HW] + (accu ? tx[(size_t)c * HW] : 0.f);
      }
#pragma unroll
      for (int j = 0; j < FLOW_KB; ++j)
        if (c0 + j < n1) tx[(size_t)(c0 + j) * HW] = v[j];
    }
  for (int k0 = 0; k0 < n2; k0 += FLOW_KB) {
    float h0[FLOW_KB], h1[FLOW_KB], gv[FLOW_KB], xv[FLOW_KB], told[FLOW_KB];
#pragma unroll
    for (int j = 0; j < FLOW_KB; ++j) {
      const int k = min(k0 + j, n2 - 1);
      h0[j] = h1[j] = gv[j] = xv[j] = told[j] = 0.f;
      if (act) {
        h1[j] = h[(size_t)(2 * k + 1) * HW];
        if (fold) h0[j] = h[(size_t)(2 * k) * HW];
        gv[j] = g[(size_t)(n1 + k) * HW];
        xv[j] = x[(size_t)(n1 + k) * HW];
        if (accu) told[j] = tx[(size_t)(n1 + k) * HW];
      }
    }
#pragma unroll
    for (int j = 0; j < FLOW_KB; ++j) {
      const int k = k0 + j;
      if (k >= n2) break;
      float g0 = 0.f, g1 = 0.f, q0 = 0.f, q1 = 0.f;
      if (act) {
        const float sg = sigmoidf_(h1[j] + 2.f);
        const float gx = gv[j] / sg;
        tx[(size_t)(n1 + k) * HW] = told[j] + gx;
        g0 = -gv[j];
        g1 = (cst - gx * xv[j]) * (1.f - sg);     // out = v / s - shift, log s; ds/dh = s (1 - s)
        if (fold) {                               // dL/dh -> dL/d(conv) = dL/dh e;  dscale = sum 3 dL/dh h;  dbias = sum dL/d(conv)
          q0 = 3.f * g0 * h0[j];
          q1 = 3.f * g1 * h1[j];
          g0 *= d.beta ? expf(d.beta[2 * k] * 3.f) : 1.f;
          g1 *= d.beta ? expf(d.beta[2 * k + 1] * 3.f) : 1.f;
        }
        th[(size_t)(2 * k) * HW] = g0;
        th[(size_t)(2 * k + 1) * HW] = g1;
      }
      if (fold) {
        const float a0 = wave_sum(g0), s0 = wave_sum(q0), a1 = wave_sum(g1), s1 = wave_sum(q1);
        if ((tid & 63) == 0) {
          part[(2 * k) * 2 + 0][tid >> 6] = a0; part[(2 * k) * 2 + 1][tid >> 6] = s0;
          part[(2 * k + 1) * 2 + 0][tid >> 6] = a1; part[(2 * k + 1) * 2 + 1][tid >> 6] = s1;
        }
      }
    }
  }
  if (!fold) return;
  __syncthreads();
  if (tid < 4 * n2) {                             // slot 2 c + {0: dbias, 1: dscale}
    const double t = ((double)part[tid][0] + (double)part[tid][1]) + ((double)part[tid][2] + (double)part[tid][3]);
    atomicAdd(&d.bn_grad[(long long)rep_of_block(d.nrep) * d.rep_stride + tid], t);
  }
}

static bool coupling_ok(const pdes_conv_desc& d) {
  return flow_common_ok(d) && d.upsample == PDES_OP_COUPLING && d.Cin == d.Cout && d.Cin >= 2 && d.Cin <= 48 && d.Hin == d.Hout &&
         d.Win == d.Wout && d.x && d.x2 && d.out && d.x2_ctot >= 2 * (d.Cin / 2) && (d.gamma || !d.beta);
}

int flow_coupling_forward(const pdes_conv_desc& d, hipStream_t st) {
  if (!coupling_ok(d)) return PDES_EINVAL;
  hipLaunchKernelGGL(flow_coupling_kernel, dim3(cdiv(d.Hin * d.Win, 256), d.B), dim3(256), 0, st, d);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

int flow_coupling_backward(const pdes_conv_desc& d, hipStream_t st) {
  if (!coupling_ok(d) || !d.g || !d.t_in || !d.t2 || (d.flags & PDES_FLOW_FORWARD) || (d.gamma && !d.bn_grad)) return PDES_EINVAL;
  hipLaunchKernelGGL(flow_coupling_bwd_kernel, dim3(cdiv(d.Hin * d.Win, 256), d.B), dim3(256), 0, st, d);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

// ------------------------------------------------------------------------------------------------- MIX
// out = (W x - bias) / weight per pixel (z -> y), or out = W (weight x + bias) (y -> z, W = the inverse matrix).
// grid (ceil(HW / PB), B), PB threads; dynamic LDS: W [C][C] + tile [C][PB + 1]
template <int PB>
__global__ __launch_bounds__(PB) void flow_mix_kernel(pdes_conv_desc d) {
  extern __shared__ __attribute__((aligned(16))) float sm_mix[];
  const int C = d.Cin, HW = d.Hin * d.Win, b = blockIdx.y, tid = threadIdx.x, p = blockIdx.x * PB + tid;
  float* Ws = sm_mix;
  float* xs = sm_mix + C * C;
  const bool fwd = (d.flags & PDES_FLOW_FORWARD) != 0;
  for (int i = tid; i < C * C; i += PB) Ws[i] = d.x2[i];
  const float* x = d.x + (size_t)b * d.x_ctot * HW;
  if (p < HW)
    for (int c = 0; c < C; ++c) {
      float v = x[(size_t)c * HW + p];
      if (fwd) v = d.p0[c] * v + d.p1[c];
      xs[c * (PB + 1) + tid] = v;
    }
  __syncthreads();
  if (p >= HW) return;
  float* o = d.out + ((size_t)b * d.out_ctot + d.out_coff) * HW + p;
  for (int oc = 0; oc < C; ++oc) {
    float a = 0.f;
    for (int c = 0; c < C; ++c) a += Ws[oc * C + c] * xs[c * (PB + 1) + tid];
    o[(size_t)oc * HW] = fwd ? a : (a - d.p1[oc]) / d.p0[oc];
  }
}

// backward of the z -> y form: gv = g / weight; t_in = W^T gv; acc += {dweight = -sum gv out, dbias = -sum gv, dW = sum gv x^T}
// dynamic LDS: W [C][C] + xs [C][PB+1] + gs [C][PB+1] + part [2][C][PB/64] floats
template <int PB>
__global__ __launch_bounds__(PB) void flow_mix_bwd_kernel(pdes_conv_desc d) {
  extern __shared__ __attribute__((aligned(16))) float sm_mix[];
  constexpr int NW = PB / 64;
  const int C = d.Cin, HW = d.Hin * d.Win, b = blockIdx.y, tid = threadIdx.x, p0 = blockIdx.x * PB;
  float* Ws = sm_mix;
  float* xs = Ws + C * C;
  float* gs = xs + C * (PB + 1);
  float* part = gs + C * (PB + 1);                       // [2][C][NW]
  for (int i = tid; i < C * C; i += PB) Ws[i] = d.x2[i];
  const float* x = d.x + (size_t)b * d.x_ctot * HW;
  const float* g = d.g + ((size_t)b * d.g_ctot + d.g_coff) * HW;
  const float* y = d.out + ((size_t)b * d.out_ctot + d.out_coff) * HW;
  const bool act = p0 + tid < HW;
  // eight channels' loads in flight together (one channel at a time, the wave reductions in between serialised the
  // round trips: 30 us per layer on the 16x16 / 8x8 levels)
  for (int c0 = 0; c0 < C; c0 += 8) {
    float xv[8], gv[8], yv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = min(c0 + j, C - 1);
      xv[j] = act ? x[(size_t)c * HW + p0 + tid] : 0.f;
      gv[j] = act ? g[(size_t)c * HW + p0 + tid] : 0.f;
      yv[j] = act ? y[(size_t)c * HW + p0 + tid] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j;
      if (c >= C) break;
      const float gq = gv[j] / d.p0[c];
      xs[c * (PB + 1) + tid] = xv[j];
      gs[c * (PB + 1) + tid] = gq;
      const float a = wave_sum(-gq * yv[j]), bsum = wave_sum(-gq);
      if ((tid & 63) == 0) { part[(0 * C + c) * NW + (tid >> 6)] = a; part[(1 * C + c) * NW + (tid >> 6)] = bsum; }
    }
  }
  __syncthreads();
  if (act) {
    float* t = d.t_in + (size_t)b * d.x_ctot * HW + p0 + tid;
    for (int c = 0; c < C; ++c) {
      float a = 0.f;
      for (int oc = 0; oc < C; ++oc) a += Ws[oc * C + c] * gs[oc * (PB + 1) + tid];
      t[(size_t)c * HW] = d.t_accumulate ? t[(size_t)c * HW] + a : a;
    }
  }
  double* acc = d.acc + (long long)rep_of_block(d.nrep) * d.rep_stride;
  for (int i = tid; i < 2 * C; i += PB) {
    float s = 0.f;
    for (int w = 0; w < NW; ++w) s += part[i * NW + w];
    atomicAdd(&acc[i], (double)s);
  }
  const int np = min(PB, HW - p0);
  for (int i = tid; i < C * C; i += PB) {
    const int oc = i / C, c = i % C;
    float s = 0.f;
    for (int q = 0; q < np; ++q) s += gs[oc * (PB + 1) + q] * xs[c * (PB + 1) + q];
    atomicAdd(&acc[2 * C + i], (double)s);
  }
}

// ---- PDES_MIX_COUPLED: affine coupling + invertible 1x1 (+ ActNorm), z -> y, in ONE launch each way.  Both operators act
// per pixel: the coupling's output u goes straight into the matrix product's LDS tile and is never stored; the backward
// pass recomputes it from the coupling's input and h (one sigmoid per channel pair) and runs the coupling's backward on the
// matrix product's input gradient while that is in registers.  dynamic LDS as for the plain kernels.
template <int PB>
__global__ __launch_bounds__(PB) void flow_coupling_mix_kernel(pdes_conv_desc d) {
  extern __shared__ __attribute__((aligned(16))) float sm_mix[];
  __shared__ double red[4];
  const int C = d.Cin, HW = d.Hin * d.Win, b = blockIdx.y, tid = threadIdx.x, p = blockIdx.x * PB + tid;
  const int n2 = C / 2, n1 = C - n2;
  float* Ws = sm_mix;
  float* xs = sm_mix + C * C;
  for (int i = tid; i < C * C; i += PB) Ws[i] = d.x2[i];
  const float* x = d.x + (size_t)b * d.x_ctot * HW + p;
  float* h = const_cast<float*>(d.h) + (size_t)b * d.h_ctot * HW + p;
  float ld = 0.f;
  if (p < HW) {
    for (int c0 = 0; c0 < n1; c0 += FLOW_KB) {
      float v[FLOW_KB];
#pragma unroll
      for (int j = 0; j < FLOW_KB; ++j) v[j] = x[(size_t)min(c0 + j, n1 - 1) * HW];
#pragma unroll
      for (int j = 0; j < FLOW_KB; ++j)
        if (c0 + j < n1) xs[(c0 + j) * (PB + 1) + tid] = v[j];
    }
    for (int k0 = 0; k0 < n2; k0 += FLOW_KB) {
      float hs[FLOW_KB], hr[FLOW_KB], xv[FLOW_KB];
#pragma unroll
      for (int j = 0; j < FLOW_KB; ++j) {
        const int k = min(k0 + j, n2 - 1);
        hs[j] = h[(size_t)(2 * k) * HW];
        hr[j] = h[(size_t)(2 * k + 1) * HW];
        xv[j] = x[(size_t)(n1 + k) * HW];
      }
#pragma unroll
      for (int j = 0; j < FLOW_KB; ++j) {
        const int k = k0 + j;
        if (k < n2) {
          float shift = hs[j], raw = hr[j];
          if (d.gamma) {
            shift = (shift + d.gamma[2 * k]) * (d.beta ? expf(d.beta[2 * k] * 3.f) : 1.f);
            raw = (raw + d.gamma[2 * k + 1]) * (d.beta ? expf(d.beta[2 * k + 1] * 3.f) : 1.f);
            h[(size_t)(2 * k) * HW] = shift;
            h[(size_t)(2 * k + 1) * HW] = raw;
          }
          const float sg = sigmoidf_(raw + 2.f);
          xs[(n1 + k) * (PB + 1) + tid] = xv[j] / sg - shift;
          ld += logf(sg);
        }
      }
    }
  }
  if (PB >= 256) {
    if (d.acc2) block_add_logp(ld, d.acc2, b, d.nrep, d.rep_stride, red);      // (a barrier inside: every thread calls)
  } else if (d.acc2) {                                                          // one wave per block
    const float ws = wave_sum(ld);
    if (tid == 0) atomicAdd(&d.acc2[(long long)rep_of_block(d.nrep) * d.rep_stride + b], (double)ws);
  }
  __syncthreads();
  if (p >= HW) return;
  float* o = d.out + ((size_t)b * d.out_ctot + d.out_coff) * HW + p;
  for (int oc = 0; oc < C; ++oc) {
    float a = 0.f;
    for (int c = 0; c < C; ++c) a += Ws[oc * C + c] * xs[c * (PB + 1) + tid];
    o[(size_t)oc * HW] = (a - d.p1[oc]) / d.p0[oc];
  }
}

// dynamic LDS: W [C][C] + xs [C][PB+1] + gs [C][PB+1] + part [2][C][PB/64] + part2 [2 n2][2][PB/64] floats
template <int PB>
__global__ __launch_bounds__(PB) void flow_mix_coupling_bwd_kernel(pdes_conv_desc d) {
  extern __shared__ __attribute__((aligned(16))) float sm_mix[];
  constexpr int NW = PB / 64;
  const int C = d.Cin, HW = d.Hin * d.Win, b = blockIdx.y, tid = threadIdx.x, p0 = blockIdx.x * PB;
  const int n2 = C / 2, n1 = C - n2;
  float* Ws = sm_mix;
  float* xs = Ws + C * C;
  float* gs = xs + C * (PB + 1);
  float* part = gs + C * (PB + 1);                       // [2][C][NW]
  float* part2 = part + 2 * C * NW;                      // [2 n2][2][NW]
  for (int i = tid; i < C * C; i += PB) Ws[i] = d.x2[i];
  const bool act = p0 + tid < HW, fold = d.gamma != nullptr, accu = d.t_accumulate != 0;
  const float* x = d.x + (size_t)b * d.x_ctot * HW + p0 + tid;
  const float* h = d.h + (size_t)b * d.h_ctot * HW + p0 + tid;
  const float* g = d.g + ((size_t)b * d.g_ctot + d.g_coff) * HW + p0 + tid;
  const float* y = d.out + ((size_t)b * d.out_ctot + d.out_coff) * HW + p0 + tid;
  float* tx = d.t_in + (size_t)b * d.x_ctot * HW + p0 + tid;
  float* th = d.th + (size_t)b * d.h_ctot * HW + p0 + tid;
  const float cst = d.cst ? d.cst[b] : 0.f;
  // (1) u, the mix's input, recomputed into xs
  for (int c0 = 0; c0 < n1; c0 += FLOW_KB) {
    float v[FLOW_KB];
#pragma unroll
    for (int j = 0; j < FLOW_KB; ++j) v[j] = act ? x[(size_t)min(c0 + j, n1 - 1) * HW] : 0.f;
#pragma unroll
    for (int j = 0; j < FLOW_KB; ++j)
      if (c0 + j < n1) xs[(c0 + j) * (PB + 1) + tid] = v[j];
  }
  for (int k0 = 0; k0 < n2; k0 += FLOW_KB) {
    float hs[FLOW_KB], hr[FLOW_KB], xv[FLOW_KB];
#pragma unroll
    for (int j = 0; j < FLOW_KB; ++j) {
      const int k = min(k0 + j, n2 - 1);
      hs[j] = act ? h[(size_t)(2 * k) * HW] : 0.f;
      hr[j] = act ? h[(size_t)(2 * k + 1) * HW] : 0.f;
      xv[j] = act ? x[(size_t)(n1 + k) * HW] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < FLOW_KB; ++j)
      if (k0 + j < n2) xs[(n1 + k0 + j) * (PB + 1) + tid] = xv[j] / sigmoidf_(hr[j] + 2.f) - hs[j];
  }
  // (2) gv = g / weight into gs; the ActNorm's {dweight, dbias} partial sums
  for (int c0 = 0; c0 < C; c0 += 8) {
    float gv[8], yv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = min(c0 + j, C - 1);
      gv[j] = act ? g[(size_t)c * HW] : 0.f;
      yv[j] = act ? y[(size_t)c * HW] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j;
      if (c >= C) break;
      const float gq = gv[j] / d.p0[c];
      gs[c * (PB + 1) + tid] = gq;
      const float a = wave_sum(-gq * yv[j]), bsum = wave_sum(-gq);
      if ((tid & 63) == 0) { part[(0 * C + c) * NW + (tid >> 6)] = a; part[(1 * C + c) * NW + (tid >> 6)] = bsum; }
    }
  }
  __syncthreads();
  // (3) dL/du = W^T gv, and the coupling's backward on it
  for (int c = 0; c < n1; ++c) {
    float a = 0.f;
    for (int oc = 0; oc < C; ++oc) a += Ws[oc * C + c] * gs[oc * (PB + 1) + tid];
    if (act) tx[(size_t)c * HW] = accu ? tx[(size_t)c * HW] + a : a;
  }
  for (int k = 0; k < n2; ++k) {
    float a = 0.f;
    for (int oc = 0; oc < C; ++oc) a += Ws[oc * C + n1 + k] * gs[oc * (PB + 1) + tid];
    float g0 = 0.f, g1 = 0.f, q0 = 0.f, q1 = 0.f;
    if (act) {
      const float h0 = h[(size_t)(2 * k) * HW], h1 = h[(size_t)(2 * k + 1) * HW], xv = x[(size_t)(n1 + k) * HW];
      const float sg = sigmoidf_(h1 + 2.f);
      const float gx = a / sg;
      tx[(size_t)(n1 + k) * HW] = accu ? tx[(size_t)(n1 + k) * HW] + gx : gx;
      g0 = -a;
      g1 = (cst - gx * xv) * (1.f - sg);
      if (fold) {
        q0 = 3.f * g0 * h0;
        q1 = 3.f * g1 * h1;
        g0 *= d.beta ? expf(d.beta[2 * k] * 3.f) : 1.f;
        g1 *= d.beta ? expf(d.beta[2 * k + 1] * 3.f) : 1.f;
      }
      th[(size_t)(2 * k) * HW] = g0;
      th[(size_t)(2 * k + 1) * HW] = g1;
    }
    if (fold) {
      const float a0 = wave_sum(g0), s0 = wave_sum(q0), a1 = wave_sum(g1), s1 = wave_sum(q1);
      if ((tid & 63) == 0) {
        float* q = part2 + (size_t)(2 * k) * 2 * NW + (tid >> 6);
        q[0] = a0; q[NW] = s0; q[2 * NW] = a1; q[3 * NW] = s1;
      }
    }
  }
  __syncthreads();
  // (4) parameter gradients: {dweight, dbias, dW} of the mix, {dbias, dscale} of the folded Conv2dZeros epilogue
  double* acc = d.acc + (long long)rep_of_block(d.nrep) * d.rep_stride;
  for (int i = tid; i < 2 * C; i += PB) {
    float s = 0.f;
    for (int w = 0; w < NW; ++w) s += part[i * NW + w];
    atomicAdd(&acc[i], (double)s);
  }
  if (fold)
    for (int i = tid; i < 4 * n2; i += PB) {
      double s = 0.0;
      for (int w = 0; w < NW; ++w) s += (double)part2[i * NW + w];
      atomicAdd(&d.bn_grad[(long long)rep_of_block(d.nrep) * d.rep_stride + i], s);
    }
  const int np = min(PB, HW - p0);
  for (int i = tid; i < C * C; i += PB) {
    const int oc = i / C, c = i % C;
    float s = 0.f;
    for (int q = 0; q < np; ++q) s += gs[oc * (PB + 1) + q] * xs[c * (PB + 1) + q];
    atomicAdd(&acc[2 * C + i], (double)s);
  }
}

static bool mix_ok(const pdes_conv_desc& d) {
  if (!(flow_common_ok(d) && d.upsample == PDES_OP_MIX && d.Cin == d.Cout && d.Cin <= 48 && d.Hin == d.Hout &&
        d.Win == d.Wout && d.x && d.x2 && d.out && d.p0 && d.p1))
    return false;
  if (d.flags & PDES_MIX_COUPLED)
    return !(d.flags & PDES_FLOW_FORWARD) && d.Cin >= 2 && d.h && d.h_ctot >= 2 * (d.Cin / 2) && (d.gamma || !d.beta);
  return true;
}

int flow_mix_forward(const pdes_conv_desc& d, hipStream_t st) {
  if (!mix_ok(d)) return PDES_EINVAL;
  const int C = d.Cin, HW = d.Hin * d.Win;
  const bool cpl = (d.flags & PDES_MIX_COUPLED) != 0;
  if (C <= 28) {
    constexpr int PB = 256;
    const size_t lds = (C * C + C * (PB + 1)) * sizeof(float);
    if (cpl) hipLaunchKernelGGL(flow_coupling_mix_kernel<PB>, dim3(cdiv(HW, PB), d.B), dim3(PB), lds, st, d);
    else hipLaunchKernelGGL(flow_mix_kernel<PB>, dim3(cdiv(HW, PB), d.B), dim3(PB), lds, st, d);
  } else {
    constexpr int PB = 64;
    const size_t lds = (C * C + C * (PB + 1)) * sizeof(float);
    if (cpl) hipLaunchKernelGGL(flow_coupling_mix_kernel<PB>, dim3(cdiv(HW, PB), d.B), dim3(PB), lds, st, d);
    else hipLaunchKernelGGL(flow_mix_kernel<PB>, dim3(cdiv(HW, PB), d.B), dim3(PB), lds, st, d);
  }
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

int flow_mix_backward(const pdes_conv_desc& d, hipStream_t st) {
  if (!mix_ok(d) || !d.g || !d.t_in || !d.acc || (d.flags & PDES_FLOW_FORWARD)) return PDES_EINVAL;
  const bool cpl = (d.flags & PDES_MIX_COUPLED) != 0;
  if (cpl && (!d.th || (d.gamma && !d.bn_grad))) return PDES_EINVAL;
  const int C = d.Cin, HW = d.Hin * d.Win, n2 = C / 2;
  if (C <= 28) {
    constexpr int PB = 256;
    const size_t lds = (C * C + 2 * C * (PB + 1) + 2 * C * (PB / 64) + (cpl ? 4 * n2 * (PB / 64) : 0)) * sizeof(float);
    if (cpl) hipLaunchKernelGGL(flow_mix_coupling_bwd_kernel<PB>, dim3(cdiv(HW, PB), d.B), dim3(PB), lds, st, d);
    else hipLaunchKernelGGL(flow_mix_bwd_kernel<PB>, dim3(cdiv(HW, PB), d.B), dim3(PB), lds, st, d);
  } else {
    constexpr int PB = 64;
    const size_t lds = (C * C + 2 * C * (PB + 1) + 2 * C * (PB / 64) + (cpl ? 4 * n2 * (PB / 64) : 0)) * sizeof(float);
    if (cpl) hipLaunchKernelGGL(flow_mix_coupling_bwd_kernel<PB>, dim3(cdiv(HW, PB), d.B), dim3(PB), lds, st, d);
    else hipLaunchKernelGGL(flow_mix_bwd_kernel<PB>, dim3(cdiv(HW, PB), d.B), dim3(PB), lds, st, d);
  }
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

// ------------------------------------------------------------------------------------------- UNSQUEEZE
// big (Cq, 2H, 2W) <-> small (4 Cq, H, W): big[c][i H + h][j W + w] = small[4 c + 2 i + j][h][w].
// to_big: big <- small (Squeeze.reverse); else small <- big (Squeeze.forward).  grid (ceil(Cq 4 HW / 256), B)
__global__ __launch_bounds__(256) void flow_requad_kernel(const float* __restrict__ src, int src_ctot, int src_coff,
                                                          float* __restrict__ dst, int dst_ctot, int dst_coff, int Cq, int H,
                                                          int W, int to_big, int accumulate) {
  const int b = blockIdx.y, HW = H * W;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Cq * 4 * HW) return;
  // enumerate in the order of the DESTINATION so that stores are coalesced
  int c, i, j, h, w;
  if (to_big) {
    const int X = idx % (2 * W), Y = (idx / (2 * W)) % (2 * H);
    c = idx / (4 * HW); i = Y / H; h = Y % H; j = X / W; w = X % W;
  } else {
    w = idx % W; h = (idx / W) % H;
    const int cs = idx / HW;
    c = cs >> 2; i = (cs >> 1) & 1; j = cs & 1;
  }
  const size_t big = ((size_t)c * 2 * H + (i * H + h)) * 2 * W + (j * W + w);
  const size_t small = ((size_t)(4 * c + 2 * i + j) * H + h) * W + w;
  if (to_big) {
    float* q = dst + ((size_t)b * dst_ctot + dst_coff) * 4 * HW + big;
    const float v = src[((size_t)b * src_ctot + src_coff) * HW + small];
    *q = accumulate ? *q + v : v;
  } else {
    float* q = dst + ((size_t)b * dst_ctot + dst_coff) * HW + small;
    const float v = src[((size_t)b * src_ctot + src_coff) * 4 * HW + big];
    *q = accumulate ? *q + v : v;
  }
}

static bool unsq_ok(const pdes_conv_desc& d) {
  if (!flow_common_ok(d) || d.upsample != PDES_OP_UNSQUEEZE || !d.x || !d.out) return false;
  if (d.flags & PDES_FLOW_FORWARD) return d.Cout == 4 * d.Cin && d.Hin == 2 * d.Hout && d.Win == 2 * d.Wout;
  return d.Cin == 4 * d.Cout && d.Hout == 2 * d.Hin && d.Wout == 2 * d.Win;
}

int flow_unsqueeze_forward(const pdes_conv_desc& d, hipStream_t st) {
  if (!unsq_ok(d)) return PDES_EINVAL;
  const bool sq = (d.flags & PDES_FLOW_FORWARD) != 0;
  const int Cq = sq ? d.Cin : d.Cout, H = sq ? d.Hout : d.Hin, W = sq ? d.Wout : d.Win;
  hipLaunchKernelGGL(flow_requad_kernel, dim3(cdiv(Cq * 4 * H * W, 256), d.B), dim3(256), 0, st, d.x, d.x_ctot, 0, d.out,
                     d.out_ctot, d.out_coff, Cq, H, W, sq ? 0 : 1, 0);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

int flow_unsqueeze_backward(const pdes_conv_desc& d, hipStream_t st) {
  if (!unsq_ok(d) || !d.g || !d.t_in || (d.flags & PDES_FLOW_FORWARD)) return PDES_EINVAL;
  hipLaunchKernelGGL(flow_requad_kernel, dim3(cdiv(d.Cout * 4 * d.Hin * d.Win, 256), d.B), dim3(256), 0, st, d.g, d.g_ctot,
                     d.g_coff, d.t_in, d.x_ctot, 0, d.Cout, d.Hin, d.Win, 0, d.t_accumulate);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

// ----------------------------------------------------------------------------------------------- GAUSS
// grid (ceil(HW / 256), B)
__global__ __launch_bounds__(256) void flow_gauss_kernel(pdes_conv_desc d) {
  __shared__ double red[4];
  const int HW = d.Hout * d.Wout, b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x, n = d.Cout;
  const bool fwd = (d.flags & PDES_FLOW_FORWARD) != 0;
  float lp = 0.f;
  if (p < HW) {
    const float* pr = d.x2 + (size_t)b * d.x2_ctot * HW + p;
    for (int c = 0; c < n; ++c) {
      const float m = pr[(size_t)c * HW];
      const float l = fminf(fmaxf(pr[(size_t)(n + c) * HW], PDES_LSD_MIN), PDES_LSD_MAX);
      float z;
      if (fwd) {
        z = d.x[((size_t)b * d.x_ctot + c) * HW + p];
        if (d.out) d.out[((size_t)b * d.out_ctot + d.out_coff + c) * HW + p] = (z - m) / expf(l);
      } else {
        z = m + expf(l) * d.p0[((size_t)b * n + c) * HW + p];
        d.out[((size_t)b * d.out_ctot + d.out_coff + c) * HW + p] = z;
      }
      const float dd = z - m;
      lp += -0.5f * (PDES_LOG2PI + l * 2.f + dd * dd / expf(l * 2.f));
    }
  }
  if (d.acc) block_add_logp(lp, d.acc, b, d.nrep, d.rep_stride, red);
}

__global__ __launch_bounds__(256) void flow_gauss_bwd_kernel(pdes_conv_desc d) {
  const int HW = d.Hout * d.Wout, b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x, n = d.Cout;
  if (p >= HW) return;
  const float cst = d.p1 ? d.p1[b] : 0.f;
  const bool detach = (d.flags & PDES_GAUSS_DETACH_LSD) != 0;
  const float* pr = d.x2 + (size_t)b * d.x2_ctot * HW + p;
  float* tp = d.t2 + (size_t)b * d.x2_ctot * HW + p;
  for (int c = 0; c < n; ++c) {
    const float m = pr[(size_t)c * HW], raw = pr[(size_t)(n + c) * HW];
    const float l = fminf(fmaxf(raw, PDES_LSD_MIN), PDES_LSD_MAX);
    const float sg = expf(l), s2 = expf(l * 2.f);
    const float eps = d.p0[((size_t)b * n + c) * HW + p];
    const float z = d.out[((size_t)b * d.out_ctot + d.out_coff + c) * HW + p];
    const float g = d.g[((size_t)b * d.g_ctot + d.g_coff + c) * HW + p];
    const float dd = z - m;
    const float gz = g - cst * dd / s2;               // dL/dz: downstream + the log-probability's own use of z
    tp[(size_t)c * HW] = gz + cst * dd / s2;          // mean: through z and explicitly (they cancel up to rounding, as in autograd)
    float gl = 0.f;
    if (!detach && raw >= PDES_LSD_MIN && raw <= PDES_LSD_MAX) gl = gz * sg * eps + cst * (dd * dd / s2 - 1.f);
    tp[(size_t)(n + c) * HW] = gl;
  }
}

static bool gauss_ok(const pdes_conv_desc& d) {
  return flow_common_ok(d) && d.upsample == PDES_OP_GAUSS && d.Hin == d.Hout && d.Win == d.Wout && d.x2 &&
         d.x2_ctot >= 2 * d.Cout && ((d.flags & PDES_FLOW_FORWARD) ? d.x != nullptr : (d.out && d.p0));
}

int flow_gauss_forward(const pdes_conv_desc& d, hipStream_t st) {
  if (!gauss_ok(d)) return PDES_EINVAL;
  hipLaunchKernelGGL(flow_gauss_kernel, dim3(cdiv(d.Hout * d.Wout, 256), d.B), dim3(256), 0, st, d);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

int flow_gauss_backward(const pdes_conv_desc& d, hipStream_t st) {
  if (!gauss_ok(d) || !d.g || !d.t2 || (d.flags & PDES_FLOW_FORWARD)) return PDES_EINVAL;
  hipLaunchKernelGGL(flow_gauss_bwd_kernel, dim3(cdiv(d.Hout * d.Wout, 256), d.B), dim3(256), 0, st, d);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

// ------------------------------------------------------------------------- parameter side of the flow
// one workgroup (256 threads) per invertible 1x1 convolution + ActNorm; C <= 48
__global__ __launch_bounds__(256) void flow_prepare_kernel(const pdes_flow_item* __restrict__ items, int need_inverse) {
  const pdes_flow_item it = items[blockIdx.x];
  const int C = it.C, tid = threadIdx.x;
  __shared__ float A[48 * 48], Bm[48 * 48];
  __shared__ double aug[48][2 * 48 + 1];       // Gauss-Jordan: [W | I]
  __shared__ double sh_ld;
  __shared__ int sh_piv;
  if (it.lu) {
    // Bm = (l * l_mask + I) (u * u_mask + diag(exp(log_s) sign_s)),  W = p Bm.  The three factors are staged in LDS first
    // (the Gauss-Jordan array is free until the inverse is asked for): with the factors read from global memory inside the
    // product loops a thread paid one memory round trip per term, 31 us for the launch at the head of every step
    float* Lf = reinterpret_cast<float*>(&aug[0][0]);
    float* Uf = Lf + 48 * 48;
    float* Pf = Uf + 48 * 48;
    static_assert(sizeof(aug) >= 3 * 48 * 48 * sizeof(float), "staging area");
    for (int i = tid; i < C * C; i += 256) {
      const int r = i / C, c = i % C;
      Lf[i] = c < r ? it.l[i] : (c == r ? 1.f : 0.f);
      Uf[i] = r < c ? it.u[i] : (r == c ? expf(it.log_s[r]) * it.sign_s[r] : 0.f);
      Pf[i] = it.p[i];
    }
    __syncthreads();
    for (int i = tid; i < C * C; i += 256) {
      const int r = i / C, c = i % C;
      float s = 0.f;
      for (int k = 0; k < C; ++k) s += Lf[r * C + k] * Uf[k * C + c];
      Bm[i] = s;
    }
    __syncthreads();
    for (int i = tid; i < C * C; i += 256) {
      const int r = i / C, c = i % C;
      float s = 0.f;
      for (int k = 0; k < C; ++k) s += Pf[r * C + k] * Bm[k * C + c];
      A[i] = s;
    }
  } else {
    for (int i = tid; i < C * C; i += 256) A[i] = it.weight[i];
  }
  __syncthreads();
  for (int i = tid; i < C * C; i += 256) it.W[i] = A[i];
  double logdet_w = 0.0;
  const bool inv = need_inverse || !it.lu;
  if (inv) {
    for (int i = tid; i < C * 2 * C; i += 256) {
      const int r = i / (2 * C), c = i % (2 * C);
      aug[r][c] = c < C ? (double)A[r * C + c] : (c - C == r ? 1.0 : 0.0);
    }
    if (tid == 0) sh_ld = 0.0;
    __syncthreads();
    for (int k = 0; k < C; ++k) {
      if (tid == 0) {                      // partial pivoting
        int best = k;
        double mx = fabs(aug[k][k]);
        for (int r = k + 1; r < C; ++r)
          if (fabs(aug[r][k]) > mx) { mx = fabs(aug[r][k]); best = r; }
        sh_piv = best;
        sh_ld += log(mx);
      }
      __syncthreads();
      const int pv = sh_piv;
      if (pv != k)
        for (int c = tid; c < 2 * C; c += 256) { const double t = aug[k][c]; aug[k][c] = aug[pv][c]; aug[pv][c] = t; }
      __syncthreads();
      const double piv = aug[k][k];
      __syncthreads();
      for (int c = tid; c < 2 * C; c += 256) aug[k][c] /= piv;
      __syncthreads();
      for (int i = tid; i < C * 2 * C; i += 256) {
        const int r = i / (2 * C), c = i % (2 * C);
        if (r != k && c > k) aug[r][c] -= aug[r][k] * aug[k][c];
      }
      __syncthreads();
      for (int r = tid; r < C; r += 256)
        if (r != k) aug[r][k] = 0.0;
      __syncthreads();
    }
    if (it.Winv)
      for (int i = tid; i < C * C; i += 256) it.Winv[i] = (float)aug[i / C][C + i % C];
    logdet_w = sh_ld;
  }
  // log-determinant: the per-channel terms by one thread each (loads in flight together), the sums in the reference's order
  __shared__ float ld_an[48], ld_ls[48];
  __syncthreads();
  if (tid < C) {
    ld_an[tid] = logf(fabsf(it.an_weight[tid]));
    ld_ls[tid] = it.lu ? it.log_s[tid] : 0.f;
  }
  __syncthreads();
  if (tid == 0) {
    double an = 0.0, ls = 0.0;
    for (int c = 0; c < C; ++c) an += (double)ld_an[c];
    if (it.lu) {
      float s = 0.f;
      for (int c = 0; c < C; ++c) s += ld_ls[c];
      ls = (double)s;
    } else {
      ls = (double)(float)logdet_w;
    }
    *it.logdet = (double)it.HW * (an - ls);
  }
}

__global__ __launch_bounds__(256) void flow_param_grads_kernel(const pdes_flow_item* __restrict__ items,
                                                               const float* __restrict__ glogp, int B, int nrep, long long rs) {
  const pdes_flow_item it = items[blockIdx.x];
  const int C = it.C, tid = threadIdx.x;
  __shared__ float dW[48 * 48], M1[48 * 48], Lf[48 * 48], Uf[48 * 48], Pf[48 * 48];
  __shared__ float red[4];
  float cb = 0.f;
  if (glogp)
    for (int b = tid; b < B; b += 256) cb += glogp[b];
  cb = wave_sum(cb);
  if ((tid & 63) == 0) red[tid >> 6] = cb;
  for (int i = tid; i < C * C; i += 256) {
    dW[i] = (float)rep_sum(it.acc, 2 * C + i, nrep, rs);
    if (it.lu) Pf[i] = it.p[i];                   // (read from global memory inside the product loop: one round trip per term)
  }
  __syncthreads();
  const float cB = (red[0] + red[1]) + (red[2] + red[3]);
  const float hw = (float)it.HW;
  for (int c = tid; c < C; c += 256) {
    it.dan_weight[c] += (float)rep_sum(it.acc, c, nrep, rs) + cB * hw / it.an_weight[c];
    it.dan_bias[c] += (float)rep_sum(it.acc, C + c, nrep, rs);
  }
  if (!it.lu) {
    for (int i = tid; i < C * C; i += 256) it.dweight[i] += dW[i] - cB * hw * it.Winv[(i % C) * C + i / C];
    return;
  }
  for (int i = tid; i < C * C; i += 256) {
    const int r = i / C, c = i % C;
    Lf[i] = c < r ? it.l[i] : (c == r ? 1.f : 0.f);
    Uf[i] = r < c ? it.u[i] : (r == c ? expf(it.log_s[r]) * it.sign_s[r] : 0.f);
    float s = 0.f;
    for (int k = 0; k < C; ++k) s += Pf[k * C + r] * dW[k * C + c];         // M1 = P^T dW
    M1[i] = s;
  }
  __syncthreads();
  for (int i = tid; i < C * C; i += 256) {
    const int r = i / C, c = i % C;
    if (c < r) {                              // dl = (M1 Uf^T) * l_mask
      float s = 0.f;
      for (int k = 0; k < C; ++k) s += M1[r * C + k] * Uf[c * C + k];
      it.dl[i] += s;
    } else {                                  // dUf = Lf^T M1: strictly upper -> du, diagonal -> dlog_s
      float s = 0.f;
      for (int k = 0; k < C; ++k) s += Lf[k * C + r] * M1[k * C + c];
      if (c > r) it.du[i] += s;
      else it.dlog_s[r] += s * expf(it.log_s[r]) * it.sign_s[r] - cB * hw;
    }
  }
}

__global__ __launch_bounds__(256) void flow_logp_kernel(const double* __restrict__ acc, const double* __restrict__ logdet,
                                                        int n_layers, float* __restrict__ logp, int B, int nrep, long long rs) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  double s = rep_sum(acc, b, nrep, rs);
  // the reference adds each layer's scalar log-determinant to the running fp32 sum; the order of fp32 additions
  // differs here (fp64 accumulation), within ~1e-7 relative of it
  for (int k = 0; k < n_layers; ++k) s += logdet[k];
  logp[b] = (float)s;
}

}  // namespace pdes

using namespace pdes;

extern "C" int pdes_flow_prepare(const pdes_flow_item* items, int n, int need_inverse, void* stream) {
  if (!items || n <= 0) return PDES_EINVAL;          // (items live in device memory: C <= 48 is the caller's contract)
  hipLaunchKernelGGL(flow_prepare_kernel, dim3(n), dim3(256), 0, static_cast<hipStream_t>(stream), items, need_inverse);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

extern "C" int pdes_flow_param_grads(const pdes_flow_item* items, int n, const float* glogp, int B, int nrep,
                                     long long rep_stride, void* stream) {
  if (!items || n <= 0 || B <= 0 || nrep != PDES_NREP) return PDES_EINVAL;
  hipLaunchKernelGGL(flow_param_grads_kernel, dim3(n), dim3(256), 0, static_cast<hipStream_t>(stream), items, glogp, B, nrep,
                     rep_stride);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

extern "C" int pdes_flow_logp(const double* acc, const double* logdet, int n_layers, float* logp, int B, int nrep,
                              long long rep_stride, void* stream) {
  if (!acc || !logp || B <= 0 || n_layers < 0 || (n_layers > 0 && !logdet) || nrep != PDES_NREP) return PDES_EINVAL;
  hipLaunchKernelGGL(flow_logp_kernel, dim3(cdiv(B, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), acc, logdet, n_layers,
                     logp, B, nrep, rep_stride);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}
