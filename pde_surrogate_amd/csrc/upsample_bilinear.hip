// `--upsample bilinear`: UpsamplingBilinear2d(scale 2, align_corners=True) between BN-ReLU and the 3x3 convolution
// of the up-transitions (reference models/codec.py:33-40, selected at :141-144 and :176-179).
//
// Unlike nearest x2 (folded into the convolution as a sub-pixel 2x2 kernel, conv_mfma_up.hip) the align_corners
// grid has the irrational-looking ratio (H-1)/(2H-1): every output pixel mixes four inputs with its own weights, so
// the resampled activation U = up(relu(bn(x))) is MATERIALISED once (4x the low-resolution map) by the descriptor
// op below and the following convolution reads U through an identity BatchNorm (gamma 1, beta 0, statistics
// {0, n (1 - eps)} -> mean 0, invstd 1; U >= 0 so the fused ReLU is the identity too) on the ordinary kernels.
//   forward  : out[b, c, i, j] = sum of 4 taps of relu((x - mean) gamma invstd + beta)
//   backward : dL/dz = up^T(dL/dU); T_x (+)= gamma dL/dz 1[z > 0]; dgamma, dbeta, finished-channel sums of T
// HBM-bound, one thread per pixel, lanes = consecutive pixels of a channel plane (coalesced).
#include "pdes_common.h"
#include "../../include/pdes_hip.h"

namespace pdes {

struct BnK { float mean, invstd, gamma, beta; };

__device__ __forceinline__ BnK up_bn_coef(const pdes_conv_desc& d, int c) {
  BnK o;
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

// source index / weights of output index i (ATen's align_corners rule, float arithmetic like the reference)
__device__ __forceinline__ void src_of(int i, int n_in, int n_out, int& i0, int& i1, float& w0, float& w1) {
  const float ratio = n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.f;
  const float s = ratio * (float)i;
  i0 = (int)s;
  if (i0 > n_in - 1) i0 = n_in - 1;
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  w1 = s - (float)i0;
  w0 = 1.f - w1;
}

// grid (ceil(Hout*Wout/256), C, B)
__global__ __launch_bounds__(256) void upsample_bilinear_fwd_kernel(pdes_conv_desc d) {
  __shared__ float cf[3];
  const int c = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  if (tid == 0) {
    const BnK k = up_bn_coef(d, c);
    cf[0] = k.mean; cf[1] = k.gamma * k.invstd; cf[2] = k.beta;
    // identity-BatchNorm statistics of the OUTPUT channel for the convolution that reads it (replica 0 only)
    if (d.out_stats && blockIdx.x == 0 && b == 0) {
      const double n = (double)d.B * d.Hout * d.Wout;
      d.out_stats[2 * (d.out_coff + c)] = 0.0;
      d.out_stats[2 * (d.out_coff + c) + 1] = n * (1.0 - (double)d.eps);
    }
  }
  __syncthreads();
  const int HWo = d.Hout * d.Wout, HWi = d.Hin * d.Win;
  const int p = blockIdx.x * 256 + tid;
  if (p >= HWo) return;
  const int oy = p / d.Wout, ox = p % d.Wout;
  int y0, y1, x0, x1;
  float wy0, wy1, wx0, wx1;
  src_of(oy, d.Hin, d.Hout, y0, y1, wy0, wy1);
  src_of(ox, d.Win, d.Wout, x0, x1, wx0, wx1);
  const float* xc = d.x + ((size_t)b * d.x_ctot + c) * HWi;
  const float mean = cf[0], scale = cf[1], beta = cf[2];
  auto z = [&](int y, int x) { return fmaxf(0.f, (xc[y * d.Win + x] - mean) * scale + beta); };
  const float top = wx0 * z(y0, x0) + wx1 * z(y0, x1);
  const float bot = wx0 * z(y1, x0) + wx1 * z(y1, x1);
  d.out[((size_t)b * d.out_ctot + d.out_coff + c) * HWo + p] = wy0 * top + wy1 * bot;
}

// grid (ceil(Hin*Win/256), C, B): one thread = one LOW-resolution pixel
__global__ __launch_bounds__(256) void upsample_bilinear_bwd_kernel(pdes_conv_desc d) {
  __shared__ float cf[4];
  __shared__ double red[4][4];
  const int c = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  if (tid == 0) {
    const BnK k = up_bn_coef(d, c);
    cf[0] = k.mean; cf[1] = k.invstd; cf[2] = k.gamma; cf[3] = k.beta;
  }
  __syncthreads();
  const int HWo = d.Hout * d.Wout, HWi = d.Hin * d.Win;
  const int p = blockIdx.x * 256 + tid;
  const bool active = p < HWi;
  const int iy = active ? p / d.Win : 0, ix = active ? p % d.Win : 0;
  const float* gc = d.g + ((size_t)b * d.g_ctot + d.g_coff + c) * HWo;
  float acc = 0.f;
  if (active) {
    // output rows / columns whose two taps can touch (iy, ix): source index of i lies in [i/2 - 1/2, i/2]
    const int ylo = max(0, 2 * iy - 2), yhi = min(d.Hout - 1, 2 * iy + 3);
    const int xlo = max(0, 2 * ix - 2), xhi = min(d.Wout - 1, 2 * ix + 3);
    for (int oy = ylo; oy <= yhi; ++oy) {
      int y0, y1; float wy0, wy1;
      src_of(oy, d.Hin, d.Hout, y0, y1, wy0, wy1);
      const float wy = (y0 == iy ? wy0 : 0.f) + (y1 == iy ? wy1 : 0.f);
      if (wy == 0.f) continue;
      for (int ox = xlo; ox <= xhi; ++ox) {
        int x0, x1; float wx0, wx1;
        src_of(ox, d.Win, d.Wout, x0, x1, wx0, wx1);
        const float wx = (x0 == ix ? wx0 : 0.f) + (x1 == ix ? wx1 : 0.f);
        if (wx != 0.f) acc += wy * wx * gc[oy * d.Wout + ox];
      }
    }
  }
  float dg = 0.f, db = 0.f, st = 0.f, sx = 0.f;
  if (active) {
    const float mean = cf[0], invstd = cf[1], gamma = cf[2], beta = cf[3];
    const size_t idx = ((size_t)b * d.x_ctot + c) * HWi + p;
    const float x = d.x[idx];
    const float y = (x - mean) * (gamma * invstd) + beta;       // same expression as the forward
    const float xh = (x - mean) * invstd;
    const float dyv = y > 0.f ? acc : 0.f;
    db = dyv;
    dg = dyv * xh;
    float t = gamma * dyv;
    if (d.t_accumulate) t += d.t_in[idx];
    d.t_in[idx] = t;
    if (c >= d.final_c0 && c < d.final_c1) { st = t; sx = t * xh; }
  }
  const int wave = tid >> 6, lane = tid & 63;
  const float r0 = wave_sum(dg), r1 = wave_sum(db), r2 = wave_sum(st), r3 = wave_sum(sx);
  if (lane == 0) { red[wave][0] = r0; red[wave][1] = r1; red[wave][2] = r2; red[wave][3] = r3; }
  __syncthreads();
  if (tid < 4) {
    const double t = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    const long long rep = (long long)rep_of_block(d.nrep) * d.rep_stride;
    if (tid < 2) atomicAdd(&d.bn_grad[rep + 2 * c + tid], t);
    else if (c >= d.final_c0 && c < d.final_c1) atomicAdd(&d.t_stats[rep + 2 * c + (tid - 2)], t);
  }
}

static bool up_desc_ok(const pdes_conv_desc& d) {
  return d.upsample == PDES_UPSAMPLE_BILINEAR_OP && d.ksize == 0 && d.Cin == d.Cout && d.Hout == 2 * d.Hin &&
         d.Wout == 2 * d.Win && d.has_bn && d.x && d.gamma && d.beta && d.nrep == PDES_NREP;
}

int upsample_bilinear_forward(const pdes_conv_desc& d, hipStream_t st) {
  if (!up_desc_ok(d) || !d.out) return PDES_EINVAL;
  if (!d.eval_mode && !d.x_stats) return PDES_EINVAL;
  dim3 grid(cdiv(d.Hout * d.Wout, 256), d.Cin, d.B);
  hipLaunchKernelGGL(upsample_bilinear_fwd_kernel, grid, dim3(256), 0, st, d);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

int upsample_bilinear_backward(const pdes_conv_desc& d, hipStream_t st) {
  if (!up_desc_ok(d) || !d.g || !d.t_in || !d.bn_grad || !d.t_stats || !d.x_stats || d.eval_mode) return PDES_EINVAL;
  dim3 grid(cdiv(d.Hin * d.Win, 256), d.Cin, d.B);
  hipLaunchKernelGGL(upsample_bilinear_bwd_kernel, grid, dim3(256), 0, st, d);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

}  // namespace pdes

// ------------------------------------------------------------------------------------------------------------------
// `--drop-rate > 0`: nn.Dropout2d after a convolution (reference codec.py:70-71 dense layer, :111-120 / :134-150
// transitions, :172-173 last decoding): whole channels of a sample are zeroed, the rest scaled by 1/(1-p).
// PDES_OP_CHANNEL_MASK descriptor: out[b, out_coff + c] *= mask[b, c] IN PLACE (mask = `w`, B x Cout floats holding 0 or
// 1/(1-p), drawn by the caller), accumulating the batch statistics of the masked channels (the convolution before it
// runs with out_stats = NULL).  Backward (pdes_conv_backward_data): g[b, g_coff + c] *= mask[b, c] in place -- the
// BatchNorm-backward finalize of these channels belongs to THIS descriptor (fin_*), the convolution's is NULL.
namespace pdes {

// grid (ceil(HW / 1024), C, B), 256 threads x float4
__global__ __launch_bounds__(256) void channel_mask_kernel(float* __restrict__ buf, int ctot, int coff, int HW,
                                                           const float* __restrict__ mask, int C, double* __restrict__ stats,
                                                           int nrep, long long rep_stride) {
  __shared__ double red[4][2];
  const int c = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const float m = mask[b * C + c];
  float4* p = reinterpret_cast<float4*>(buf + ((size_t)b * ctot + coff + c) * HW);
  const int i = blockIdx.x * 256 + tid;
  float s = 0.f, q = 0.f;
  if (i < HW / 4) {
    float4 v = p[i];
    v.x *= m; v.y *= m; v.z *= m; v.w *= m;
    p[i] = v;
    s = (v.x + v.y) + (v.z + v.w);
    q = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (!stats) return;
  const float ws = wave_sum(s), wq = wave_sum(q);
  if ((tid & 63) == 0) { red[tid >> 6][0] = ws; red[tid >> 6][1] = wq; }
  __syncthreads();
  if (tid < 2) {
    const double t = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    atomicAdd(&stats[(long long)rep_of_block(nrep) * rep_stride + 2 * (coff + c) + tid], t);
  }
}

static bool mask_desc_ok(const pdes_conv_desc& d) {
  return d.upsample == PDES_OP_CHANNEL_MASK && d.ksize == 0 && d.Cin == d.Cout && d.Hout == d.Hin && d.Wout == d.Win &&
         d.w && (d.Hout * d.Wout) % 4 == 0 && d.nrep == PDES_NREP;
}

int channel_mask_forward(const pdes_conv_desc& d, hipStream_t st) {
  if (!mask_desc_ok(d) || !d.out || !aligned16(d.out)) return PDES_EINVAL;
  const int HW = d.Hout * d.Wout;
  hipLaunchKernelGGL(channel_mask_kernel, dim3(cdiv(HW / 4, 256), d.Cout, d.B), dim3(256), 0, st, d.out, d.out_ctot,
                     d.out_coff, HW, d.w, d.Cout, d.out_stats, d.nrep, d.rep_stride);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

int channel_mask_backward(const pdes_conv_desc& d, hipStream_t st) {
  if (!mask_desc_ok(d) || !d.g || !aligned16(d.g)) return PDES_EINVAL;
  const int HW = d.Hout * d.Wout;
  hipLaunchKernelGGL(channel_mask_kernel, dim3(cdiv(HW / 4, 256), d.Cout, d.B), dim3(256), 0, st, const_cast<float*>(d.g),
                     d.g_ctot, d.g_coff, HW, d.w, d.Cout, (double*)nullptr, d.nrep, d.rep_stride);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

}  // namespace pdes
