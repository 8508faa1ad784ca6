// Small table-driven kernels around the convolutions: BatchNorm backward finalisation, running
// statistics, fp64->fp32 parameter gradients, weight packing, and the flat Adam step.
// References: nn.BatchNorm2d semantics as used by models/codec.py (train: batch stats, biased var
// for normalisation, unbiased var into running_var, momentum 0.1, eps 1e-5);
// torch.optim.Adam as called in train_codec_mixed_residual.py:151-152,239.
#include <stdlib.h>
#include <hip/hip_ext.h>
#include "pdes_common.h"
#include "pdes_options.h"
#include "../../include/pdes_hip.h"
#include "pack_kernels.h"

namespace pdes {

// T -> dL/dx in place: g = invstd * (T - mean(T) - xhat * mean(T*xhat))
// grid: (ceil(HW/256)?, c1-c0, B) -> use flat: x = HW tiles, y = channel, z = sample
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(float* __restrict__ t, const float* __restrict__ x,
                                                              const double* __restrict__ x_stats,
                                                              const double* __restrict__ t_stats, int B, int ctot,
                                                              int c0, int HW, float eps, int nrep, long long rs, int early,
                                                              const float* __restrict__ add, const float* __restrict__ coef) {
  const int c = c0 + blockIdx.y, b = blockIdx.z;
  // {mean, invstd} of the channel from the table its forward consumers published (batch_mean_invstd), when it is there:
  // no replica sums of x, no fp64 square root / division on the one thread everybody waits for
  float2 ce = make_float2(0.f, 0.f);
  if (coef) ce = reinterpret_cast<const float2*>(coef)[c];
  const bool have = ce.y > 0.f;
  __shared__ float sc[4];
  __shared__ double sums[4];
  const size_t base = ((size_t)b * ctot + c) * HW;
  // HW is a multiple of 4 for every supported feature map (>= 8x8); vectorise when aligned.  The first element's
  // loads do not depend on the statistics: issue them BEFORE the reduce -> fp64 -> LDS chain so that the two
  // latencies overlap (this kernel runs 27 times per step on the critical stream and is pure latency)
  const bool vec = (HW & 3) == 0;
  const int i0 = blockIdx.x * 256 + threadIdx.x;
  float4 tv0 = make_float4(0.f, 0.f, 0.f, 0.f), xv0 = tv0;
  if (vec && early && i0 < HW / 4) {
    tv0 = reinterpret_cast<const float4*>(t + base)[i0];
    xv0 = reinterpret_cast<const float4*>(x + base)[i0];
  }
  if (threadIdx.x < 64) {
    // 4 quantities x PDES_NREP (<= 16) replicas: one load per lane, then a 16-lane shuffle reduction
    const int q = threadIdx.x >> 4, r = threadIdx.x & 15;
    const double* src = (q < 2 ? x_stats : t_stats) + (long long)r * rs + 2 * c + (q & 1);
    double v = (r < PDES_NREP && !(have && q < 2)) ? *src : 0.0;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 16);
    if (r == 0) sums[q] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double inv_n = 1.0 / ((double)B * HW);
    if (have) {
      sc[0] = ce.x;
      sc[1] = ce.y;
    } else {
      const double m = sums[0] * inv_n;
      double var = sums[1] * inv_n - m * m;
      var = var < 0.0 ? 0.0 : var;
      sc[0] = (float)m;
      sc[1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    sc[2] = (float)(sums[2] * inv_n);
    sc[3] = (float)(sums[3] * inv_n);
  }
  __syncthreads();
  const float mean = sc[0], invstd = sc[1], m1 = sc[2], m2 = sc[3];
  if (vec) {
    float4* t4 = reinterpret_cast<float4*>(t + base);
    const float4* x4 = reinterpret_cast<const float4*>(x + base);
    for (int i = i0; i < HW / 4; i += gridDim.x * 256) {
      float4 tv = (early && i == i0) ? tv0 : t4[i];
      const float4 xv = (early && i == i0) ? xv0 : x4[i];
      tv.x = invstd * (tv.x - m1 - (xv.x - mean) * invstd * m2);
      tv.y = invstd * (tv.y - m1 - (xv.y - mean) * invstd * m2);
      tv.z = invstd * (tv.z - m1 - (xv.z - mean) * invstd * m2);
      tv.w = invstd * (tv.w - m1 - (xv.w - mean) * invstd * m2);
      if (add) {            // gradient from consumers that read the activation without a BatchNorm (pdes_conv_desc.g_add)
        const float4 a = reinterpret_cast<const float4*>(add + base)[i];
        tv.x += a.x; tv.y += a.y; tv.z += a.z; tv.w += a.w;
      }
      t4[i] = tv;
    }
  } else {
    for (int i = i0; i < HW; i += gridDim.x * 256)
      t[base + i] = invstd * (t[base + i] - m1 - (x[base + i] - mean) * invstd * m2) + (add ? add[base + i] : 0.f);
  }
}

// (Cout,Cin,kk) -> w_fwd (Cin,kk,cout_pad) and w_bwd (Cout,kk,cin_pad); pads are pre-zeroed once.
__global__ __launch_bounds__(256) void pack_weights_kernel(const pdes_pack_item* __restrict__ items) {
  pack_direct_item(items[blockIdx.y], blockIdx.x, gridDim.x);
}

// every packed image of the network in ONE launch: blockIdx.y walks the four tables back to back
__global__ __launch_bounds__(256) void pack_all_kernel(const pdes_pack_item* __restrict__ a, int na,
                                                       const pdes_mfma_pack_item* __restrict__ m, int nm,
                                                       const pdes_up_pack_item* __restrict__ u, int nu,
                                                       const pdes_b3_pack_item* __restrict__ b3, int nb,
                                                       const pdes_b3up_pack_item* __restrict__ bu) {
  const int y = blockIdx.y;
  if (y < na) pack_direct_item(a[y], blockIdx.x, gridDim.x);
  else if (y < na + nm) pack_mfma_item(m[y - na], blockIdx.x, gridDim.x);
  else if (y < na + nm + nu) pack_up_item(u[y - na - nm], blockIdx.x, gridDim.x);
  else if (y < na + nm + nu + nb) pack_b3_item(b3[y - na - nm - nu], blockIdx.x, gridDim.x);
  else pack_b3up_item(bu[y - na - nm - nu - nb], blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(256) void bn_update_running_kernel(const pdes_bn_item* __restrict__ items, float momentum,
                                                                int nrep, long long rs) {
  const pdes_bn_item it = items[blockIdx.y];
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c == 0 && it.num_batches_tracked) *it.num_batches_tracked += 1;
  if (c >= it.C) return;
  const double n = (double)it.count;
  const double m = rep_sum(it.x_stats, 2 * c, nrep, rs) / n;
  double var = rep_sum(it.x_stats, 2 * c + 1, nrep, rs) / n - m * m;
  var = var < 0.0 ? 0.0 : var;
  const double unbiased = it.count > 1 ? var * n / (n - 1.0) : var;
  it.run_mean[c] = (float)((1.0 - momentum) * it.run_mean[c] + momentum * m);
  it.run_var[c] = (float)((1.0 - momentum) * it.run_var[c] + momentum * unbiased);
}

__global__ __launch_bounds__(256) void bn_param_grads_kernel(const pdes_bn_item* __restrict__ items, int nrep, long long rs) {
  const pdes_bn_item it = items[blockIdx.y];
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= it.C) return;
  it.dgamma[c] += (float)rep_sum(it.bn_grad, 2 * c, nrep, rs);
  it.dbeta[c] += (float)rep_sum(it.bn_grad, 2 * c + 1, nrep, rs);
}

// End-of-step table kernel of the training loop: per BatchNorm layer the parameter gradients and (optionally) the
// running statistics, plus -- in the extra block row -- the fixed-order fp64 reduction of the per-image loss partials
// into terms[5] = {total, const, cont, dir, neu} and their per-epoch accumulator.  One launch instead of four small
// ones (bn_update_running, darcy_loss_finalize, a tensor add, bn_param_grads) on the serial stream.
struct LossTail { const float* partials; float* terms; double* accum; int B; double inv_n, inv_dir, inv_neu; float w[4]; };
__global__ __launch_bounds__(256) void step_tail_kernel(const pdes_bn_item* __restrict__ items, int n_bn, float momentum,
                                                        int update_running, LossTail lt, int nrep, long long rs) {
  if ((int)blockIdx.y == n_bn) {
    if (blockIdx.x != 0) return;
    __shared__ double sh[4][4];
    double acc[4] = {0, 0, 0, 0};
    for (int b = threadIdx.x; b < lt.B; b += 256) {
      const float4 v = reinterpret_cast<const float4*>(lt.partials)[b];
      acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = wave_sum(acc[i]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) sh[threadIdx.x >> 6][i] = acc[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double t[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) t[i] = sh[0][i] + sh[1][i] + sh[2][i] + sh[3][i];
      const double lc = t[0] * lt.inv_n, lcn = t[1] * lt.inv_n, ld = t[2] * lt.inv_dir, ln = t[3] * lt.inv_neu;
      // the same fp32 values darcy_loss_finalize writes; the accumulator adds exactly those
      const float o[5] = {(float)(lt.w[0] * lc + lt.w[1] * lcn + lt.w[2] * ld + lt.w[3] * ln), (float)lc, (float)lcn,
                          (float)ld, (float)ln};
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        if (lt.terms) lt.terms[i] = o[i];
        if (lt.accum) lt.accum[i] += (double)o[i];
      }
    }
    return;
  }
  const pdes_bn_item it = items[blockIdx.y];
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (update_running && c == 0 && it.num_batches_tracked) *it.num_batches_tracked += 1;
  if (c >= it.C) return;
  it.dgamma[c] += (float)rep_sum(it.bn_grad, 2 * c, nrep, rs);
  it.dbeta[c] += (float)rep_sum(it.bn_grad, 2 * c + 1, nrep, rs);
  if (update_running) {
    const double n = (double)it.count;
    const double m = rep_sum(it.x_stats, 2 * c, nrep, rs) / n;
    double var = rep_sum(it.x_stats, 2 * c + 1, nrep, rs) / n - m * m;
    var = var < 0.0 ? 0.0 : var;
    const double unbiased = it.count > 1 ? var * n / (n - 1.0) : var;
    it.run_mean[c] = (float)((1.0 - momentum) * it.run_mean[c] + momentum * m);
    it.run_var[c] = (float)((1.0 - momentum) * it.run_var[c] + momentum * unbiased);
  }
}

struct AdamHyper { float lr, b1, b2, eps, wd, bc1, bc2_sqrt; };

// torch.optim.Adam (non-amsgrad, non-maximize); the host computes the bias corrections in
// double exactly as torch does: bc1 = 1 - beta1^step, bc2_sqrt = sqrt(1 - beta2^step)
template <bool ZERO>
__device__ __forceinline__ void adam_update(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                            float* __restrict__ v, const AdamHyper& h, float gscale, long long n) {
  const float step_size = h.lr / h.bc1;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float gi = g[i] * gscale;
    if (ZERO) g[i] = 0.f;          // the next step accumulates into a clean buffer: no separate fill launch
    const float pi = p[i];
    if (h.wd != 0.f) gi += h.wd * pi;
    const float mi = m[i] + (1.f - h.b1) * (gi - m[i]);          // lerp, as torch does
    const float vi = h.b2 * v[i] + (1.f - h.b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / h.bc2_sqrt + h.eps;
    p[i] = pi - step_size * (mi / denom);
  }
}

// hyper-parameters in device memory (a captured hipGraph replays with new values) ...
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   const float* __restrict__ hyper, float gscale, long long n) {
  const AdamHyper h = {hyper[0], hyper[1], hyper[2], hyper[3], hyper[4], hyper[5], hyper[6]};
  adam_update<false>(p, const_cast<float*>(g), m, v, h, gscale, n);
}
// ... or by value in the kernel arguments (eager steps: nothing to copy, nothing to race with)
template <bool ZERO>
__global__ __launch_bounds__(256) void adam_kernel_v(float* __restrict__ p, float* __restrict__ g,
                                                     float* __restrict__ m, float* __restrict__ v, AdamHyper h,
                                                     float gscale, long long n, double* __restrict__ clear, long long nclear) {
  adam_update<ZERO>(p, g, m, v, h, gscale, n);
  // the fp64 statistics arena of the step that just ended (its last reader, pdes_step_tail, ran before this kernel):
  // cleared here so that the next forward needs no fill launch of its own
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nclear; i += (long long)gridDim.x * 256) clear[i] = 0.0;
}

}  // namespace pdes

using namespace pdes;

extern "C" int pdes_stat_replicas(void) { return PDES_NREP; }

namespace pdes {
// `done` (nullable): an event that completes WITH this kernel -- it rides on the dispatch packet's own completion signal
// (hipExtLaunchKernelGGL stop event) instead of a barrier packet behind it.  pdes_backward forks its weight-gradient
// stream from it: measured with tools/archive/proto/event_gap.hip, hipEventRecord costs the NEXT kernel of the recording
// stream 3-5 us (27 times per step on the finalize -> data-gradient chain), the completion-signal form costs nothing.
int bn_backward_finalize_launch(const pdes_context* ctx, float* t, const float* x, const double* x_stats,
                                const double* t_stats, int B, int ctot, int c0, int c1, int HW, float eps, int nrep,
                                long long rep_stride, hipStream_t st, hipEvent_t done, const float* add, const float* coef) {
  if (!t || !x || !x_stats || !t_stats || B <= 0 || c1 <= c0 || c0 < 0 || c1 > ctot || HW <= 0 || nrep != PDES_NREP) return PDES_EINVAL;
  if (!aligned16(t) || !aligned16(x)) return PDES_EALIGN;
  dim3 grid(cdiv(cdiv(HW, 4), 256), c1 - c0, B), block(256);
  OptScope scope(ctx);
  const int early = 1;        // the T / x loads are issued before the statistics chain (round 1: 2.056 -> 2.040 ms per step)
  if (done)
    hipExtLaunchKernelGGL(bn_bwd_finalize_kernel, grid, block, 0, st, nullptr, done, 0, t, x, x_stats, t_stats, B, ctot,
                          c0, HW, eps, nrep, rep_stride, early, add, coef);
  else
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, grid, block, 0, st, t, x, x_stats, t_stats, B, ctot, c0, HW, eps, nrep,
                       rep_stride, early, add, coef);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}
}  // namespace pdes

extern "C" int pdes_bn_backward_finalize(const pdes_context* ctx, float* t, const float* x, const double* x_stats, const double* t_stats,
                                         int B, int ctot, int c0, int c1, int HW, float eps, int nrep,
                                         long long rep_stride, void* stream) {
  return bn_backward_finalize_launch(ctx, t, x, x_stats, t_stats, B, ctot, c0, c1, HW, eps, nrep, rep_stride,
                                     static_cast<hipStream_t>(stream), nullptr, nullptr, nullptr);
}

extern "C" int pdes_pack_weights(const pdes_pack_item* items, int n, int max_elems, void* stream) {
  if (!items || n <= 0 || max_elems <= 0) return PDES_EINVAL;
  int gx = cdiv(max_elems, 256);
  gx = gx > 64 ? 64 : gx;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(gx, n), dim3(256), 0, static_cast<hipStream_t>(stream), items);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

extern "C" int pdes_pack_all(const pdes_pack_item* items, int n, const pdes_mfma_pack_item* mitems, int nm,
                             const pdes_up_pack_item* uitems, int nu, const pdes_b3_pack_item* bitems, int nb,
                             int max_elems, void* stream) {
  return pdes_pack_all2(items, n, mitems, nm, uitems, nu, bitems, nb, nullptr, 0, max_elems, stream);
}

extern "C" int pdes_pack_all2(const pdes_pack_item* items, int n, const pdes_mfma_pack_item* mitems, int nm,
                              const pdes_up_pack_item* uitems, int nu, const pdes_b3_pack_item* bitems, int nb,
                              const pdes_b3up_pack_item* buitems, int nbu, int max_elems, void* stream) {
  if (n < 0 || nm < 0 || nu < 0 || nb < 0 || nbu < 0 || n + nm + nu + nb + nbu <= 0 || max_elems <= 0) return PDES_EINVAL;
  if ((n && !items) || (nm && !mitems) || (nu && !uitems) || (nb && !bitems) || (nbu && !buitems)) return PDES_EINVAL;
  // one element per thread per iteration is a dependent div/mod + gather chain: enough blocks that the
  // largest image (~0.5 M elements) needs 4 iterations, the small ones exit after one
  int gx = cdiv(max_elems, 256);
  gx = gx > 512 ? 512 : gx;
  hipLaunchKernelGGL(pack_all_kernel, dim3(gx, n + nm + nu + nb + nbu), dim3(256), 0, static_cast<hipStream_t>(stream), items, n,
                     mitems, nm, uitems, nu, bitems, nb, buitems);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

extern "C" int pdes_bn_update_running(const pdes_bn_item* items, int n, int max_c, float momentum, int nrep,
                                      long long rep_stride, void* stream) {
  if (!items || n <= 0 || max_c <= 0 || nrep != PDES_NREP) return PDES_EINVAL;
  hipLaunchKernelGGL(bn_update_running_kernel, dim3(cdiv(max_c, 256), n), dim3(256), 0,
                     static_cast<hipStream_t>(stream), items, momentum, nrep, rep_stride);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

extern "C" int pdes_bn_param_grads(const pdes_bn_item* items, int n, int max_c, int nrep, long long rep_stride,
                                   void* stream) {
  if (!items || n <= 0 || max_c <= 0 || nrep != PDES_NREP) return PDES_EINVAL;
  hipLaunchKernelGGL(bn_param_grads_kernel, dim3(cdiv(max_c, 256), n), dim3(256), 0,
                     static_cast<hipStream_t>(stream), items, nrep, rep_stride);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

extern "C" int pdes_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const float* hyper,
                              float grad_scale, long long n, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !hyper || n <= 0) return PDES_EINVAL;
  long long gx = (n + 255) / 256;
  gx = gx > 2048 ? 2048 : gx;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)gx), dim3(256), 0, static_cast<hipStream_t>(stream), param, grad,
                     exp_avg, exp_avg_sq, hyper, grad_scale, n);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

extern "C" int pdes_adam_step_host(float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                                   const float* hyper_host, float grad_scale, int zero_grad, long long n, void* stream) {
  return pdes_adam_step_host2(param, grad, exp_avg, exp_avg_sq, hyper_host, grad_scale, zero_grad, n, nullptr, 0, stream);
}

extern "C" int pdes_adam_step_host2(float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                                    const float* hyper_host, float grad_scale, int zero_grad, long long n,
                                    double* clear, long long nclear, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !hyper_host || n <= 0 || nclear < 0 || (nclear && !clear)) return PDES_EINVAL;
  const AdamHyper h = {hyper_host[0], hyper_host[1], hyper_host[2], hyper_host[3], hyper_host[4], hyper_host[5],
                       hyper_host[6]};
  if (!(h.bc1 > 0.f) || !(h.bc2_sqrt > 0.f)) return PDES_EINVAL;
  long long gx = (n + 255) / 256;
  gx = gx > 2048 ? 2048 : gx;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (zero_grad)
    hipLaunchKernelGGL(adam_kernel_v<true>, dim3((unsigned)gx), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq, h,
                       grad_scale, n, clear, nclear);
  else
    hipLaunchKernelGGL(adam_kernel_v<false>, dim3((unsigned)gx), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq, h,
                       grad_scale, n, clear, nclear);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

extern "C" int pdes_step_tail(const pdes_bn_item* items, int n, int max_c, float momentum, int update_running,
                              const float* partials, int B, int H, int W, float w_const, float w_cont, float w_dir,
                              float w_neu, float* terms, double* terms_accum, int nrep, long long rep_stride,
                              void* stream) {
  if (!items || n <= 0 || max_c <= 0 || nrep != PDES_NREP) return PDES_EINVAL;
  if (partials && (B <= 0 || H <= 0 || W <= 0 || !aligned16(partials))) return PDES_EINVAL;
  LossTail lt;
  lt.partials = partials; lt.terms = terms; lt.accum = terms_accum;
  lt.B = partials ? pdes_darcy_loss_partial_rows(B, H, W, 0) : 0;       // rows of the loss launch (flags without NO_TB / UNCORRECTED)
  if (partials && lt.B <= 0) return PDES_ENOSUP;
  lt.inv_n = partials ? 1.0 / ((double)B * H * W) : 0.0;
  lt.inv_dir = partials ? 1.0 / ((double)B * H) : 0.0;
  lt.inv_neu = partials ? 1.0 / (2.0 * B * W) : 0.0;
  lt.w[0] = w_const; lt.w[1] = w_cont; lt.w[2] = w_dir; lt.w[3] = w_neu;
  hipLaunchKernelGGL(step_tail_kernel, dim3(cdiv(max_c, 256), n + (partials ? 1 : 0)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), items, n, momentum, update_running, lt, nrep, rep_stride);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}
