// Test-time metrics (NRMSE, R^2) and the mean-squared-error loss of the data-driven harness.
//   pdes_test_metrics : train_codec_mixed_residual.py:180-183,196-197 (err2_sum, relative l2, r2 numerator), the
//                       same lines of train_codec_max_likelihood.py:173-176,189-190
//   pdes_mse_loss     : F.mse_loss(output, target) + its autograd backward wrt output
//                       (train_codec_max_likelihood.py:170,203-204)
// Both are HBM-bound streaming reductions: 16-byte loads, fp32 per thread, fp64 from the wave reduce on, per-block /
// per-image partials combined in a FIXED order by a one-block finalize (deterministic, no float atomics).
#include "pdes_common.h"
#include "../../include/pdes_hip.h"

namespace pdes {

constexpr int MSE_BLOCK = 256;
constexpr int MSE_MAX_BLOCKS = 2048;

__device__ __forceinline__ double block_sum(double v, double* sm) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) sm[w] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < nw; ++i) s += sm[i];
  __syncthreads();
  return s;                      // valid on thread 0
}

__global__ __launch_bounds__(MSE_BLOCK) void mse_partial_kernel(const float* __restrict__ o, const float* __restrict__ t,
                                                                float* __restrict__ g, double* __restrict__ partials,
                                                                long long n, float gscale) {
  __shared__ double sm[MSE_BLOCK / 64];
  const long long n4 = n >> 2;
  const float4* o4 = reinterpret_cast<const float4*>(o);
  const float4* t4 = reinterpret_cast<const float4*>(t);
  float4* g4 = reinterpret_cast<float4*>(g);
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * MSE_BLOCK + threadIdx.x; i < n4; i += (long long)gridDim.x * MSE_BLOCK) {
    const float4 a = o4[i], b = t4[i];
    const float4 d = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
    acc += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
    if (g) g4[i] = make_float4(gscale * d.x, gscale * d.y, gscale * d.z, gscale * d.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {          // tail (n not a multiple of 4)
    const long long i = (n4 << 2) + threadIdx.x;
    const float d = o[i] - t[i];
    acc += d * d;
    if (g) g[i] = gscale * d;
  }
  const double s = block_sum((double)acc, sm);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

__global__ __launch_bounds__(64) void mse_finalize_kernel(const double* __restrict__ partials, int nblk, double inv_n,
                                                          float* __restrict__ loss_out, double* __restrict__ accum) {
  if (threadIdx.x) return;
  double s = 0.0;
  for (int i = 0; i < nblk; ++i) s += partials[i];
  s *= inv_n;
  if (loss_out) loss_out[0] = (float)s;
  if (accum) accum[0] += s;
}

// one workgroup per (channel, image) plane: {sum (o-t)^2, sum t^2}
__global__ __launch_bounds__(256) void test_metrics_kernel(const float* __restrict__ o, const float* __restrict__ t,
                                                           float* __restrict__ per_image, int C, int HW) {
  __shared__ double sm[4];
  const int c = blockIdx.x, b = blockIdx.y;
  const size_t base = ((size_t)b * C + c) * HW;
  float e = 0.f, q = 0.f;
  if ((HW & 3) == 0) {
    const float4* o4 = reinterpret_cast<const float4*>(o + base);
    const float4* t4 = reinterpret_cast<const float4*>(t + base);
    for (int i = threadIdx.x; i < HW / 4; i += 256) {
      const float4 a = o4[i], r = t4[i];
      const float dx = a.x - r.x, dy = a.y - r.y, dz = a.z - r.z, dw = a.w - r.w;
      e += dx * dx + dy * dy + dz * dz + dw * dw;
      q += r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w;
    }
  } else {
    for (int i = threadIdx.x; i < HW; i += 256) {
      const float d = o[base + i] - t[base + i];
      e += d * d;
      q += t[base + i] * t[base + i];
    }
  }
  const double es = block_sum((double)e, sm);
  const double qs = block_sum((double)q, sm);
  if (threadIdx.x == 0) {
    per_image[((size_t)b * C + c) * 2 + 0] = (float)es;
    per_image[((size_t)b * C + c) * 2 + 1] = (float)qs;
  }
}

// accum[c] += sum_b sqrt(err2/t2); accum[C+c] += sum_b err2; accum[2C] += B   (fixed order over b)
__global__ __launch_bounds__(64) void test_metrics_accum_kernel(const float* __restrict__ per_image, int B, int C,
                                                                double* __restrict__ accum) {
  const int c = threadIdx.x;
  if (c < C) {
    double rel = 0.0, e2 = 0.0;
    for (int b = 0; b < B; ++b) {
      const float e = per_image[((size_t)b * C + c) * 2 + 0], q = per_image[((size_t)b * C + c) * 2 + 1];
      rel += (double)sqrtf(e / q);            // fp32 like the reference's torch.sqrt(err2_sum / (target**2).sum)
      e2 += (double)e;
    }
    accum[c] += rel;
    accum[C + c] += e2;
  }
  if (c == 0) accum[2 * C] += (double)B;
}

}  // namespace pdes

using namespace pdes;

extern "C" int pdes_mse_partials(long long n) {
  if (n <= 0) return PDES_EINVAL;
  const long long blocks = ((n >> 2) + MSE_BLOCK - 1) / MSE_BLOCK;
  return (int)(blocks < 1 ? 1 : (blocks > MSE_MAX_BLOCKS ? MSE_MAX_BLOCKS : blocks));
}

extern "C" int pdes_mse_loss(const float* output, const float* target, float* grad_out, double* partials,
                             float* loss_out, double* loss_accum, long long n, void* stream) {
  if (!output || !target || !partials || n <= 0) return PDES_EINVAL;
  if (!aligned16(output) || !aligned16(target) || (grad_out && !aligned16(grad_out))) return PDES_EALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nblk = pdes_mse_partials(n);
  hipLaunchKernelGGL(mse_partial_kernel, dim3(nblk), dim3(MSE_BLOCK), 0, st, output, target, grad_out, partials, n,
                     (float)(2.0 / (double)n));
  PDES_LAUNCH_CHECK();
  if (loss_out || loss_accum) {
    hipLaunchKernelGGL(mse_finalize_kernel, dim3(1), dim3(64), 0, st, partials, nblk, 1.0 / (double)n, loss_out, loss_accum);
    PDES_LAUNCH_CHECK();
  }
  return PDES_OK;
}

extern "C" int pdes_test_metrics(const float* output, const float* target, float* per_image, double* accum, int B,
                                 int C, int HW, void* stream) {
  if (!output || !target || !per_image || B <= 0 || C <= 0 || C > 64 || HW <= 0) return PDES_EINVAL;
  if ((HW & 3) == 0 && (!aligned16(output) || !aligned16(target))) return PDES_EALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(test_metrics_kernel, dim3(C, B), dim3(256), 0, st, output, target, per_image, C, HW);
  PDES_LAUNCH_CHECK();
  if (accum) {
    hipLaunchKernelGGL(test_metrics_accum_kernel, dim3(1), dim3(64), 0, st, per_image, B, C, accum);
    PDES_LAUNCH_CHECK();
  }
  return PDES_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Inner products of the L-BFGS curvature history (config 5: the two-loop recursion of torch.optim.LBFGS,
// reference solve_conv_mixed_residual.py:124, restated on Gram matrices in pde_surrogate_amd/lbfgs.py):
//   out[s][r][j] = sum over the s-th slice of k of W[r][k] * V[j][k],   r < rows (<= 2 m + 1), j < nv (<= 4)
// W is (rows, ld) row-major, V is (nv, n): one bandwidth-bound pass over the history (rows * n * 4 bytes) instead of a
// GEMM with a 101 x 3 output tile and K = n (one workgroup in a BLAS library).  Per-slice fp64 partials are summed by
// the caller in a fixed order (deterministic).
namespace pdes {
template <int NV>
__global__ __launch_bounds__(256) void multi_dot_kernel(const float* __restrict__ W, long long ld, const float* __restrict__ V,
                                                        long long n, double* __restrict__ out, int rows) {
  __shared__ double sm[4];
  const int r = blockIdx.y, s = blockIdx.x, ns = gridDim.x;
  const long long n4 = n >> 2;
  const long long lo = n4 * s / ns, hi = n4 * (s + 1) / ns;
  const float4* w4 = reinterpret_cast<const float4*>(W + (long long)r * ld);
  float acc[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) acc[j] = 0.f;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const float4 a = w4[i];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const float4 b = reinterpret_cast<const float4*>(V + (long long)j * n)[i];
      acc[j] += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
  }
  if (s == ns - 1 && threadIdx.x < (n & 3)) {               // tail elements
    const long long i = (n4 << 2) + threadIdx.x;
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] += W[(long long)r * ld + i] * V[(long long)j * n + i];
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const double t = block_sum((double)acc[j], sm);
    if (threadIdx.x == 0) out[((long long)s * rows + r) * NV + j] = t;
  }
}
}  // namespace pdes

extern "C" int pdes_multi_dot(const float* W, long long ld, int rows, const float* V, int nv, long long n,
                              double* partials, int nsplit, void* stream) {
  if (!W || !V || !partials || rows <= 0 || nv < 1 || nv > 4 || n <= 0 || nsplit <= 0 || ld < n) return PDES_EINVAL;
  if (!aligned16(W) || !aligned16(V) || (ld & 3) || ((n & 3) && nv > 1)) return PDES_EALIGN;   // rows of W and V 16-byte aligned
  hipStream_t st = static_cast<hipStream_t>(stream);
  dim3 grid(nsplit, rows), block(256);
  switch (nv) {
    case 1: hipLaunchKernelGGL(multi_dot_kernel<1>, grid, block, 0, st, W, ld, V, n, partials, rows); break;
    case 2: hipLaunchKernelGGL(multi_dot_kernel<2>, grid, block, 0, st, W, ld, V, n, partials, rows); break;
    case 3: hipLaunchKernelGGL(multi_dot_kernel<3>, grid, block, 0, st, W, ld, V, n, partials, rows); break;
    default: hipLaunchKernelGGL(multi_dot_kernel<4>, grid, block, 0, st, W, ld, V, n, partials, rows); break;
  }
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}
