// Shared device/host helpers for the pde-surrogate MI355X (gfx950) kernels.
// Everything here is written for CDNA4 only: 64-lane wavefronts, DPP row shifts, 160 KiB LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PDES_OK 0
#define PDES_EINVAL (-1)      // bad argument (null pointer, size <= 0)
#define PDES_ENOSUP (-2)      // shape/option the kernels do not implement
#define PDES_EALIGN (-3)      // pointer not 16-byte aligned

// enqueue-only launches: report the launch error code (>0 = hipError_t), never synchronise
#define PDES_LAUNCH_CHECK()                          \
  do {                                               \
    hipError_t e__ = hipGetLastError();              \
    if (e__ != hipSuccess) return (int)e__;          \
  } while (0)

namespace pdes {

constexpr int WAVE = 64;

__device__ __forceinline__ float dpp_row_shr1(float v) {   // lane i <- lane i-1 (within 16-lane row)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_row_shl1(float v) {   // lane i <- lane i+1 (within 16-lane row)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x101, 0xf, 0xf, true));
}

// streaming (non-temporal) 16-byte global accesses: data touched exactly once should not displace
// reusable lines in L2 / Infinity Cache
typedef float nvec4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float4* p) {
  const nvec4 v = __builtin_nontemporal_load(reinterpret_cast<const nvec4*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void nt_store4(float4* p, const float4& v) {
  nvec4 t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<nvec4*>(p));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// fp64 statistics are accumulated into one of `nrep` replicas of the arena (replica stride in
// doubles) to keep same-address atomic contention low; readers sum the replicas.
// The replica count is a compile-time constant (descriptors must carry nrep == PDES_NREP) so the
// loads below are issued back to back and their latency is paid once, not nrep times.
// 8 replicas (round 3, same process alternated on one box: 1.712-1.713 ms per step against 1.725-1.740 with 16 and
// 1.740-1.757 with 4: every reader sums the replicas in its prologue, every writer queues behind the atomics of its
// replica; rounds 1-2 ran 16).  At most 16: bn_bwd_finalize / the flow finalize reduce them with a 16-lane shuffle.
#ifndef PDES_NREP
#define PDES_NREP 8
#endif
static_assert(PDES_NREP >= 1 && PDES_NREP <= 16, "replica count");
__device__ __forceinline__ double rep_sum(const double* p, int idx, int /*nrep*/, long long stride) {
  double a[PDES_NREP];
#pragma unroll
  for (int r = 0; r < PDES_NREP; ++r) a[r] = p[(long long)r * stride + idx];
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < PDES_NREP; ++r) s += a[r];
  return s;
}
// Batch mean / inverse standard deviation of channel `c` of a buffer.  `coef` (may be NULL) is the per-channel {mean,
// invstd} table of the buffer, zeroed with the statistics arena at the start of every step: an entry with invstd > 0 was
// published earlier in THIS step from the completed sums (same expression, same bits), otherwise the replicas are summed
// here and -- from one workgroup of the launch (`publish`) -- the entry is written for the kernels that follow.  One
// 8-byte access per entry: no torn reads.
struct MeanInv { float mean, invstd; };
__device__ __forceinline__ MeanInv batch_mean_invstd(float* coef, const double* x_stats, long long rs, double n, float eps,
                                                     int c, bool publish) {
  MeanInv o;
  if (coef) {
    const float2 e = reinterpret_cast<const float2*>(coef)[c];
    if (e.y > 0.f) { o.mean = e.x; o.invstd = e.y; return o; }
  }
  const double m = rep_sum(x_stats, 2 * c, PDES_NREP, rs) / n;
  double var = rep_sum(x_stats, 2 * c + 1, PDES_NREP, rs) / n - m * m;
  var = var < 0.0 ? 0.0 : var;
  o.mean = (float)m;
  o.invstd = (float)(1.0 / sqrt(var + (double)eps));
  if (coef && publish) reinterpret_cast<float2*>(coef)[c] = make_float2(o.mean, o.invstd);
  return o;
}
__device__ __forceinline__ int rep_of_block(int nrep) {
  return (int)((blockIdx.x + 7u * blockIdx.y + 3u * blockIdx.z) % (unsigned)nrep);
}

__host__ __device__ __forceinline__ int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace pdes
