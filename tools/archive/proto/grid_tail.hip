// Micro-benchmark: is a grid-wide barrier INSIDE the data-gradient kernel cheaper than the kernel boundary in front of
// the BatchNorm-backward finalize?  pdes_backward's serial chain is [data gradient_i -> finalize_{i-1}] x 27: the
// finalize needs batch-wide sums (sum T, sum T xhat) that every workgroup of the data gradient adds to, i.e. a
// grid-wide dependency, today expressed as a kernel boundary (+ a 7 us latency-bound kernel that re-reads T and x).
//   mode 0: producer kernel (dummy matrix work, writes T, fp64 statistics atomics) + separate finalize kernel
//   mode 1: producer kernel whose workgroups keep their T values in registers, arrive on a counter after their
//           atomics, spin (bounded) until the whole grid has arrived, read the statistics and store g directly
// Reported: us per chain link for both, and the largest difference between the two results.
// Built by hand: hipcc --offload-arch=gfx950 -O2 tools/archive/proto/grid_tail.hip -o /tmp/grid_tail   (not part of the library)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>

constexpr int NREP = 16;
constexpr int C = 16, HW = 1024;

__device__ __forceinline__ int rep_of_block() { return (blockIdx.x + blockIdx.y * 7) & (NREP - 1); }

// grid (8 tiles, B, Z); 256 threads.  Workgroups with blockIdx.z == 0 "hold" the 16 final channels: a thread owns 4
// consecutive pixels of 2 channels (tile = 128 pixels x 16 channels).
template <int MODE>
__global__ __launch_bounds__(256) void producer(float* __restrict__ t, const float* __restrict__ x, double* __restrict__ stats,
                                                unsigned* __restrict__ counter, unsigned target, int work, int* __restrict__ err) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, tile = blockIdx.x;
  float acc = 1.f + 1e-3f * (tid & 7);
  for (int k = 0; k < work; ++k) acc = acc * 1.00001f + 1e-4f;       // stands for the matrix loop
  const bool holder = blockIdx.z == 0;
  float4 tv[2], xv[2];
  if (holder) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = (lane & 15), p = tile * 128 + ((wave * 2 + j) * 4 + (lane >> 4)) * 4;
      const size_t idx = ((size_t)b * C + c) * HW + p;
      xv[j] = *reinterpret_cast<const float4*>(x + idx);
      const float s = acc * 1e-3f;
      tv[j] = make_float4(xv[j].x * s + 0.1f, xv[j].y * s - 0.2f, xv[j].z * s, xv[j].w * s + 0.05f);
    }
    float st = 0.f, sx = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      st += (tv[j].x + tv[j].y) + (tv[j].z + tv[j].w);
      sx += (tv[j].x * xv[j].x + tv[j].y * xv[j].y) + (tv[j].z * xv[j].z + tv[j].w * xv[j].w);
    }
    st += __shfl_xor(st, 16, 64); st += __shfl_xor(st, 32, 64);
    sx += __shfl_xor(sx, 16, 64); sx += __shfl_xor(sx, 32, 64);
    if (lane < 16) {
      double* s = stats + (size_t)rep_of_block() * 2 * C;
      atomicAdd(&s[2 * lane], (double)st);
      atomicAdd(&s[2 * lane + 1], (double)sx);
    }
  }
  if (MODE == 0) {
    if (holder) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = (lane & 15), p = tile * 128 + ((wave * 2 + j) * 4 + (lane >> 4)) * 4;
        *reinterpret_cast<float4*>(t + ((size_t)b * C + c) * HW + p) = tv[j];
      }
    }
    return;
  }
  // ---- MODE 1: per WORKGROUP arrive (after its waves' atomics) on one of 16 sub-counters (same-address atomics
  // serialise at ~35 ns each: 1024 wave arrivals on one word cost 43 us); the last arrival of a sub-counter bumps the
  // top counter that the holders poll
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned old = __hip_atomic_fetch_add(counter + 1 + rep_of_block(), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == target - 1) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!holder) return;
  if (tid == 0) {
    int n = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < NREP) {
      __builtin_amdgcn_s_sleep(1);
      if (++n > (1 << 22)) { *err = 1; break; }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  // statistics: lane group g = lane >> 4 reads replicas 4 g .. 4 g + 3 of its channel, both sums
  const int c = lane & 15, g = lane >> 4;
  double s0 = 0.0, s1 = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const double* s = stats + (size_t)(4 * g + r) * 2 * C;
    s0 += __hip_atomic_load(&s[2 * c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s1 += __hip_atomic_load(&s[2 * c + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  s0 += __shfl_xor(s0, 16, 64); s0 += __shfl_xor(s0, 32, 64);
  s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
  const float inv_n = 1.f / (float)(gridDim.y * HW);
  const float m1 = (float)s0 * inv_n, m2 = (float)s1 * inv_n;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int p = tile * 128 + ((wave * 2 + j) * 4 + (lane >> 4)) * 4;
    float4 o;
    o.x = tv[j].x - m1 - xv[j].x * m2; o.y = tv[j].y - m1 - xv[j].y * m2;
    o.z = tv[j].z - m1 - xv[j].z * m2; o.w = tv[j].w - m1 - xv[j].w * m2;
    *reinterpret_cast<float4*>(t + ((size_t)b * C + c) * HW + p) = o;
  }
}

// the stand-alone finalize (shape of bn_bwd_finalize_kernel): grid (1, C, B)
__global__ __launch_bounds__(256) void finalize(float* __restrict__ t, const float* __restrict__ x, const double* __restrict__ stats, int B) {
  const int c = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  __shared__ double sums[2];
  __shared__ float sc[2];
  const size_t base = ((size_t)b * C + c) * HW;
  float4 tv = reinterpret_cast<const float4*>(t + base)[tid], xv = reinterpret_cast<const float4*>(x + base)[tid];
  if (tid < 32) {
    const int q = tid >> 4, r = tid & 15;
    double v = stats[(size_t)r * 2 * C + 2 * c + q];
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 16);
    if (r == 0) sums[q] = v;
  }
  __syncthreads();
  if (tid == 0) { const float inv_n = 1.f / (float)(B * HW); sc[0] = (float)sums[0] * inv_n; sc[1] = (float)sums[1] * inv_n; }
  __syncthreads();
  const float m1 = sc[0], m2 = sc[1];
  tv.x = tv.x - m1 - xv.x * m2; tv.y = tv.y - m1 - xv.y * m2; tv.z = tv.z - m1 - xv.z * m2; tv.w = tv.w - m1 - xv.w * m2;
  reinterpret_cast<float4*>(t + base)[tid] = tv;
}

int main() {
  const int B = 32, N = 200;
  const size_t n = (size_t)B * C * HW;
  float *x, *t0, *t1;
  double* stats;
  unsigned* counter;
  int* err;
  hipMalloc(&x, n * 4); hipMalloc(&t0, n * 4); hipMalloc(&t1, n * 4);
  hipMalloc(&stats, (size_t)N * NREP * 2 * C * 8); hipMalloc(&counter, N * 32 * 4); hipMalloc(&err, 4);
  std::vector<float> hx(n);
  for (size_t i = 0; i < n; ++i) hx[i] = sinf(0.37f * (float)(i % 9973)) + 0.1f;
  hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
  hipMemset(err, 0, 4);
  hipStream_t st;
  hipStreamCreate(&st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int Z : {1, 3})
    for (int work : {0, 500, 1500}) {
      float ms[2];
      for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
          hipMemsetAsync(stats, 0, (size_t)N * NREP * 2 * C * 8, st);
          hipMemsetAsync(counter, 0, N * 32 * 4, st);
          hipEventRecord(e0, st);
          const dim3 grid(8, B, Z);
          for (int i = 0; i < N; ++i) {
            double* s = stats + (size_t)i * NREP * 2 * C;
            if (mode == 0) {
              hipLaunchKernelGGL(producer<0>, grid, dim3(256), 0, st, t0, x, s, counter + 32 * i, 0u, work, err);
              hipLaunchKernelGGL(finalize, dim3(1, C, B), dim3(256), 0, st, t0, x, s, B);
            } else {
              hipLaunchKernelGGL(producer<1>, grid, dim3(256), 0, st, t1, x, s, counter + 32 * i, (unsigned)(8 * B * Z / NREP), work, err);
            }
          }
          hipEventRecord(e1, st);
          hipEventSynchronize(e1);
          float m;
          hipEventElapsedTime(&m, e0, e1);
          best = m < best ? m : best;
        }
        ms[mode] = best;
      }
      std::vector<float> a(n), b(n);
      hipMemcpy(a.data(), t0, n * 4, hipMemcpyDeviceToHost);
      hipMemcpy(b.data(), t1, n * 4, hipMemcpyDeviceToHost);
      double md = 0, mx = 0;
      for (size_t i = 0; i < n; ++i) { md = fmax(md, fabs((double)a[i] - b[i])); mx = fmax(mx, fabs((double)a[i])); }
      int herr = 0;
      hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
      printf("Z=%d work=%d: separate finalize %.2f us/link, grid-tail %.2f us/link, max |diff| %.3g (max |g| %.3g), spin timeout %d\n", Z,
             work, ms[0] * 1e3f / N, ms[1] * 1e3f / N, md, mx, herr);
    }
  return 0;
}
