// Micro-benchmark: what does the accumulator layout of v_mfma_f32_16x16x4_f32 cost the epilogue's memory traffic?
// A data-gradient epilogue reads x and writes T for (B, C, H, W) = (32, 49, 64, 64) (51 MB).  Pattern (a) is what the
// kernels do: lane (n = lane & 15, g = lane >> 4) touches 4 consecutive pixels of channel n -> a wave instruction covers
// 16 channels x 64 contiguous bytes.  Pattern (b): the same bytes with lanes along pixels (after an LDS transpose):
// a wave instruction covers 4 (channel, row) pairs x 256 contiguous bytes.  (c): plain streaming copy for reference.
// Built by hand: hipcc --offload-arch=gfx950 -O2 tools/archive/proto/epilogue_pattern.hip -o /tmp/epilogue_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>

constexpr int B = 32, C = 49, H = 64, W = 64, HW = H * W;

// workgroup = 4 rows x 32 px of one image (8 M-tiles), wave w = N-tile w (16 channels)
__global__ __launch_bounds__(256) void pat_a(const float* __restrict__ x, float* __restrict__ t) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y;
  const int oy0 = (blockIdx.x / 2) * 4, ox0 = (blockIdx.x % 2) * 32;
  const int ci = wave * 16 + (lane & 15), kq = lane >> 4;
  if (ci >= C) return;
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    const size_t idx = ((size_t)b * C + ci) * HW + (size_t)(oy0 + (mt >> 1)) * W + ox0 + (mt & 1) * 16 + 4 * kq;
    float4 v = *reinterpret_cast<const float4*>(x + idx);
    v.x = v.x > 0.f ? v.x * 1.5f : 0.f; v.y = v.y > 0.f ? v.y * 1.5f : 0.f; v.z = v.z > 0.f ? v.z * 1.5f : 0.f; v.w = v.w > 0.f ? v.w * 1.5f : 0.f;
    *reinterpret_cast<float4*>(t + idx) = v;
  }
}
// same tile, lanes along pixels: lane -> (pair = lane >> 3 of 8 (channel,row) pairs, 4 px at 4 (lane & 7))
__global__ __launch_bounds__(256) void pat_b(const float* __restrict__ x, float* __restrict__ t) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y;
  const int oy0 = (blockIdx.x / 2) * 4, ox0 = (blockIdx.x % 2) * 32;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int pair = it * 8 + (lane >> 3);           // 64 (channel, row) pairs per wave: 16 channels x 4 rows
    const int ci = wave * 16 + (pair >> 2), row = pair & 3;
    if (ci >= C) continue;
    const size_t idx = ((size_t)b * C + ci) * HW + (size_t)(oy0 + row) * W + ox0 + 4 * (lane & 7);
    float4 v = *reinterpret_cast<const float4*>(x + idx);
    v.x = v.x > 0.f ? v.x * 1.5f : 0.f; v.y = v.y > 0.f ? v.y * 1.5f : 0.f; v.z = v.z > 0.f ? v.z * 1.5f : 0.f; v.w = v.w > 0.f ? v.w * 1.5f : 0.f;
    *reinterpret_cast<float4*>(t + idx) = v;
  }
}
__global__ __launch_bounds__(256) void pat_c(const float4* __restrict__ x, float4* __restrict__ t, int n4) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
    float4 v = x[i];
    v.x = v.x > 0.f ? v.x * 1.5f : 0.f; v.y = v.y > 0.f ? v.y * 1.5f : 0.f; v.z = v.z > 0.f ? v.z * 1.5f : 0.f; v.w = v.w > 0.f ? v.w * 1.5f : 0.f;
    t[i] = v;
  }
}

int main() {
  const size_t n = (size_t)B * C * HW;
  float *x, *t;
  hipMalloc(&x, n * 4); hipMalloc(&t, n * 4);
  hipMemset(x, 0, n * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const dim3 grid(2 * (H / 4), B);
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, 0);
      for (int k = 0; k < 100; ++k) {
        if (mode == 0) hipLaunchKernelGGL(pat_a, grid, dim3(256), 0, 0, x, t);
        else if (mode == 1) hipLaunchKernelGGL(pat_b, grid, dim3(256), 0, 0, x, t);
        else hipLaunchKernelGGL(pat_c, dim3(2048), dim3(256), 0, 0, (const float4*)x, (float4*)t, (int)(n / 4));
      }
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("pattern %c: %.2f us per pass, %.0f GB/s\n", 'a' + mode, ms * 10.f, 2.0 * n * 4 / (ms * 1e-5) / 1e9);
    }
  }
  return 0;
}
