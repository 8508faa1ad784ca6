// Micro-benchmark: can v_mfma_f32_16x16x32_bf16 and ordinary VALU work overlap on one SIMD of gfx950?
//   mode 0: every wave runs N MFMAs (4 independent accumulators)
//   mode 1: every wave runs 4 N VALU FMAs (8 independent chains)
//   mode 2: every wave runs both, interleaved in program order (1 MFMA, 4 FMAs)
//   mode 3: waves 0-3 run the MFMAs, waves 4-7 the FMAs (overlap ACROSS the waves of one SIMD; needs >= 2 waves per SIMD)
// One workgroup per CU, `wpc` waves per workgroup (4 = one per SIMD, 8 = two per SIMD); time = wall_clock / s_memtime
// cycles of wave 0.  Built by hand: hipcc --offload-arch=gfx950 -O2 tools/archive/proto/mfma_valu_overlap.hip -o tools/archive/proto/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ void k(float* out, long long* cyc, int n) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v8bf a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
  v4f acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = 1.f + 0.01f * (lane + i);
  const float m = 1.0001f, c = 1e-4f;
  const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && ((wave >> 2) & 1) == 0);
  const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && ((wave >> 2) & 1) == 1);
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (MODE == 2) {
    for (int i = 0; i < n; i += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) f[(j & 1) * 4 + q] = f[(j & 1) * 4 + q] * m + c;
      }
    }
  } else {
    if (do_m)
      for (int i = 0; i < n; i += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
      }
    if (do_v)
      for (int i = 0; i < n; i += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) f[(j & 1) * 4 + q] = f[(j & 1) * 4 + q] * m + c;
      }
  }
  const long long t1 = __builtin_readcyclecounter();
  __syncthreads();
  float s = 0.f;
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  if (threadIdx.x == 256 && blockIdx.x == 0) cyc[1] = t1 - t0;
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 16);
  const int n = 4096;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wpc : {4, 8}) {
    for (int mode = 0; mode < 4; ++mode) {
      float best = 1e9f; long long hc[2] = {0, 0};
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * wpc), 0, 0, out, cyc, n);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * wpc), 0, 0, out, cyc, n);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(64 * wpc), 0, 0, out, cyc, n);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(64 * wpc), 0, 0, out, cyc, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost);
      }
      printf("waves/WG %d mode %d: %.1f us  (wave0 %lld, wave1 %lld cycle-counter ticks; %d MFMAs and/or %d FMAs per wave)\n", wpc, mode,
             best * 1e3f, hc[0], hc[1], n, 4 * n);
    }
  }
  return 0;
}
