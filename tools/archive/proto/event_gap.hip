// Micro-benchmark: what does one fork event cost ON THE MAIN CHAIN?  A chain of small dependent kernels on stream A,
//   (a) back to back,
//   (b) with hipEventRecord(e, A) + hipStreamWaitEvent(B, e) after every kernel (B otherwise idle),
//   (c) as (b) plus a small kernel on B behind every wait (what pdes_backward does with its weight gradients),
//   (d) as (c) but one event per FOUR kernels,
//   (e) as (c) but the event is the kernel's own completion signal (hipExtLaunchKernelGGL stopEvent): no barrier packet.
// Built by hand: hipcc --offload-arch=gfx950 -O2 tools/archive/proto/event_gap.hip -o tools/archive/proto/event_gap  (not part of the library)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <vector>

__global__ void small(float* p, int n, int iters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    float v = p[i];
    for (int k = 0; k < iters; ++k) v = v * 1.0001f + 0.5f;
    p[i] = v;
  }
}

int main() {
  const int n = 1 << 16, N = 2000;
  float *a, *b;
  hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
  hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4);
  hipStream_t A, B;
  hipStreamCreate(&A); hipStreamCreate(&B);
  std::vector<hipEvent_t> ev(N);
  for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  hipEvent_t t0, t1;
  hipEventCreate(&t0); hipEventCreate(&t1);
  for (int iters : {400, 1600, 4800}) {          // ~2 us and ~10 us kernels
    for (int mode = 0; mode < 5; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        hipDeviceSynchronize();
        hipEventRecord(t0, A);
        for (int i = 0; i < N; ++i) {
          if (mode == 4) {
            hipExtLaunchKernelGGL(small, dim3(n / 256), dim3(256), 0, A, nullptr, ev[i], 0, a, n, iters);
            hipStreamWaitEvent(B, ev[i], 0);
            hipLaunchKernelGGL(small, dim3(n / 256), dim3(256), 0, B, b, n, iters);
            continue;
          }
          hipLaunchKernelGGL(small, dim3(n / 256), dim3(256), 0, A, a, n, iters);
          const bool fork = mode == 1 || mode == 2 || (mode == 3 && i % 4 == 3);
          if (fork) {
            hipEventRecord(ev[i], A);
            hipStreamWaitEvent(B, ev[i], 0);
            if (mode >= 2) hipLaunchKernelGGL(small, dim3(n / 256), dim3(256), 0, B, b, n, iters);
          }
        }
        hipEventRecord(t1, A);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, t0, t1);
        if (rep) printf("kernel iters %3d mode %d: %.2f us per chain link\n", iters, mode, ms * 1e3 / N);
      }
    }
  }
  return 0;
}
