// Micro-benchmark: cross-stream release WITHOUT an event on the producer's stream.  Stream A runs a chain of small
// dependent kernels; after each one, stream B may run a kernel of its own (what pdes_backward does with the weight
// gradients).  Ways to tell B that A's kernel i has finished:
//   (0) nothing (B idle): the bare chain
//   (1) the kernel's completion signal as an event (hipExtLaunchKernelGGL stopEvent) + hipStreamWaitEvent(B)
//   (2) the kernel itself publishes i+1 to a signal-memory word when its last block retires; B waits with
//       hipStreamWaitValue32(B, flag, i+1, Gte): a packet-processor poll on B's queue, nothing on A's
// Also checks (2) for correctness: B's kernel reads what A's kernel wrote.
// Built by hand: hipcc --offload-arch=gfx950 -O2 tools/archive/proto/wait_value.hip -o /tmp/wait_value   (not part of the library)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <vector>

__global__ void small(float* p, int n, int iters, unsigned* counter, unsigned* flag, unsigned epoch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    float v = p[i];
    for (int k = 0; k < iters; ++k) v = v * 1.0001f + 0.5f;
    p[i] = v;
  }
  if (flag) {
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned done = atomicAdd(counter, 1u);
      if (done == gridDim.x - 1) {
        *counter = 0;
        __threadfence_system();
        __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

__global__ void check(const float* a, float* out, int n, float expect_min) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && a[i] < expect_min) atomicAdd(out, 1.f);       // counts elements A had not written yet
}

int main() {
  int can = 0;
  hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  const int n = 1 << 16, N = 1000;
  float *a, *b, *bad;
  unsigned *counter, *flag;
  hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&bad, 4); hipMalloc(&counter, 4);
  hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4); hipMemset(bad, 0, 4); hipMemset(counter, 0, 4);
  if (hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory) != hipSuccess) { printf("signal memory: failed\n"); return 1; }
  hipStream_t A, B;
  hipStreamCreate(&A); hipStreamCreate(&B);
  std::vector<hipEvent_t> ev(N);
  for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  hipEvent_t t0, t1;
  hipEventCreate(&t0); hipEventCreate(&t1);
  unsigned epoch = 0;
  for (int iters : {400, 1600}) {
    for (int mode = 0; mode < 3; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        hipDeviceSynchronize();
        hipEventRecord(t0, A);
        for (int i = 0; i < N; ++i) {
          if (mode == 0) {
            hipLaunchKernelGGL(small, dim3(n / 256), dim3(256), 0, A, a, n, iters, (unsigned*)nullptr, (unsigned*)nullptr, 0u);
          } else if (mode == 1) {
            hipExtLaunchKernelGGL(small, dim3(n / 256), dim3(256), 0, A, nullptr, ev[i], 0, a, n, iters, (unsigned*)nullptr,
                                  (unsigned*)nullptr, 0u);
            hipStreamWaitEvent(B, ev[i], 0);
            hipLaunchKernelGGL(small, dim3(n / 256), dim3(256), 0, B, b, n, iters, (unsigned*)nullptr, (unsigned*)nullptr, 0u);
          } else {
            ++epoch;
            hipLaunchKernelGGL(small, dim3(n / 256), dim3(256), 0, A, a, n, iters, counter, flag, epoch);
            const hipError_t e = hipStreamWaitValue32(B, flag, epoch, hipStreamWaitValueGte, 0xffffffffu);
            if (e != hipSuccess) { printf("hipStreamWaitValue32: %s\n", hipGetErrorString(e)); return 1; }
            hipLaunchKernelGGL(small, dim3(n / 256), dim3(256), 0, B, b, n, iters, (unsigned*)nullptr, (unsigned*)nullptr, 0u);
          }
        }
        hipEventRecord(t1, A);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, t0, t1);
        if (rep) printf("kernel iters %4d mode %d: %.2f us per chain link (stream A)\n", iters, mode, ms * 1e3 / N);
      }
    }
  }
  // correctness of mode 2: A fills a with a value, B checks it right behind the wait
  for (int trial = 0; trial < 200; ++trial) {
    hipMemsetAsync(a, 0, n * 4, A);
    ++epoch;
    hipLaunchKernelGGL(small, dim3(n / 256), dim3(256), 0, A, a, n, 3000, counter, flag, epoch);
    hipStreamWaitValue32(B, flag, epoch, hipStreamWaitValueGte, 0xffffffffu);
    hipLaunchKernelGGL(check, dim3(n / 256), dim3(256), 0, B, a, bad, n, 0.4f);
    hipDeviceSynchronize();
  }
  float h = -1;
  hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("elements B saw before A wrote them (200 trials): %.0f\n", h);
  return 0;
}
