// What does it cost the MAIN chain to release work on a second stream after one of its kernels?  The backward pass
// forks 27 times per step (finalize -> weight gradient on a side stream); the kernel behind a fork starts ~5 us late.
// One "layer" = F (~5 us) ; notify ; D (~20 us) on the main stream, and wait ; W (~20 us) on the side stream.
// Variants of "notify / wait", GPU time per layer (wall clock around a deep queue, host cost excluded by construction):
//   hipcc --offload-arch=gfx950 -O2 tools/archive/proto/fork_gap.hip -o /tmp/fork_gap && /tmp/fork_gap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>

__global__ void k_work(float* p, int n) {
  float a = p[threadIdx.x];
  for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f;
  p[threadIdx.x + blockIdx.x * blockDim.x] = a;
}
static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  float *a, *b;
  CK(hipMalloc(&a, 1 << 24)); CK(hipMalloc(&b, 1 << 24));
  CK(hipMemset(a, 0, 1 << 24)); CK(hipMemset(b, 0, 1 << 24));
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t s, w;
  CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithPriority(&w, hipStreamNonBlocking, lo));
  const int L = 27, STEPS = 30, NF = 1500, ND = 6000;
  hipEvent_t evA[L], evB[L], join;
  for (int i = 0; i < L; ++i) {
    CK(hipEventCreateWithFlags(&evA[i], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&evB[i], hipEventDisableTiming | hipEventDisableSystemFence));
  }
  CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  uint32_t* flag = nullptr;
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  if (can) CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory));
  if (flag) *flag = 0;
  uint32_t seq = 0;
  auto F = [&](hipEvent_t stop) {
    if (stop) hipExtLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, s, nullptr, stop, 0, a, NF);
    else hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, s, a, NF);
  };
  auto D = [&]() { hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, s, a, ND); };
  auto W = [&]() { hipLaunchKernelGGL(k_work, dim3(128), dim3(256), 0, w, b, ND); };
  auto run = [&](const char* name, auto&& layer) {
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
      hipDeviceSynchronize();
      const double t0 = now_us();
      for (int st = 0; st < STEPS; ++st) {
        for (int i = 0; i < L; ++i) layer(i);
        hipEventRecord(join, w);
        hipStreamWaitEvent(s, join, 0);
      }
      hipDeviceSynchronize();
      const double d = (now_us() - t0) / (STEPS * L);
      best = d < best ? d : best;
    }
    printf("%-78s %6.2f us per layer\n", name, best);
  };
  run("F ; D only (no fork, no side stream)", [&](int) { F(nullptr); D(); });
  run("F ; D on main, W on the side stream without any dependency", [&](int) { F(nullptr); D(); W(); });
  run("F with stop event ; D   (nobody waits for the event)", [&](int i) { F(evA[i]); D(); });
  run("F with stop event (no system fence) ; D   (nobody waits)", [&](int i) { F(evB[i]); D(); });
  run("F with stop event ; side waits ; D ; W", [&](int i) { F(evA[i]); hipStreamWaitEvent(w, evA[i], 0); D(); W(); });
  run("F with stop event (no system fence) ; side waits ; D ; W", [&](int i) { F(evB[i]); hipStreamWaitEvent(w, evB[i], 0); D(); W(); });
  run("F ; hipEventRecord ; side waits ; D ; W", [&](int i) { F(nullptr); hipEventRecord(evA[i], s); hipStreamWaitEvent(w, evA[i], 0); D(); W(); });
  run("F ; hipEventRecord (no system fence) ; side waits ; D ; W", [&](int i) { F(nullptr); hipEventRecord(evB[i], s); hipStreamWaitEvent(w, evB[i], 0); D(); W(); });
  if (flag) {
    run("F ; hipStreamWriteValue32 ; D   (nobody waits)", [&](int) { F(nullptr); hipStreamWriteValue32(s, flag, ++seq, 0); D(); });
    run("F ; hipStreamWriteValue32 ; side hipStreamWaitValue32(>=) ; D ; W", [&](int) {
      F(nullptr); hipStreamWriteValue32(s, flag, ++seq, 0); hipStreamWaitValue32(w, flag, seq, hipStreamWaitValueGte, 0xffffffffu); D(); W();
    });
  } else {
    printf("hipStreamWaitValue32 not supported on this device\n");
  }
  return 0;
}
