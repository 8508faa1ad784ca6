// Host-side cost of getting work to the GPU on this runtime (MI355X, ROCm 7): what does ONE launch cost the calling
// thread, and what does a hipGraph replay of the same chain cost?  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/archive/proto/launch_cost.hip -o /tmp/launch_cost && /tmp/launch_cost
// Each figure: host microseconds per launch while the queue is shallow (bursts of 64 launches after a synchronise),
// and the GPU-side time per kernel of the chain (events).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

struct Big { int v[100]; };                         // 400 bytes by value, like pdes_conv_desc
__global__ void k_small(float* p) { if (p && threadIdx.x == 9999) p[0] = 1.f; }
__global__ void k_big(Big b, float* p) { if (p && threadIdx.x == 9999) p[0] = (float)b.v[3]; }
__global__ void k_work(float* p, int n) {           // ~10 us of work: a chain like the training step's kernels
  float a = p[threadIdx.x];
  for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f;
  p[threadIdx.x + blockIdx.x * blockDim.x] = a;
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  float* buf;
  CK(hipMalloc(&buf, 1 << 24));
  CK(hipMemset(buf, 0, 1 << 24));
  hipStream_t s, s2;
  CK(hipStreamCreate(&s));
  CK(hipStreamCreate(&s2));
  Big big{};
  const int NB = 64, REP = 20;
  auto burst = [&](const char* name, auto&& fn) {
    double best = 1e30, sum = 0;
    for (int r = 0; r < REP; ++r) {
      hipDeviceSynchronize();
      const double t0 = now_us();
      for (int i = 0; i < NB; ++i) fn(i);
      const double t1 = now_us();
      hipDeviceSynchronize();
      const double d = (t1 - t0) / NB;
      best = d < best ? d : best;
      if (r >= 2) sum += d;
    }
    printf("%-64s host %6.2f us/launch (min %5.2f)\n", name, sum / (REP - 2), best);
  };
  burst("hipLaunchKernelGGL, 8-byte args, empty kernel", [&](int) { hipLaunchKernelGGL(k_small, dim3(256), dim3(256), 0, s, buf); });
  burst("hipLaunchKernelGGL, 400-byte struct by value", [&](int) { hipLaunchKernelGGL(k_big, dim3(256), dim3(256), 0, s, big, buf); });
  burst("... with 32 KiB dynamic LDS", [&](int) { hipLaunchKernelGGL(k_big, dim3(256), dim3(256), 32768, s, big, buf); });
  hipEvent_t ev[64];
  for (auto& evi : ev) CK(hipEventCreateWithFlags(&evi, hipEventDisableTiming));
  burst("hipExtLaunchKernelGGL with a stop event (completion signal)",
        [&](int i) { hipExtLaunchKernelGGL(k_big, dim3(256), dim3(256), 0, s, nullptr, ev[i % 64], 0, big, buf); });
  burst("launch + hipEventRecord + hipStreamWaitEvent(other stream)", [&](int i) {
    hipLaunchKernelGGL(k_big, dim3(256), dim3(256), 0, s, big, buf);
    hipEventRecord(ev[i % 64], s);
    hipStreamWaitEvent(s2, ev[i % 64], 0);
  });
  burst("hipEventRecord alone", [&](int i) { hipEventRecord(ev[i % 64], s); });
  burst("hipStreamWaitEvent alone", [&](int i) { hipStreamWaitEvent(s2, ev[i % 64], 0); });
  burst("10 us kernel, 400-byte args (queue never empty)", [&](int) { hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, s, buf, 3000); });

  // back pressure: how many launches can be outstanding before the host blocks?
  for (int depth : {256, 1024, 4096, 16384}) {
    hipDeviceSynchronize();
    const double t0 = now_us();
    for (int i = 0; i < depth; ++i) hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, s, buf, 3000);
    const double t1 = now_us();
    hipDeviceSynchronize();
    const double t2 = now_us();
    printf("%6d launches of a ~10 us kernel without sync: host %7.2f us/launch, until done %7.2f us/launch\n", depth,
           (t1 - t0) / depth, (t2 - t0) / depth);
  }

  // hipGraph: a linear chain of N kernel nodes captured from the stream
  for (int n : {30, 120}) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_big, dim3(256), dim3(256), 0, s, big, buf);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 3; ++w) hipGraphLaunch(ge, s);
    hipDeviceSynchronize();
    double host = 0, total = 0;
    const int R = 20;
    for (int r = 0; r < R; ++r) {
      hipDeviceSynchronize();
      const double t0 = now_us();
      hipGraphLaunch(ge, s);
      const double t1 = now_us();
      hipStreamSynchronize(s);
      const double t2 = now_us();
      host += t1 - t0; total += t2 - t0;
    }
    printf("hipGraphLaunch, linear chain of %3d empty kernels: host %7.1f us per replay (%5.2f us/node), until done %7.1f us (%5.2f us/node)\n",
           n, host / R, host / R / n, total / R, total / R / n);
    // the same chain eagerly, for the GPU-side comparison
    hipDeviceSynchronize();
    const double t0 = now_us();
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_big, dim3(256), dim3(256), 0, s, big, buf);
    const double t1 = now_us();
    hipStreamSynchronize(s);
    const double t2 = now_us();
    printf("eager,          linear chain of %3d empty kernels: host %7.1f us (%5.2f us/launch), until done %7.1f us (%5.2f us/node)\n",
           n, t1 - t0, (t1 - t0) / n, t2 - t0, (t2 - t0) / n);
    hipGraphExecDestroy(ge);
    hipGraphDestroy(g);
  }
  // forked graph: main chain of 60 + a side chain of 30 forked / joined every other node (the backward pass's shape)
  {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 30; ++i) {
      hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, s, buf, 1500);
      hipEventRecord(ev[i], s);
      hipStreamWaitEvent(s2, ev[i], 0);
      hipLaunchKernelGGL(k_work, dim3(128), dim3(256), 0, s2, buf + (1 << 20), 3000);
      hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, s, buf, 1500);
    }
    hipEventRecord(ev[63], s2);
    hipStreamWaitEvent(s, ev[63], 0);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 3; ++w) hipGraphLaunch(ge, s);
    hipDeviceSynchronize();
    double host = 0, total = 0;
    const int R = 20;
    for (int r = 0; r < R; ++r) {
      hipDeviceSynchronize();
      const double t0 = now_us();
      hipGraphLaunch(ge, s);
      const double t1 = now_us();
      hipStreamSynchronize(s);
      const double t2 = now_us();
      host += t1 - t0; total += t2 - t0;
    }
    printf("forked graph (60 main + 30 side kernels, 30 forks): host %7.1f us per replay, until done %7.1f us\n", host / R, total / R);
    double eh = 0, et = 0;
    for (int r = 0; r < R; ++r) {
      hipDeviceSynchronize();
      const double t0 = now_us();
      for (int i = 0; i < 30; ++i) {
        hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, s, buf, 1500);
        hipEventRecord(ev[i], s);
        hipStreamWaitEvent(s2, ev[i], 0);
        hipLaunchKernelGGL(k_work, dim3(128), dim3(256), 0, s2, buf + (1 << 20), 3000);
        hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, s, buf, 1500);
      }
      hipEventRecord(ev[63], s2);
      hipStreamWaitEvent(s, ev[63], 0);
      const double t1 = now_us();
      hipStreamSynchronize(s);
      const double t2 = now_us();
      eh += t1 - t0; et += t2 - t0;
    }
    printf("the same eagerly on two streams:                     host %7.1f us,            until done %7.1f us\n", eh / R, et / R);
    hipGraphExecDestroy(ge);
    hipGraphDestroy(g);
  }
  return 0;
}
