// What can a pure streaming kernel reach with the loss kernel's traffic mix?  Per pixel the fused Sobel + Darcy loss
// reads 4 planes (K, u, sigma1, sigma2) and writes 3 (the gradient): 28 B.  This kernel does exactly that with float4
// accesses and no arithmetic to speak of, at the roofline measurement's size (B = 16384 images of 64 x 64: 1.88 GB), with
// plain and with non-temporal accesses; plus a 1:1 copy of the same total bytes.
// Built by hand: hipcc --offload-arch=gfx950 -O2 tools/archive/proto/stream_mix.hip -o /tmp/stream_mix   (not part of the library)
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float nvec4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void mix43(const nvec4* __restrict__ k, const nvec4* __restrict__ y, nvec4* __restrict__ g,
                                             size_t n4) {           // n4 = float4 per plane
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const size_t img = i / 1024, p = i % 1024;                     // 64 x 64 / 4 float4 per plane and image
    const nvec4* yb = y + img * 3072 + p;
    nvec4 a, b, c, d;
    if (NT) {
      a = __builtin_nontemporal_load(k + i); b = __builtin_nontemporal_load(yb);
      c = __builtin_nontemporal_load(yb + 1024); d = __builtin_nontemporal_load(yb + 2048);
    } else { a = k[i]; b = yb[0]; c = yb[1024]; d = yb[2048]; }
    nvec4* gb = g + img * 3072 + p;
    const nvec4 r0 = a + b, r1 = c * a, r2 = d - b;
    if (NT) { __builtin_nontemporal_store(r0, gb); __builtin_nontemporal_store(r1, gb + 1024); __builtin_nontemporal_store(r2, gb + 2048); }
    else { gb[0] = r0; gb[1024] = r1; gb[2048] = r2; }
  }
}
__global__ __launch_bounds__(256) void copy11(const nvec4* __restrict__ s, nvec4* __restrict__ d, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
    __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i);
}

int main() {
  const size_t B = 16384, n4 = B * 1024;
  nvec4 *k, *y, *g;
  hipMalloc(&k, n4 * 16); hipMalloc(&y, 3 * n4 * 16); hipMalloc(&g, 3 * n4 * 16);
  hipMemset(k, 0, n4 * 16); hipMemset(y, 0, 3 * n4 * 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const double bytes = 7.0 * n4 * 16;
  for (int mode = 0; mode < 3; ++mode)
    for (int grid : {2048, 8192, 32768}) {
      for (int rep = 0; rep < 2; ++rep) {
        for (int w = 0; w < (rep ? 0 : 50); ++w) {
          if (mode == 0) hipLaunchKernelGGL(mix43<false>, dim3(grid), dim3(256), 0, 0, k, y, g, n4);
        }
        hipEventRecord(e0, 0);
        for (int it = 0; it < 50; ++it) {
          if (mode == 0) hipLaunchKernelGGL(mix43<false>, dim3(grid), dim3(256), 0, 0, k, y, g, n4);
          else if (mode == 1) hipLaunchKernelGGL(mix43<true>, dim3(grid), dim3(256), 0, 0, k, y, g, n4);
          else hipLaunchKernelGGL(copy11, dim3(grid), dim3(256), 0, 0, y, g, 3 * n4);
        }
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double by = mode == 2 ? 6.0 * n4 * 16 : bytes;
        if (rep) printf("%s grid %5d: %.1f us per pass, %.0f GB/s (%.3f of 8 TB/s)\n",
                        mode == 0 ? "4R:3W plain  " : (mode == 1 ? "4R:3W nontemp" : "copy 1:1 nt  "), grid, ms * 20.f,
                        by / (ms / 50 * 1e-3) / 1e9, by / (ms / 50 * 1e-3) / 8e12);
      }
    }
  return 0;
}
