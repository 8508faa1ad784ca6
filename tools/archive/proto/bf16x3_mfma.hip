// Prototype / measurement only (NOT part of libpdes_hip.so): can fp32-accurate contractions run faster on the bf16
// matrix pipe of gfx950 than on the f32 pipe?  Each fp32 operand x is split into three bf16 terms
//   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)        (8 + 8 + 8 = 24 significant bits)
// and a product a*b is accumulated in fp32 from the six cross terms of weight >= 2^-16:
//   ah*bh + ah*bm + am*bh + ah*bl + al*bh + am*bm        (dropped: am*bl, al*bm, al*bl ~ 2^-24 and below)
// Kernel A: v_mfma_f32_16x16x4_f32, K = 4 per instruction.  Kernel B: v_mfma_f32_16x16x32_bf16, K = 32, x6.
// Both: one wave = 8 M-tiles x 2 N-tiles of accumulators (the register tile of the wide-layer convolution kernel),
// A operand from LDS, B operand from global memory, K loop of `ksteps` x 32.
// Build + run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/bf16x3 tools/archive/proto/bf16x3_mfma.hip && /tmp/bf16x3
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

static inline unsigned short f2bf(float x) {          // host: round to nearest even
  unsigned u;
  std::memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static inline float bf2f(unsigned short h) {
  unsigned u = (unsigned)h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

constexpr int MT = 8, NT = 2, KC = 32;      // per wave: 128 x 32 outputs; K chunk of 32

// A: [M = 128][K] fp32 (row-major), B: [K][N = 32] fp32.  One workgroup = 4 waves computing the SAME tile with
// different K ranges would need a reduction; here every wave computes its own copy (throughput measurement) and
// wave 0 of block 0 writes the result (accuracy check).
__global__ __launch_bounds__(256) void gemm_f32(const float* __restrict__ A, const float* __restrict__ B,
                                                float* __restrict__ C, int K, int reps) {
  extern __shared__ float lds[];                 // [K][128] : k-major so that a lane reads lds[k][i]
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < K * 128; i += 256) lds[i] = A[(i % 128) * K + i / 128];
  __syncthreads();
  v4f acc[MT][NT];
  for (int m = 0; m < MT; ++m) for (int n = 0; n < NT; ++n) acc[m][n] = (v4f){0, 0, 0, 0};
  for (int rep = 0; rep < reps; ++rep)
  for (int k0 = 0; k0 < K; k0 += 4) {
    const int k = k0 + (lane >> 4);
    float b[NT];
    for (int n = 0; n < NT; ++n) b[n] = B[k * 32 + n * 16 + (lane & 15)];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float a = lds[k * 128 + m * 16 + (lane & 15)];
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[n], acc[m][n], 0, 0, 0);
    }
  }
  if (blockIdx.x == 0 && tid < 64)
    for (int m = 0; m < MT; ++m) for (int n = 0; n < NT; ++n) for (int r = 0; r < 4; ++r)
      C[(m * 16 + (lane >> 4) * 4 + r) * 32 + n * 16 + (lane & 15)] = acc[m][n][r];
}

// pre-split operands: Asplit[3][M = 128][K] bf16 (k contiguous), Bsplit[3][N = 32][K] bf16 (k contiguous)
__global__ __launch_bounds__(256) void gemm_bf16x3(const unsigned short* __restrict__ As, const unsigned short* __restrict__ Bs,
                                                   float* __restrict__ C, int K, int terms, int reps) {
  extern __shared__ unsigned short ldsb[];       // [3][128][K] bf16, row pitch K (+8 pad to spread banks)
  const int tid = threadIdx.x, lane = tid & 63;
  const int P = K + 8;
  for (int i = tid; i < 3 * 128 * K; i += 256) {
    const int pl = i / (128 * K), r = (i / K) % 128, k = i % K;
    ldsb[(pl * 128 + r) * P + k] = As[i];
  }
  __syncthreads();
  v4f acc[MT][NT];
  for (int m = 0; m < MT; ++m) for (int n = 0; n < NT; ++n) acc[m][n] = (v4f){0, 0, 0, 0};
  for (int rep = 0; rep < reps; ++rep)
  for (int k0 = 0; k0 < K; k0 += KC) {
    const int k = k0 + 8 * (lane >> 4);
    v8bf b[3][NT];
    for (int p = 0; p < 3; ++p)
      for (int n = 0; n < NT; ++n)
        b[p][n] = *reinterpret_cast<const v8bf*>(Bs + ((size_t)p * 32 + n * 16 + (lane & 15)) * K + k);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      v8bf a[3];
      for (int p = 0; p < 3; ++p)
        a[p] = *reinterpret_cast<const v8bf*>(ldsb + ((size_t)p * 128 + m * 16 + (lane & 15)) * P + k);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        v4f c = acc[m][n];
        if (terms >= 6) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1][n], c, 0, 0, 0);   // smallest terms first
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0][n], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2][n], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0][n], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1][n], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0][n], c, 0, 0, 0);
        acc[m][n] = c;
      }
    }
  }
  if (blockIdx.x == 0 && tid < 64)
    for (int m = 0; m < MT; ++m) for (int n = 0; n < NT; ++n) for (int r = 0; r < 4; ++r)
      C[(m * 16 + (lane >> 4) * 4 + r) * 32 + n * 16 + (lane & 15)] = acc[m][n][r];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e, __FILE__, __LINE__); exit(1); } } while (0)

int main() {
  const int K = 128, M = 128, N = 32, REPS = 32;      // timing: the K loop is repeated REPS times per launch
  std::vector<float> A(M * K), B(K * N);
  srand(1);
  for (auto& v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (auto& v : B) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.1f;
  std::vector<double> ref(M * N, 0.0);
  for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * B[k * N + j]; ref[i * N + j] = s; }
  std::vector<unsigned short> As(3 * M * K), Bs(3 * N * K);
  for (int i = 0; i < M; ++i) for (int k = 0; k < K; ++k) {
    float x = A[i * K + k]; unsigned short h = f2bf(x); float r1 = x - bf2f(h); unsigned short m = f2bf(r1); float r2 = r1 - bf2f(m);
    As[(0 * M + i) * K + k] = h; As[(1 * M + i) * K + k] = m; As[(2 * M + i) * K + k] = f2bf(r2);
  }
  for (int k = 0; k < K; ++k) for (int j = 0; j < N; ++j) {
    float x = B[k * N + j]; unsigned short h = f2bf(x); float r1 = x - bf2f(h); unsigned short m = f2bf(r1); float r2 = r1 - bf2f(m);
    Bs[(0 * N + j) * K + k] = h; Bs[(1 * N + j) * K + k] = m; Bs[(2 * N + j) * K + k] = f2bf(r2);
  }
  float *dA, *dB, *dC; unsigned short *dAs, *dBs;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, M * N * 4));
  CK(hipMalloc(&dAs, As.size() * 2)); CK(hipMalloc(&dBs, Bs.size() * 2));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dAs, As.data(), As.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dBs, Bs.data(), Bs.size() * 2, hipMemcpyHostToDevice));
  std::vector<float> C(M * N);
  auto err = [&](const char* tag) {
    CK(hipMemcpy(C.data(), dC, M * N * 4, hipMemcpyDeviceToHost));
    double num = 0, den = 0, mx = 0;
    for (int i = 0; i < M * N; ++i) { double d = C[i] - ref[i]; num += d * d; den += ref[i] * ref[i]; mx = fmax(mx, fabs(d)); }
    printf("%-28s rel-L2 error vs fp64 %.3e, max abs %.3e\n", tag, sqrt(num / den), mx);
  };
  const int blocks = 256, iters = 20;                      // one 4-wave workgroup per CU (one wave per SIMD)
  const size_t ldsA = (size_t)K * 128 * 4, ldsB = (size_t)3 * 128 * (K + 8) * 2;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto bench = [&](const char* tag, auto launch, double flops_per_wave) {
    launch(); CK(hipGetLastError()); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < iters; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double t = ms * 1e-3 / iters;
    printf("%-28s %.1f us per launch, %.1f fp32-equivalent TFLOP/s (includes the LDS fill of every workgroup)\n", tag, t * 1e6,
           flops_per_wave * 4 * blocks / t / 1e12);
  };
  const double fl = 2.0 * M * N * K * REPS;
  bench("f32 MFMA 16x16x4", [&] { hipLaunchKernelGGL(gemm_f32, dim3(blocks), dim3(256), ldsA, 0, dA, dB, dC, K, REPS); }, fl);
  hipLaunchKernelGGL(gemm_f32, dim3(1), dim3(256), ldsA, 0, dA, dB, dC, K, 1); CK(hipDeviceSynchronize());
  err("f32 MFMA 16x16x4");
  bench("bf16x3, 6 terms (16x16x32)", [&] { hipLaunchKernelGGL(gemm_bf16x3, dim3(blocks), dim3(256), ldsB, 0, dAs, dBs, dC, K, 6, REPS); }, fl);
  hipLaunchKernelGGL(gemm_bf16x3, dim3(1), dim3(256), ldsB, 0, dAs, dBs, dC, K, 6, 1); CK(hipDeviceSynchronize());
  err("bf16x3, 6 terms");
  bench("bf16x3, 5 terms", [&] { hipLaunchKernelGGL(gemm_bf16x3, dim3(blocks), dim3(256), ldsB, 0, dAs, dBs, dC, K, 5, REPS); }, fl);
  hipLaunchKernelGGL(gemm_bf16x3, dim3(1), dim3(256), ldsB, 0, dAs, dBs, dC, K, 5, 1); CK(hipDeviceSynchronize());
  err("bf16x3, 5 terms");
  return 0;
}
