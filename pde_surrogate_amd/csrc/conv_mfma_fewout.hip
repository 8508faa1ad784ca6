// 5x5 convolution with very few output channels (the last layer of DenseED: 49 -> 3, reference
// models/codec.py:163-188) on the f32 matrix cores, gfx950.
//
// With 3 output channels an implicit GEMM whose N dimension is "output channels" fills 3 of the 16 MFMA
// columns.  Here N = (output channel, kernel column kx): 3 x 5 = 15 columns.  For one output row y
//     P[x'][(co,kx)] = sum_{ci, ky} z[ci][y + ky - 2][x'] * W[co][ci][ky][kx]        (x' = input column)
// is accumulated over input channels and kernel rows by 5 MFMAs per 4 channels instead of 25, and
//     out[co][y][x] = sum_kx P[x + kx - 2][(co,kx)]
// is a 5-term shift-add done once per tile through LDS.  GEMM roles per v_mfma_f32_16x16x4_f32:
// M = 16 consecutive input columns x' (A: one ds_read_b32 per lane from the BN+ReLU'd LDS tile), N = (co,kx)
// (B: gathered straight from the (Cout,Cin,5,5) weight tensor, 5 loads per k-step), K = 4 input channels.
// Workgroup = R output rows of one sample, full width; its 4 waves split K (one k-step of every 16-channel
// chunk each) and are summed through LDS before the shift-add.  Staging is the straight-line, two-stage
// register pipeline of conv_mfma.hip, but through ONE LDS tile (two barriers per chunk): with two tile buffers and the
// whole exchange area a workgroup held 69 KB and the 1024 workgroups of the 64 x 64 layer ran as two rounds of two per
// CU; with 35 KB four are resident and cover each other's barriers.
#include <stdlib.h>
#include "pdes_common.h"
#include "pdes_options.h"
#include "../../include/pdes_hip.h"

namespace pdes {

typedef float v4f __attribute__((ext_vector_type(4)));

template <int NMT, int R>     // NMT: M-tiles per row = ceil((W + 4) / 16); R: output rows per workgroup
struct FewGeo {
  static constexpr int W = NMT == 5 ? 64 : (NMT == 3 ? 32 : 16);
  static constexpr int ROWS = R + 4;
  static constexpr int LDW = 16 * NMT + 4;                    // cols 2,3 = x' -2,-1; interior from col 4 (16-B aligned)
  static constexpr int CS = ((ROWS * LDW - 16 + 31) / 32) * 32 + 16;   // == 16 (mod 32): see conv_mfma.hip
  static constexpr int NV4 = 16 * ROWS * (W / 4);
  static constexpr int NPV = (NV4 + 255) / 256;
  static_assert(16 * NMT >= W + 4 && CS >= ROWS * LDW, "tile geometry");
};

template <int NMT, int R>
__global__ __launch_bounds__(256) void conv5_fewout_fwd_kernel(pdes_conv_desc d, int xcd_map) {
  using G = FewGeo<NMT, R>;
  constexpr int W = G::W;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Workgroups go to the 8 XCDs round-robin by their linear id, and each XCD has its own L2: with (row block, image) =
  // blockIdx, the row blocks of ONE image -- which re-read each other's halo rows: (R + 4) / R = 3 x the input at R = 2 -- sit
  // on 8 different L2s and every re-read goes to the Infinity Cache (100 MB per launch for the 49 -> 3 layer: the kernel
  // ran at that bandwidth, not at its matrix time).  Remapped so that an XCD takes whole images (round 6).
  int b = blockIdx.y, rb = blockIdx.x;
  if (xcd_map && (gridDim.y & 7) == 0) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y, xcd = lin & 7, j = lin >> 3;
    b = xcd + 8 * (j / (int)gridDim.x);
    rb = j % (int)gridDim.x;
  }
  const int oy0 = rb * R;
  const int H = d.Hin, HW = H * W;
  const int kpad = (d.Cin + 15) & ~15, nchunk = kpad / 16, ksteps = kpad / 4;
  float4* cf4 = reinterpret_cast<float4*>(smem);            // [kpad] {mean, gamma*invstd, beta, -}
  float* tile = smem + 4 * kpad;                            // [16][CS]

  for (int c = tid; c < kpad; c += 256) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < d.Cin) {
      float mean, invstd;
      if (d.eval_mode) {
        mean = d.run_mean[c];
        invstd = (float)(1.0 / sqrt((double)d.run_var[c] + (double)d.eps));
      } else {
        const MeanInv mi = batch_mean_invstd(d.coef, d.x_stats, d.rep_stride, (double)d.B * HW, d.eps, c, (blockIdx.x | blockIdx.y) == 0);
        mean = mi.mean;
        invstd = mi.invstd;
      }
      v = make_float4(mean, d.gamma[c] * invstd, d.beta[c], 0.f);
    }
    cf4[c] = v;
  }
  // pad columns (x' < 0 and x' >= W) are never written by the staging: zero both buffers once
  for (int i = tid; i < 16 * G::CS; i += 256) tile[i] = 0.f;

  // ---- staging geometry (chunk independent)
  const float* xb = d.x + (size_t)b * d.x_ctot * HW;
  int vg[G::NPV], vl[G::NPV];
  unsigned vrow = 0;
#pragma unroll
  for (int i = 0; i < G::NPV; ++i) {
    const int e = tid + 256 * i;
    const int ch = e / (G::ROWS * (W / 4)), rem = e % (G::ROWS * (W / 4));
    const int r = rem / (W / 4), j = rem % (W / 4);
    const int y = oy0 - 2 + r;
    vg[i] = min(max(y, 0), H - 1) * W + 4 * j;
    vl[i] = e < G::NV4 ? ch * G::CS + r * G::LDW + 4 + 4 * j : -1;
    if (e < G::NV4 && y >= 0 && y < H) vrow |= 1u << i;
  }
  struct Stage { float4 pv[G::NPV]; };
  Stage sA, sB;
  auto issue = [&](int chunk, Stage& st) __attribute__((always_inline)) {
    const float* src = xb + (size_t)chunk * 16 * HW;
    const int cmax = d.Cin - chunk * 16 - 1;
#pragma unroll
    for (int i = 0; i < G::NPV; ++i) {
      const int ch = min((tid + 256 * i) / (G::ROWS * (W / 4)), cmax);
      st.pv[i] = *reinterpret_cast<const float4*>(src + ch * HW + vg[i]);
    }
  };
  auto commit = [&](int chunk, const Stage& st) __attribute__((always_inline)) {
    float* t = tile;
    const int crem = d.Cin - chunk * 16;
#pragma unroll
    for (int i = 0; i < G::NPV; ++i) {
      if (vl[i] >= 0) {
        const int ch = (tid + 256 * i) / (G::ROWS * (W / 4));
        const bool ok = ((vrow >> i) & 1u) && ch < crem;
        const float4 k = cf4[chunk * 16 + ch];
        float4 z = st.pv[i];
        z.x = ok ? fmaxf(0.f, (z.x - k.x) * k.y + k.z) : 0.f;
        z.y = ok ? fmaxf(0.f, (z.y - k.x) * k.y + k.z) : 0.f;
        z.z = ok ? fmaxf(0.f, (z.z - k.x) * k.y + k.z) : 0.f;
        z.w = ok ? fmaxf(0.f, (z.w - k.x) * k.y + k.z) : 0.f;
        *reinterpret_cast<float4*>(t + vl[i]) = z;
      }
    }
  };

  // B operand gathered from the weight tensor: lane (k = lane>>4, n = lane&15), n = co*5 + kx
  const int bn = lane & 15, bk = lane >> 4;
  const int bco = min(bn / 5, d.Cout - 1), bkx = bn % 5;
  const float bmask = (bn < 5 * d.Cout) ? 1.f : 0.f;
  float bA[5], bB[5];
  auto load_b = [&](int kstep, float (&dst)[5]) __attribute__((always_inline)) {
    const int ci = min(4 * min(kstep, ksteps - 1) + bk, d.Cin - 1);        // rows >= Cin meet a zero A operand
    const float* wp = d.w + ((size_t)(bco * d.Cin + ci) * 5) * 5 + bkx;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) dst[ky] = wp[ky * 5];
  };

  v4f acc[R][NMT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int j = 0; j < NMT; ++j) acc[r][j] = (v4f){0.f, 0.f, 0.f, 0.f};

  load_b(wave, bA);
  issue(0, sA);
  issue(min(1, nchunk - 1), sB);
  __syncthreads();                 // cf4 and the zeroed pads are visible
  commit(0, sA);
  __syncthreads();

  const int a_lane = (lane >> 4) * G::CS + 2 + (lane & 15);
  auto step = [&](int chunk, Stage& sfree, const Stage& snext, float (&b0)[5], float (&b1)[5]) __attribute__((always_inline)) {
    load_b((chunk + 1) * 4 + wave, b1);
    issue(min(chunk + 2, nchunk - 1), sfree);
    const float* tk = tile + wave * 4 * G::CS + a_lane;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
      const float bw = b0[ky] * bmask;
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < NMT; ++j)
          acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(tk[(r + ky) * G::LDW + 16 * j], bw, acc[r][j], 0, 0, 0);
    }
    __syncthreads();                                   // every wave is done with this chunk's tile
    if (chunk + 1 < nchunk) {
      commit(chunk + 1, snext);
      __syncthreads();
    }
  };
  {
    int chunk = 0;
    for (; chunk + 1 < nchunk; chunk += 2) { step(chunk, sA, sB, bA, bB); step(chunk + 1, sB, sA, bB, bA); }
    if (chunk < nchunk) step(chunk, sA, sB, bA, bB);
  }

  // ---- sum the 4 K-split waves and shift-add the kernel columns, one output row at a time.  Exchange layout
  // red[w][p][n] with a pixel pitch of 17 floats: the readers below walk consecutive pixels p (stride 17 -> 32 distinct banks)
  float* red = tile;
  constexpr int PP = 17, NPX = 16 * NMT;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (r) __syncthreads();                            // the previous row's readers are done
#pragma unroll
    for (int j = 0; j < NMT; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        red[(wave * NPX + 16 * j + (lane >> 4) * 4 + q) * PP + (lane & 15)] = acc[r][j][q];
    __syncthreads();
    const int nout = d.Cout * W;
    for (int o = tid; o < nout; o += 256) {
      const int x = o % W, co = o / W;
      float s = 0.f;
#pragma unroll
      for (int kx = 0; kx < 5; ++kx) {
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red[(w * NPX + x + kx) * PP + co * 5 + kx];   // P index = x' + 2 = x + kx
      }
      d.out[((size_t)b * d.out_ctot + d.out_coff + co) * HW + (size_t)(oy0 + r) * W + x] = s;
    }
  }
}

// PDES_ENOSUP when the layer is not of this shape (the caller tries the generic kernels next)
int conv_forward_fewout(const pdes_conv_desc& d, hipStream_t st, bool dry) {       // dry: capability query only
  if (d.ksize != 5 || d.stride != 1 || d.pad != 2 || d.upsample || !d.has_bn || d.Cout * 5 > 16 || d.Cin < 16)
    return PDES_ENOSUP;
  if (d.out_stats || d.Hin != d.Hout || d.Win != d.Wout || d.nrep != PDES_NREP || !d.w) return PDES_ENOSUP;
  if (!(d.Win == 64 || d.Win == 32 || d.Win == 16) || d.Hin % 2) return PDES_ENOSUP;
  if (dry) return PDES_OK;
  const int kpad = (d.Cin + 15) & ~15;
  const int R = 2;            // rows per workgroup (4 measured +0.4 % on the step)
  dim3 grid(d.Hin / R, d.B), block(256);
#define PDES_FEW_LAUNCH(NMT_, R_)                                                                    \
  do {                                                                                                \
    using G = FewGeo<NMT_, R_>;                                                                       \
    size_t fl = (size_t)16 * G::CS;                                                                   \
    const size_t red = (size_t)4 * 16 * NMT_ * 17;                                                     \
    if (red > fl) fl = red;                                                                           \
    hipLaunchKernelGGL((conv5_fewout_fwd_kernel<NMT_, R_>), grid, block, (4 * (size_t)kpad + fl) * sizeof(float), st, d, opt().xcd_map); \
  } while (0)
  if (R == 4) {
    if (d.Win == 64) PDES_FEW_LAUNCH(5, 4);
    else if (d.Win == 32) PDES_FEW_LAUNCH(3, 4);
    else PDES_FEW_LAUNCH(2, 4);
  } else {
    if (d.Win == 64) PDES_FEW_LAUNCH(5, 2);
    else if (d.Win == 32) PDES_FEW_LAUNCH(3, 2);
    else PDES_FEW_LAUNCH(2, 2);
  }
#undef PDES_FEW_LAUNCH
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

}  // namespace pdes
