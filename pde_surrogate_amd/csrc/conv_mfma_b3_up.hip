// Nearest-x2 upsampling + 3x3 convolution in SUB-PIXEL form (four 2x2 convolutions of the low-resolution map with
// effective weights, see conv_mfma_up.hip) on the bf16 matrix pipe with fp32 accuracy (three-way bf16 split of both
// operands, six cross products, see conv_mfma_b3.hip): the forward of TransUp*.conv2 and LastTransUp.conv2
// (reference models/codec.py:141-146, :176-181).
//
// Same tile and staging as conv_mfma_b3.hip: the 3x3 halo tile of the LOW-res map, pixel-major / channel-minor,
// [plane hi|mid|lo][row][pixel][32 channels] bf16, BatchNorm + ReLU + split applied on the way in.  Every (M-tile,
// N-tile) keeps FOUR accumulators (one per output parity); the 9 tile positions are walked once per 32-channel chunk,
// a position's A fragments (3 planes x MT tiles) are read once and feed the 1, 2 or 4 parities that use it
// (16 (parity, tap) pairs per chunk = 16 B-operand sets, streamed through two register sets in that order).
// One N-tile per wave (4 accumulators x 4 M-tiles = 64 registers), gridDim.z covers the N-tiles in groups of 4.
// The epilogue interleaves the column parities into float4 stores of the hi-res rows 2y and 2y + 1.
#include "pdes_common.h"
#include "pdes_options.h"
#include "../../include/pdes_hip.h"
#include "pack_kernels.h"

namespace pdes {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

template <int TWG>
struct B3UGeo {
  static constexpr int MT = 4, TH = MT / TWG, TW = 16 * TWG;
  static constexpr int ROWS = TH + 2, PW = TW + 2, KC = 32;
  static constexpr int PLANE = ROWS * PW * KC;                 // bf16 elements of one plane of one buffer
  static constexpr int BUF = 3 * PLANE;
  static constexpr int QX = TW / 4;
  static constexpr int NI = ROWS * QX * 4;                     // interior work items (row, quad, channel octet)
  static constexpr int NHI = ROWS * 2 * 4;                     // halo work items (row, side, channel octet)
  static_assert(NI + NHI <= 256, "one work item per thread");
  static constexpr int FPW = PW + 2, FCS = ROWS * FPW;         // K tail on the f32 pipe: [4 channels][ROWS][FPW] floats
};

struct BnBU { float mean, invstd, gamma, beta; };
__device__ __forceinline__ BnBU bn_coef_b3u(const pdes_conv_desc& d, int c, bool publish = false) {
  BnBU o;
  if (d.eval_mode) {
    o.mean = d.run_mean[c];
    o.invstd = (float)(1.0 / sqrt((double)d.run_var[c] + (double)d.eps));
  } else {
    const MeanInv mi = batch_mean_invstd(d.coef, d.x_stats, d.rep_stride, (double)d.B * d.Hin * d.Win, d.eps, c, publish);
    o.mean = mi.mean;
    o.invstd = mi.invstd;
  }
  o.gamma = d.gamma[c];
  o.beta = d.beta[c];
  return o;
}

// grid: (tiles of the LOW-res map, B, ceil(N-tiles / 4)); dynamic LDS: [kpad32] float4 coefficients + 2 buffers
template <int TWG>
__global__ __launch_bounds__(256, 2) void conv_b3_up_fwd_kernel(pdes_conv_desc d, const unsigned short* __restrict__ wb,
                                                                int nt_total, int tail_on) {
  using G = B3UGeo<TWG>;
  constexpr int MT = G::MT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b3u[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int ntp = (nt_total + 3) & ~3;
  const int nt_w = blockIdx.z * 4 + wave;                      // this wave's N-tile (< ntp)
  const int kC = d.Cin, H = d.Hin, W = d.Win, HW = H * W;      // low-res map
  const int Wh = d.Wout, HWh = d.Hout * d.Wout;
  const float* kbase = d.x + (size_t)b * d.x_ctot * HW;
  const int nchunk = (kC + G::KC - 1) / G::KC, kpad = nchunk * G::KC;
  float4* cf4 = reinterpret_cast<float4*>(smem_b3u);
  unsigned short* tile = reinterpret_cast<unsigned short*>(smem_b3u + 16 * kpad);
  // K tail (98 = 3 x 32 + 2 input channels): the <= 4 channels of the last chunk on v_mfma_f32_16x16x4_f32, one per
  // (position, parity) pair instead of six bf16 ones (conv_mfma_b3.hip); the effective weights are summed in registers
  const int rtail = kC - G::KC * (nchunk - 1);
  const bool tail = tail_on && nchunk >= 2 && rtail <= 4;
  const int nb = tail ? nchunk - 1 : nchunk;
  float* ftile = reinterpret_cast<float*>(tile + 2 * G::BUF);
  if (tail)
    for (int i = tid; i < 4 * G::FCS; i += 256) ftile[i] = 0.f;
  const int tiles_x = W / G::TW;
  const int oy0 = (blockIdx.x / tiles_x) * G::TH, ox0 = (blockIdx.x % tiles_x) * G::TW;

  for (int c = tid; c < kpad; c += 256) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < d.Cin) { const BnBU k = bn_coef_b3u(d, c, (blockIdx.x | blockIdx.y | blockIdx.z) == 0); v = make_float4(k.mean, k.gamma * k.invstd, k.beta, 0.f); }
    cf4[c] = v;
  }

  // ---- staging geometry: one work item per thread (as conv_mfma_b3.hip)
  const bool interior = tid < G::NI, halo = !interior && tid < G::NI + G::NHI;
  int it_r, it_c, it_o;
  if (interior) { it_o = tid & 3; it_c = 1 + 4 * ((tid >> 2) % G::QX); it_r = tid / (4 * G::QX); }
  else { const int t = tid - G::NI; it_o = t & 3; it_c = ((t >> 2) & 1) ? G::PW - 1 : 0; it_r = (t >> 3) % G::ROWS; }
  const int gy = oy0 - 1 + it_r, gx = ox0 - 1 + it_c;
  const bool row_ok = gy >= 0 && gy < H;
  const bool px_ok = row_ok && (interior || (halo && gx >= 0 && gx < W));
  const int goff = min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1);
  const int lds_off = (it_r * G::PW + it_c) * G::KC;
  // LDS swizzle of conv_mfma_b3.hip (octet bit 1 ^= bit 2 of the pixel's tile column: conflict-free fragment reads)
  auto swz_oct = [](int col, int oct) __attribute__((always_inline)) { return oct ^ (((col >> 2) & 1) << 1); };

  struct Stage { float4 v[8]; };
  Stage sA;                           // ONE register stage: the next chunk's loads have a whole matrix phase to land
  auto issue = [&](int chunk, Stage& st) __attribute__((always_inline)) {
    const int c0 = chunk * G::KC + 8 * it_o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float* p = kbase + (size_t)min(c0 + j, kC - 1) * HW + goff;
      if (interior) st.v[j] = *reinterpret_cast<const float4*>(p);
      else st.v[j].x = *p;
    }
  };
  auto commit = [&](int chunk, int buf, const Stage& st) __attribute__((always_inline)) {
    if (!(interior || halo)) return;
    const int c0 = chunk * G::KC + 8 * it_o;
    unsigned short* t = tile + buf * G::BUF + lds_off;
    const int npx = interior ? 4 : 1;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (p >= npx) break;
      u32 hw[4], mw[4], lw[4];
      float xv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float x = p == 0 ? st.v[j].x : (p == 1 ? st.v[j].y : (p == 2 ? st.v[j].z : st.v[j].w));
        const bool ok = px_ok && c0 + j < kC;
        const float4 k = cf4[min(c0 + j, kpad - 1)];
        x = fmaxf(0.f, (x - k.x) * k.y + k.z);
        xv[j] = ok ? x : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) split3_pair(xv[2 * j], xv[2 * j + 1], hw[j], mw[j], lw[j]);
      unsigned short* q = t + p * G::KC + 8 * swz_oct(it_c + p, it_o);
      *reinterpret_cast<uint4*>(q) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
      *reinterpret_cast<uint4*>(q + G::PLANE) = make_uint4(mw[0], mw[1], mw[2], mw[3]);
      *reinterpret_cast<uint4*>(q + 2 * G::PLANE) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    }
  };
  auto commit_tail = [&](const Stage& st) __attribute__((always_inline)) {
    if (!(interior || halo) || it_o != 0) return;
    const int c0 = (nchunk - 1) * G::KC;
    float* t = ftile + it_r * G::FPW + it_c;
    const int npx = interior ? 4 : 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < rtail) {
        const float4 k = cf4[c0 + j];
        float xv[4] = {st.v[j].x, st.v[j].y, st.v[j].z, st.v[j].w};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          if (p >= npx) break;
          t[j * G::FCS + p] = px_ok ? fmaxf(0.f, (xv[p] - k.x) * k.y + k.z) : 0.f;
        }
      }
    }
  };

  v4f acc[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[mt][q] = (v4f){0.f, 0.f, 0.f, 0.f};

  // B operand: image [(chunk*16 + j)*ntp + nt][plane][64 lanes][8 bf16], j = (position, parity) pairs in the walking
  // order of the loop below (pack_b3up_item)
  v8bf bS[2][3];
  auto load_b = [&](int cj, v8bf (&dst)[3]) __attribute__((always_inline)) {
    const int cc = min(cj, nchunk * 16 - 1);
    const unsigned short* p = wb + (((size_t)cc * ntp + nt_w) * 3 * 64 + lane) * 8;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) dst[pl] = *reinterpret_cast<const v8bf*>(p + (size_t)pl * 64 * 8);
  };
  int a_lane_k[3];                   // per tap column tx: the lane's pixel is tile column tx + (lane & 15) (+ 16 per M-tile)
#pragma unroll
  for (int tx = 0; tx < 3; ++tx) a_lane_k[tx] = (lane & 15) * G::KC + 8 * swz_oct(tx + (lane & 15), lane >> 4);

  // tail: B operand of the f32 MFMA = Weff_p[a][b][co = lane & 15][tail channel lane >> 4], from the 9 taps of the tensor
  float wf[4][4];                    // [parity][a * 2 + b]
  auto load_tail_b = [&]() __attribute__((always_inline)) {
    const int ch = lane >> 4, n = nt_w * 16 + (lane & 15);
    const bool ok = ch < rtail && n < d.Cout;
    const float* wp = d.w + ((size_t)min(n, d.Cout - 1) * d.Cin + (kC - rtail + min(ch, rtail - 1))) * 9;
    float w9[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) { const float v = wp[t]; w9[t] = ok ? v : 0.f; }
#pragma unroll
    for (int pp = 0; pp < 4; ++pp)
#pragma unroll
      for (int ab = 0; ab < 4; ++ab) {
        const int rm = weff_mask(pp >> 1, ab >> 1), cm = weff_mask(pp & 1, ab & 1);
        float sum = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
            if ((rm >> ky) & (cm >> kx) & 1) sum += w9[ky * 3 + kx];
        wf[pp][ab] = sum;
      }
  };
  auto tail_mma = [&]() __attribute__((always_inline)) {
    const float* fa = ftile + (lane >> 4) * G::FCS + (lane & 15);
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
      for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const float a = fa[((mt / TWG) + ty) * G::FPW + (mt % TWG) * 16 + tx];
#pragma unroll
          for (int ddy = 0; ddy < 2; ++ddy)
#pragma unroll
            for (int ddx = 0; ddx < 2; ++ddx) {
              const int ia = ty - ddy, ib = tx - ddx;
              if (ia < 0 || ia > 1 || ib < 0 || ib > 1) continue;
              const int pp = ddy * 2 + ddx;
              acc[mt][pp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wf[pp][ia * 2 + ib], acc[mt][pp], 0, 0, 0);
            }
        }
  };

  load_b(0, bS[0]);
  issue(0, sA);
  __syncthreads();                   // coefficients visible
  commit(0, 0, sA);
  __syncthreads();

  auto step = [&](int chunk) __attribute__((always_inline)) {
    const int buf = chunk & 1;
    const unsigned short* tb = tile + buf * G::BUF;
    issue(min(chunk + 1, nchunk - 1), sA);
    if (tail && chunk + 1 == nb) load_tail_b();            // in flight during the last bf16 chunk
    __builtin_amdgcn_sched_barrier(0);
    int j = 0;                         // compile-time after unrolling: index of the (position, parity) pair
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        v8bf ah[MT], am[MT], al[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const unsigned short* ap = tb + (((mt / TWG) + ty) * G::PW + (mt % TWG) * 16 + tx) * G::KC + a_lane_k[tx];
          ah[mt] = *reinterpret_cast<const v8bf*>(ap);
          am[mt] = *reinterpret_cast<const v8bf*>(ap + G::PLANE);
          al[mt] = *reinterpret_cast<const v8bf*>(ap + 2 * G::PLANE);
        }
#pragma unroll
        for (int ddy = 0; ddy < 2; ++ddy)
#pragma unroll
          for (int ddx = 0; ddx < 2; ++ddx) {
            const int ia = ty - ddy, ib = tx - ddx;            // position inside the 2x2 effective kernel of this parity
            if (ia < 0 || ia > 1 || ib < 0 || ib > 1) continue;
            const int pp = ddy * 2 + ddx;
            load_b(chunk * 16 + j + 1, bS[(j + 1) & 1]);       // next pair's weights stream in behind this pair's MFMAs
            const v8bf (&bw)[3] = bS[j & 1];
            // six cross terms, smallest first; the M-tiles alternate so that consecutive MFMAs are independent
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][pp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[mt], bw[1], acc[mt][pp], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][pp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mt], bw[0], acc[mt][pp], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][pp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt], bw[2], acc[mt][pp], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][pp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[mt], bw[0], acc[mt][pp], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][pp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt], bw[1], acc[mt][pp], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][pp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt], bw[0], acc[mt][pp], 0, 0, 0);
            ++j;
          }
      }
    if (chunk + 1 < nb) commit(chunk + 1, buf ^ 1, sA);
    else if (tail) commit_tail(sA);
    __syncthreads();
  };
  for (int chunk = 0; chunk < nb; ++chunk) step(chunk);
  if (tail) tail_mma();

  // ---- epilogue: accumulator = D[pixel (lane >> 4) * 4 + r][channel lane & 15] per parity
  const int px = (lane >> 4) * 4;
  const int cn = nt_w * 16 + (lane & 15);
  float s = 0.f, q = 0.f;
  if (nt_w < nt_total && cn < d.Cout) {
    float* ob = d.out + ((size_t)b * d.out_ctot + d.out_coff + cn) * HWh;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int oy = oy0 + mt / TWG, ox = ox0 + (mt % TWG) * 16 + px;          // low-res
#pragma unroll
      for (int ddy = 0; ddy < 2; ++ddy) {
        const v4f v0 = acc[mt][ddy * 2], v1 = acc[mt][ddy * 2 + 1];
        float* row = ob + (size_t)(2 * oy + ddy) * Wh + 2 * ox;
        *reinterpret_cast<float4*>(row) = make_float4(v0[0], v1[0], v0[1], v1[1]);
        *reinterpret_cast<float4*>(row + 4) = make_float4(v0[2], v1[2], v0[3], v1[3]);
#pragma unroll
        for (int r = 0; r < 4; ++r) { s += v0[r] + v1[r]; q += v0[r] * v0[r] + v1[r] * v1[r]; }
      }
    }
  }
  if (d.out_stats) {
    s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
    q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
    if (lane < 16 && nt_w < nt_total && cn < d.Cout) {
      double* os = d.out_stats + (long long)rep_of_block(d.nrep) * d.rep_stride;
#ifndef PDES_FW_NOATOM          // (component-timing build: EXPERIMENTS.md round 4)
      atomicAdd(&os[2 * (d.out_coff + cn)], (double)s);
      atomicAdd(&os[2 * (d.out_coff + cn) + 1], (double)q);
#endif
    }
  }
}

__global__ __launch_bounds__(256) void pack_b3up_kernel(const pdes_b3up_pack_item* __restrict__ items) {
  pack_b3up_item(items[blockIdx.y], blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------- host dispatch
static bool b3up_shape_ok(const pdes_conv_desc& d) {
  if (d.ksize != 3 || d.stride != 1 || d.pad != 1 || d.upsample != PDES_UPSAMPLE_NEAREST || !d.has_bn || d.nrep != PDES_NREP) return false;
  if (d.Hout != 2 * d.Hin || d.Wout != 2 * d.Win) return false;
  if (d.Cin < 64 || d.Cout < 32) return false;
  const int W = d.Win, H = d.Hin;
  // low-res maps of at least 32 columns only: 98 -> 49 at 32x32 -> 64x64 gains (67.9 -> 52.1 us stand-alone), 100 -> 100
  // at 16x16 -> 32x32 does not (37.2 vs 39.0 us: 256 small workgroups, latency bound either way)
  if (W < 32 || W % 32) return false;
  return H % 2 == 0;
}

int conv_forward_b3_up(const pdes_conv_desc& d, hipStream_t st, bool dry) {        // dry: capability query only
  if (!(opt().mfma_b3 & 4) || !d.wbu_fwd || !b3up_shape_ok(d)) return PDES_ENOSUP;
  if (dry) return PDES_OK;
  const int nchunk = (d.Cin + 31) / 32, kpad = nchunk * 32, nt_total = (d.Cout + 15) / 16;
  const int twg = d.Win >= 32 ? 2 : 1;
  const int tail_on = (opt().b3_tail && d.w) ? 1 : 0;
  dim3 grid((d.Win / (16 * twg)) * (d.Hin / (4 / twg)), d.B, (nt_total + 3) / 4), block(256);
  if (twg == 2) {
    const size_t lds = 16 * (size_t)kpad + 2 * (size_t)B3UGeo<2>::BUF * 2 + 4 * (size_t)B3UGeo<2>::FCS * sizeof(float);
    hipLaunchKernelGGL((conv_b3_up_fwd_kernel<2>), grid, block, lds, st, d, d.wbu_fwd, nt_total, tail_on);
  } else {
    const size_t lds = 16 * (size_t)kpad + 2 * (size_t)B3UGeo<1>::BUF * 2 + 4 * (size_t)B3UGeo<1>::FCS * sizeof(float);
    hipLaunchKernelGGL((conv_b3_up_fwd_kernel<1>), grid, block, lds, st, d, d.wbu_fwd, nt_total, tail_on);
  }
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

}  // namespace pdes

using namespace pdes;

extern "C" int pdes_pack_weights_b3up(const pdes_b3up_pack_item* items, int n, int max_elems, void* stream) {
  if (!items || n <= 0 || max_elems <= 0) return PDES_EINVAL;
  int gx = cdiv(max_elems, 256);
  gx = gx > 256 ? 256 : gx;
  hipLaunchKernelGGL(pack_b3up_kernel, dim3(gx, n), dim3(256), 0, static_cast<hipStream_t>(stream), items);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

extern "C" int pdes_b3up_image_elems(int Cout, int Cin, long long* fwd_elems, long long* bwd_elems) {
  if (Cout <= 0 || Cin <= 0 || !fwd_elems || !bwd_elems) return PDES_EINVAL;
  const long long ntf = (((Cout + 15) / 16) + 3) & ~3;
  *fwd_elems = (long long)((Cin + 31) / 32) * 16 * ntf * 3 * 64 * 8;
  const long long ntb = (((Cin + 15) / 16) + 7) & ~7;
  *bwd_elems = (long long)4 * ((Cout + 31) / 32) * 4 * ntb * 3 * 64 * 8;      // (parity, 32-channel chunk) x 4 taps
  return PDES_OK;
}
