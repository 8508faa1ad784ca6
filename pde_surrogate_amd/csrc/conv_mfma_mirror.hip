// "Mirror" (write-once) data gradient of a dense block (reference models/codec.py:43-86, autograd of _DenseBlock).
//
// Layer j of a block reads channels [0, C_j) of the block buffer and writes its 16 at C_j.  The per-layer data
// gradient (conv_mfma.hip, MODE_BWD) adds layer j's term to the accumulator T of ALL C_j input channels: a block of L
// layers reads and writes T O(L^2) times (47 of the 77 MB the 180 -> 16 layer moves).  Here the sum is turned round:
//     T[group c] += sum_{j > c} gamma_jc * 1[BN_j(x)_c > 0] * convT_j(g_j)[c]
// is ONE launch per group c, issued where the per-layer kernel of layer c+1 was (its g has just been finalised; the
// g of the later layers were finalised before).  T of a group is read and written once, x once, and the terms of the
// later layers accumulate in registers.
//
// The launch is the data gradient of a VIRTUAL convolution whose output channels are the outputs of layers
// c+1 .. L side by side -- which they are in the block buffer, so the K operand (g) is staged exactly as in
// conv_mfma.hip: chunks of 16 channels = one layer each, double-buffered LDS tile, two register stages.  What differs
// is the ReLU mask: it belongs to (layer, channel), so the accumulators are folded into the running sum -- mask,
// gamma, dgamma / dbeta partial sums -- after every chunk instead of once at the end.
//
// Two wave layouts: a group of one N-tile (the 16 outputs of a layer) splits the M-tiles of the workgroup over its
// four waves (MSPLIT); the block input (several N-tiles) gives each wave an N-tile, as in conv_mfma.hip.
// The B operand comes from an image packed for (group, later layers) by pack_mir_item (pack_kernels.h):
//     [(kstep * 9 + tap) * ntp + nt][kq * 16 + n] = W_{j'}[4 (kstep % 4) + kq][n0 + 16 nt + n][8 - tap],  j' = j + kstep / 4
#include <stdlib.h>
#include "pdes_common.h"
#include "pdes_options.h"
#include "../../include/pdes_hip.h"
#include "conv_mfma_tile.h"

namespace pdes {

struct MirArgs {
  int nj;                        // layers (= K chunks) of this launch
  int n0, n1;                    // channels of the block buffer whose T this launch completes
  int ntp;                       // N-tiles per (kstep, tap) of the packed image
  const float* gamma[PDES_MIRROR_MAX];
  const float* beta[PDES_MIRROR_MAX];
  double* bn_grad[PDES_MIRROR_MAX];
};

template <int TWG, int MT, bool MSPLIT, bool PIPE>
__global__ __launch_bounds__(256, 2)
void conv_mirror_kernel(pdes_conv_desc d, const float* __restrict__ wm, MirArgs ma) {
  using G = TileGeo<3, TWG, MT, 1>;
  constexpr int KK = 9;
  constexpr int MTW = MSPLIT ? MT / 4 : MT;        // M-tiles a wave accumulates
  static_assert(MT % 4 == 0, "M-split: MT / 4 tiles per wave");
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int nt_w = MSPLIT ? 0 : (int)blockIdx.z * 4 + wave;       // this wave's N-tile
  const int mt_w0 = MSPLIT ? wave * MTW : 0;                       // ... and its first M-tile
  const int ntp = ma.ntp, nchunk = ma.nj;

  const int kH = d.Hout, kW = d.Wout, HWs = kH * kW;
  const float* kbase = d.g + ((size_t)b * d.g_ctot + d.g_coff) * HWs;
  // [per-(layer, wave) partial sums: (PDES_MIRROR_MAX + 1) rows x 4 waves x 16 channels x 2][tile buffers]
  float* sred = smem;
  float* tile = smem + (PDES_MIRROR_MAX + 1) * 128;
  const int tiles_x = kW / G::TW;
  const int oy0 = (blockIdx.x / tiles_x) * G::TH, ox0 = (blockIdx.x % tiles_x) * G::TW;

  // ---- staging geometry (per thread, chunk independent): as in conv_mfma.hip, data-gradient view
  const bool halo_live = tiles_x > 1;
  int vg[G::NPV], vl[G::NPV], hg[G::NPH], hl[G::NPH];
  unsigned vrow = 0, hval = 0;
#pragma unroll
  for (int i = 0; i < G::NPV; ++i) {
    const int e = tid + 256 * i;
    const int ch = e / (G::ROWS * (G::TWI / 4)), rem = e % (G::ROWS * (G::TWI / 4));
    const int r = rem / (G::TWI / 4), j = rem % (G::TWI / 4);
    const int cy = oy0 - G::PADL + r, cx = ox0 + 4 * j;
    const bool ok = e < G::NV4 && cy >= 0 && cy < kH;
    vg[i] = min(max(cy, 0), kH - 1) * kW + cx;
    vl[i] = e < G::NV4 ? ch * G::CS + r * G::LDW + G::COL0 + 4 * j : -1;
    if (ok) vrow |= 1u << i;
  }
#pragma unroll
  for (int i = 0; i < G::NPH; ++i) {
    const int e = tid + 256 * i;
    const int ch = e / (G::ROWS * G::NHC), rem = e % (G::ROWS * G::NHC);
    const int r = rem / G::NHC, h = rem % G::NHC;
    const int cy = oy0 - G::PADL + r;
    const int cx = h < G::NL ? ox0 - G::NL + h : ox0 + G::TWI + (h - G::NL);
    const int lc = h < G::NL ? G::COL0 - G::NL + h : G::COL0 + G::TWI + (h - G::NL);
    const bool ok = e < G::NH && cy >= 0 && cy < kH && cx >= 0 && cx < kW;
    hg[i] = min(max(cy, 0), kH - 1) * kW + min(max(cx, 0), kW - 1);
    hl[i] = e < G::NH ? ch * G::CS + r * G::LDW + lc : -1;
    if (ok) hval |= 1u << i;
  }
  struct Stage { float4 pv[G::NPV]; float ph[G::NPH]; };
  Stage sA, sB;
  auto issue = [&](int chunk, Stage& st) __attribute__((always_inline)) {
    const float* src = kbase + (size_t)chunk * 16 * HWs;
#pragma unroll
    for (int i = 0; i < G::NPV; ++i) {
      const int ch = min((tid + 256 * i) / (G::ROWS * (G::TWI / 4)), 15);
      st.pv[i] = *reinterpret_cast<const float4*>(src + ch * HWs + vg[i]);
    }
    if (halo_live) {
#pragma unroll
      for (int i = 0; i < G::NPH; ++i) {
        const int ch = min((tid + 256 * i) / (G::ROWS * G::NHC), 15);
        st.ph[i] = src[ch * HWs + hg[i]];
      }
    }
  };
  auto commit = [&](int buf, const Stage& st) __attribute__((always_inline)) {
    float* t = tile + buf * (G::KC * G::CS);
#pragma unroll
    for (int i = 0; i < G::NPV; ++i)
      if (vl[i] >= 0)
        *reinterpret_cast<float4*>(t + vl[i]) = ((vrow >> i) & 1u) ? st.pv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (halo_live) {
#pragma unroll
      for (int i = 0; i < G::NPH; ++i)
        if (hl[i] >= 0) t[hl[i]] = ((hval >> i) & 1u) ? st.ph[i] : 0.f;
    }
  };
  if (!halo_live) {                  // full-width tiles: the halo columns lie outside the image, zero once
#pragma unroll
    for (int i = 0; i < G::NPH; ++i)
      if (hl[i] >= 0) { tile[hl[i]] = 0.f; tile[G::KC * G::CS + hl[i]] = 0.f; }
  }

  // ---- this lane's channel of the group, its statistics, x and T of the wave's pixels
  const int nC = ma.n1 - ma.n0;
  const bool wave_live = nt_w * 16 < nC;                                  // an N-split wave past the last N-tile idles
  const int cl = nt_w * 16 + (lane & 15);
  const bool ch_live = cl < nC;
  const int ci = ma.n0 + min(cl, nC - 1);                                 // clamped: loads stay in range
  const int HWi = d.Hin * d.Win;
  const int px = (lane >> 4) * 4;
  const float* xb = d.x + ((size_t)b * d.x_ctot + ci) * HWi;
  float* tb2 = d.t_in + ((size_t)b * d.x_ctot + ci) * HWi;
  int pix[MTW], aoff[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    const int mt = mt_w0 + m;
    pix[m] = (oy0 + mt / TWG) * d.Win + ox0 + (mt % TWG) * 16 + px;
    aoff[m] = (mt / TWG) * G::LDW + (G::COL0 - G::PADL) + (mt % TWG) * 16;
  }

  // B operand, two register sets.  N-split: the next k-step streams in while this one runs on the matrix pipe (36
  // MFMAs).  M-split: a k-step is only 9 * MTW MFMAs, too short to cover a trip to L2, so a set holds a whole chunk
  // (36 values) and the NEXT chunk streams in under this chunk's 36 * MTW MFMAs.
  const int ksteps = 4 * nchunk;
  constexpr int BS = MSPLIT ? 4 * KK : KK;
  float bA[BS], bB[BS];
  auto load_b = [&](int kstep, float* dst) __attribute__((always_inline)) {
    const float* wp = wm + ((size_t)min(kstep, ksteps - 1) * KK * ntp + min(nt_w, ntp - 1)) * 64 + lane;
#pragma unroll
    for (int t = 0; t < KK; ++t) dst[t] = wp[(size_t)t * ntp * 64];
  };
  auto load_b_chunk = [&](int chunk, float* dst) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 4; ++s) load_b(min(chunk, nchunk - 1) * 4 + s, dst + s * KK);
  };
  v4f acc[MTW], tsum[MTW];
  float4 xq[MTW];
  if constexpr (MSPLIT) load_b_chunk(0, bA); else load_b(0, bA);
  issue(0, sA);
  const BnC kst = bn_coef_m(d, ci);                  // mean / invstd of the channel (gamma, beta: per layer below)
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    xq[m] = *reinterpret_cast<const float4*>(xb + pix[m]);
    const float4 t0 = d.t_accumulate ? *reinterpret_cast<const float4*>(tb2 + pix[m]) : make_float4(0.f, 0.f, 0.f, 0.f);
    tsum[m] = (v4f){t0.x, t0.y, t0.z, t0.w};
    acc[m] = (v4f){0.f, 0.f, 0.f, 0.f};
  }
  if constexpr (PIPE) issue(min(1, nchunk - 1), sB);
  commit(0, sA);
  __syncthreads();

  const int a_lane = (lane >> 4) * G::CS + (lane & 15);
  auto mfma_kstep = [&](const float* tk, const float* bw) __attribute__((always_inline)) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int m = 0; m < MTW; ++m)
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(tk[aoff[m] + ky * G::LDW + kx], bw[ky * 3 + kx], acc[m], 0, 0, 0);
  };
  // sred rows: {dgamma, dbeta} of layer (first + chunk), [chunk][wave][16 channels][2]; row nchunk: {sum T, sum T xhat}

  auto step = [&](int chunk, Stage& sfree, const Stage& snext, float* b0, float* b1) __attribute__((always_inline)) {
    const int buf = chunk & 1;
    const float* tb = tile + buf * (G::KC * G::CS) + a_lane;
    // BatchNorm weight / bias of layer (first + chunk) for this lane's channel: in flight under the matrix work
    const float gam = ma.gamma[chunk][ci], bet = ma.beta[chunk][ci];
    if constexpr (MSPLIT) {          // b0: this chunk's weights; b1 receives the next chunk's
      load_b_chunk(chunk + 1, b1);
      if constexpr (PIPE) issue(min(chunk + 2, nchunk - 1), sfree);
#pragma unroll
      for (int s = 0; s < 4; ++s) mfma_kstep(tb + s * 4 * G::CS, b0 + s * KK);
    } else {                         // b0: the chunk's first k-step; the sets alternate per k-step and end where they began
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        load_b(chunk * 4 + s + 1, (s & 1) ? b0 : b1);
        if constexpr (PIPE) { if (s == 0) issue(min(chunk + 2, nchunk - 1), sfree); }
        if (wave_live) mfma_kstep(tb + s * 4 * G::CS, (s & 1) ? b1 : b0);
      }
    }
    // fold: ReLU mask and gamma of THIS layer, into the running sum; dgamma / dbeta partial sums of this layer
    const float scale = gam * kst.invstd;
    float dg = 0.f, db = 0.f;
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      const float xs[4] = {xq[m].x, xq[m].y, xq[m].z, xq[m].w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float y = (xs[r] - kst.mean) * scale + bet;
        const float xh = (xs[r] - kst.mean) * kst.invstd;
        const float dyv = (y > 0.f) ? acc[m][r] : 0.f;
        db += dyv; dg += dyv * xh;
        tsum[m][r] += gam * dyv;
      }
      acc[m] = (v4f){0.f, 0.f, 0.f, 0.f};
    }
    dg += __shfl_xor(dg, 16, 64); dg += __shfl_xor(dg, 32, 64);
    db += __shfl_xor(db, 16, 64); db += __shfl_xor(db, 32, 64);
    if (lane < 16) {
      float* q = sred + ((chunk * 4 + wave) * 16 + lane) * 2;
      q[0] = ch_live ? dg : 0.f; q[1] = ch_live ? db : 0.f;
    }
    if constexpr (PIPE) { if (chunk + 1 < nchunk) commit(buf ^ 1, snext); }
    __syncthreads();
  };
  if constexpr (!PIPE) {
    step(0, sA, sA, bA, bB);
  } else {
    int c = 0;
    for (; c + 1 < nchunk; c += 2) {
      step(c, sA, sB, bA, bB);
      if constexpr (MSPLIT) step(c + 1, sB, sA, bB, bA);
      else step(c + 1, sB, sA, bA, bB);
    }
    if (c < nchunk) step(c, sA, sB, bA, bB);
  }

  // ---- T of the group: written once; {sum T, sum T xhat} for its finalize
  float st = 0.f, sx = 0.f;
  if (ch_live && (MSPLIT || wave_live)) {
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      const float xs[4] = {xq[m].x, xq[m].y, xq[m].z, xq[m].w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float xh = (xs[r] - kst.mean) * kst.invstd;
        st += tsum[m][r]; sx += tsum[m][r] * xh;
      }
      *reinterpret_cast<float4*>(tb2 + pix[m]) = make_float4(tsum[m][0], tsum[m][1], tsum[m][2], tsum[m][3]);
    }
  }
  st += __shfl_xor(st, 16, 64); st += __shfl_xor(st, 32, 64);
  sx += __shfl_xor(sx, 16, 64); sx += __shfl_xor(sx, 32, 64);
  if (lane < 16) {
    float* q = sred + ((nchunk * 4 + wave) * 16 + lane) * 2;
    q[0] = st; q[1] = sx;
  }
  __syncthreads();
  // one fp64 atomic per (row, channel, term) and workgroup (M-split: the four waves hold the same channels)
  const long long ro = (long long)rep_of_block(d.nrep) * d.rep_stride;
  constexpr int NW = MSPLIT ? 1 : 4;
  for (int e = tid; e < (nchunk + 1) * NW * 32; e += 256) {
    const int k = e & 1, c = (e >> 1) & 15, w = (e >> 5) % NW, row = e / (32 * NW);
    float v;
    if (MSPLIT) {
      const float* q = sred + (row * 4 * 16 + c) * 2 + k;
      v = (q[0] + q[32]) + (q[64] + q[96]);
    } else {
      v = sred[((row * 4 + w) * 16 + c) * 2 + k];
    }
    const int clc = (MSPLIT ? 0 : ((int)blockIdx.z * 4 + w) * 16) + c;
    if (clc >= nC) continue;
    double* dst = row < nchunk ? ma.bn_grad[row] : d.t_stats;
    atomicAdd(&dst[ro + 2 * (ma.n0 + clc) + k], (double)v);
  }
}

// descs[0 .. nj): the block's layers from the one whose output gradient has just been finalised to the block's last;
// descs[0].final_c0 .. final_c1 is the group this launch completes
static int mirror_check(const pdes_conv_desc* ds, int nj) {
  const pdes_conv_desc& d = ds[0];
  if (nj < 1 || nj > PDES_MIRROR_MAX || !d.wm_mir || d.nrep != PDES_NREP) return PDES_ENOSUP;
  for (int k = 0; k < nj; ++k) {
    const pdes_conv_desc& e = ds[k];
    if (e.ksize != 3 || e.stride != 1 || e.pad != 1 || e.upsample || !e.has_bn || e.eval_mode || e.Cout != 16) return PDES_ENOSUP;
    if (e.x != d.x || e.t_in != d.t_in || e.g != d.g || e.out != d.x || e.x_ctot != d.x_ctot || e.g_ctot != d.x_ctot) return PDES_ENOSUP;
    if (e.Cin != d.Cin + 16 * k || e.g_coff != e.Cin || e.Hin != d.Hin || e.Win != d.Win || e.Hout != d.Hin || e.Wout != d.Win)
      return PDES_ENOSUP;
    if (!e.gamma || !e.beta || !e.bn_grad || !e.t_in) return PDES_ENOSUP;
  }
  if (d.final_c1 != d.Cin || d.final_c0 < 0 || d.final_c0 >= d.final_c1) return PDES_ENOSUP;
  const int W = d.Win, H = d.Hin;
  if (W % 16 || (W >= 32 && W % 32)) return PDES_ENOSUP;
  return H % (W >= 32 ? 2 : 4) == 0 ? PDES_OK : PDES_ENOSUP;
}

int conv_backward_data_mirror(const pdes_conv_desc* ds, int nj, hipStream_t st, bool dry) {
  const int chk = mirror_check(ds, nj);
  if (chk) return chk;
  const pdes_conv_desc& d = ds[0];
  MirArgs ma;
  ma.nj = nj; ma.n0 = d.final_c0; ma.n1 = d.final_c1; ma.ntp = d.mir_ntp;
  for (int k = 0; k < PDES_MIRROR_MAX; ++k) {
    const pdes_conv_desc& e = ds[k < nj ? k : nj - 1];
    ma.gamma[k] = e.gamma; ma.beta[k] = e.beta; ma.bn_grad[k] = e.bn_grad;
  }
  const int ntr = (ma.n1 - ma.n0 + 15) / 16;
  if (ma.ntp != (ntr == 1 ? 1 : ((ntr + 7) & ~7))) return PDES_EINVAL;
  const bool msplit = ntr == 1;
  const int W = d.Win, H = d.Hin, twg = W >= 32 ? 2 : 1;
  const long long tiles8 = (long long)(W / (16 * twg)) * (H / (8 / twg)) * d.B;
  // N-split waves hold accumulator, running sum and x of all their M-tiles: 8 M-tiles do not fit the register file
  // at two workgroups per CU
  const int mt = (msplit && H % (8 / twg) == 0 && tiles8 >= 256) ? 8 : 4;
  if (H % (mt / twg)) return PDES_ENOSUP;
  // LDS: the partial-sum rows + two tile buffers; TileGeo<3, twg, mt, 1>::CS
  const int rows = mt / twg + 2, ldw = ((4 + 16 * twg + 1 + 3) / 4) * 4, cs = ((rows * ldw - 16 + 31) / 32) * 32 + 16;
  const size_t lds = ((size_t)2 * 16 * cs + (size_t)(PDES_MIRROR_MAX + 1) * 4 * 32) * sizeof(float);
  if (lds > 160 * 1024) return PDES_ENOSUP;
  if (dry) return PDES_OK;
  dim3 grid((W / (16 * twg)) * (H / (mt / twg)), d.B, msplit ? 1 : (ntr + 3) / 4), block(256);
#define PDES_MIR(TWG_, MT_, MS_)                                                                              \
  if (twg == TWG_ && mt == MT_ && msplit == MS_) {                                                             \
    static_assert(TileGeo<3, TWG_, MT_, 1>::KC == 16, "chunk = one layer");                                   \
    if (cs != TileGeo<3, TWG_, MT_, 1>::CS) return PDES_EINVAL;                                                \
    if (nj > 1)                                                                                                \
      hipLaunchKernelGGL((conv_mirror_kernel<TWG_, MT_, MS_, true>), grid, block, lds, st, d, d.wm_mir, ma);   \
    else                                                                                                       \
      hipLaunchKernelGGL((conv_mirror_kernel<TWG_, MT_, MS_, false>), grid, block, lds, st, d, d.wm_mir, ma);  \
  }
  PDES_MIR(2, 8, true) PDES_MIR(2, 4, true) PDES_MIR(2, 4, false)
  PDES_MIR(1, 8, true) PDES_MIR(1, 4, true) PDES_MIR(1, 4, false)
#undef PDES_MIR
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

}  // namespace pdes

using namespace pdes;

extern "C" int pdes_mirror_image_floats(int nj, int n0, int n1, long long* floats, int* ntp) {
  if (nj < 1 || nj > PDES_MIRROR_MAX || n0 < 0 || n1 <= n0 || !floats || !ntp) return PDES_EINVAL;
  const int ntr = (n1 - n0 + 15) / 16;
  *ntp = ntr == 1 ? 1 : ((ntr + 7) & ~7);
  *floats = (long long)4 * nj * 9 * *ntp * 64;
  return PDES_OK;
}

extern "C" int pdes_mirror_check(const pdes_conv_desc* descs, int nj) {
  if (!descs) return PDES_EINVAL;
  return conv_backward_data_mirror(descs, nj, nullptr, true);
}
