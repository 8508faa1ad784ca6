// Weight gradient of the 1x1 / 3x3 stride-1 convolutions on the f32 matrix cores (gfx950):
//     dW[co][ci][tap] = sum_{b, pixel} g[b][co][pixel] * z[b][ci][pixel + tap],  z = relu(bn(x)) (+ nearest x2)
// (autograd of F.conv2d wrt its weight for reference models/codec.py:66-69, :103-150, :163-188).
//
// GEMM roles per v_mfma_f32_16x16x4_f32: M = 16 input channels (A = z, one ds_read_b32 per lane),
// N = 16 output channels (B = g), K = 4 consecutive pixels of an image row.  The K dimension
// (B*H*W pixels) is what has to be split for parallelism: a workgroup owns ONE 16-channel M-tile,
// NTW N-tiles and a run of pixel tiles of one sample; its 4 waves take different rows of each
// pixel tile and are summed through LDS at the end.  Partial results of the pixel splits go to a
// scratch buffer and are reduced in a fixed order by a second kernel (deterministic, no float
// atomics).  LDS images are [channel][pixel] with a channel stride == 2 (mod 32) dwords so that
// the 16 channels x 2 pixels of a ds_read_b32 lane group hit 32 distinct banks.
#include <stdlib.h>
#include "pdes_common.h"
#include "pdes_options.h"
#include "../../include/pdes_hip.h"

namespace pdes {
int conv_backward_weight_1x1(const pdes_conv_desc& d, int splits_per_image, hipStream_t st);   // conv_mfma_1x1.hip
bool first_layer_shape(const pdes_conv_desc& d);                                                 // conv_direct.hip
bool wgrad_b3_applies(const pdes_conv_desc& d);                                                  // conv_mfma_wgrad_b3.hip
int wgrad_b3_splits(const pdes_conv_desc& d);
int conv_backward_weight_b3(const pdes_conv_desc& d, hipStream_t st);

typedef float v4f __attribute__((ext_vector_type(4)));

template <int KS, int TWG, int S>
struct WGeo {
  static constexpr int TH = 8 / TWG, TW = 16 * TWG;
  static constexpr int PADL = (KS - 1) / 2;
  static constexpr int ROWS = (TH - 1) * S + KS;
  static constexpr int TWI = S * TW;                          // interior input columns (float4 traffic)
  static constexpr int NL = PADL, NR = KS - PADL - S;         // halo columns
  static constexpr int COL0 = 4;
  static constexpr int LDW = ((COL0 + TWI + NR + 1) / 2) * 2; // even row pitch (8-byte aligned b64 stores)
  static constexpr int CS = ((ROWS * LDW + 29) / 32) * 32 + 2;  // channel stride == 2 (mod 32) dwords
  static constexpr int GS = TH * TW + 2;                       // g image channel stride (130)
  static constexpr int NV4 = 16 * ROWS * (TWI / 4);
  static constexpr int NPV = (NV4 + 255) / 256;
  static constexpr int NHC = (NL + NR) > 0 ? (NL + NR) : 1;
  static constexpr int NH = 16 * ROWS * (NL + NR);
  static constexpr int NPH = (NH + 255) / 256;
  static_assert(CS % 32 == 2 && GS % 32 == 2 && CS >= ROWS * LDW && NR >= 0, "LDS geometry");
};

// PIPE: > 2 pixel tiles per workgroup, prefetch two ahead.
// FEW (5x5, Cout*5 <= 16): the N dimension is (output channel, kernel column kx) instead of 16 output channels
// (conv_mfma_fewout.hip): 5 accumulators (kernel rows) and 5 MFMAs per pixel k-step instead of 25, the B
// operand is the gradient tile read with a per-lane column shift of -kx.
// GF (one N-tile, the dense blocks' 16-output-channel layers): `g` still holds the accumulator T of the layer's output
// channels; the BatchNorm-backward finalize g = invstd (T - mean(T) - xhat mean(T xhat)) is applied while the gradient
// tile is staged (x = the raw activation `out`, read beside T) -- the expression of bn_bwd_finalize_kernel, which then
// is not launched for this layer (pdes_backward2, option PDES_FIN_ONLOAD).
template <int KS, int TWG, int NTW, int S, bool PIPE, bool FEW, bool GF = false>
__global__ __launch_bounds__(256, (NTW == 1 && KS <= 3 && S == 1) ? 3 : 1) void conv_mfma_wgrad_kernel(pdes_conv_desc d, float* __restrict__ part, int tpw,
                                                             int n_ngroups, int co_off) {
  static_assert(!GF || (NTW == 1 && !FEW), "finalize on load: one N-tile, generic form");
  using G = WGeo<KS, TWG, S>;
  constexpr int KK = KS * KS;
  constexpr int NPG4 = 16 * NTW * G::TH * G::TW / 4 / 256;      // g float4 per thread per tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  static_assert(!FEW || (KS == 5 && NTW == 1 && S == 1), "few-output form: 5x5, stride 1");
  constexpr int GROW = G::TW + 8;                                // FEW: gradient row with 4 zero columns either side
  constexpr int GPL = ((G::TH * GROW - 8 + 31) / 32) * 32 + 8;  // FEW: plane stride == 8 (mod 32); plane 3 stays zero
  constexpr int GAREA = FEW ? 4 * GPL : 16 * NTW * G::GS;
  constexpr int LDSB = 16 * G::CS + GAREA;                       // one buffer: z image then g image
  float* zt = smem;                                            // [2][ [16][CS] | [16*NTW][GS] ]
  float* gt = smem + 16 * G::CS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Hc = d.Hin, Wc = d.Win;                             // conv-input size (nearest-x2: the _up kernel)
  const int tiles_x = d.Wout / G::TW, tps = tiles_x * (d.Hout / G::TH);                 // tiles of the OUTPUT map
  const int groups = tps / tpw;
  const int b = blockIdx.x / groups, tg = blockIdx.x % groups;
  const int mtile = blockIdx.y / n_ngroups, ng = blockIdx.y % n_ngroups;
  const int ci0 = mtile * 16, co0 = co_off + ng * 16 * NTW;   // co_off: first channel of this launch's N range
  const int HWi = d.Hin * d.Win, HWo = d.Hout * d.Wout;

  // BN coefficients of this thread's staging channels are per element; keep the 16 channels' in LDS-free regs:
  // every thread needs (mean, scale, beta) of channel ch(e) for its NPZ elements -> recompute from a tiny table
  __shared__ float cf[16][3];
  if (tid < 16) {
    const int c = ci0 + tid;
    float m = 0.f, s = 0.f, bt = 0.f;
    if (c < d.Cin) {
      double mean, invstd;
      if (d.eval_mode) { mean = d.run_mean[c]; invstd = 1.0 / sqrt((double)d.run_var[c] + (double)d.eps); }
      else {           // (the table entry, or the replica sums; published by the workgroups of the first pixel split)
        const MeanInv mi = batch_mean_invstd(d.coef, d.x_stats, d.rep_stride, (double)d.B * HWi, d.eps, c, blockIdx.x == 0);
        mean = mi.mean; invstd = mi.invstd;
      }
      m = (float)mean; s = d.gamma[c] * (float)invstd; bt = d.beta[c];
    }
    cf[tid][0] = m; cf[tid][1] = s; cf[tid][2] = bt;
  }
  __shared__ float4 gfc[GF ? 16 : 1];        // GF: {mean, invstd, mean(T), mean(T xhat)} of the 16 gradient channels
  // one statistic load per thread, issued here (thread = (channel, {sum T, sum T xhat}, replica)); reduced with shuffles
  // behind the first tiles' loads (gf_table)
  double gf_sv = 0.0;
  float2 gf_ce = make_float2(0.f, 0.f);
  if constexpr (GF) {
    static_assert(PDES_NREP == 8, "finalize on load: 16 channels x 2 sums x 8 replicas = one load per thread");
    const int c = d.g_coff + min(co_off + (int)(blockIdx.y % n_ngroups) * 16 * NTW + (tid >> 4), d.Cout - 1);
    gf_sv = d.fin_tstats[(long long)(tid & 7) * d.rep_stride + 2 * c + ((tid >> 3) & 1)];
    if (d.fin_coef) gf_ce = reinterpret_cast<const float2*>(d.fin_coef)[c];
  }
  auto gf_table = [&]() __attribute__((always_inline)) {
    if constexpr (GF) {
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) gf_sv += __shfl_xor(gf_sv, o, 64);
      const double sx = __shfl_down(gf_sv, 8, 64);
      if ((tid & 15) == 0) {
        const double n = (double)d.B * d.Hout * d.Wout, inv_n = 1.0 / n;
        MeanInv mi;
        mi.mean = gf_ce.x; mi.invstd = gf_ce.y;
        if (!(gf_ce.y > 0.f))
          mi = batch_mean_invstd(nullptr, d.fin_xstats, d.rep_stride, n, d.eps,
                                 d.g_coff + min(co_off + (int)(blockIdx.y % n_ngroups) * 16 * NTW + (tid >> 4), d.Cout - 1), false);
        gfc[tid >> 4] = make_float4(mi.mean, mi.invstd, (float)(gf_sv * inv_n), (float)(sx * inv_n));
      }
    }
  };

  if (FEW) {                     // pad columns and the zero plane are never written by the staging
    for (int i = tid; i < GAREA; i += 256) { gt[i] = 0.f; if (PIPE) gt[LDSB + i] = 0.f; }
  }
  const float* xb = d.x + ((size_t)b * d.x_ctot + ci0) * HWi;
  const float* gb = d.g + ((size_t)b * d.g_ctot + d.g_coff + co0) * HWo;
  const float* xob = GF ? d.out + ((size_t)b * d.out_ctot + d.out_coff + co0) * HWo : nullptr;    // GF: raw activation beside g
  const int crem = d.Cin - ci0, corem = d.Cout - co0;

  const bool halo_live = (G::NL + G::NR) > 0 && tiles_x > 1;
  // Register stages hold RAW loads (addresses clamped into the image, no select on the loaded value):
  // anything that consumes a load right after issuing it would drain vmcnt and serialise the prefetch
  // with the matrix work.  Validity is applied when a stage is committed to LDS.
  constexpr int NPHS = G::NPH > 0 ? G::NPH : 1;
  struct Stage { float4 pv[G::NPV]; float4 pg[NPG4]; float ph[NPHS]; float4 px[GF ? NPG4 : 1]; };
  Stage sA, sB;
  auto issue = [&](int tile, Stage& st) __attribute__((always_inline)) {
    const int oy0 = (tile / tiles_x) * G::TH, ox0 = (tile % tiles_x) * G::TW;
#pragma unroll
    for (int i = 0; i < G::NPV; ++i) {
      const int e = tid + 256 * i;
      const int ch = e / (G::ROWS * (G::TWI / 4)), rem = e % (G::ROWS * (G::TWI / 4));
      const int r = rem / (G::TWI / 4), j = rem % (G::TWI / 4);
      const int cy = oy0 * S - G::PADL + r, cx = ox0 * S + 4 * j;
      const int chc = min(ch, crem - 1), cyc = min(max(cy, 0), Hc - 1);
      st.pv[i] = *reinterpret_cast<const float4*>(xb + (size_t)chc * HWi + cyc * d.Win + cx);
    }
    if (halo_live) {
#pragma unroll
      for (int i = 0; i < G::NPH; ++i) {
        const int e = tid + 256 * i;
        const int ch = e / (G::ROWS * G::NHC), rem = e % (G::ROWS * G::NHC);
        const int r = rem / G::NHC, h = rem % G::NHC;
        const int cy = oy0 * S - G::PADL + r;
        const int cx = h < G::NL ? ox0 * S - G::NL + h : ox0 * S + G::TWI + (h - G::NL);
        const int chc = min(ch, crem - 1), cyc = min(max(cy, 0), Hc - 1), cxc = min(max(cx, 0), Wc - 1);
        st.ph[i] = xb[(size_t)chc * HWi + cyc * d.Win + cxc];
      }
    }
#pragma unroll
    for (int i = 0; i < NPG4; ++i) {
      const int e = tid + 256 * i;                          // float4 index: channel-major, then pixel
      const int ch = e / (G::TH * G::TW / 4), p4 = e % (G::TH * G::TW / 4);
      const int oy = oy0 + (4 * p4) / G::TW, ox = ox0 + (4 * p4) % G::TW;
      const size_t off = (size_t)min(ch, corem - 1) * HWo + oy * d.Wout + ox;
      st.pg[i] = *reinterpret_cast<const float4*>(gb + off);
      if constexpr (GF) st.px[i] = *reinterpret_cast<const float4*>(xob + off);
    }
  };
  auto bnrelu = [&](float x, int ch, bool ok) __attribute__((always_inline)) {
    return ok ? fmaxf(0.f, (x - cf[ch][0]) * cf[ch][1] + cf[ch][2]) : 0.f;
  };
  auto commit = [&](int tile, int buf, const Stage& st) __attribute__((always_inline)) {
    const int oy0 = (tile / tiles_x) * G::TH, ox0 = (tile % tiles_x) * G::TW;
    float* ztb = zt + buf * LDSB;
    float* gtb = gt + buf * LDSB;
#pragma unroll
    for (int i = 0; i < G::NPV; ++i) {
      const int e = tid + 256 * i;
      if (e < G::NV4) {
        const int ch = e / (G::ROWS * (G::TWI / 4)), rem = e % (G::ROWS * (G::TWI / 4));
        const int r = rem / (G::TWI / 4), j = rem % (G::TWI / 4);
        const int cy = oy0 * S - G::PADL + r;
        const bool ok = ch < crem && cy >= 0 && cy < Hc;       // outside the image: 0, not relu(bn(0))
        float* dst = ztb + ch * G::CS + r * G::LDW + G::COL0 + 4 * j;      // 8-byte aligned
        const float4 x = st.pv[i];
        *reinterpret_cast<float2*>(dst) = make_float2(bnrelu(x.x, ch, ok), bnrelu(x.y, ch, ok));
        *reinterpret_cast<float2*>(dst + 2) = make_float2(bnrelu(x.z, ch, ok), bnrelu(x.w, ch, ok));
      }
    }
#pragma unroll
    for (int i = 0; i < G::NPH; ++i) {
      const int e = tid + 256 * i;
      if (e < G::NH) {
        const int ch = e / (G::ROWS * G::NHC), rem = e % (G::ROWS * G::NHC);
        const int r = rem / G::NHC, h = rem % G::NHC;
        const int cy = oy0 * S - G::PADL + r;
        const int cx = h < G::NL ? ox0 * S - G::NL + h : ox0 * S + G::TWI + (h - G::NL);
        const bool ok = halo_live && ch < crem && cy >= 0 && cy < Hc && cx >= 0 && cx < Wc;
        const int lc = h < G::NL ? G::COL0 - G::NL + h : G::COL0 + G::TWI + (h - G::NL);
        ztb[ch * G::CS + r * G::LDW + lc] = bnrelu(st.ph[i], ch, ok);
      }
    }
#pragma unroll
    for (int i = 0; i < NPG4; ++i) {
      const int e = tid + 256 * i;
      const int ch = e / (G::TH * G::TW / 4), p4 = e % (G::TH * G::TW / 4);
      const bool ok = ch < corem;
      if constexpr (FEW) {
        if (ok) {
          const int row = (4 * p4) / G::TW, col = (4 * p4) % G::TW;
          *reinterpret_cast<float4*>(gtb + ch * GPL + row * GROW + 4 + col) = st.pg[i];
        }
        continue;
      }
      float* dst = gtb + ch * G::GS + 4 * p4;                               // 8-byte aligned (GS even)
      float4 gv = st.pg[i];
      if constexpr (GF) {
        const float4 k = gfc[min(ch, 15)], x = st.px[i];
        gv.x = k.y * (gv.x - k.z - (x.x - k.x) * k.y * k.w);
        gv.y = k.y * (gv.y - k.z - (x.y - k.x) * k.y * k.w);
        gv.z = k.y * (gv.z - k.z - (x.z - k.x) * k.y * k.w);
        gv.w = k.y * (gv.w - k.z - (x.w - k.x) * k.y * k.w);
      }
      *reinterpret_cast<float2*>(dst) = ok ? make_float2(gv.x, gv.y) : make_float2(0.f, 0.f);
      *reinterpret_cast<float2*>(dst + 2) = ok ? make_float2(gv.z, gv.w) : make_float2(0.f, 0.f);
    }
  };

  constexpr int NACC = FEW ? KS : KK;
  v4f acc[NACC][NTW];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[t][nt] = (v4f){0.f, 0.f, 0.f, 0.f};

  // wave w owns rows [w*TH/4, (w+1)*TH/4) of each pixel tile
  constexpr int RPW = (G::TH >= 4) ? G::TH / 4 : 1;
  const int a_lane = (lane & 15) * G::CS + (lane >> 4) * S;   // A: i = ci, k = pixel offset
  const int b_lane = (lane & 15) * G::GS + (lane >> 4);       // B: j = co, k = pixel offset

  const int tile0 = tg * tpw;
  auto mfma_tile = [&](int buf) __attribute__((always_inline)) {
    const float* ztb = zt + buf * LDSB;
    const float* gtb = gt + buf * LDSB;
    if constexpr (FEW) {
      const int n = lane & 15, bco = n < 15 ? min(n / 5, 3) : 3, bkx = n % 5;     // column 15: the zero plane
      const int few_b = bco * GPL + 4 + (lane >> 4) - bkx;
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int row = wave * RPW + rr;
#pragma unroll
        for (int ks = 0; ks <= G::TW / 4; ++ks) {            // x' = ox0 - 2 + 4 ks + k: one k-step of halo
          const float bv = gtb[few_b + row * GROW + 4 * ks];
#pragma unroll
          for (int ky = 0; ky < KS; ++ky) {
            const float a = ztb[a_lane + (row + ky) * G::LDW + (G::COL0 - G::PADL) + 4 * ks];
            acc[ky][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc[ky][0], 0, 0, 0);
          }
        }
      }
      return;
    }
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int row = wave * RPW + rr;
#pragma unroll
      for (int ks = 0; ks < G::TW / 4; ++ks) {
        float bv[NTW];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) bv[nt] = gtb[b_lane + nt * 16 * G::GS + row * G::TW + 4 * ks];
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
          for (int kx = 0; kx < KS; ++kx) {
            const float a = ztb[a_lane + (row * S + ky) * G::LDW + (G::COL0 - G::PADL) + 4 * ks * S + kx];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
              acc[ky * KS + kx][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[nt], acc[ky * KS + kx][nt], 0, 0, 0);
          }
      }
    }
  };
  if constexpr (!PIPE) {
    // <= 2 tiles per workgroup: one register stage, one LDS buffer (smaller footprint -> more resident workgroups)
    issue(tile0, sA);
    gf_table();
    __syncthreads();               // cf visible
    for (int tt = 0; tt < tpw; ++tt) {
      commit(tile0 + tt, 0, sA);
      __syncthreads();
      if (tt + 1 < tpw) issue(tile0 + tt + 1, sA);
      mfma_tile(0);
      __syncthreads();             // every wave is done with the LDS images before the next commit
    }
  } else {
    // double-buffered LDS images, two register stages: tile t+2 is requested before the MFMAs of tile t.
    // The request is unconditional (index clamped): a runtime test would merge wait states at the join and
    // over-wait, and peeled copies of the loop body cost registers (occupancy) -- both measured slower.
    issue(tile0, sA);
    issue(tile0 + 1, sB);
    gf_table();
    __syncthreads();               // cf visible
    commit(tile0, 0, sA);
    __syncthreads();
    auto step = [&](int tt, Stage& sfree, const Stage& snext) __attribute__((always_inline)) {
      const int buf = tt & 1;
      // (-DPDES_WG_NOSTAGE / -DPDES_WG_NOMFMA: component timing builds, EXPERIMENTS.md round 4 -- never shipped)
#ifndef PDES_WG_NOSTAGE
      issue(tile0 + min(tt + 2, tpw - 1), sfree);
#endif
#ifndef PDES_WG_NOMFMA
      mfma_tile(buf);
#endif
#ifndef PDES_WG_NOSTAGE
      if (tt + 1 < tpw) commit(tile0 + tt + 1, buf ^ 1, snext);
#endif
      __syncthreads();
    };
    int tt = 0;
    for (; tt + 1 < tpw; tt += 2) { step(tt, sA, sB); step(tt + 1, sB, sA); }
    if (tt < tpw) step(tt, sA, sB);
  }

  // ---- sum the 4 waves through LDS, then write this pixel split's partial dW
  float* red = smem;               // [4][NACC*NTW*4][64]
  constexpr int NR = NACC * NTW * 4;
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * NR + (t * NTW + nt) * 4 + r) * 64 + lane] = acc[t][nt][r];
  __syncthreads();
  float* pout = part + (size_t)blockIdx.x * d.Cout * d.Cin * KK;
  for (int q = wave; q < NR; q += 4) {
    const float s = red[(0 * NR + q) * 64 + lane] + red[(1 * NR + q) * 64 + lane] +
                    red[(2 * NR + q) * 64 + lane] + red[(3 * NR + q) * 64 + lane];
    const int r = q & 3, nt = (q >> 2) % NTW, t = (q >> 2) / NTW;
    const int ci = ci0 + (lane >> 4) * 4 + r;
    if constexpr (FEW) {             // t = kernel row ky, column n = (co, kx)
      const int n = lane & 15, co = n / 5, kx = n % 5;
      if (n < 15 && co < d.Cout && ci < d.Cin) pout[(((size_t)co * d.Cin + ci) * KS + t) * KS + kx] = s;
    } else {
      const int co = co0 + nt * 16 + (lane & 15);
      if (co < d.Cout && ci < d.Cin) pout[((size_t)co * d.Cin + ci) * KK + t] = s;
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Weight gradient of nearest-x2 + 3x3 convolutions in sub-pixel form (see conv_mfma_up.hip):
//   dWeff[p][a][b][co][ci] = sum_{y,x} G_p[co][y][x] * z[ci][y+a+dy-1][x+b+dx-1],  G_p[Y][X] = G[2Y+dy][2X+dx]
//   dW[ky][kx] = sum_{dy,dx} dWeff[(dy,dx)][a(ky,dy)][b(kx,dx)]
// 16 MFMAs per low-res pixel k-step instead of 36; the operand images are the low-res z halo tile
// and the four de-interleaved parity sub-images of the hi-res gradient.
template <int TWG, int NTW>
__global__ __launch_bounds__(256) void conv_mfma_wgrad_up_kernel(pdes_conv_desc d, float* __restrict__ part, int tpw,
                                                                int n_ngroups, int co_off) {
  using G = WGeo<3, TWG, 1>;
  constexpr int NPX = G::TH * G::TW;                            // low-res pixels per tile (128)
  constexpr int NPG = 16 * NTW * (NPX / 4) * 2 / 256;          // (channel, pixel quad, dy) items per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* zt = smem;                                            // [16][CS]
  float* gt = smem + 16 * G::CS;                               // [4 parities][16*NTW][GS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Hl = d.Hin, Wl = d.Win, HWl = Hl * Wl, Wh = d.Wout, HWh = d.Hout * d.Wout;
  const int tiles_x = Wl / G::TW, tps = tiles_x * (Hl / G::TH);
  const int groups = tps / tpw;
  const int b = blockIdx.x / groups, tg = blockIdx.x % groups;
  const int mtile = blockIdx.y / n_ngroups, ng = blockIdx.y % n_ngroups;
  const int ci0 = mtile * 16, co0 = co_off + ng * 16 * NTW;   // co_off: first channel of this launch's N range
  __shared__ float cf[16][3];
  if (tid < 16) {
    const int c = ci0 + tid;
    float m = 0.f, sc = 0.f, bt = 0.f;
    if (c < d.Cin) {
      double mean, invstd;
      if (d.eval_mode) { mean = d.run_mean[c]; invstd = 1.0 / sqrt((double)d.run_var[c] + (double)d.eps); }
      else {           // (the table entry, or the replica sums; published by the workgroups of the first pixel split)
        const MeanInv mi = batch_mean_invstd(d.coef, d.x_stats, d.rep_stride, (double)d.B * HWl, d.eps, c, blockIdx.x == 0);
        mean = mi.mean; invstd = mi.invstd;
      }
      m = (float)mean; sc = d.gamma[c] * (float)invstd; bt = d.beta[c];
    }
    cf[tid][0] = m; cf[tid][1] = sc; cf[tid][2] = bt;
  }
  const float* xb = d.x + ((size_t)b * d.x_ctot + ci0) * HWl;
  const float* gb = d.g + ((size_t)b * d.g_ctot + d.g_coff + co0) * HWh;
  const int crem = d.Cin - ci0, corem = d.Cout - co0;
  const bool halo_live = tiles_x > 1;

  // raw loads only (clamped addresses, no select on loaded values: see conv_mfma_wgrad_kernel); validity and
  // the parity de-interleave happen when the registers are committed to LDS
  float4 pv[G::NPV], ph0[NPG], ph1[NPG];
  float ph[G::NPH];
  auto issue = [&](int tile) __attribute__((always_inline)) {
    const int oy0 = (tile / tiles_x) * G::TH, ox0 = (tile % tiles_x) * G::TW;
#pragma unroll
    for (int i = 0; i < G::NPV; ++i) {
      const int e = tid + 256 * i;
      const int ch = e / (G::ROWS * (G::TWI / 4)), rem = e % (G::ROWS * (G::TWI / 4));
      const int r = rem / (G::TWI / 4), j = rem % (G::TWI / 4);
      const int cy = oy0 - 1 + r, cx = ox0 + 4 * j;
      const int chc = min(ch, crem - 1), cyc = min(max(cy, 0), Hl - 1);
      pv[i] = *reinterpret_cast<const float4*>(xb + (size_t)chc * HWl + cyc * Wl + cx);
    }
    if (halo_live) {
#pragma unroll
      for (int i = 0; i < G::NPH; ++i) {
        const int e = tid + 256 * i;
        const int ch = e / (G::ROWS * G::NHC), rem = e % (G::ROWS * G::NHC);
        const int r = rem / G::NHC, h = rem % G::NHC;
        const int cy = oy0 - 1 + r, cx = h == 0 ? ox0 - 1 : ox0 + G::TW;
        const int chc = min(ch, crem - 1), cyc = min(max(cy, 0), Hl - 1), cxc = min(max(cx, 0), Wl - 1);
        ph[i] = xb[(size_t)chc * HWl + cyc * Wl + cxc];
      }
    }
#pragma unroll
    for (int i = 0; i < NPG; ++i) {
      const int e = tid + 256 * i;                       // (channel, low-res pixel quad, dy)
      const int dy = e & 1, q = e >> 1;
      const int ch = q / (NPX / 4), p4 = q % (NPX / 4);
      const int y = oy0 + (4 * p4) / G::TW, x = ox0 + (4 * p4) % G::TW;
      const size_t off = (size_t)min(ch, corem - 1) * HWh + (size_t)(2 * y + dy) * Wh + 2 * x;
      const float* src = gb + off;
      ph0[i] = *reinterpret_cast<const float4*>(src);
      ph1[i] = *reinterpret_cast<const float4*>(src + 4);
    }
  };
  auto bnrelu = [&](float x, int ch, bool ok) __attribute__((always_inline)) {
    return ok ? fmaxf(0.f, (x - cf[ch][0]) * cf[ch][1] + cf[ch][2]) : 0.f;
  };
  auto commit = [&](int tile) __attribute__((always_inline)) {
    const int oy0 = (tile / tiles_x) * G::TH, ox0 = (tile % tiles_x) * G::TW;
#pragma unroll
    for (int i = 0; i < G::NPV; ++i) {
      const int e = tid + 256 * i;
      if (e < G::NV4) {
        const int ch = e / (G::ROWS * (G::TWI / 4)), rem = e % (G::ROWS * (G::TWI / 4));
        const int r = rem / (G::TWI / 4), j = rem % (G::TWI / 4);
        const int cy = oy0 - 1 + r;
        const bool ok = ch < crem && cy >= 0 && cy < Hl;
        float* dst = zt + ch * G::CS + r * G::LDW + G::COL0 + 4 * j;
        const float4 x = pv[i];
        *reinterpret_cast<float2*>(dst) = make_float2(bnrelu(x.x, ch, ok), bnrelu(x.y, ch, ok));
        *reinterpret_cast<float2*>(dst + 2) = make_float2(bnrelu(x.z, ch, ok), bnrelu(x.w, ch, ok));
      }
    }
#pragma unroll
    for (int i = 0; i < G::NPH; ++i) {
      const int e = tid + 256 * i;
      if (e < G::NH) {
        const int ch = e / (G::ROWS * G::NHC), rem = e % (G::ROWS * G::NHC);
        const int r = rem / G::NHC, h = rem % G::NHC;
        const int cy = oy0 - 1 + r, cx = h == 0 ? ox0 - 1 : ox0 + G::TW;
        const bool ok = halo_live && ch < crem && cy >= 0 && cy < Hl && cx >= 0 && cx < Wl;
        zt[ch * G::CS + r * G::LDW + (h == 0 ? G::COL0 - 1 : G::COL0 + G::TW)] = bnrelu(ph[i], ch, ok);
      }
    }
#pragma unroll
    for (int i = 0; i < NPG; ++i) {
      const int e = tid + 256 * i;
      const int dy = e & 1, q = e >> 1;
      const int ch = q / (NPX / 4), p4 = q % (NPX / 4);
      const bool v = ch < corem;
      const float4 h0 = ph0[i], h1 = ph1[i];
      float* d0 = gt + ((dy * 2 + 0) * 16 * NTW + ch) * G::GS + 4 * p4;
      float* d1 = gt + ((dy * 2 + 1) * 16 * NTW + ch) * G::GS + 4 * p4;
      *reinterpret_cast<float2*>(d0) = v ? make_float2(h0.x, h0.z) : make_float2(0.f, 0.f);       // dx = 0
      *reinterpret_cast<float2*>(d0 + 2) = v ? make_float2(h1.x, h1.z) : make_float2(0.f, 0.f);
      *reinterpret_cast<float2*>(d1) = v ? make_float2(h0.y, h0.w) : make_float2(0.f, 0.f);       // dx = 1
      *reinterpret_cast<float2*>(d1 + 2) = v ? make_float2(h1.y, h1.w) : make_float2(0.f, 0.f);
    }
  };

  v4f acc[16][NTW];
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[t][nt] = (v4f){0.f, 0.f, 0.f, 0.f};
  constexpr int RPW = (G::TH >= 4) ? G::TH / 4 : 1;
  const int a_lane = (lane & 15) * G::CS + (lane >> 4);
  const int b_lane = (lane & 15) * G::GS + (lane >> 4);
  const int tile0 = tg * tpw;
  issue(tile0);
  __syncthreads();
  for (int tt = 0; tt < tpw; ++tt) {
    commit(tile0 + tt);
    __syncthreads();
    issue(tile0 + min(tt + 1, tpw - 1));
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int row = wave * RPW + rr;
#pragma unroll
      for (int ks = 0; ks < G::TW / 4; ++ks) {
        float bv[4][NTW];
#pragma unroll
        for (int pp = 0; pp < 4; ++pp)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt)
            bv[pp][nt] = gt[b_lane + (pp * 16 * NTW + nt * 16) * G::GS + row * G::TW + 4 * ks];
#pragma unroll
        for (int ty = 0; ty < 3; ++ty)
#pragma unroll
          for (int tx = 0; tx < 3; ++tx) {
            const float a = zt[a_lane + (row + ty) * G::LDW + (G::COL0 - 1) + 4 * ks + tx];
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
              for (int dx = 0; dx < 2; ++dx) {
                const int ia = ty - dy, ib = tx - dx;
                if (ia < 0 || ia > 1 || ib < 0 || ib > 1) continue;
                const int pp = dy * 2 + dx, q = pp * 4 + ia * 2 + ib;
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                  acc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[pp][nt], acc[q][nt], 0, 0, 0);
              }
          }
      }
    }
    __syncthreads();
  }

  // fold the 16 effective-kernel gradients into the 9 taps, then sum the 4 waves through LDS
  v4f w9[9][NTW];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        v4f s = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            const int ia = dy == 0 ? (ky == 0 ? 0 : 1) : (ky == 2 ? 1 : 0);
            const int ib = dx == 0 ? (kx == 0 ? 0 : 1) : (kx == 2 ? 1 : 0);
            s += acc[(dy * 2 + dx) * 4 + ia * 2 + ib][nt];
          }
        w9[ky * 3 + kx][nt] = s;
      }
  float* red = smem;
  constexpr int NR = 9 * NTW * 4;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * NR + (t * NTW + nt) * 4 + r) * 64 + lane] = w9[t][nt][r];
  __syncthreads();
  float* pout = part + (size_t)blockIdx.x * d.Cout * d.Cin * 9;
  for (int q = wave; q < NR; q += 4) {
    const float s = red[(0 * NR + q) * 64 + lane] + red[(1 * NR + q) * 64 + lane] +
                    red[(2 * NR + q) * 64 + lane] + red[(3 * NR + q) * 64 + lane];
    const int r = q & 3, nt = (q >> 2) % NTW, t = (q >> 2) / NTW;
    const int co = co0 + nt * 16 + (lane & 15), ci = ci0 + (lane >> 4) * 4 + r;
    if (co < d.Cout && ci < d.Cin) pout[((size_t)co * d.Cin + ci) * 9 + t] = s;
  }
}

// dw[i] += sum_s part[s][i], fixed order
__global__ __launch_bounds__(64) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                           int n, int nsplit) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = 0;
  for (; k + 4 <= nsplit; k += 4) {
    s0 += part[(size_t)k * n + i];
    s1 += part[(size_t)(k + 1) * n + i];
    s2 += part[(size_t)(k + 2) * n + i];
    s3 += part[(size_t)(k + 3) * n + i];
  }
  for (; k < nsplit; ++k) s0 += part[(size_t)k * n + i];
  dw[i] += (s0 + s1) + (s2 + s3);
}

// dw[i] += sum_s part[s][i] for a whole table of layers: ONE launch at the end of the backward pass
__global__ __launch_bounds__(64) void wgrad_reduce_all_kernel(const pdes_reduce_item* __restrict__ items) {
  const pdes_reduce_item it = items[blockIdx.y];
  for (int i = blockIdx.x * 64 + threadIdx.x; i < it.n; i += gridDim.x * 64) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 4 <= it.nsplit; k += 4) {
      s0 += it.part[(size_t)k * it.n + i];
      s1 += it.part[(size_t)(k + 1) * it.n + i];
      s2 += it.part[(size_t)(k + 2) * it.n + i];
      s3 += it.part[(size_t)(k + 3) * it.n + i];
    }
    for (; k < it.nsplit; ++k) s0 += it.part[(size_t)k * it.n + i];
    it.dw[i] += (s0 + s1) + (s2 + s3);
  }
}

bool wgrad_small_applies(const pdes_conv_desc& d);     // conv_small.hip
int wgrad_small_splits(const pdes_conv_desc& d);
// tile / split plan shared by the launcher and pdes_conv_wgrad_plan
struct WgradPlan { int twg, tps, tpw, nsplit, ntw, ngroups, gy; long long per; };
static bool wgrad_plan(const pdes_conv_desc& d, WgradPlan* p) {
  const int KK = d.ksize * d.ksize;
  const bool up = d.upsample && d.ksize == 3;            // sub-pixel form: tiles of the LOW-res map
  const int Wt = up ? d.Win : d.Wout, Ht = up ? d.Hin : d.Hout;
  p->twg = Wt >= 32 ? 2 : 1;
  p->tps = (Wt / (16 * p->twg)) * (Ht / (8 / p->twg));
  const int mtiles = (d.Cin + 15) / 16, ntiles = (d.Cout + 15) / 16;
  p->ntw = ntiles >= 2 ? 2 : 1;
  p->ngroups = (ntiles + p->ntw - 1) / p->ntw;
  p->gy = mtiles * p->ngroups;
  p->per = (long long)d.Cout * d.Cin * KK;
  if (wgrad_b3_applies(d)) {                             // bf16 x3 kernel for the wide layers: its own split count
    p->nsplit = wgrad_b3_splits(d);
    p->tpw = p->tps;
    if ((long long)p->nsplit * p->per * 4 <= d.ws_bytes) return true;
  }
  // pixel tiles per workgroup: as few as possible while (a) the grid has at most ~256 workgroups (PDES_WGRAD_WGS;
  // stand-alone the kernels are fastest with ~768, but they run beside the data-gradient chain on a second
  // stream and smaller grids leave it more of the chip: 2.203 / 2.182 / 2.179 / 2.177 / 2.204 ms per step at
  // 768 / 512 / 384 / 256 / 128; with the final kernel set, separate processes on one box: 2.033 / 2.020 / 2.008 /
  // 2.025 / 2.032 ms at 448 / 320 / 256 / 192 / 128) and
  // (b) the partial buffer fits the scratch
  p->tpw = p->tps;
  const int wg_target = opt().wgrad_wgs;
  for (int cand = 1; cand <= p->tps; cand *= 2) {
    if (p->tps % cand) continue;
    const long long ns = (long long)d.B * (p->tps / cand);
    if (ns * p->per * 4 > d.ws_bytes) continue;
    if (ns * p->gy <= wg_target || cand == p->tps) { p->tpw = cand; break; }
  }
  p->nsplit = d.B * (p->tps / p->tpw);
  return (long long)p->nsplit * p->per * 4 <= d.ws_bytes;
}

template <int KS, int S>
static int launch_wgrad(const pdes_conv_desc& d, hipStream_t st) {
  WgradPlan pl;
  if (!wgrad_plan(d, &pl)) return PDES_ENOSUP;
  const int twg = pl.twg, ntw = pl.ntw, ngroups = pl.ngroups, gy = pl.gy, tpw = pl.tpw, nsplit = pl.nsplit;
  const long long per = pl.per;
  (void)gy;
  const int mtiles = (d.Cin + 15) / 16, ntiles = (d.Cout + 15) / 16;
  dim3 block(256);
  // GY_: grid.y = M-tiles x N-groups of this launch; NGR_: N-groups; COFF_: first output channel of the launch
#define PDES_WG_LAUNCH(TWG_, NTW_, NGR_, COFF_)                                                             \
  do {                                                                                                        \
    const int ngroups_l = (NGR_);                                                                             \
    dim3 grid(nsplit, mtiles * ngroups_l);                                                                    \
    using G = WGeo<KS, TWG_, S>;                                                                              \
    size_t lds = (size_t)(tpw > 2 ? 2 : 1) * (16 * G::CS + 16 * NTW_ * G::GS) * sizeof(float);                \
    const size_t red = (size_t)4 * KS * KS * NTW_ * 4 * 64 * sizeof(float);                                   \
    if (red > lds) lds = red;                                                                                 \
    if constexpr (KS == 5 && NTW_ == 1 && S == 1) {                                                           \
      if (d.Cout * 5 <= 16) {                    /* few-output form; its LDS need is below the generic one */ \
        if (tpw > 2)                                                                                          \
          hipLaunchKernelGGL((conv_mfma_wgrad_kernel<KS, TWG_, NTW_, S, true, true>), grid, block, lds, st, d, d.ws, tpw, ngroups_l, (COFF_)); \
        else                                                                                                  \
          hipLaunchKernelGGL((conv_mfma_wgrad_kernel<KS, TWG_, NTW_, S, false, true>), grid, block, lds, st, d, d.ws, tpw, ngroups_l, (COFF_)); \
        break;                                                                                                \
      }                                                                                                       \
    }                                                                                                         \
    if (d.g_fused) {                               /* finalize on load: the one-N-tile 3x3 layers (GF) */   \
      if constexpr (KS == 3 && S == 1 && NTW_ == 1) {                                                         \
        if (tpw > 2)                                                                                          \
          hipLaunchKernelGGL((conv_mfma_wgrad_kernel<KS, TWG_, NTW_, S, true, false, true>), grid, block, lds, st, d, d.ws, tpw, ngroups_l, (COFF_)); \
        else                                                                                                  \
          hipLaunchKernelGGL((conv_mfma_wgrad_kernel<KS, TWG_, NTW_, S, false, false, true>), grid, block, lds, st, d, d.ws, tpw, ngroups_l, (COFF_)); \
        break;                                                                                                \
      }                                                                                                       \
      return PDES_ENOSUP;                                                                                     \
    }                                                                                                         \
    if (tpw > 2)                                                                                              \
      hipLaunchKernelGGL((conv_mfma_wgrad_kernel<KS, TWG_, NTW_, S, true, false>), grid, block, lds, st, d, d.ws, tpw, ngroups_l, (COFF_)); \
    else                                                                                                      \
      hipLaunchKernelGGL((conv_mfma_wgrad_kernel<KS, TWG_, NTW_, S, false, false>), grid, block, lds, st, d, d.ws, tpw, ngroups_l, (COFF_)); \
  } while (0)
  // an odd number of N-tiles >= 3: pairs with the two-tile kernel, the last tile with the one-tile kernel
  // (instead of a padding tile: 98 output channels are 7 tiles, not 8)
  // (only when the padding tile is a small share of a big layer: a second launch costs a few us)
  const bool split_odd = ntw == 2 && (ntiles & 1) && ntiles >= 7 && !(KS == 5);
  if (split_odd) {
    if (twg == 2) { PDES_WG_LAUNCH(2, 2, ntiles / 2, 0); PDES_WG_LAUNCH(2, 1, 1, 16 * (ntiles - 1)); }
    else { PDES_WG_LAUNCH(1, 2, ntiles / 2, 0); PDES_WG_LAUNCH(1, 1, 1, 16 * (ntiles - 1)); }
  } else if (twg == 2) { if (ntw == 2) PDES_WG_LAUNCH(2, 2, ngroups, 0); else PDES_WG_LAUNCH(2, 1, ngroups, 0); }
  else { if (ntw == 2) PDES_WG_LAUNCH(1, 2, ngroups, 0); else PDES_WG_LAUNCH(1, 1, ngroups, 0); }
#undef PDES_WG_LAUNCH
  PDES_LAUNCH_CHECK();
  if (!d.ws_defer) {      // otherwise the caller reduces every layer at once with pdes_wgrad_reduce_all
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv((int)per, 64)), dim3(64), 0, st, d.ws, d.dw, (int)per, nsplit);
    PDES_LAUNCH_CHECK();
  }
  return PDES_OK;
}

static bool wgrad_shape_ok(const pdes_conv_desc& d) {
  if (!d.has_bn || !(d.ksize == 5 || d.ksize == 3 || d.ksize == 1) || d.pad != (d.ksize - 1) / 2) return false;
  if (d.stride != 1 && !(d.stride == 2 && d.ksize == 3 && !d.upsample)) return false;
  if (d.upsample && d.ksize != 3) return false;
  if (d.Cin < 16) return false;
  const bool up = d.upsample && d.ksize == 3;
  const int W = up ? d.Win : d.Wout, H = up ? d.Hin : d.Hout;
  return !(W % 16 || (W >= 32 ? (W % 32 || H % 4) : (H % 8)));
}

static int launch_wgrad_up(const pdes_conv_desc& d, hipStream_t st) {
  WgradPlan pl;
  if (!wgrad_plan(d, &pl)) return PDES_ENOSUP;
  const int mtiles = (d.Cin + 15) / 16, ntiles = (d.Cout + 15) / 16;
  dim3 block(256);
#define PDES_WGU_LAUNCH(TWG_, NTW_, NGR_, COFF_)                                                            \
  do {                                                                                                        \
    const int ngroups_l = (NGR_);                                                                             \
    dim3 grid(pl.nsplit, mtiles * ngroups_l);                                                                 \
    using G = WGeo<3, TWG_, 1>;                                                                               \
    size_t lds = (size_t)(16 * G::CS + 4 * 16 * NTW_ * G::GS) * sizeof(float);                                \
    const size_t red = (size_t)4 * 9 * NTW_ * 4 * 64 * sizeof(float);                                         \
    if (red > lds) lds = red;                                                                                 \
    hipLaunchKernelGGL((conv_mfma_wgrad_up_kernel<TWG_, NTW_>), grid, block, lds, st, d, d.ws, pl.tpw, ngroups_l, (COFF_)); \
  } while (0)
  const bool split_odd = pl.ntw == 2 && (ntiles & 1) && ntiles >= 7;
  if (split_odd) {
    if (pl.twg == 2) { PDES_WGU_LAUNCH(2, 2, ntiles / 2, 0); PDES_WGU_LAUNCH(2, 1, 1, 16 * (ntiles - 1)); }
    else { PDES_WGU_LAUNCH(1, 2, ntiles / 2, 0); PDES_WGU_LAUNCH(1, 1, 1, 16 * (ntiles - 1)); }
  } else if (pl.twg == 2) { if (pl.ntw == 2) PDES_WGU_LAUNCH(2, 2, pl.ngroups, 0); else PDES_WGU_LAUNCH(2, 1, pl.ngroups, 0); }
  else { if (pl.ntw == 2) PDES_WGU_LAUNCH(1, 2, pl.ngroups, 0); else PDES_WGU_LAUNCH(1, 1, pl.ngroups, 0); }
#undef PDES_WGU_LAUNCH
  PDES_LAUNCH_CHECK();
  if (!d.ws_defer) {
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv((int)pl.per, 64)), dim3(64), 0, st, d.ws, d.dw, (int)pl.per, pl.nsplit);
    PDES_LAUNCH_CHECK();
  }
  return PDES_OK;
}

// dry = true: only report whether this implementation would take the descriptor (nothing is enqueued)
int conv_backward_weight_mfma(const pdes_conv_desc& d, hipStream_t st, bool dry) {
  if (!d.ws || !d.has_bn || !(d.ksize == 5 || d.ksize == 3 || d.ksize == 1) || d.pad != (d.ksize - 1) / 2)
    return PDES_ENOSUP;
  if (d.stride != 1 && !(d.stride == 2 && d.ksize == 3 && !d.upsample)) return PDES_ENOSUP;
  if (d.nrep != PDES_NREP) return PDES_EINVAL;
  if (!wgrad_shape_ok(d)) return PDES_ENOSUP;
  // finalize on load: the one-N-tile stride-1 3x3 layers (the GF instantiations); PDES_OP_COPY has its own (flow_ops.hip)
  if (d.g_fused && !(d.ksize == 3 && d.stride == 1 && !d.upsample && d.Cout <= 16 && !wgrad_b3_applies(d) && d.fin_tstats &&
                     d.fin_xstats && d.out && d.g_ctot == d.out_ctot && d.g_coff == d.out_coff))
    return PDES_ENOSUP;
  if (dry) { WgradPlan pl; return wgrad_plan(d, &pl) ? PDES_OK : PDES_ENOSUP; }
  if (d.ksize == 1 && d.stride == 1 && !d.upsample) {      // conv_mfma_1x1.hip: one split per image
    WgradPlan pl;
    if (wgrad_plan(d, &pl) && pl.nsplit % d.B == 0) {
      const int rc = conv_backward_weight_1x1(d, pl.nsplit / d.B, st);
      if (rc == PDES_OK && !d.ws_defer) {
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv((int)pl.per, 64)), dim3(64), 0, st, d.ws, d.dw, (int)pl.per, pl.nsplit);
        PDES_LAUNCH_CHECK();
      }
      if (rc != PDES_ENOSUP) return rc;
    }
  }
  if (wgrad_b3_applies(d)) {
    WgradPlan pl;
    const int rc = conv_backward_weight_b3(d, st);
    if (rc == PDES_OK && !d.ws_defer && wgrad_plan(d, &pl)) {
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv((int)pl.per, 64)), dim3(64), 0, st, d.ws, d.dw, (int)pl.per, pl.nsplit);
      PDES_LAUNCH_CHECK();
    }
    if (rc != PDES_ENOSUP) return rc;
  }
  if (d.upsample) return launch_wgrad_up(d, st);
  if (d.stride == 2) return launch_wgrad<3, 2>(d, st);
  if (d.ksize == 5) return launch_wgrad<5, 1>(d, st);
  return d.ksize == 3 ? launch_wgrad<3, 1>(d, st) : launch_wgrad<1, 1>(d, st);
}

}  // namespace pdes

using namespace pdes;

extern "C" int pdes_conv_wgrad_plan(const pdes_context* ctx, const pdes_conv_desc* d, int* nsplit, long long* floats) {
  if (!d || !nsplit || !floats) return PDES_EINVAL;
  OptScope scope(ctx);
  WgradPlan pl;
  if (first_layer_shape(*d)) {                 // 7x7 stride-2 first convolution: one partial per image (deterministic)
    *nsplit = d->B;
    *floats = (long long)d->B * d->Cout * d->Cin * 49;
    return (*floats) * 4 <= d->ws_bytes ? PDES_OK : PDES_ENOSUP;
  }
  if (opt().conv_direct) return PDES_ENOSUP;   // VALU kernels forced: no partials
  if (wgrad_small_applies(*d)) {               // 3x3 on an 8x8 map (conv_small.hip): one partial per four images
    *nsplit = wgrad_small_splits(*d);
    *floats = (long long)(*nsplit) * d->Cout * d->Cin * 9;
    return (*floats) * 4 <= d->ws_bytes ? PDES_OK : PDES_ENOSUP;
  }
  if (!wgrad_shape_ok(*d) || !wgrad_plan(*d, &pl)) return PDES_ENOSUP;
  *nsplit = pl.nsplit;
  *floats = (long long)pl.nsplit * pl.per;
  return PDES_OK;
}

extern "C" int pdes_wgrad_reduce_all(const pdes_reduce_item* items, int n, int max_n, void* stream) {
  if (!items || n <= 0 || max_n <= 0) return PDES_EINVAL;
  int gx = cdiv(max_n, 64);
  gx = gx > 1024 ? 1024 : gx;
  hipLaunchKernelGGL(wgrad_reduce_all_kernel, dim3(gx, n), dim3(64), 0, static_cast<hipStream_t>(stream), items);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}
