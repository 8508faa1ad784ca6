// Implicit-GEMM convolutions on the f32 matrix cores (v_mfma_f32_16x16x4_f32, exact f32 = an fmaf
// chain) for the 1x1 / 3x3 / 5x5 convolutions of DenseED (reference models/codec.py:43-188), gfx950.
//
// A k x k convolution is k*k pointwise (1x1) contractions over shifted views of one LDS tile:
//     out[pixel][co] += sum_ci z[ci][pixel*stride + tap] * W[co][ci][tap]
// GEMM roles per MFMA (16x16x4): M = 16 consecutive output pixels of one image row (A operand, one
// ds_read_b32 per lane from the LDS tile), N = 16 output channels (B operand, one coalesced global
// load per lane from a pre-packed weight image), K = 4 input channels.  Accumulator lane layout:
// col = lane&15 = output channel, rows (lane>>4)*4+r = 4 consecutive pixels -> one float4 store.
//
// Workgroup = 256 threads = 4 waves, output tile = MT M-tiles (MT = 8: 4 rows x 32 px or 8 rows x
// 16 px; MT = 4 halves the rows when the grid would otherwise not fill the 256 CUs) of ONE sample.
// Input channels are processed in chunks of 16: the chunk's halo tile is BatchNorm+ReLU'd (and
// nearest-x2 upsampled) on the way into LDS, double buffered; the next chunk's global loads are
// issued before the current chunk's MFMAs and committed after them (register prefetch).
// LDS channel stride is == 16 (mod 32) dwords (odd for stride 2) so the 2 x 16 lanes of a
// ds_read_b32 group never collide.  Waves split the work either by K (dense layers, Cout = 16:
// each wave takes one k-step of every chunk, partial sums are combined through LDS once at the
// end) or by N (wide layers: each wave owns NT_W of the output-channel tiles).
// The epilogue accumulates the fp64 {sum, sum^2} per output channel that the consumer BatchNorms
// need (replicated accumulators, one atomic per channel per workgroup).
//
// The same kernel computes the data gradient (MODE_BWD): the "input" is dL/d(out) (raw, no BN;
// zero-inserted for a stride-2 convolution), the weights are the transposed / tap-flipped image,
// and the epilogue applies the ReLU mask and gamma, accumulates into T, and reduces dgamma / dbeta /
// {sum T, sum T xhat}.
#include <stdlib.h>
#include <hip/hip_ext.h>
#include "pdes_common.h"
#include "pdes_options.h"
#include "../../include/pdes_hip.h"
#include "pack_kernels.h"

namespace pdes {

typedef float v4f __attribute__((ext_vector_type(4)));

struct BnC { float mean, invstd, gamma, beta; };
__device__ __forceinline__ BnC bn_coef_m(const pdes_conv_desc& d, int c, bool publish = false) {
  BnC o;
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

template <int KS, int TWG, int MT, int S>
struct TileGeo {
  static constexpr int TH = MT / TWG, TW = 16 * TWG;          // output tile (pixels)
  static constexpr int PADL = (KS - 1) / 2;
  static constexpr int ROWS = (TH - 1) * S + KS;              // input rows of the tile
  static constexpr int TWI = S * TW;                          // "interior" input columns: 16-B aligned in
                                                              // global memory and in LDS -> float4 traffic
  static constexpr int NL = PADL, NR = KS - PADL - S;         // halo columns left / right of the interior
  static constexpr int COL0 = 4;                              // LDS column of the first interior element
  static constexpr int LDW = ((COL0 + TWI + NR + 3) / 4) * 4; // row pitch (multiple of 4 dwords)
  // channel stride: == 16 (mod 32) dwords so the two 16-lane halves of a ds_read_b32 group hit
  // disjoint banks (stride-2 lanes step by 2 dwords: 16 mod 32 keeps them disjoint as well)
  static constexpr int CS = ((ROWS * LDW - 16 + 31) / 32) * 32 + 16;
  static constexpr int KC = 16;                               // input channels per chunk
  static constexpr int NV4 = KC * ROWS * (TWI / 4);           // interior float4 per chunk
  static constexpr int NPV = (NV4 + 255) / 256;
  static constexpr int NHC = (NL + NR) > 0 ? (NL + NR) : 1;   // halo columns per row (>= 1 to keep index math defined)
  static constexpr int NH = KC * ROWS * (NL + NR);            // halo scalars per chunk
  static constexpr int NPH = (NH + 255) / 256;
  static_assert(MT % TWG == 0 && CS >= ROWS * LDW && NL <= COL0 && NR >= 0, "tile geometry");
};

#ifdef PDES_TRACE
__device__ unsigned long long pdes_trace_buf[16];
#define TR(i) do { if (trace_on) pdes_trace_buf[i] = wall_clock64(); } while (0)
#define TRACC(i, t0) do { tacc[i - 8] += wall_clock64() - (t0); } while (0)
#else
#define TR(i)
#define TRACC(i, t0)
#endif

// pdes_backward2 may ask the NEXT data-gradient launch of this file to carry a completion signal (an event that completes
// with the kernel, hipExtLaunchKernelGGL's stop event): the fork of the following layer's weight gradient then costs no
// barrier packet behind the kernel.  Consumed by the launch; a kernel family that does not look at it leaves it pending.
static thread_local hipEvent_t tl_stop_event = nullptr;
void set_dgrad_stop_event(hipEvent_t e) { tl_stop_event = e; }
bool dgrad_stop_event_pending() { return tl_stop_event != nullptr; }
hipEvent_t take_dgrad_stop_event() { hipEvent_t e = tl_stop_event; tl_stop_event = nullptr; return e; }       // (conv_small.hip)

enum { MODE_FWD = 0, MODE_BWD = 1 };
enum { KV_PLAIN = 0, KV_ZEROINS2 = 2 };   // K-operand view: as stored / zero-inserted x2 (stride-2 data gradient)

// NG = 2 (forward of the 16-output-channel layers only): TWO K-split wave groups of 4 waves each; group g stages and
// multiplies the chunks g, g+2, ... out of its own double-buffered LDS tile, halving the serial chunk chain of a
// workgroup (a dense layer at batch 32 has one workgroup per CU, i.e. otherwise one wave per SIMD).
// GF (data gradient of a layer with ONE chunk of output channels, the dense blocks' 16-channel layers): `g` still holds
// the accumulator T of the layer's output channels and the BatchNorm-backward finalize
//     g = invstd (T - mean(T) - xhat mean(T xhat))        (bn_bwd_finalize_kernel, the same expression)
// is applied while the tile is staged (x = the raw activation `out`, read beside T): no finalize launch in front of
// this kernel (pdes_backward2, option PDES_FIN_ONLOAD).
template <int KS, int TWG, int MT, int S, int WAVES_K, int NT_W, int MODE, int KM, bool PIPE, int NG, bool GF = false, bool TPI = false>
__global__ __launch_bounds__(256 * NG, NG == 2 ? 2 : ((NT_W == 1 && KS != 5 && S == 1) ? 3 : 1))
void conv_mfma_kernel(pdes_conv_desc d, const float* __restrict__ wm, int nt_total) {
  static_assert(NG == 1 || (NG == 2 && WAVES_K == 4 && MODE == MODE_FWD && PIPE), "wave groups: K-split forward only");
  static_assert(!GF || (MODE == MODE_BWD && !PIPE && KM == KV_PLAIN && NG == 1), "finalize on load: one-chunk data gradient");
  using G = TileGeo<KS, TWG, MT, S>;
  const int ntp = (nt_total + 7) & ~7;       // N-tiles of the packed weight image (zero padded)
  constexpr int KK = KS * KS;
  constexpr int KSW = 4 / WAVES_K;          // k-steps of a chunk handled by one wave
  static_assert(WAVES_K == 1 || MT >= 2, "K-split waves: MT/4 M-tiles per wave at the end (MT = 2: one tile each for two waves)");
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;      // ids inside the wave group
  const int grp = NG == 2 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;
  const int gwave = (int)threadIdx.x >> 6;                                  // wave id inside the workgroup
#ifdef PDES_TRACE
  const bool trace_on = tid == 0 && blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2 && blockIdx.z == 0;
  if (trace_on) { for (int i = 0; i < 16; ++i) pdes_trace_buf[i] = 0; }
  unsigned long long tt = 0, tacc[3] = {0, 0, 0};
#endif
  TR(0);
  const int wk = wave % WAVES_K, wn = wave / WAVES_K;
  const int b = blockIdx.y;
  const int nt_base = (blockIdx.z * (4 / WAVES_K) + wn) * NT_W;   // first N-tile of this wave

  // ---- K operand (staged through LDS) and the output side
  const float* kbase;
  int kC, kH, kW, kHc, kWc;
  constexpr int kmode = KM;     // compile time: a runtime view switch inside the load path costs a vmcnt(0) per chunk
  int Hout, Wout;
  if (MODE == MODE_FWD) {
    kC = d.Cin; kH = d.Hin; kW = d.Win;
    kbase = d.x + (size_t)b * d.x_ctot * d.Hin * d.Win;
    Hout = d.Hout; Wout = d.Wout;
  } else {
    kC = d.Cout; kH = d.Hout; kW = d.Wout;
    kbase = d.g + ((size_t)b * d.g_ctot + d.g_coff) * d.Hout * d.Wout;
    Hout = d.Hin; Wout = d.Win;
  }
  kHc = kmode ? 2 * kH : kH;
  kWc = kmode ? 2 * kW : kW;
  const int kpad = (kC + 15) & ~15;
  const int nchunk = kpad / 16;
  float* tile0 = smem + (MODE == MODE_FWD ? 4 * kpad : 0);             // [kpad] float4 per-channel coefficients first
  float* tile = tile0 + grp * (2 * G::KC * G::CS);                     // this wave group's two tile buffers

  const int tiles_x = Wout / G::TW;
  const int oy0 = (blockIdx.x / tiles_x) * G::TH, ox0 = (blockIdx.x % tiles_x) * G::TW;

  // GF: {mean, invstd, mean(T), mean(T xhat)} of the 16 staged channels (the batch statistics of the OUTPUT buffer)
  __shared__ float4 gfc[GF ? 16 : 1];
  const float* xo_base = nullptr;           // GF: the raw activation beside g (same layout: out_ctot == g_ctot, out_coff == g_coff)
  // one statistic load per thread, issued FIRST (thread = (channel, {sum T, sum T xhat}, replica)): their round trip runs
  // under the geometry arithmetic and the tile loads; reduced with shuffles in front of the first barrier
  double gf_sv = 0.0;
  float2 gf_ce = make_float2(0.f, 0.f);
  if constexpr (GF) {
    static_assert(PDES_NREP == 8, "finalize on load: 16 channels x 2 sums x 8 replicas = one load per thread");
    xo_base = d.out + ((size_t)b * d.out_ctot + d.out_coff) * d.Hout * d.Wout;
    const int c = d.g_coff + min((int)threadIdx.x >> 4, d.Cout - 1);
    gf_sv = d.fin_tstats[(long long)(threadIdx.x & 7) * d.rep_stride + 2 * c + ((threadIdx.x >> 3) & 1)];
    if (d.fin_coef) gf_ce = reinterpret_cast<const float2*>(d.fin_coef)[c];
  }
  // FWD: per-channel {mean, gamma*invstd, beta, -} as one float4 (a single ds_read_b128 per staged float4)
  float4* cf4 = reinterpret_cast<float4*>(smem);
  if (MODE == MODE_FWD) {
    for (int c = threadIdx.x; c < kpad; c += 256 * NG) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < d.Cin) { const BnC k = bn_coef_m(d, c, (blockIdx.x | blockIdx.y | blockIdx.z) == 0); v = make_float4(k.mean, k.gamma * k.invstd, k.beta, 0.f); }
      cf4[c] = v;
    }
  }

  // ---- staging: rows of the halo tile as aligned float4 (interior) + a few halo scalars.
  // Geometry is per thread and chunk independent: compute it once.
  const int HWs = kH * kW;
  const bool halo_live = (G::NL + G::NR) > 0 && tiles_x > 1;    // full-width tiles: halo columns are padding
  int vg[G::NPV], vl[G::NPV];        // interior float4: global offset (within the chunk), LDS offset (-1: none)
  int hg[G::NPH > 0 ? G::NPH : 1], hl[G::NPH > 0 ? G::NPH : 1];
  unsigned vrow = 0, hval = 0, hzero = 0;
#pragma unroll
  for (int i = 0; i < G::NPV; ++i) {
    const int e = tid + 256 * i;
    const int ch = e / (G::ROWS * (G::TWI / 4)), rem = e % (G::ROWS * (G::TWI / 4));
    const int r = rem / (G::TWI / 4), j = rem % (G::TWI / 4);
    const int cy = oy0 * S - G::PADL + r, cx = ox0 * S + 4 * j;
    const bool ok = e < G::NV4 && cy >= 0 && cy < kHc;
    const int cyc = min(max(cy, 0), kHc - 1);
    const int sy = kmode ? (cyc >> 1) : cyc, sx = kmode ? (cx >> 1) : cx;
    vg[i] = sy * kW + sx;                               // always a valid address (row clamped)
    vl[i] = e < G::NV4 ? ch * G::CS + r * G::LDW + G::COL0 + 4 * j : -1;
    if (ok && !(kmode == KV_ZEROINS2 && (cy & 1))) vrow |= 1u << i;
  }
#pragma unroll
  for (int i = 0; i < G::NPH; ++i) {
    const int e = tid + 256 * i;
    const int ch = e / (G::ROWS * G::NHC), rem = e % (G::ROWS * G::NHC);
    const int r = rem / G::NHC, h = rem % G::NHC;
    const int cy = oy0 * S - G::PADL + r;
    const int cx = h < G::NL ? ox0 * S - G::NL + h : ox0 * S + G::TWI + (h - G::NL);
    const int lc = h < G::NL ? G::COL0 - G::NL + h : G::COL0 + G::TWI + (h - G::NL);
    bool ok = e < G::NH && cy >= 0 && cy < kHc && cx >= 0 && cx < kWc;
    if (kmode == KV_ZEROINS2) ok = ok && !((cy | cx) & 1);
    const int cyc = min(max(cy, 0), kHc - 1), cxc = min(max(cx, 0), kWc - 1);
    const int sy = kmode ? (cyc >> 1) : cyc, sx = kmode ? (cxc >> 1) : cxc;
    hg[i] = sy * kW + sx;
    hl[i] = e < G::NH ? ch * G::CS + r * G::LDW + lc : -1;
    if (ok) hval |= 1u << i;
  }
  (void)hzero;

  // two register stages: the loads of chunk c+2 are issued before the MFMAs of chunk c, so a tile has
  // two chunks of matrix work (not one) to arrive from L2/HBM before it is committed to LDS
  constexpr int NPHS = G::NPH > 0 ? G::NPH : 1;
  struct Stage {
    float4 pv[G::NPV]; float ph[NPHS];
    float4 xv[GF ? G::NPV : 1]; float xh[GF ? NPHS : 1];      // GF: the raw activation at the same places
  };
  Stage sA, sB;
  // loads are unconditional (row offsets are clamped into the image above, the channel is clamped
  // here); validity is applied when the registers are committed to LDS
  auto issue = [&](int chunk, Stage& st) __attribute__((always_inline)) {
    float4 (&pv)[G::NPV] = st.pv;
    float (&ph)[NPHS] = st.ph;
    const float* src = kbase + (size_t)chunk * 16 * HWs;
    const int cmax = kC - chunk * 16 - 1;           // last valid channel of this chunk
#pragma unroll
    for (int i = 0; i < G::NPV; ++i) {
      const int ch = min((tid + 256 * i) / (G::ROWS * (G::TWI / 4)), cmax);
      const float* p = src + ch * HWs + vg[i];
      if constexpr (kmode == KV_PLAIN) {
        pv[i] = *reinterpret_cast<const float4*>(p);
        if constexpr (GF) st.xv[i] = *reinterpret_cast<const float4*>(xo_base + ch * HWs + vg[i]);
      } else {                                       // raw pair; expanded to (x, 0, y, 0) at commit time
        const float2 t = *reinterpret_cast<const float2*>(p);
        pv[i].x = t.x; pv[i].y = t.y;
      }
    }
    if (halo_live) {
#pragma unroll
      for (int i = 0; i < G::NPH; ++i) {
        const int ch = min((tid + 256 * i) / (G::ROWS * G::NHC), cmax);
        ph[i] = src[ch * HWs + hg[i]];
        if constexpr (GF) st.xh[i] = xo_base[ch * HWs + hg[i]];
      }
    }
  };
  auto commit = [&](int chunk, int buf, const Stage& st) __attribute__((always_inline)) {
    const float4 (&pv)[G::NPV] = st.pv;
    const float (&ph)[NPHS] = st.ph;
    float* t = tile + buf * (G::KC * G::CS);
    const int crem = kC - chunk * 16;
#pragma unroll
    for (int i = 0; i < G::NPV; ++i) {
      if (vl[i] >= 0) {
        float4 z = kmode == KV_PLAIN ? pv[i] : make_float4(pv[i].x, 0.f, pv[i].y, 0.f);
        const int ch = (tid + 256 * i) / (G::ROWS * (G::TWI / 4));
        const bool ok = ((vrow >> i) & 1u) && ch < crem;
        if (MODE == MODE_FWD) {
          const float4 k = cf4[chunk * 16 + ch];
          z.x = ok ? fmaxf(0.f, (z.x - k.x) * k.y + k.z) : 0.f;
          z.y = ok ? fmaxf(0.f, (z.y - k.x) * k.y + k.z) : 0.f;
          z.z = ok ? fmaxf(0.f, (z.z - k.x) * k.y + k.z) : 0.f;
          z.w = ok ? fmaxf(0.f, (z.w - k.x) * k.y + k.z) : 0.f;
        } else if (!ok) {
          z = make_float4(0.f, 0.f, 0.f, 0.f);
        } else if constexpr (GF) {
          const float4 k = gfc[ch], x = st.xv[i];
          z.x = k.y * (z.x - k.z - (x.x - k.x) * k.y * k.w);
          z.y = k.y * (z.y - k.z - (x.y - k.x) * k.y * k.w);
          z.z = k.y * (z.z - k.z - (x.z - k.x) * k.y * k.w);
          z.w = k.y * (z.w - k.z - (x.w - k.x) * k.y * k.w);
        }
        *reinterpret_cast<float4*>(t + vl[i]) = z;
      }
    }
    if (halo_live) {
#pragma unroll
      for (int i = 0; i < G::NPH; ++i) {
        if (hl[i] >= 0) {
          float z = ph[i];
          const int ch = (tid + 256 * i) / (G::ROWS * G::NHC);
          const bool ok = ((hval >> i) & 1u) && ch < crem;
          if (MODE == MODE_FWD) {
            const float4 k = cf4[chunk * 16 + ch];
            z = ok ? fmaxf(0.f, (z - k.x) * k.y + k.z) : 0.f;
          } else if (!ok) {
            z = 0.f;
          } else if constexpr (GF) {
            const float4 k = gfc[ch];
            z = k.y * (z - k.z - (st.xh[i] - k.x) * k.y * k.w);
          }
          t[hl[i]] = z;
        }
      }
    }
  };
  // halo columns that are always outside the image are zeroed once (both buffers)
  if (!halo_live && (G::NL + G::NR) > 0) {
#pragma unroll
    for (int i = 0; i < G::NPH; ++i)
      if (hl[i] >= 0) {
        tile[hl[i]] = 0.f;
        if constexpr (PIPE) tile[G::KC * G::CS + hl[i]] = 0.f;      // !PIPE: one chunk, one buffer (the launch allocates one)
      }
  }

  v4f acc[MT][NT_W];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT_W; ++nt) acc[mt][nt] = (v4f){0.f, 0.f, 0.f, 0.f};

  TR(1);
  const int a_lane = (lane >> 4) * G::CS + (lane & 15) * S;
  // B operand: packed image [(kstep*KK + tap)*ntp + nt][64] (ntp = N-tiles padded to a multiple of 8 with
  // zero tiles, so no bounds test is needed), one coalesced load per (tap, N-tile), held in two register
  // sets: while k-step s runs on the matrix pipe from one set, the next k-step streams into the other.
  static_assert(KK * NT_W <= 25, "two B register sets must fit");
  const int wks = WAVES_K == 4 ? __builtin_amdgcn_readfirstlane(wk) : 0;    // wave-uniform -> SGPR
  const int ksteps = kpad / 4;
  float bA[KK][NT_W], bB[KK][NT_W];
  // TP ("tile pipelined" data gradient: ONE chunk of <= 16 output channels, one N-tile per wave -- the dense blocks' layers):
  // the weights of all four k-steps stay in registers (36 values) and the loop runs M-tile by M-tile, each tile's epilogue
  // (mask, gamma, T +=, store) straight behind its 36 MFMAs.  With the k-step loop outside (all 8 tiles accumulating, one
  // epilogue at the end) every workgroup of the launch -- one round on the chip -- loaded, multiplied and stored in
  // lock-step: three phases in sequence (27 us for 180 <- 16 channels at 32 x 32: 9 of loads, 11.5 of MFMAs, 5 of stores);
  // tile by tile the stores of one tile and the returning loads of the next run under the MFMAs of the tiles between.
  // Stand-alone (B = 32) it gains where several workgroups share a CU (180 <- 16 channels at 32 x 32: 28.0 -> 25.7 us,
  // 128 <- 16: 21.6 -> 20.7) and loses where one workgroup per CU is latency bound (48 <- 16: 13.6 -> 15.1, 16 x 16 maps
  // +0.3 ... 0.6 us); inside the step -- beside the weight-gradient streams -- it wins everywhere: same process, threshold
  // on the input channels 0 (never) / 48 / 96 / 128: 1.6580 / 1.6442 / 1.6429 / 1.6548 ms per step, the 16-wide maps on top
  // 1.6440 -> 1.6372.  Selected by the launcher (TPI; options PDES_DG_TILEPIPE, PDES_DG_TILEPIPE16).
  static_assert(!TPI || (MODE == MODE_BWD && !PIPE && NT_W == 1 && WAVES_K == 1 && KS == 3 && KM == KV_PLAIN && NG == 1), "tile pipeline");
  constexpr bool TP = TPI;
  float bC[TP ? KK : 1][NT_W], bD[TP ? KK : 1][NT_W];
  auto load_b = [&](int kstep, float (&dst)[KK][NT_W]) {
    const float* wp = wm + ((size_t)min(kstep, ksteps - 1) * KK * ntp + nt_base) * 64 + lane;
#pragma unroll
    for (int t = 0; t < KK; ++t)
#pragma unroll
      for (int nt = 0; nt < NT_W; ++nt) dst[t][nt] = wp[(size_t)(t * ntp + nt) * 64];
  };
  auto mfma_kstep = [&](const float* tk, const float (&bw)[KK][NT_W]) {
#pragma unroll
    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          // zero-inserted view (stride-2 data gradient): tile rows start on an even row, data sits on even rows of
          // the view only, so for half of the (M-tile row, ky) pairs the A operand is identically zero: skip them
          if constexpr (KM == KV_ZEROINS2) { if ((((mt / TWG) + ky) & 1) == 0) continue; }
          const float a = tk[((mt / TWG) * S + ky) * G::LDW + (G::COL0 - G::PADL) + (mt % TWG) * 16 * S + kx];
#pragma unroll
          for (int nt = 0; nt < NT_W; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[ky * KS + kx][nt], acc[mt][nt], 0, 0, 0);
        }
      }
  };

  // The loop body below is straight-line (every global load is unconditional, indices are clamped
  // instead): vmcnt is an in-order counter, and only without divergent paths can the compiler wait for
  // exactly the loads a k-step needs instead of draining the prefetches that were just issued.
  // virtual step v of this wave group <-> chunk grp + NG * v (clamped for the loads, tested for the work)
  auto cidx = [&](int v) __attribute__((always_inline)) { return min(grp + NG * v, nchunk - 1); };
  load_b(cidx(0) * 4 + wks, bA);
  if constexpr (TP) { load_b(1, bB); load_b(2, bC); load_b(3, bD); }
  issue(cidx(0), sA);
  // data gradient: the BatchNorm coefficients of this wave's input channels are only needed in the epilogue;
  // their 32 replica loads + fp64 arithmetic are issued here so that they overlap the matrix work
  BnC kepi[MODE == MODE_BWD ? NT_W : 1];
  if constexpr (MODE == MODE_BWD) {
#pragma unroll
    for (int nt = 0; nt < NT_W; ++nt) {
      const int ci = min((nt_base + nt) * 16 + (lane & 15), d.Cin - 1);
      kepi[nt] = bn_coef_m(d, ci);
    }
  }
  if constexpr (PIPE) issue(cidx(1), sB);      // !PIPE: exactly one chunk, no second stage at all
  // data gradient with one N-tile per wave: x and T of the wave's output pixels are requested HERE, so that their trip
  // to L2 / HBM runs under the matrix loop instead of at the head of the epilogue (where it was half of the kernel:
  // 8.9 of 18.3 us for 16 -> 180 channels at 32 x 32)
  constexpr bool EPI_PRE = MODE == MODE_BWD && NT_W == 1 && WAVES_K == 1;
  constexpr int TPD = 3;                      // TP: tiles of x / T in flight ahead of the tile being multiplied
  float4 xpre[EPI_PRE ? MT : 1], tpre[EPI_PRE ? MT : 1];
  size_t epi_cb = 0;
  auto epi_load = [&](int mt) __attribute__((always_inline)) {
    const size_t idx = epi_cb + (size_t)(oy0 + mt / TWG) * d.Win + ox0 + (mt % TWG) * 16;
    xpre[mt] = *reinterpret_cast<const float4*>(d.x + idx);
#ifdef PDES_DG_NOTLOAD           // (component-timing build: the data gradient without its T loads)
    tpre[mt] = make_float4(0.f, 0.f, 0.f, 0.f);
#else
    tpre[mt] = d.t_accumulate ? *reinterpret_cast<const float4*>(d.t_in + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
#endif
  };
  if constexpr (EPI_PRE) {
    const int cic = min(nt_base * 16 + (lane & 15), d.Cin - 1);
    epi_cb = ((size_t)b * d.x_ctot + cic) * (size_t)(d.Hin * d.Win) + (size_t)((lane >> 4) * 4);
    // (GF without the tile pipeline: requested behind the commit below instead -- the staged T / x tiles and all eight
    //  x / T quads do not fit the register budget of three workgroups per CU together)
    if constexpr (!GF || TP) {
#pragma unroll
      for (int mt = 0; mt < (TP ? (TPD < MT ? TPD : MT) : MT); ++mt) epi_load(mt);
    }
  }
  if constexpr (GF) {
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) gf_sv += __shfl_xor(gf_sv, o, 64);          // the 8 replicas (fixed order)
    const double sx = __shfl_down(gf_sv, 8, 64);                                // lane & 15 == 0: {sum T} here, {sum T xhat} 8 lanes up
    if ((threadIdx.x & 15) == 0) {
      const int k16 = threadIdx.x >> 4;
      const double n = (double)d.B * d.Hout * d.Wout, inv_n = 1.0 / n;
      MeanInv mi;
      mi.mean = gf_ce.x; mi.invstd = gf_ce.y;
      if (!(gf_ce.y > 0.f))                    // (not published: never on the training path, every channel has a forward consumer)
        mi = batch_mean_invstd(nullptr, d.fin_xstats, d.rep_stride, n, d.eps, d.g_coff + min(k16, d.Cout - 1), false);
      gfc[k16] = make_float4(mi.mean, mi.invstd, (float)(gf_sv * inv_n), (float)(sx * inv_n));
    }
  }
  __syncthreads();                 // cf visible
  TR(2);
  if (grp < nchunk) commit(cidx(0), 0, sA);
  if constexpr (EPI_PRE && GF && !TP) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) epi_load(mt);
  }
  __syncthreads();
  TR(3);

  // one chunk: `sfree` = the (free) register stage that receives step v+2, `snext` = the stage holding
  // step v+1; `b0` holds the weights of this chunk's first k-step, `b1` is the other weight set
  auto step = [&](int v, Stage& sfree, const Stage& snext, float (&b0)[KK][NT_W], float (&b1)[KK][NT_W])
      __attribute__((always_inline)) {
    const int buf = v & 1;
    const int chunk = grp + NG * v;            // may lie past the last chunk for the group with fewer chunks
#ifdef PDES_TRACE
    tt = wall_clock64();
#endif
    const float* tb = tile + buf * (G::KC * G::CS) + a_lane;
    if constexpr (WAVES_K == 4) {
      load_b(cidx(v + 1) * 4 + wks, b1);
      if constexpr (PIPE) issue(cidx(v + 2), sfree);
      if (NG == 1 || chunk < nchunk) mfma_kstep(tb + wks * 4 * G::CS, b0);
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        load_b(chunk * 4 + s + 1, (s & 1) ? b0 : b1);
        if constexpr (PIPE) { if (s == 0) issue(min(chunk + 2, nchunk - 1), sfree); }
        if ((chunk * 4 + s) * 4 < kC)            // scalar: skip k-steps that lie entirely in the zero padding
          mfma_kstep(tb + s * 4 * G::CS, (s & 1) ? b1 : b0);
      }
    }
    TRACC(8, tt);      // issue + MFMAs
#ifdef PDES_TRACE
    tt = wall_clock64();
#endif
    if constexpr (PIPE) { if (chunk + NG < nchunk) commit(chunk + NG, buf ^ 1, snext); }
    TRACC(9, tt);      // commit
#ifdef PDES_TRACE
    tt = wall_clock64();
#endif
    __syncthreads();
    TRACC(10, tt);     // barrier wait
  };
  if constexpr (TP) {
    // M-tile by M-tile: 36 MFMAs, then the tile's epilogue
    const float* tb = tile + a_lane;
    const int HWi = d.Hin * d.Win;
    const int ci = nt_base * 16 + (lane & 15);
    const BnC k = kepi[0];
    const float scale = k.gamma * k.invstd;
    const bool live = ci < d.Cin, fin = ci >= d.final_c0 && ci < d.final_c1;
    float* tb2 = d.t_in + (size_t)b * d.x_ctot * HWi + (size_t)min(ci, d.Cin - 1) * HWi + (size_t)((lane >> 4) * 4);
    float dg = 0.f, db = 0.f, st = 0.f, sx = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      v4f a4 = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        if (s4 * 4 < kC) {                      // (scalar: k-steps that lie entirely in the zero padding are skipped)
          const float (&bw)[KK][NT_W] = s4 == 0 ? bA : (s4 == 1 ? bB : (s4 == 2 ? bC : bD));
#pragma unroll
          for (int ky = 0; ky < KS; ++ky)
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
              const float a = tb[s4 * 4 * G::CS + ((mt / TWG) * S + ky) * G::LDW + (G::COL0 - G::PADL) + (mt % TWG) * 16 * S + kx];
#ifdef PDES_DG_NOMFMA            // (component-timing build: what the dense data gradients' MFMAs cost the step)
              a4[0] += a * bw[ky * KS + kx][0];
#else
              a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[ky * KS + kx][0], a4, 0, 0, 0);
#endif
            }
        }
      }
      if (mt + TPD < MT) epi_load(mt + TPD);     // (requested behind this tile's MFMAs, consumed TPD tiles later)
      if (live) {
        const float xs[4] = {xpre[mt].x, xpre[mt].y, xpre[mt].z, xpre[mt].w};
        float ts[4] = {tpre[mt].x, tpre[mt].y, tpre[mt].z, tpre[mt].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float y = (xs[r] - k.mean) * scale + k.beta;
          const float xh = (xs[r] - k.mean) * k.invstd;
          const float dyv = (y > 0.f) ? a4[r] : 0.f;
          db += dyv; dg += dyv * xh;
          ts[r] += k.gamma * dyv;
          if (fin) { st += ts[r]; sx += ts[r] * xh; }
        }
#ifdef PDES_DG_NOTSTORE          // (component-timing build: the data gradient without its T stores)
        if (ts[0] == 123.456f)
#endif
        *reinterpret_cast<float4*>(tb2 + (size_t)(oy0 + mt / TWG) * d.Win + ox0 + (mt % TWG) * 16) = make_float4(ts[0], ts[1], ts[2], ts[3]);
      }
    }
    dg += __shfl_xor(dg, 16, 64); dg += __shfl_xor(dg, 32, 64);
    db += __shfl_xor(db, 16, 64); db += __shfl_xor(db, 32, 64);
    st += __shfl_xor(st, 16, 64); st += __shfl_xor(st, 32, 64);
    sx += __shfl_xor(sx, 16, 64); sx += __shfl_xor(sx, 32, 64);
    if (lane < 16 && live) {
      const long long ro = (long long)rep_of_block(d.nrep) * d.rep_stride;
#ifndef PDES_DG_NOATOM
      atomicAdd(&d.bn_grad[ro + 2 * ci], (double)dg);
      atomicAdd(&d.bn_grad[ro + 2 * ci + 1], (double)db);
#endif
      if (fin) {
        atomicAdd(&d.t_stats[ro + 2 * ci], (double)st);
        atomicAdd(&d.t_stats[ro + 2 * ci + 1], (double)sx);
      }
    }
    return;
  } else if constexpr (!PIPE) {
    step(0, sA, sA, bA, bB);
  } else {
    const int nv = (nchunk + NG - 1) / NG;     // the same number of steps (barriers) for every wave group
    int v = 0;
    for (; v + 1 < nv; v += 2) {
      step(v, sA, sB, bA, bB);
      if constexpr (WAVES_K == 4) step(v + 1, sB, sA, bB, bA);
      else step(v + 1, sB, sA, bA, bB);
    }
    if (v < nv) step(v, sA, sB, bA, bB);
  }

  TR(4);
  // ---- combine the K-split partial sums: wave w ends up owning M-tiles [w*MT/4, (w+1)*MT/4)
  constexpr int NWV = 4 * NG;                                        // K-split waves of the workgroup
  constexpr int NOWN = (WAVES_K == 4) ? ((MT >= NWV) ? NWV : (MT >= 4 ? 4 : MT)) : 1;   // waves that own output M-tiles afterwards
  constexpr int MT_OWN = (WAVES_K == 4) ? MT / NOWN : MT;
  const int mt0 = (WAVES_K == 4) ? MT_OWN * gwave : 0;
  const bool owner = WAVES_K != 4 || gwave < NOWN;
  if (WAVES_K == 4) {
    float* red = tile0;                        // [NWV waves][MT][4 r][64 lanes]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((gwave * MT + mt) * 4 + r) * 64 + lane] = acc[mt][0][r];
    __syncthreads();
    if (owner) {
#pragma unroll
      for (int j = 0; j < MT_OWN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s = 0.f;
#pragma unroll
          for (int w = 0; w < NWV; ++w) s += red[((w * MT + mt0 + j) * 4 + r) * 64 + lane];
          acc[j][0][r] = s;                    // acc[0..MT_OWN) now hold the wave's own M-tiles
        }
    }
  }

  TR(5);
  const int HWo = Hout * Wout;
  const int px = (lane >> 4) * 4;              // first of the lane's 4 consecutive pixels in the M-tile
  if (MODE == MODE_FWD) {
#pragma unroll
    for (int nt = 0; nt < NT_W; ++nt) {
      const int co = (nt_base + nt) * 16 + (lane & 15);
      float s = 0.f, q = 0.f;
      if (co < d.Cout && owner) {
        float* ob = d.out + ((size_t)b * d.out_ctot + d.out_coff + co) * HWo;
#pragma unroll
        for (int j = 0; j < MT_OWN; ++j) {
          const int mt = mt0 + j;
          const v4f v = acc[j][nt];
          const int oy = oy0 + mt / TWG, ox = ox0 + (mt % TWG) * 16 + px;
          *reinterpret_cast<float4*>(ob + (size_t)oy * Wout + ox) = make_float4(v[0], v[1], v[2], v[3]);
          s += (v[0] + v[1]) + (v[2] + v[3]);
          q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
      }
      if (d.out_stats) {
        s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
        q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
        double* os = d.out_stats + (long long)rep_of_block(d.nrep) * d.rep_stride;
        if (WAVES_K == 4) {
          // the four waves hold partial sums of the SAME 16 channels: combine through LDS -> one
          // pair of atomics per channel per workgroup
          float* sred = tile0 + NWV * MT * 4 * 64;   // past the accumulator exchange area
          if (lane < 16 && owner) { sred[(gwave * 16 + lane) * 2] = s; sred[(gwave * 16 + lane) * 2 + 1] = q; }
          __syncthreads();
          if (gwave == 0 && lane < 32) {
            const int c = lane >> 1, w = lane & 1;
            float t = 0.f;
#pragma unroll
            for (int ow = 0; ow < NOWN; ++ow) t += sred[(ow * 16 + c) * 2 + w];
            const int cc = nt_base * 16 + c;
#ifndef PDES_FW_NOATOM          // (component-timing build: EXPERIMENTS.md round 4)
            if (cc < d.Cout) atomicAdd(&os[2 * (d.out_coff + cc) + w], (double)t);
#endif
          }
        } else if (lane < 16 && co < d.Cout) {
          atomicAdd(&os[2 * (d.out_coff + co)], (double)s);
          atomicAdd(&os[2 * (d.out_coff + co) + 1], (double)q);
        }
      }
    }
  } else {
    // data gradient epilogue (nearest-x2 layers run on the sub-pixel kernels of conv_mfma_up.hip)
    const int HWi = d.Hin * d.Win;
    const float* xb = d.x + (size_t)b * d.x_ctot * HWi;
    float* tb2 = d.t_in + (size_t)b * d.x_ctot * HWi;
#pragma unroll
    for (int nt = 0; nt < NT_W; ++nt) {
      const int ci = (nt_base + nt) * 16 + (lane & 15);
      float dg = 0.f, db = 0.f, st = 0.f, sx = 0.f;
      if (ci < d.Cin) {
        const BnC k = kepi[nt];
        const float scale = k.gamma * k.invstd;
        const bool fin = ci >= d.final_c0 && ci < d.final_c1;
        {
          // all loads of this N-tile first, then the arithmetic and the stores: the store to T may alias the next
          // loads for the compiler, which otherwise serialises MT_OWN load -> wait -> store round trips
          float4 xq[MT_OWN], tq[MT_OWN];
#pragma unroll
          for (int j = 0; j < MT_OWN; ++j) {
            const int mt = mt0 + j;
            const size_t idx = (size_t)ci * HWi + (size_t)(oy0 + mt / TWG) * d.Win + ox0 + (mt % TWG) * 16 + px;
            if constexpr (EPI_PRE) { xq[j] = xpre[mt]; tq[j] = tpre[mt]; continue; }
            xq[j] = *reinterpret_cast<const float4*>(xb + idx);
            tq[j] = d.t_accumulate ? *reinterpret_cast<const float4*>(tb2 + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int j = 0; j < MT_OWN; ++j) {
            const int mt = mt0 + j;
            const v4f v = acc[j][nt];
            const size_t idx = (size_t)ci * HWi + (size_t)(oy0 + mt / TWG) * d.Win + ox0 + (mt % TWG) * 16 + px;
            const float xs[4] = {xq[j].x, xq[j].y, xq[j].z, xq[j].w};
            float ts[4] = {tq[j].x, tq[j].y, tq[j].z, tq[j].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float y = (xs[r] - k.mean) * scale + k.beta;
              const float xh = (xs[r] - k.mean) * k.invstd;
              const float dyv = (y > 0.f) ? v[r] : 0.f;
              db += dyv; dg += dyv * xh;
              ts[r] += k.gamma * dyv;
              if (fin) { st += ts[r]; sx += ts[r] * xh; }
            }
            *reinterpret_cast<float4*>(tb2 + idx) = make_float4(ts[0], ts[1], ts[2], ts[3]);
          }
        }
      }
      dg += __shfl_xor(dg, 16, 64); dg += __shfl_xor(dg, 32, 64);
      db += __shfl_xor(db, 16, 64); db += __shfl_xor(db, 32, 64);
      st += __shfl_xor(st, 16, 64); st += __shfl_xor(st, 32, 64);
      sx += __shfl_xor(sx, 16, 64); sx += __shfl_xor(sx, 32, 64);
      if (lane < 16 && ci < d.Cin) {
        const long long ro = (long long)rep_of_block(d.nrep) * d.rep_stride;
#ifndef PDES_DG_NOATOM          // (component-timing build: EXPERIMENTS.md round 4)
        atomicAdd(&d.bn_grad[ro + 2 * ci], (double)dg);
        atomicAdd(&d.bn_grad[ro + 2 * ci + 1], (double)db);
#endif
        if (ci >= d.final_c0 && ci < d.final_c1) {
          atomicAdd(&d.t_stats[ro + 2 * ci], (double)st);
          atomicAdd(&d.t_stats[ro + 2 * ci + 1], (double)sx);
        }
      }
    }
  }
  TR(6);
#ifdef PDES_TRACE
  if (trace_on) { pdes_trace_buf[8] = tacc[0]; pdes_trace_buf[9] = tacc[1]; pdes_trace_buf[10] = tacc[2]; }
#endif
}

// ------------------------------------------------------------------------------------------------
// weight images for the MFMA kernels, rebuilt from the live weights every step (one launch for the
// whole network; NT = N-tile count rounded up to a multiple of 8, padding tiles are zero):
//                 forward  [(kstep*KK + tap)*NT + nt][kq*16 + n] = W[n + 16 nt][4 kstep + kq][tap]
//                 backward [(kstep*KK + tap)*NT + nt][kq*16 + n] = W[4 kstep + kq][n + 16 nt][KK-1-tap]
__global__ __launch_bounds__(256) void pack_mfma_kernel(const pdes_mfma_pack_item* __restrict__ items) {
  pack_mfma_item(items[blockIdx.y], blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------- host dispatch
// (W, H) = size of the map the kernel tiles: the output map (forward) or the input map (data gradient)
static bool mfma_shape_ok(const pdes_conv_desc& d, bool bwd, int* W, int* H) {
  if (!(d.ksize == 5 || d.ksize == 3 || d.ksize == 1) || d.pad != (d.ksize - 1) / 2) return false;
  if (d.upsample) return false;     // conv_mfma_up.hip (sub-pixel form) or the direct kernels
  if (d.stride != 1 && !(d.stride == 2 && d.ksize == 3 && d.Hin == 2 * d.Hout && d.Win == 2 * d.Wout))
    return false;
  if (!d.has_bn) return false;
  *W = bwd ? d.Win : d.Wout;
  *H = bwd ? d.Hin : d.Hout;
  if (*W % 16 || (*W >= 32 && *W % 32)) return false;
  const int twg = *W >= 32 ? 2 : 1;
  return *H % (8 / twg) == 0;
}


template <int KS, int S, int MODE, int KM>
static int launch_mfma(const pdes_conv_desc& d, const float* wm, int W, int H, hipStream_t st, bool dry = false) {
  const bool bwd = MODE == MODE_BWD;
  const int kC = bwd ? d.Cout : d.Cin, nC = bwd ? d.Cin : d.Cout;
  // finalize on load: the one-chunk 3x3 data gradient only (GF instantiations below); PDES_OP_COPY has its own (flow_ops.hip)
  constexpr bool gf_able = MODE == MODE_BWD && KS == 3 && S == 1 && KM == KV_PLAIN;
  if (d.g_fused && !(gf_able && kC <= 16 && nC > 16 && d.fin_tstats && d.fin_xstats && d.out && d.g_ctot == d.out_ctot &&
                     d.g_coff == d.out_coff))
    return PDES_ENOSUP;
  const int kpad = (kC + 15) & ~15, nchunk = kpad / 16;
  const int nt_total = (nC + 15) / 16;
  const int twg = W >= 32 ? 2 : 1;
  // wave roles and tile size.  One N-tile (dense layers): the four waves split K.  Otherwise the waves split
  // N, NTW tiles each, the rest of N over gridDim.z; among (M-tiles per workgroup, NTW) = (8,2) (8,1) (4,2) (4,1)
  // take the first that puts a workgroup on every CU (small maps: staging a tile twice is cheaper than idle CUs),
  // else the one with the most workgroups.
  int wk, ntw, gz = 1, mt = 8;
  const long long tiles8 = (long long)(W / (16 * twg)) * (H / (8 / twg)) * d.B;
  const bool mt4_ok = KS != 5 && H % (4 / twg) == 0;
  if (nt_total == 1) {
    wk = 4; ntw = 1;
    if (tiles8 < 256 && mt4_ok) mt = 4;
    // round 6: the forward of the 16x16 dense layers at batch 32 is 128 workgroups of MT = 4 on 256 CUs -- tiles of 2 rows x
    // 16 pixels put a workgroup on every CU and halve its serial chain of MFMAs (18 per wave and chunk; the two halo rows
    // are staged twice as often: out of L2).  PDES_MFMA_MT2=0 keeps MT = 4.
    if (!bwd && KS == 3 && S == 1 && mt == 4 && twg == 1 && tiles8 * 2 < 256 && H % 2 == 0 && nchunk >= 3 && opt().mfma_mt2) mt = 2;
    // (round 6, measured and removed: the 32-wide maps' forward -- exactly one MT = 8 workgroup per CU at batch 32 -- on tiles of
    //  2 rows x 32 pixels: 188 -> 207 us over the twelve layers, step +1.1 %; experiments/round6.md)
  } else {
    wk = 1;
    const bool ntw2_ok = KS != 5 && kpad > 16 && nt_total > 4;
    const int cand[4][2] = {{8, 2}, {8, 1}, {4, 2}, {4, 1}};
    long long best = -1;
    mt = 8; ntw = 1;
    for (int c = 0; c < 4; ++c) {
      const int cm = cand[c][0], cn = cand[c][1];
      if ((cn == 2 && !ntw2_ok) || (cm == 4 && !mt4_ok)) continue;
      const long long wgs = tiles8 * (8 / cm) * ((nt_total + 4 * cn - 1) / (4 * cn));
      if (wgs >= 256) { mt = cm; ntw = cn; best = wgs; break; }
      if (wgs > best) { mt = cm; ntw = cn; best = wgs; }
    }
    gz = (nt_total + 4 * ntw - 1) / (4 * ntw);
  }
  const int th = mt / twg;
  if (H % th) return PDES_ENOSUP;
  dim3 grid((W / (16 * twg)) * (H / th), d.B, gz), block(256);
  size_t lds = 0;
  int rc = PDES_ENOSUP;
  bool break_out = false;
#define PDES_TRY(TWG_, MT_, WK_, NTW_)                                                                       \
  if (twg == TWG_ && mt == MT_ && wk == WK_ && ntw == NTW_) {                                                 \
    using G = TileGeo<KS, TWG_, MT_, S>;                                                                      \
    const size_t cf_f = bwd ? 0 : 4 * (size_t)kpad;                                                          \
    const int ng = (WK_ == 4 && !bwd && S == 1 && nchunk >= 3) ? 2 : 1;                        \
    /* one chunk needs ONE tile buffer: half the LDS -> the 1024 workgroups of the 3 -> 49 channel 5x5 data gradient  \
       are resident at once instead of running a second, quarter-full round */                                       \
    size_t fl = cf_f + (size_t)ng * ((bwd && nchunk == 1) ? 1 : 2) * G::KC * G::CS;   /* (= the !PIPE instantiation) */ \
    const size_t red = cf_f + (size_t)4 * ng * MT_ * 4 * 64 + 256;                                            \
    if (WK_ == 4 && red > fl) fl = red;                                                                       \
    lds = fl * sizeof(float);                                                                                 \
    if (dry) {                                                                                                \
    } else if constexpr (MODE == MODE_FWD) {                                                                  \
      if constexpr (WK_ == 4 && S == 1) {                                                                     \
        if (ng == 2) {                                                                                        \
          hipLaunchKernelGGL((conv_mfma_kernel<KS, TWG_, MT_, S, WK_, NTW_, MODE, KM, true, 2>), grid, dim3(512), \
                             lds, st, d, wm, nt_total);                                                       \
          rc = PDES_OK;                                                                                       \
          break_out = true;                                                                                   \
        }                                                                                                     \
      }                                                                                                       \
      if (!break_out)                                                                                         \
      hipLaunchKernelGGL((conv_mfma_kernel<KS, TWG_, MT_, S, WK_, NTW_, MODE, KM, true, 1>), grid, block, lds, st, \
                         d, wm, nt_total);                                                                    \
    } else {                                                                                                  \
      if (nchunk > 1)                                                                                         \
        hipLaunchKernelGGL((conv_mfma_kernel<KS, TWG_, MT_, S, WK_, NTW_, MODE, KM, true, 1>), grid, block, lds, st, \
                           d, wm, nt_total);                                                                  \
      else if (d.g_fused) {                                                                                   \
        if constexpr (gf_able && WK_ == 1 && NTW_ == 1) {                                                     \
          hipEvent_t se = tl_stop_event;                                                                      \
          tl_stop_event = nullptr;                                                                            \
          constexpr bool tp_able = true;                                                                      \
          const int tp_min = (TWG_ == 2 && MT_ == 8) ? opt().dg_tilepipe : opt().dg_tilepipe16;              \
          if (tp_min > 0 && nC >= tp_min) {                                                                         \
            if (se)                                                                                           \
              hipExtLaunchKernelGGL((conv_mfma_kernel<KS, TWG_, MT_, S, WK_, NTW_, MODE, KM, false, 1, true, tp_able>), grid, block, lds, st, \
                                    nullptr, se, 0, d, wm, nt_total);                                         \
            else                                                                                              \
              hipLaunchKernelGGL((conv_mfma_kernel<KS, TWG_, MT_, S, WK_, NTW_, MODE, KM, false, 1, true, tp_able>), grid, block, lds, st, \
                                 d, wm, nt_total);                                                            \
          } else if (se)                                                                                      \
            hipExtLaunchKernelGGL((conv_mfma_kernel<KS, TWG_, MT_, S, WK_, NTW_, MODE, KM, false, 1, true>), grid, block, lds, st, \
                                  nullptr, se, 0, d, wm, nt_total);                                           \
          else                                                                                                \
            hipLaunchKernelGGL((conv_mfma_kernel<KS, TWG_, MT_, S, WK_, NTW_, MODE, KM, false, 1, true>), grid, block, lds, st, \
                               d, wm, nt_total);                                                              \
        } else                                                                                                \
          return PDES_ENOSUP;                                                                                 \
      } else {                                                                                                \
        constexpr bool tp_able2 = gf_able && WK_ == 1 && NTW_ == 1;                                           \
        const int tp_min2 = (TWG_ == 2 && MT_ == 8) ? opt().dg_tilepipe : opt().dg_tilepipe16;               \
        if (tp_able2 && tp_min2 > 0 && nC >= tp_min2)                                                                            \
          hipLaunchKernelGGL((conv_mfma_kernel<KS, TWG_, MT_, S, WK_, NTW_, MODE, KM, false, 1, false, tp_able2>), grid, block, lds, st, \
                             d, wm, nt_total);                                                                \
        else                                                                                                  \
          hipLaunchKernelGGL((conv_mfma_kernel<KS, TWG_, MT_, S, WK_, NTW_, MODE, KM, false, 1>), grid, block, lds, st, \
                             d, wm, nt_total);                                                                \
      }                                                                                                       \
    }                                                                                                         \
    rc = PDES_OK;                                                                                             \
  }
  if constexpr (S == 1) {
    PDES_TRY(2, 8, 4, 1) PDES_TRY(2, 8, 1, 1)
    PDES_TRY(1, 8, 4, 1) PDES_TRY(1, 8, 1, 1)
    if constexpr (KS != 5) {
      PDES_TRY(2, 8, 1, 2) PDES_TRY(1, 8, 1, 2)
      PDES_TRY(2, 4, 4, 1) PDES_TRY(2, 4, 1, 1) PDES_TRY(2, 4, 1, 2)
      PDES_TRY(1, 4, 4, 1) PDES_TRY(1, 4, 1, 1) PDES_TRY(1, 4, 1, 2)
      if constexpr (KS == 3 && MODE == MODE_FWD) { PDES_TRY(1, 2, 4, 1) }
    }
  } else {
    PDES_TRY(2, 8, 1, 1) PDES_TRY(2, 8, 1, 2) PDES_TRY(1, 8, 1, 1) PDES_TRY(1, 8, 1, 2)
    PDES_TRY(2, 4, 1, 1) PDES_TRY(2, 4, 1, 2) PDES_TRY(1, 4, 1, 1) PDES_TRY(1, 4, 1, 2)
  }
#undef PDES_TRY
  if (rc || dry) return rc;
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

// returns PDES_ENOSUP when the shape is not covered (caller falls back to the direct kernels)
int conv_forward_mfma(const pdes_conv_desc& d, hipStream_t st, bool dry) {
  if (d.nrep != PDES_NREP) return PDES_EINVAL;
  int W, H;
  if (!d.wm_fwd || !mfma_shape_ok(d, false, &W, &H) || d.Cin < 16) return PDES_ENOSUP;
  if (d.stride == 2) {
    if ((d.Cout + 15) / 16 == 1) return PDES_ENOSUP;
    return launch_mfma<3, 2, MODE_FWD, KV_PLAIN>(d, d.wm_fwd, W, H, st, dry);
  }
  if (d.ksize == 5) return launch_mfma<5, 1, MODE_FWD, KV_PLAIN>(d, d.wm_fwd, W, H, st, dry);
  return d.ksize == 3 ? launch_mfma<3, 1, MODE_FWD, KV_PLAIN>(d, d.wm_fwd, W, H, st, dry)
                      : launch_mfma<1, 1, MODE_FWD, KV_PLAIN>(d, d.wm_fwd, W, H, st, dry);
}

// dry = true: only report whether this implementation would take the descriptor (nothing is enqueued)
int conv_backward_data_mfma(const pdes_conv_desc& d, hipStream_t st, bool dry) {
  int W, H;
  if (!d.wm_bwd || !mfma_shape_ok(d, true, &W, &H) || d.eval_mode) return PDES_ENOSUP;
  // a stride-2 convolution's data gradient is the unit-stride gather over the zero-inserted dL/d(out)
  if (d.stride == 2) return launch_mfma<3, 1, MODE_BWD, KV_ZEROINS2>(d, d.wm_bwd, W, H, st, dry);
  if (d.ksize == 5) return launch_mfma<5, 1, MODE_BWD, KV_PLAIN>(d, d.wm_bwd, W, H, st, dry);
  return d.ksize == 3 ? launch_mfma<3, 1, MODE_BWD, KV_PLAIN>(d, d.wm_bwd, W, H, st, dry)
                      : launch_mfma<1, 1, MODE_BWD, KV_PLAIN>(d, d.wm_bwd, W, H, st, dry);
}

}  // namespace pdes

using namespace pdes;

#ifdef PDES_TRACE
extern "C" int pdes_debug_trace(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pdes_trace_buf), 16 * sizeof(unsigned long long));
}
#endif

extern "C" int pdes_pack_weights_mfma(const pdes_mfma_pack_item* items, int n, int max_elems, void* stream) {
  if (!items || n <= 0 || max_elems <= 0) return PDES_EINVAL;
  int gx = cdiv(max_elems, 256);
  gx = gx > 128 ? 128 : gx;
  hipLaunchKernelGGL(pack_mfma_kernel, dim3(gx, n), dim3(256), 0, static_cast<hipStream_t>(stream), items);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}
