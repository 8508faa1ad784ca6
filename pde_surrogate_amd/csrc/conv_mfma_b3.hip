// 3x3 stride-1 convolutions of the WIDE layers (forward and data gradient) on the bf16 matrix pipe with
// fp32 accuracy: every fp32 operand x is split into three bf16 terms
//     hi = bf16(x),  mid = bf16(x - hi),  lo = bf16(x - hi - mid)            (8 + 8 + 8 = 24 significant bits)
// and a product a*b is accumulated in fp32 from the six cross terms of weight >= 2^-16
//     am*bm + al*bh + ah*bl + am*bh + ah*bm + ah*bh                          (small terms first)
// with v_mfma_f32_16x16x32_bf16 (K = 32 channels per instruction, ~17 cycles) instead of v_mfma_f32_16x16x4_f32
// (K = 4, 32 cycles): 6 instructions replace 8 at about half their cost each.  Stand-alone measurement
// (tools/archive/proto/bf16x3_mfma.hip): 212 vs 104 fp32-equivalent TFLOP/s, rel-L2 error vs fp64 1.5e-7 vs 2.0e-7.
//
// GEMM roles as in conv_mfma.hip (M = 16 pixels of an image row, N = 16 output channels, K = input channels),
// but the LDS tile is pixel-major / channel-minor -- [plane hi|mid|lo][row][pixel][32 channels] bf16 -- so that the
// A operand of a lane (8 consecutive channels of one pixel) is ONE aligned ds_read_b128 and a wave reads one
// contiguous KiB.  Staging transposes on the way: a thread loads the same 4 pixels of 8 channels (8 float4),
// applies BatchNorm+ReLU (forward), splits, and writes 8 channels x 3 planes per pixel (ds_write_b128).
// The weights are split once per step by the pack kernel into the B-operand image
//     [chunk of 32 channels][tap][N-tile][plane][64 lanes][8 bf16].
// Workgroup = 256 threads = 4 waves, N-split (two N-tiles per wave), 8 M-tiles; chunks of 32 channels,
// double-buffered LDS, two register stages, straight-line main loop (see conv_mfma.hip for why).
// Reference: models/codec.py:163-175 (LastTransUp.conv1, 196 -> 98 channels at 32 x 32) and its autograd.
#include <stdlib.h>
#include "pdes_common.h"
#include "pdes_options.h"
#include "../../include/pdes_hip.h"
#include "pack_kernels.h"

namespace pdes {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

enum { B3_FWD = 0, B3_BWD = 1, B3_UPBWD = 2 };
// B3_UPBWD: data gradient of nearest-x2 upsampling + 3x3 (sub-pixel form, conv_mfma_up.hip): the K operand is the four
// parity sub-images G_p[Y][X] = g[2Y + dy][2X + dx] of the output gradient (each a low-resolution map, gathered with
// stride 2 on the way into LDS), a virtual chunk = (parity, 32 output channels), and each chunk uses the 4 taps of its
// parity's effective 2x2 kernel at tile offsets (2 - a - dy, 2 - b - dx) instead of 9:
//     dz[y][x] = sum_p sum_{a,b} Weff_p[a][b] . G_p[y - a - dy + 1][x - b - dx + 1]

template <int TWG, int MT_>
struct B3Geo {
  static constexpr int MT = MT_, TH = MT / TWG, TW = 16 * TWG;
  static constexpr int ROWS = TH + 2, PW = TW + 2, KC = 32;
  static constexpr int PLANE = ROWS * PW * KC;                 // bf16 elements of one plane of one buffer
  static constexpr int BUF = 3 * PLANE;
  static constexpr int QX = TW / 4;                            // pixel quads per row
  static constexpr int NI = ROWS * QX * 4;                     // interior work items (row, quad, channel octet)
  static constexpr int NHI = ROWS * 2 * 4;                     // halo work items (row, side, channel octet)
  static_assert(NI + NHI <= 256, "one work item per thread");
  // K tail on the f32 pipe: [4 channels][ROWS][FPW] floats behind the two bf16 buffers
  static constexpr int FPW = PW + 2, FCS = ROWS * FPW;
};

struct BnB { float mean, invstd, gamma, beta; };
__device__ __forceinline__ BnB bn_coef_b3(const pdes_conv_desc& d, int c, bool publish = false) {
  BnB o;
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

// grid: (tiles of the map, B, ceil(N-tiles / 8)); dynamic LDS: [kpad32] float4 coefficients (forward) + 2 buffers
template <int TWG, int MTP, int MODE, bool APIPE = false>
__global__ __launch_bounds__(256, MTP == 4 ? 2 : 1) void conv_mfma_b3_kernel(pdes_conv_desc d, const unsigned short* __restrict__ wb,
                                                          int nt_total, int tail_on_x) {
  const int tail_on = tail_on_x & 1;                     // (bit 1: XCD-aware workgroup order, below)
  using G = B3Geo<TWG, MTP>;
  constexpr int MT = G::MT, NT_W = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b3[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // (image, tile, N-tile group) of this workgroup.  As launched -- blockIdx = (tile, image, group) -- the tiles of one image,
  // which re-read each other's halo rows, go to eight different XCDs (workgroups are dealt round-robin by linear id, each XCD
  // has its own L2), and the N-tile groups, which re-read the WHOLE operand tile, run a full grid plane apart.  With bit 1 of
  // the last argument (PDES_XCD_MAP) an XCD takes whole images, and on it the groups of a tile run back to back, then the
  // next tile of the same image: the re-reads hit L2 instead of the Infinity Cache (round 6).
  int b = blockIdx.y, tile_id = blockIdx.x, zg = blockIdx.z;
  if ((tail_on_x & 2) && (gridDim.y & 7) == 0) {
    const int gx = gridDim.x, gz = gridDim.z;
    const int lin = blockIdx.x + gx * (blockIdx.y + (int)gridDim.y * blockIdx.z), j = lin >> 3;
    zg = j % gz;
    tile_id = (j / gz) % gx;
    b = (lin & 7) + 8 * (j / (gz * gx));
  }
  const int ntp = (nt_total + 7) & ~7;
  const int nt_base = (zg * 4 + wave) * NT_W;

  constexpr bool UPB = MODE == B3_UPBWD;
  constexpr int NTAP = UPB ? 4 : 9;
  const float* kbase;
  int kC, H, W;
  if (MODE == B3_FWD) {
    kC = d.Cin; H = d.Hin; W = d.Win;
    kbase = d.x + (size_t)b * d.x_ctot * H * W;
  } else if (MODE == B3_BWD) {
    kC = d.Cout; H = d.Hout; W = d.Wout;
    kbase = d.g + ((size_t)b * d.g_ctot + d.g_coff) * H * W;
  } else {                                    // the tiles are those of the LOW-resolution map
    kC = d.Cout; H = d.Hin; W = d.Win;
    kbase = d.g + ((size_t)b * d.g_ctot + d.g_coff) * 4 * H * W;
  }
  const int HW = UPB ? 4 * H * W : H * W;     // plane stride of the K operand
  const int nch1 = (kC + G::KC - 1) / G::KC, kpad = nch1 * G::KC;
  const int nchunk = UPB ? 4 * nch1 : nch1;   // virtual chunks: (parity, 32 channels)
  // K tail.  196 = 6 x 32 + 4 input channels (forward) and 98 = 3 x 32 + 2 gradient channels (data gradient): the last
  // 32-channel chunk would spend 9 taps x 6 MFMAs on 4 resp. 2 real channels (1/7 resp. 1/4 of the kernel's matrix cycles).
  // Up to four tail channels go through ONE v_mfma_f32_16x16x4_f32 per (tap, M-tile, N-tile) instead -- exact fp32, a
  // third of the cycles, no split -- from a small fp32 tile, the weights read straight from the (Cout, Cin, 3, 3) tensor.
  const int rtail = kC - G::KC * (nch1 - 1);
  const bool tail = !UPB && tail_on && nch1 >= 2 && rtail <= 4;
  const int nb = tail ? nchunk - 1 : nchunk;  // chunks on the bf16 pipe
  float4* cf4 = reinterpret_cast<float4*>(smem_b3);                                   // forward only
  unsigned short* tile = reinterpret_cast<unsigned short*>(smem_b3 + (MODE == B3_FWD ? 16 * kpad : 0));
  float* ftile = reinterpret_cast<float*>(tile + 2 * G::BUF);
  if (tail)
    for (int i = tid; i < 4 * G::FCS; i += 256) ftile[i] = 0.f;      // channels >= rtail stay zero (their weights are zero, 0 * NaN is not)
  const int tiles_x = W / G::TW;
  const int oy0 = (tile_id / tiles_x) * G::TH, ox0 = (tile_id % tiles_x) * G::TW;

  if (MODE == B3_FWD) {
    for (int c = tid; c < kpad; c += 256) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < d.Cin) { const BnB k = bn_coef_b3(d, c, (blockIdx.x | blockIdx.y | blockIdx.z) == 0); v = make_float4(k.mean, k.gamma * k.invstd, k.beta, 0.f); }
      cf4[c] = v;
    }
  }

  // ---- staging geometry: one work item per thread
  const bool interior = tid < G::NI, halo = !interior && tid < G::NI + G::NHI;
  int it_r, it_c, it_o;                 // tile row, first tile column, channel octet
  if (interior) { it_o = tid & 3; it_c = 1 + 4 * ((tid >> 2) % G::QX); it_r = tid / (4 * G::QX); }
  else { const int t = tid - G::NI; it_o = t & 3; it_c = ((t >> 2) & 1) ? G::PW - 1 : 0; it_r = (t >> 3) % G::ROWS; }
  const int gy = oy0 - 1 + it_r, gx = ox0 - 1 + it_c;
  const bool row_ok = gy >= 0 && gy < H;
  const bool px_ok = row_ok && (interior || (halo && gx >= 0 && gx < W));
  const int goff = UPB ? 4 * W * min(max(gy, 0), H - 1) + 2 * min(max(gx, 0), W - 1)       // + dy * 2W + dx per parity
                       : min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1);
  const int lds_off = (it_r * G::PW + it_c) * G::KC;                  // bf16 elements, plane 0, channel octet 0 of the pixel
  // LDS swizzle.  A pixel is 64 bytes (32 channels), an A fragment read is 16 consecutive pixels x one 16-byte octet per lane:
  // ds_read_b128 is serviced in four NON-contiguous 16-lane groups (MI355X_MICROARCH.md), and with the linear layout pixels
  // i and i + 12 (resp. i + 4) of one group fall on the same 16 banks -- a 2-way conflict on EVERY fragment read, 50-58 % of
  // the LDS cycles of this kernel in the SQ counters.  XOR-ing bit 1 of the octet index with bit 2 of the pixel's tile column
  // makes all four groups conflict free for every tap offset (exhaustive check in tools/archive/lds_swizzle_check.py); the writers
  // below use the same map.
  auto swz_oct = [](int col, int oct) __attribute__((always_inline)) { return oct ^ (((col >> 2) & 1) << 1); };

  struct Stage { float4 v[8]; };
  Stage sA, sB;
  auto issue = [&](int chunk, Stage& st) __attribute__((always_inline)) {
    const int cq = UPB ? chunk % nch1 : chunk, par = UPB ? chunk / nch1 : 0;
    const int c0 = cq * G::KC + 8 * it_o;
    const int poff = UPB ? (par >> 1) * 2 * W + (par & 1) : 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float* p = kbase + (size_t)min(c0 + j, kC - 1) * HW + goff + poff;
      if (UPB) {                                                     // every other pixel of the high-resolution row
        st.v[j].x = p[0];
        if (interior) { st.v[j].y = p[2]; st.v[j].z = p[4]; st.v[j].w = p[6]; }
      } else if (interior) st.v[j] = *reinterpret_cast<const float4*>(p);
      else st.v[j].x = *p;                                           // halo (and idle threads: a valid dummy load)
    }
  };
  auto commit = [&](int chunk, int buf, const Stage& st) __attribute__((always_inline)) {
    if (!(interior || halo)) return;
    const int c0 = (UPB ? chunk % nch1 : chunk) * G::KC + 8 * it_o;
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
        if (MODE == B3_FWD) {
          const float4 k = cf4[min(c0 + j, kpad - 1)];
          x = fmaxf(0.f, (x - k.x) * k.y + k.z);
        }
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
  auto commit_tail = [&](const Stage& st) __attribute__((always_inline)) {       // the tail chunk's <= 4 channels as fp32
    if (!(interior || halo) || it_o != 0) return;
    const int c0 = (nch1 - 1) * G::KC;
    float* t = ftile + it_r * G::FPW + it_c;
    const int npx = interior ? 4 : 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < rtail) {
        float xv[4] = {st.v[j].x, st.v[j].y, st.v[j].z, st.v[j].w};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          if (p >= npx) break;
          float x = xv[p];
          if (MODE == B3_FWD) {
            const float4 k = cf4[c0 + j];
            x = fmaxf(0.f, (x - k.x) * k.y + k.z);
          }
          t[j * G::FCS + p] = px_ok ? x : 0.f;
        }
      }
    }
  };

  v4f acc[MT][NT_W];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT_W; ++nt) acc[mt][nt] = (v4f){0.f, 0.f, 0.f, 0.f};

  // B operand: image [(chunk*9 + tap)*ntp + nt][plane][64 lanes][8 bf16]; one 16-byte load per (N-tile, plane)
  v8bf bA[3][NT_W], bB[3][NT_W];
  auto load_b = [&](int ct, v8bf (&dst)[3][NT_W]) __attribute__((always_inline)) {     // ct = chunk * 9 + tap (clamped)
    const int cc = min(ct, nchunk * NTAP - 1);
    const unsigned short* p = wb + (((size_t)cc * ntp + nt_base) * 3 * 64 + lane) * 8;
#pragma unroll
    for (int nt = 0; nt < NT_W; ++nt)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) dst[pl][nt] = *reinterpret_cast<const v8bf*>(p + ((size_t)nt * 3 + pl) * 64 * 8);
  };
  // pixel i = lane & 15, channel octet = lane >> 4 (swizzled by the pixel's tile column, which depends on the tap's kx)
  int a_lane_k[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) a_lane_k[kx] = (lane & 15) * G::KC + 8 * swz_oct(kx + (lane & 15), lane >> 4);
  auto a_lane_of = [&](int kx) __attribute__((always_inline)) { return kx == 0 ? a_lane_k[0] : (kx == 1 ? a_lane_k[1] : a_lane_k[2]); };
  auto mfma_tap = [&](const unsigned short* tb, int ky, int kx, const v8bf (&bw)[3][NT_W]) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const unsigned short* ap = tb + (((mt / TWG) + ky) * G::PW + (mt % TWG) * 16 + kx) * G::KC + a_lane_of(kx);
      const v8bf ah = *reinterpret_cast<const v8bf*>(ap);
      const v8bf am = *reinterpret_cast<const v8bf*>(ap + G::PLANE);
      const v8bf al = *reinterpret_cast<const v8bf*>(ap + 2 * G::PLANE);
      // six cross terms, smallest first; the two N-tiles alternate so that consecutive MFMAs are independent
#pragma unroll
      for (int nt = 0; nt < NT_W; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bw[1][nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
      for (int nt = 0; nt < NT_W; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bw[0][nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
      for (int nt = 0; nt < NT_W; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bw[2][nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
      for (int nt = 0; nt < NT_W; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bw[0][nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
      for (int nt = 0; nt < NT_W; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bw[1][nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
      for (int nt = 0; nt < NT_W; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bw[0][nt], acc[mt][nt], 0, 0, 0);
    }
  };

  // data gradient: the BatchNorm coefficients of the epilogue are fetched before the matrix loop
  BnB kepi[NT_W];
  if (MODE != B3_FWD) {
#pragma unroll
    for (int nt = 0; nt < NT_W; ++nt) kepi[nt] = bn_coef_b3(d, min((nt_base + nt) * 16 + (lane & 15), d.Cin - 1));
  }

  load_b(0, bA);
  issue(0, sA);
  issue(min(1, nchunk - 1), sB);
  __syncthreads();                   // coefficients visible
  commit(0, 0, sA);
  __syncthreads();

  // one chunk: 9 taps, the weights of the next tap stream into the other register set while this tap multiplies.
  // 9 is odd, so the roles of the two sets swap from chunk to chunk (b0 = set holding tap 0 of this chunk).
  // A-operand pipeline (PDES_B3_APIPE): the three fragments of the NEXT (tap, M-tile) are read from LDS before the twelve
  // MFMAs of the current one are issued (a scheduling barrier pins the order), so their ~100-cycle latency hides behind
  // 192 cycles of matrix work instead of stalling every few instructions (the compiler otherwise reads just in time)
  // tile offset (rows, columns) of tap t: the 3x3 position, or (sub-pixel data gradient) position (2 - a - dy, 2 - b - dx)
  // of tap (a, b) = (t >> 1, t & 1) of parity `par`
  auto tap_ky = [&](int t, int par) __attribute__((always_inline)) { return UPB ? 2 - (t >> 1) - (par >> 1) : t / 3; };
  auto tap_kx = [&](int t, int par) __attribute__((always_inline)) { return UPB ? 2 - (t & 1) - (par & 1) : t % 3; };
  auto lda = [&](const unsigned short* tb, int t, int mt, int par, v8bf (&a)[3]) __attribute__((always_inline)) {
    const int kx = tap_kx(t, par);
    const unsigned short* ap = tb + (((mt / TWG) + tap_ky(t, par)) * G::PW + (mt % TWG) * 16 + kx) * G::KC + a_lane_of(kx);
    a[0] = *reinterpret_cast<const v8bf*>(ap);
    a[1] = *reinterpret_cast<const v8bf*>(ap + G::PLANE);
    a[2] = *reinterpret_cast<const v8bf*>(ap + 2 * G::PLANE);
  };
  auto mfma12 = [&](int mt, const v8bf (&a)[3], const v8bf (&bw)[3][NT_W]) __attribute__((always_inline)) {
#pragma unroll
    for (int nt = 0; nt < NT_W; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], bw[1][nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < NT_W; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], bw[0][nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < NT_W; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bw[2][nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < NT_W; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], bw[0][nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < NT_W; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bw[1][nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < NT_W; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bw[0][nt], acc[mt][nt], 0, 0, 0);
  };
  // tail: B operand of v_mfma_f32_16x16x4_f32 = W[k = tail channel lane >> 4][n = lane & 15] of tap t, from the weight tensor
  float bft[9][NT_W];
  auto load_tail_b = [&]() __attribute__((always_inline)) {
    const int ch = lane >> 4, kc = kC - rtail + min(ch, rtail - 1);
#pragma unroll
    for (int nt = 0; nt < NT_W; ++nt) {
      const int n = (nt_base + nt) * 16 + (lane & 15);
      const int nC = MODE == B3_FWD ? d.Cout : d.Cin, nc = min(n, nC - 1);
      const bool ok = ch < rtail && n < nC;
      const float* wp = MODE == B3_FWD ? d.w + ((size_t)nc * d.Cin + kc) * 9 : d.w + ((size_t)kc * d.Cin + nc) * 9;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float v = wp[MODE == B3_FWD ? t : 8 - t];
        bft[t][nt] = ok ? v : 0.f;
      }
    }
  };
  auto tail_mma = [&]() __attribute__((always_inline)) {
    const float* fa = ftile + (lane >> 4) * G::FCS + (lane & 15);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float a = fa[((mt / TWG) + t / 3) * G::FPW + (mt % TWG) * 16 + t % 3];
#pragma unroll
        for (int nt = 0; nt < NT_W; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bft[t][nt], acc[mt][nt], 0, 0, 0);
      }
  };
  const bool apipe = APIPE;
  auto step = [&](int chunk, Stage& sfree, const Stage& snext, v8bf (&b0)[3][NT_W], v8bf (&b1)[3][NT_W])
      __attribute__((always_inline)) {
    const int buf = chunk & 1;
    const unsigned short* tb = tile + buf * G::BUF;
    const int par = UPB ? chunk / nch1 : 0;
    issue(min(chunk + 2, nchunk - 1), sfree);
    if (!UPB && tail && chunk + 1 == nb) load_tail_b();      // in flight during the last bf16 chunk
    if (apipe) {
      v8bf a0[3], a1[3];
      lda(tb, 0, 0, par, a0);
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        load_b(chunk * NTAP + t + 1, (t & 1) ? b0 : b1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int k = t * MT + mt;                       // flat index of this (tap, M-tile); the next one is k + 1
          if (k + 1 < NTAP * MT) lda(tb, (k + 1) / MT, (k + 1) % MT, par, (k & 1) ? a0 : a1);
          __builtin_amdgcn_sched_barrier(0);
          mfma12(mt, (k & 1) ? a1 : a0, (t & 1) ? b1 : b0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        load_b(chunk * NTAP + t + 1, (t & 1) ? b0 : b1);
        mfma_tap(tb, tap_ky(t, par), tap_kx(t, par), (t & 1) ? b1 : b0);
      }
    }
    if (chunk + 1 < nb) commit(chunk + 1, buf ^ 1, snext);
    else if (!UPB && tail) commit_tail(snext);               // (chunk + 1 == nb: the tail chunk's loads are in `snext`)
    __syncthreads();
  };
  {
    int chunk = 0;
    for (; chunk + 1 < nb; chunk += 2) {
      step(chunk, sA, sB, bA, bB);
      if constexpr (NTAP & 1) step(chunk + 1, sB, sA, bB, bA);       // an odd tap count swaps the roles of the weight sets
      else step(chunk + 1, sB, sA, bA, bB);
    }
    if (chunk < nb) step(chunk, sA, sB, bA, bB);
    if (!UPB && tail) tail_mma();
  }

  // ---- epilogues (accumulator layout = v_mfma_f32_16x16x4_f32's: col = lane & 15, rows (lane >> 4) * 4 + r)
  const int px = (lane >> 4) * 4;
  if (MODE == B3_FWD) {
    const int HWo = d.Hout * d.Wout;
#pragma unroll
    for (int nt = 0; nt < NT_W; ++nt) {
      const int co = (nt_base + nt) * 16 + (lane & 15);
      float s = 0.f, q = 0.f;
      if (co < d.Cout) {
        float* ob = d.out + ((size_t)b * d.out_ctot + d.out_coff + co) * HWo;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const v4f v = acc[mt][nt];
          const int oy = oy0 + mt / TWG, ox = ox0 + (mt % TWG) * 16 + px;
          *reinterpret_cast<float4*>(ob + (size_t)oy * d.Wout + ox) = make_float4(v[0], v[1], v[2], v[3]);
          s += (v[0] + v[1]) + (v[2] + v[3]);
          q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
      }
      if (d.out_stats) {
        s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
        q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
        if (lane < 16 && co < d.Cout) {
          double* os = d.out_stats + (long long)rep_of_block(d.nrep) * d.rep_stride;
#ifndef PDES_FW_NOATOM          // (component-timing build: EXPERIMENTS.md round 4)
          atomicAdd(&os[2 * (d.out_coff + co)], (double)s);
          atomicAdd(&os[2 * (d.out_coff + co) + 1], (double)q);
#endif
        }
      }
    }
  } else {
    const int HWi = d.Hin * d.Win;
    const float* xb = d.x + (size_t)b * d.x_ctot * HWi;
    float* tb2 = d.t_in + (size_t)b * d.x_ctot * HWi;
#pragma unroll
    for (int nt = 0; nt < NT_W; ++nt) {
      const int ci = (nt_base + nt) * 16 + (lane & 15);
      float dg = 0.f, db = 0.f, st = 0.f, sx = 0.f;
      if (ci < d.Cin) {
        const BnB k = kepi[nt];
        const float scale = k.gamma * k.invstd;
        const bool fin = ci >= d.final_c0 && ci < d.final_c1;
        float4 xq[MT], tq[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const size_t idx = (size_t)ci * HWi + (size_t)(oy0 + mt / TWG) * d.Win + ox0 + (mt % TWG) * 16 + px;
          xq[mt] = *reinterpret_cast<const float4*>(xb + idx);
          tq[mt] = d.t_accumulate ? *reinterpret_cast<const float4*>(tb2 + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const v4f v = acc[mt][nt];
          const size_t idx = (size_t)ci * HWi + (size_t)(oy0 + mt / TWG) * d.Win + ox0 + (mt % TWG) * 16 + px;
          const float xs[4] = {xq[mt].x, xq[mt].y, xq[mt].z, xq[mt].w};
          float ts[4] = {tq[mt].x, tq[mt].y, tq[mt].z, tq[mt].w};
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
      dg += __shfl_xor(dg, 16, 64); dg += __shfl_xor(dg, 32, 64);
      db += __shfl_xor(db, 16, 64); db += __shfl_xor(db, 32, 64);
      st += __shfl_xor(st, 16, 64); st += __shfl_xor(st, 32, 64);
      sx += __shfl_xor(sx, 16, 64); sx += __shfl_xor(sx, 32, 64);
      if (lane < 16 && ci < d.Cin) {
        const long long ro = (long long)rep_of_block(d.nrep) * d.rep_stride;
        atomicAdd(&d.bn_grad[ro + 2 * ci], (double)dg);
        atomicAdd(&d.bn_grad[ro + 2 * ci + 1], (double)db);
        if (ci >= d.final_c0 && ci < d.final_c1) {
          atomicAdd(&d.t_stats[ro + 2 * ci], (double)st);
          atomicAdd(&d.t_stats[ro + 2 * ci + 1], (double)sx);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// split weight images, rebuilt from the live weights every step:
//   forward  [(chunk*9 + tap)*NT + nt][plane][lane][j] = split_plane( W[co = 16 nt + (lane&15)][ci = 32 chunk + 8 (lane>>4) + j][tap] )
//   backward [(chunk*9 + tap)*NT + nt][plane][lane][j] = split_plane( W[co = 32 chunk + 8 (lane>>4) + j][ci = 16 nt + (lane&15)][8 - tap] )
// NT = N-tile count rounded up to a multiple of 8 (zero tiles), chunks of 32 K-channels (zero beyond the tensor).
__global__ __launch_bounds__(256) void pack_b3_kernel(const pdes_b3_pack_item* __restrict__ items) {
  pack_b3_item(items[blockIdx.y], blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------- host dispatch
static bool b3_enabled() {
  return (opt().mfma_b3 & 1) != 0;
}

// the layers this kernel takes: 3x3, stride 1, no upsampling, wide on both sides of the contraction
static bool b3_shape_ok(const pdes_conv_desc& d, bool bwd) {
  if (d.ksize != 3 || d.stride != 1 || d.pad != 1 || d.upsample || !d.has_bn || d.nrep != PDES_NREP) return false;
  if (d.Hin != d.Hout || d.Win != d.Wout) return false;
  const int kC = bwd ? d.Cout : d.Cin, nC = bwd ? d.Cin : d.Cout;
  if (kC < 64 || nC < 80) return false;
  const int W = d.Win, H = d.Hin;
  if (W % 16 || (W >= 32 && W % 32)) return false;
  return H % (W >= 32 ? 4 : 8) == 0;
}

template <int MODE>
static int launch_b3(const pdes_conv_desc& d, const unsigned short* wb, hipStream_t st) {
  const bool bwd = MODE != B3_FWD;
  const int kC = bwd ? d.Cout : d.Cin, nC = bwd ? d.Cin : d.Cout;
  const int nchunk = (kC + 31) / 32, kpad = nchunk * 32, nt_total = (nC + 15) / 16;
  const int W = d.Win, H = d.Hin, twg = W >= 32 ? 2 : 1;
  // 8 M-tiles per workgroup (82 KB of LDS: one workgroup per CU) or 4 (52 KB: two to three per CU, whose staging
  // and matrix phases overlap)
  int mt = 4;
  if (H % (mt / twg)) mt = 8;
  dim3 grid((W / (16 * twg)) * (H / (mt / twg)), d.B, (nt_total + 7) / 8), block(256);
  const size_t cf = bwd ? 0 : 16 * (size_t)kpad;
#define PDES_B3_LAUNCH(TWG_, MT_)                                                                             \
  do {                                                                                                        \
    using GL = B3Geo<TWG_, MT_>;                                                                              \
    const size_t lds = cf + 2 * (size_t)GL::BUF * 2 + 4 * (size_t)GL::FCS * sizeof(float);                    \
    const int tail_on = ((opt().b3_tail && d.w) ? 1 : 0) | (opt().xcd_map ? 2 : 0);                                                     \
    if (MT_ == 4)       /* A-operand fragments of the next (tap, M-tile) read before this one's MFMAs */     \
      hipLaunchKernelGGL((conv_mfma_b3_kernel<TWG_, MT_, MODE, true>), grid, block, lds, st, d, wb, nt_total, tail_on);  \
    else                                                                                                      \
      hipLaunchKernelGGL((conv_mfma_b3_kernel<TWG_, MT_, MODE, false>), grid, block, lds, st, d, wb, nt_total, tail_on); \
  } while (0)
  if (twg == 2) { if (mt == 8) PDES_B3_LAUNCH(2, 8); else PDES_B3_LAUNCH(2, 4); }
  else { if (mt == 8) PDES_B3_LAUNCH(1, 8); else PDES_B3_LAUNCH(1, 4); }
#undef PDES_B3_LAUNCH
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

int conv_forward_b3(const pdes_conv_desc& d, hipStream_t st, bool dry) {
  if (!b3_enabled() || !d.wb_fwd || !b3_shape_ok(d, false)) return PDES_ENOSUP;
  if (dry) return PDES_OK;
  return launch_b3<B3_FWD>(d, d.wb_fwd, st);
}

// data gradient of nearest-x2 + 3x3 in the sub-pixel form (B3_UPBWD)
int conv_backward_data_b3_up(const pdes_conv_desc& d, hipStream_t st, bool dry) {
  if (!(opt().mfma_b3 & 8) || !d.wbu_bwd || d.eval_mode || d.g_fused) return PDES_ENOSUP;
  if (d.ksize != 3 || d.stride != 1 || d.pad != 1 || d.upsample != PDES_UPSAMPLE_NEAREST || !d.has_bn || d.nrep != PDES_NREP)
    return PDES_ENOSUP;
  if (d.Hout != 2 * d.Hin || d.Wout != 2 * d.Win || d.Cout < 32 || d.Cin < 64) return PDES_ENOSUP;
  const int W = d.Win, H = d.Hin;
  if (W % 16 || (W >= 32 && W % 32) || H % (W >= 32 ? 4 : 8)) return PDES_ENOSUP;
  if (dry) return PDES_OK;
  return launch_b3<B3_UPBWD>(d, d.wbu_bwd, st);
}

// dry = true: only report whether this implementation would take the descriptor
int conv_backward_data_b3(const pdes_conv_desc& d, hipStream_t st, bool dry) {
  if (!b3_enabled() || !d.wb_bwd || !b3_shape_ok(d, true) || d.eval_mode || d.g_fused) return PDES_ENOSUP;
  if (dry) return PDES_OK;
  return launch_b3<B3_BWD>(d, d.wb_bwd, st);
}

}  // namespace pdes

using namespace pdes;

extern "C" int pdes_pack_weights_b3(const pdes_b3_pack_item* items, int n, int max_elems, void* stream) {
  if (!items || n <= 0 || max_elems <= 0) return PDES_EINVAL;
  int gx = cdiv(max_elems, 256);
  gx = gx > 256 ? 256 : gx;
  hipLaunchKernelGGL(pack_b3_kernel, dim3(gx, n), dim3(256), 0, static_cast<hipStream_t>(stream), items);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

// floats... sizes of the two images of one layer, in bf16 elements (the caller allocates 16-byte aligned buffers)
extern "C" int pdes_b3_image_elems(int Cout, int Cin, long long* fwd_elems, long long* bwd_elems) {
  if (Cout <= 0 || Cin <= 0 || !fwd_elems || !bwd_elems) return PDES_EINVAL;
  const long long ntf = (((Cout + 15) / 16) + 7) & ~7, ntb = (((Cin + 15) / 16) + 7) & ~7;
  *fwd_elems = (long long)((Cin + 31) / 32) * 9 * ntf * 3 * 64 * 8;
  *bwd_elems = (long long)((Cout + 31) / 32) * 9 * ntb * 3 * 64 * 8;
  return PDES_OK;
}
