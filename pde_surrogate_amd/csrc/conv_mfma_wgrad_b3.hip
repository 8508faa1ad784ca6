// Weight gradient of the WIDE 3x3 stride-1 convolutions on the bf16 matrix pipe with fp32 accuracy (three-way bf16
// split of both operands, six cross products -- see conv_mfma_b3.hip for the arithmetic and its adversarial test):
//     dW[co][ci][ky][kx] = sum_{b, y, x} g[b][co][y][x] * z[b][ci][y + ky - 1][x + kx - 1],   z = relu(bn(x))
// (autograd of F.conv2d wrt its weight, reference models/codec.py:169-170 LastTransUp.conv1: 196 -> 98 at 32 x 32 --
// with 160 us the largest single kernel of the step on the f32 pipe).
//
// GEMM roles per v_mfma_f32_16x16x32_bf16: M = 16 input channels (A = z), N = 16 output channels (B = g),
// K = 32 consecutive pixels = ONE ROW of a 32-wide map.  Both operands are pixel-contiguous in NCHW, so a lane's
// 8 k-elements are 8 consecutive pixels of one channel: the LDS images are [plane hi|mid|lo][row][channel][32 px]
// bf16 and a fragment is one aligned ds_read_b128 per lane (a wave reads one contiguous KiB, conflict free).
// The kx shift of z would misalign those reads, so z is kept in THREE column-shifted copies (z[x-1], z[x], z[x+1],
// zero beyond the row), built at staging time from the neighbour lanes' packed words; the ky shift is a row of the
// 3-slot ring the workgroup slides down the image.  g needs no shift: its three planes of one row are loaded into
// registers once per row and reused by all 9 taps.
// Workgroup = 256 threads = 4 waves: ONE 16-channel M-tile, ALL N-tiles (wave w owns tiles w and w + 4), a run of rows
// of one image; per row 9 taps x 2 tiles x 6 = 108 MFMAs per wave against 27 + 6 fragment reads.  71 KB of LDS: two
// workgroups per CU.  The (co, ci, tap) block of partial sums goes through LDS to the split-K scratch in runs of
// 16 x 9 contiguous floats; the fixed-order reduce is the one of conv_mfma_wgrad.hip (deterministic).
#include "pdes_common.h"
#include "pdes_options.h"
#include "../../include/pdes_hip.h"
#include "pack_kernels.h"

namespace pdes {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

namespace wb3 {
constexpr int W = 32;                      // map width = pixels per k-step
constexpr int ZROW = 16 * W;               // bf16 elements of one ring slot (16 channels x 32 pixels)
constexpr int ZPLANE = 3 * ZROW;           // 3 ring slots
constexpr int ZCOPY = 3 * ZPLANE;          // 3 planes
constexpr int ZSIZE = 3 * ZCOPY;           // 3 column-shifted copies: 13,824 elements = 27,648 B
// A [16 channels][32 pixels] bf16 tile is 16 rows of 64 B = four 16-B slots (one per MFMA k-group).  ds_read_b128 is
// serviced in four NON-contiguous 16-lane groups (MI355X_MICROARCH.md, LDS: {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...)
// against 64 banks = 16 slots: with the linear layout rows r and r + 12 of one k-group share a slot (2-way conflict on
// every fragment read -- the kernel was LDS bound at twice the conflict-free time).  XOR-ing the slot with
// F[row >> 2], F = {0, 3, 2, 1}, makes the 16 lanes of every group hit 16 distinct slots; writers use the same map.
__device__ __forceinline__ int swz(int row, int kg) { return kg ^ ((0x6C >> (2 * (row >> 2))) & 3); }   // F packed: 0,3,2,1
}  // namespace wb3

// grid: (B * H / rows, M-tiles); dynamic LDS: Z + G[2][3][nco][32] (bf16), reused for the [nco][16][9] fp32 epilogue
// PF: rows of lead of the operand loads (1 or 2 register stages)
template <int PF>
__global__ __launch_bounds__(256, 2) void conv_wgrad_b3_kernel(pdes_conv_desc d, float* __restrict__ part, int rows) {
  using namespace wb3;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_wb3[];
  unsigned short* Z = reinterpret_cast<unsigned short*>(smem_wb3);            // [kx][plane][slot][16][32]
  unsigned short* G = Z + ZSIZE;                                              // [buf][plane][nco][32]
  __shared__ float cf[16][3];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = d.Hout, HW = H * W;
  const int ntiles = (d.Cout + 15) >> 4, nco = ntiles * 16;
  const int hs_n = H / rows;
  const int b = blockIdx.x / hs_n, y0 = (blockIdx.x % hs_n) * rows;
  const int ci0 = blockIdx.y * 16;
  const long long per = (long long)d.Cout * d.Cin * 9;

  if (tid < 16) {                         // BatchNorm coefficients of this workgroup's 16 input channels
    const int c = ci0 + tid;
    float m = 0.f, s = 0.f, bt = 0.f;
    if (c < d.Cin) {
      double mean, invstd;
      if (d.eval_mode) { mean = d.run_mean[c]; invstd = 1.0 / sqrt((double)d.run_var[c] + (double)d.eps); }
      else {           // (the table entry, or the replica sums; published by the workgroups of the first pixel split)
        const MeanInv mi = batch_mean_invstd(d.coef, d.x_stats, d.rep_stride, (double)d.B * HW, d.eps, c, blockIdx.x == 0);
        mean = mi.mean; invstd = mi.invstd;
      }
      m = (float)mean; s = d.gamma[c] * (float)invstd; bt = d.beta[c];
    }
    cf[tid][0] = m; cf[tid][1] = s; cf[tid][2] = bt;
  }

  const float* xb = d.x + ((size_t)b * d.x_ctot + ci0) * HW;
  const float* gb = d.g + ((size_t)b * d.g_ctot + d.g_coff) * HW;
  const int crem = d.Cin - ci0;
  // staging roles: threads 0..127 own one float4 of the z row (channel zc, columns 4 zj ..); everyone owns up to four
  // float4 of the g row (channel e >> 3, columns 4 (e & 7) ..)
  const bool zt = tid < 128;
  const int zc = (tid >> 3) & 15, zj = tid & 7;
  const int ng4 = nco * 8;
  struct Stage { float4 z; float4 g[4]; };
  Stage s;
  // raw loads only (clamped addresses): validity is applied at commit, nothing consumes a load right after its issue
  auto issue = [&](int y, Stage& st) __attribute__((always_inline)) {
    const int zr = min(max(y + 1, 0), H - 1), gr = min(max(y, 0), H - 1);
    st.z = *reinterpret_cast<const float4*>(xb + (size_t)min(zc, crem - 1) * HW + zr * W + 4 * zj);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = min(tid + 256 * i, ng4 - 1);
      const int co = min(e >> 3, d.Cout - 1);
      st.g[i] = *reinterpret_cast<const float4*>(gb + (size_t)co * HW + gr * W + 4 * (e & 7));
    }
  };
  auto commit = [&](int y, const Stage& st) __attribute__((always_inline)) {
    if (zt) {                              // z row y + 1 -> ring slot, three column-shifted copies
      const int row = y + 1;
      const bool ok = row >= 0 && row < H && zc < crem;
      const float mean = cf[zc][0], sc = cf[zc][1], bt = cf[zc][2];
      float v[4] = {st.z.x, st.z.y, st.z.z, st.z.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = ok ? fmaxf(0.f, (v[i] - mean) * sc + bt) : 0.f;
      u32 w0[3], w1[3];
      split3_pair(v[0], v[1], w0[0], w0[1], w0[2]);        // w?[plane]: pixels (0,1) and (2,3) of this quad
      split3_pair(v[2], v[3], w1[0], w1[1], w1[2]);
      const int slot = (row + 3) % 3;
      unsigned short* zp = Z + slot * ZROW + zc * W + 8 * swz(zc, zj >> 1) + 4 * (zj & 1);
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        u32 prev1 = __shfl_up(w1[p], 1, 64), next0 = __shfl_down(w0[p], 1, 64);
        if (zj == 0) prev1 = 0u;           // zero padding left of column 0 / right of column 31
        if (zj == 7) next0 = 0u;
        const u32 mid = (w0[p] >> 16) | (w1[p] << 16);     // pixels (1,2)
        unsigned short* q = zp + p * ZPLANE;
        *reinterpret_cast<uint2*>(q) = make_uint2((prev1 >> 16) | (w0[p] << 16), mid);              // kx = 0: z[x - 1]
        *reinterpret_cast<uint2*>(q + ZCOPY) = make_uint2(w0[p], w1[p]);                             // kx = 1: z[x]
        *reinterpret_cast<uint2*>(q + 2 * ZCOPY) = make_uint2(mid, (w1[p] >> 16) | (next0 << 16));  // kx = 2: z[x + 1]
      }
    }
    unsigned short* gbuf = G + (y & 1) * 3 * nco * W;      // g row y -> buffer y & 1
    const bool rok = y >= 0 && y < H;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i;
      if (e < ng4) {
        const int co = e >> 3;
        const bool ok = rok && co < d.Cout;
        u32 a[3], c2[3];
        split3_pair(ok ? st.g[i].x : 0.f, ok ? st.g[i].y : 0.f, a[0], a[1], a[2]);
        split3_pair(ok ? st.g[i].z : 0.f, ok ? st.g[i].w : 0.f, c2[0], c2[1], c2[2]);
        unsigned short* q = gbuf + co * W + 8 * swz(co & 15, (e & 7) >> 1) + 4 * (e & 1);
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(q + p * nco * W) = make_uint2(a[p], c2[p]);
      }
    }
  };

  v4f acc[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[t][nt] = (v4f){0.f, 0.f, 0.f, 0.f};

  __syncthreads();                          // coefficients visible
  issue(y0 - 2, s); commit(y0 - 2, s);      // z row y0 - 1
  issue(y0 - 1, s); commit(y0 - 1, s);      // z row y0
  const int frag = (lane & 15) * W + 8 * swz(lane & 15, lane >> 4);     // a lane's 8 pixels inside a [16][32] tile
  const int ylast = y0 + rows - 1;
  // the matrix work of row y: 108 MFMAs per wave
  auto mma = [&](int y) __attribute__((always_inline)) {
    const unsigned short* gbuf = G + (y & 1) * 3 * nco * W;
    v8bf bh[2], bm[2], bl[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int tile = min(wave + 4 * nt, ntiles - 1);
      const unsigned short* q = gbuf + tile * 16 * W + frag;
      bh[nt] = *reinterpret_cast<const v8bf*>(q);
      bm[nt] = *reinterpret_cast<const v8bf*>(q + nco * W);
      bl[nt] = *reinterpret_cast<const v8bf*>(q + 2 * nco * W);
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int slot = (y + ky + 2) % 3;    // row y - 1 + ky
      // the three kx taps of this kernel row together: six independent accumulators (3 taps x 2 N-tiles) per cross
      // term, so no MFMA waits for the one before it on the same accumulator
      v8bf ah[3], am[3], al[3];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const unsigned short* zp = Z + kx * ZCOPY + slot * ZROW + frag;
        ah[kx] = *reinterpret_cast<const v8bf*>(zp);
        am[kx] = *reinterpret_cast<const v8bf*>(zp + ZPLANE);
        al[kx] = *reinterpret_cast<const v8bf*>(zp + 2 * ZPLANE);
      }
#define PDES_WB3_TERM(A_, B_)                                                                                   \
      _Pragma("unroll") for (int kx = 0; kx < 3; ++kx)                                                          \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                        \
          acc[ky * 3 + kx][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A_[kx], B_[nt], acc[ky * 3 + kx][nt], 0, 0, 0)
      PDES_WB3_TERM(am, bm);                // six cross terms, smallest first
      PDES_WB3_TERM(al, bh);
      PDES_WB3_TERM(ah, bl);
      PDES_WB3_TERM(am, bh);
      PDES_WB3_TERM(ah, bm);
      PDES_WB3_TERM(ah, bh);
#undef PDES_WB3_TERM
    }
  };
  // one row: commit its operands (loaded PF rows ago), start the loads of row y + PF into the freed registers, matrix
  // work.  Two rows of lead instead of one: 119 -> 117 us (the kernel is not latency bound: see DESIGN.md, matrix and
  // vector instructions of one SIMD do not overlap, and this kernel issues about as many cycles of each).
  auto row = [&](int y, Stage& st) __attribute__((always_inline)) {
    __syncthreads();                        // the fragment reads of row y - 1 are done: its oldest slot may be overwritten
    commit(y, st);                          // z row y + 1, g row y
    __syncthreads();
    issue(min(y + PF, ylast), st);          // in flight during the matrix work of PF rows
    __builtin_amdgcn_sched_barrier(0);      // (without it the scheduler sinks these loads to their use and every row
                                            //  pays the full memory latency)
    mma(y);
  };
  if constexpr (PF == 1) {
    issue(y0, s);
    for (int y = y0; y <= ylast; ++y) row(y, s);
  } else {
    Stage s2;
    issue(y0, s);
    issue(min(y0 + 1, ylast), s2);
    for (int y = y0; y <= ylast; y += 2) {
      row(y, s);
      if (y + 1 <= ylast) row(y + 1, s2);
    }
  }

  // ---- epilogue: accumulator tile = D[ci = (lane >> 4) * 4 + r][co = lane & 15]; through LDS as [co][ci][tap]
  __syncthreads();
  float* outl = reinterpret_cast<float*>(smem_wb3);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int tile = wave + 4 * nt;
    if (tile < ntiles) {
      const int co = tile * 16 + (lane & 15);
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) outl[(co * 16 + (lane >> 4) * 4 + r) * 9 + t] = acc[t][nt][r];
    }
  }
  __syncthreads();
  float* pb = part + (size_t)blockIdx.x * per;
  const int run = min(16, crem) * 9;                       // contiguous floats per output channel
  for (int e = tid; e < d.Cout * run; e += 256) {
    const int co = e / run, k = e - co * run;
    pb[((size_t)co * d.Cin + ci0) * 9 + k] = outl[co * 144 + k];
  }
}

// ------------------------------------------------------------------------------------------------
// The same kernel for nearest-x2 + 3x3 (reference models/codec.py:130-150, :172-176 LastTransUp.conv2: 98 -> 49 at
// 32 x 32 -> 64 x 64) in SUB-PIXEL form (the arithmetic of conv_mfma_wgrad_up_kernel, conv_mfma_wgrad.hip):
//     dWeff[p][a][b][co][ci] = sum_{b, y, x} G_p[co][y][x] * z[ci][y + a + py - 1][x + b + px - 1],   G_p[y][x] = g[2y + py][2x + px]
//     dW[ky][kx] = sum_{py, px} dWeff[(py, px)][a(ky, py)][b(kx, px)]
// 16 (parity, a, b) products on the LOW-res pixels instead of 9 taps on four times as many (4/9 of the flops), at the
// bf16 rate.  K = 32 low-res pixels = one row of z; the z ring and its three column-shifted copies are unchanged (row
// shift a + py - 1, column shift b + px - 1, both in {-1, 0, +1}); the two hi-res rows 2y, 2y + 1 of g are de-interleaved
// into the four parity rows on the way into LDS (a thread owns 8 hi-res columns of one row = 4 pixels of two parities).
// One N-tile per wave (64 output channels per workgroup, the rest over gridDim.z): 16 accumulators, 96 MFMAs per wave and
// row.  LDS: Z 27 KB + G [4][3][64][32] 48 KB, G single-buffered (both barriers of a row lie between its write and the
// next one): two workgroups per CU.
// WL = 16 (TransUp1.conv2: 100 -> 100 at 16 x 16 -> 32 x 32): a k-row is a PAIR of map rows (32 contiguous pixels); the
// column-shifted copies are zero padded at both 16-pixel boundaries, and a row shift moves a lane's 8-pixel group by two
// groups -- into the neighbouring k-row's ring slot for half of the lanes (still one aligned 16-byte read).
template <int WL>
__global__ __launch_bounds__(256, 2) void conv_wgrad_b3_up_kernel(pdes_conv_desc d, float* __restrict__ part, int rows) {
  using namespace wb3;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_wb3[];
  unsigned short* Z = reinterpret_cast<unsigned short*>(smem_wb3);            // [kx][plane][slot][16][32]
  unsigned short* G = Z + ZSIZE;                                              // [parity][plane][nco][32]
  __shared__ float cf[16][3];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = d.Hin * WL / W, HW = H * W, WH = 2 * WL, HWh = 4 * HW;         // k-rows (32 low-res pixels each), hi-res width / map
  const int co0 = blockIdx.z * 64, corem = d.Cout - co0;                      // this workgroup's output channels
  const int ntiles = min((corem + 15) >> 4, 4), nco = ntiles * 16;
  const int hs_n = H / rows;
  const int b = blockIdx.x / hs_n, y0 = (blockIdx.x % hs_n) * rows;
  const int ci0 = blockIdx.y * 16;
  const long long per = (long long)d.Cout * d.Cin * 9;

  if (tid < 16) {                         // BatchNorm coefficients of this workgroup's 16 input channels
    const int c = ci0 + tid;
    float m = 0.f, s = 0.f, bt = 0.f;
    if (c < d.Cin) {
      double mean, invstd;
      if (d.eval_mode) { mean = d.run_mean[c]; invstd = 1.0 / sqrt((double)d.run_var[c] + (double)d.eps); }
      else {           // (the table entry, or the replica sums; published by the workgroups of the first pixel split)
        const MeanInv mi = batch_mean_invstd(d.coef, d.x_stats, d.rep_stride, (double)d.B * HW, d.eps, c, blockIdx.x == 0);
        mean = mi.mean; invstd = mi.invstd;
      }
      m = (float)mean; s = d.gamma[c] * (float)invstd; bt = d.beta[c];
    }
    cf[tid][0] = m; cf[tid][1] = s; cf[tid][2] = bt;
  }

  const float* xb = d.x + ((size_t)b * d.x_ctot + ci0) * HW;
  const float* gb = d.g + ((size_t)b * d.g_ctot + d.g_coff + co0) * HWh;
  const int crem = d.Cin - ci0;
  // staging roles: threads 0..127 own one float4 of the z row; everyone owns up to four items of the g rows, an item =
  // (channel e >> 4, row parity (e >> 3) & 1, low-res pixels 4 (e & 7) .. of the k-row = 8 hi-res columns of one hi-res row)
  const bool zt = tid < 128;
  const int zc = (tid >> 3) & 15, zj = tid & 7;
  const int ng = nco * 16;
  struct Stage { float4 z; float4 g[4][2]; };
  Stage s;
  auto issue = [&](int y, Stage& st) __attribute__((always_inline)) {
    const int zr = min(max(y + 1, 0), H - 1), gr = min(max(y, 0), H - 1);
    st.z = *reinterpret_cast<const float4*>(xb + (size_t)min(zc, crem - 1) * HW + zr * W + 4 * zj);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = min(tid + 256 * i, ng - 1);
      const int co = min(e >> 4, corem - 1), py = (e >> 3) & 1, jj = e & 7;          // jj: the item's 4 low-res pixels 4 jj ..
      const float* src = WL == 32 ? gb + (size_t)co * HWh + (2 * gr + py) * WH + 8 * jj
                                  : gb + (size_t)co * HWh + (4 * gr + 2 * (jj >> 2) + py) * WH + 8 * (jj & 3);
      st.g[i][0] = *reinterpret_cast<const float4*>(src);
      st.g[i][1] = *reinterpret_cast<const float4*>(src + 4);
    }
  };
  auto commit = [&](int y, const Stage& st, bool with_g) __attribute__((always_inline)) {
    if (zt) {                              // z row y + 1 -> ring slot, three column-shifted copies
      const int row = y + 1;
      const bool ok = row >= 0 && row < H && zc < crem;
      const float mean = cf[zc][0], sc = cf[zc][1], bt = cf[zc][2];
      float v[4] = {st.z.x, st.z.y, st.z.z, st.z.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = ok ? fmaxf(0.f, (v[i] - mean) * sc + bt) : 0.f;
      u32 w0[3], w1[3];
      split3_pair(v[0], v[1], w0[0], w0[1], w0[2]);
      split3_pair(v[2], v[3], w1[0], w1[1], w1[2]);
      const int slot = (row + 3) % 3;
      unsigned short* zp = Z + slot * ZROW + zc * W + 8 * swz(zc, zj >> 1) + 4 * (zj & 1);
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        u32 prev1 = __shfl_up(w1[p], 1, 64), next0 = __shfl_down(w0[p], 1, 64);
        constexpr int ZM = WL / 4 - 1;     // float4 per map row - 1: zero padding left / right of every map row
        if ((zj & ZM) == 0) prev1 = 0u;
        if ((zj & ZM) == ZM) next0 = 0u;
        const u32 mid = (w0[p] >> 16) | (w1[p] << 16);
        unsigned short* q = zp + p * ZPLANE;
        *reinterpret_cast<uint2*>(q) = make_uint2((prev1 >> 16) | (w0[p] << 16), mid);              // kx = 0: z[x - 1]
        *reinterpret_cast<uint2*>(q + ZCOPY) = make_uint2(w0[p], w1[p]);                             // kx = 1: z[x]
        *reinterpret_cast<uint2*>(q + 2 * ZCOPY) = make_uint2(mid, (w1[p] >> 16) | (next0 << 16));  // kx = 2: z[x + 1]
      }
    }
    if (!with_g) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i;
      if (e < ng) {
        const int co = e >> 4, py = (e >> 3) & 1, jj = e & 7;
        const bool ok = co < corem;
        const float4 a = st.g[i][0], c = st.g[i][1];
        unsigned short* q = G + (size_t)(py * 2) * 3 * nco * W + co * W + 8 * swz(co & 15, jj >> 1) + 4 * (jj & 1);
        u32 h0[3], h1[3];
        split3_pair(ok ? a.x : 0.f, ok ? a.z : 0.f, h0[0], h0[1], h0[2]);      // px = 0: hi-res columns 0, 2 | 4, 6
        split3_pair(ok ? c.x : 0.f, ok ? c.z : 0.f, h1[0], h1[1], h1[2]);
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(q + p * nco * W) = make_uint2(h0[p], h1[p]);
        split3_pair(ok ? a.y : 0.f, ok ? a.w : 0.f, h0[0], h0[1], h0[2]);      // px = 1: hi-res columns 1, 3 | 5, 7
        split3_pair(ok ? c.y : 0.f, ok ? c.w : 0.f, h1[0], h1[1], h1[2]);
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(q + (3 + p) * nco * W) = make_uint2(h0[p], h1[p]);
      }
    }
  };

  v4f acc[16];                              // [(py, px)][a][b]
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};

  __syncthreads();                          // coefficients visible
  issue(y0 - 2, s); commit(y0 - 2, s, false);      // z row y0 - 1
  issue(y0 - 1, s); commit(y0 - 1, s, false);      // z row y0
  issue(y0, s);
  const int frag = (lane & 15) * W + 8 * swz(lane & 15, lane >> 4);
  // WL = 16: the A fragment of row shift ky - 1 is the lane's group moved by 2 (ky - 1) groups: afrag[ky] inside a slot,
  // adk[ky] = k-row offset of that slot (-1, 0, +1)
  int afrag[3], adk[3];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int g2 = (lane >> 4) + (WL == 16 ? 2 * (ky - 1) : 0);
    adk[ky] = WL == 16 ? (g2 < 0 ? -1 : (g2 > 3 ? 1 : 0)) : ky - 1;
    afrag[ky] = (lane & 15) * W + 8 * swz(lane & 15, g2 & 3);
  }
  const int tile = min(wave, ntiles - 1);
  const int ylast = y0 + rows - 1;
  for (int y = y0; y <= ylast; ++y) {
    __syncthreads();                        // the fragment reads of row y - 1 are done
    commit(y, s, true);                     // z row y + 1, the four parity rows of the hi-res rows under k-row y
    __syncthreads();
    issue(min(y + 1, ylast), s);            // in flight during the matrix work below
    __builtin_amdgcn_sched_barrier(0);
    v8bf bh[4], bm[4], bl[4];
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const unsigned short* q = G + (size_t)pp * 3 * nco * W + tile * 16 * W + frag;
      bh[pp] = *reinterpret_cast<const v8bf*>(q);
      bm[pp] = *reinterpret_cast<const v8bf*>(q + nco * W);
      bl[pp] = *reinterpret_cast<const v8bf*>(q + 2 * nco * W);
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {        // row shift ky - 1 of z
      const int slot = (y + adk[ky] + 3) % 3;       // (lane dependent for WL = 16)
      v8bf ah[3], am[3], al[3];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const unsigned short* zp = Z + kx * ZCOPY + slot * ZROW + afrag[ky];
        ah[kx] = *reinterpret_cast<const v8bf*>(zp);
        am[kx] = *reinterpret_cast<const v8bf*>(zp + ZPLANE);
        al[kx] = *reinterpret_cast<const v8bf*>(zp + 2 * ZPLANE);
      }
      // (py, a) with a + py == ky, (px, b) with b + px == kx: 4 / 8 / 4 products for ky = 0 / 1 / 2, all on distinct
      // accumulators inside one cross term
#define PDES_WB3U_TERM(A_, B_)                                                                                   \
      _Pragma("unroll") for (int py = 0; py < 2; ++py) {                                                         \
        const int ia = ky - py;                                                                                  \
        if (ia < 0 || ia > 1) continue;                                                                          \
        _Pragma("unroll") for (int kx = 0; kx < 3; ++kx)                                                         \
          _Pragma("unroll") for (int px = 0; px < 2; ++px) {                                                     \
            const int ib = kx - px;                                                                              \
            if (ib < 0 || ib > 1) continue;                                                                      \
            const int q = (py * 2 + px) * 4 + ia * 2 + ib;                                                       \
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A_[kx], B_[py * 2 + px], acc[q], 0, 0, 0);          \
          }                                                                                                      \
      }
      PDES_WB3U_TERM(am, bm);               // six cross terms, smallest first
      PDES_WB3U_TERM(al, bh);
      PDES_WB3U_TERM(ah, bl);
      PDES_WB3U_TERM(am, bh);
      PDES_WB3U_TERM(ah, bm);
      PDES_WB3U_TERM(ah, bh);
#undef PDES_WB3U_TERM
    }
  }

  // ---- epilogue: the 16 effective-kernel gradients fold into the 9 taps; accumulator tile = D[ci = (lane >> 4) * 4 + r]
  // [co = lane & 15]; through LDS as [co][ci][tap]
  __syncthreads();
  float* outl = reinterpret_cast<float*>(smem_wb3);
  if (wave < ntiles) {
    const int co = wave * 16 + (lane & 15);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        v4f sum = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
          for (int px = 0; px < 2; ++px) {
            const int ia = py == 0 ? (ky == 0 ? 0 : 1) : (ky == 2 ? 1 : 0);
            const int ib = px == 0 ? (kx == 0 ? 0 : 1) : (kx == 2 ? 1 : 0);
            sum += acc[(py * 2 + px) * 4 + ia * 2 + ib];
          }
#pragma unroll
        for (int r = 0; r < 4; ++r) outl[(co * 16 + (lane >> 4) * 4 + r) * 9 + ky * 3 + kx] = sum[r];
      }
  }
  __syncthreads();
  float* pb = part + (size_t)blockIdx.x * per;
  const int run = min(16, crem) * 9;                       // contiguous floats per output channel
  for (int e = tid; e < min(corem, 64) * run; e += 256) {
    const int co = e / run, k = e - co * run;
    pb[((size_t)(co0 + co) * d.Cin + ci0) * 9 + k] = outl[co * 144 + k];
  }
}

// ------------------------------------------------------------------------------- host side
static bool wgrad_b3_up_shape(const pdes_conv_desc& d) {      // nearest-x2 + 3x3 from a 32- or 16-wide map
  if (!(opt().mfma_b3 & 16) || d.upsample != PDES_UPSAMPLE_NEAREST || d.Wout != 2 * d.Win || d.Hout != 2 * d.Hin) return false;
  if (!(d.Win == 32 || (d.Win == 16 && d.Hin % 2 == 0))) return false;
  return d.Cin >= 64 && d.Cout >= 32 && d.Cout <= 128;
}

bool wgrad_b3_applies(const pdes_conv_desc& d) {
  if (!(opt().mfma_b3 & 2) || d.ksize != 3 || d.stride != 1 || d.pad != 1 || !d.has_bn || d.g_fused || d.nrep != PDES_NREP) return false;
  if (d.upsample) return wgrad_b3_up_shape(d);
  if (d.Win != wb3::W || d.Wout != wb3::W || d.Hin != d.Hout) return false;
  return d.Cin >= 64 && d.Cout >= 32 && d.Cout <= 128;
}

// rows per workgroup: the whole image (one split per sample) unless that leaves most of the chip idle
int wgrad_b3_splits(const pdes_conv_desc& d) {
  const int mtiles = (d.Cin + 15) / 16 * (d.upsample ? (d.Cout + 63) / 64 : 1);
  const int H = d.Hin * d.Win / wb3::W;                   // k-rows of 32 pixels
  int hs = 1;
  while (d.B * hs * mtiles < 384 && hs * 2 <= H / 8 && H % (hs * 2) == 0) hs *= 2;
  return d.B * hs;
}

int conv_backward_weight_b3(const pdes_conv_desc& d, hipStream_t st) {
  if (!wgrad_b3_applies(d) || !d.ws) return PDES_ENOSUP;
  const int nsplit = wgrad_b3_splits(d), rows = d.Hin * d.Win / wb3::W / (nsplit / d.B);
  const long long per = (long long)d.Cout * d.Cin * 9;
  if ((long long)nsplit * per * 4 > d.ws_bytes) return PDES_ENOSUP;
  const int ntiles = (d.Cout + 15) / 16, nco = ntiles * 16;
  const size_t epi = (size_t)nco * 144 * sizeof(float);
  dim3 grid(nsplit, (d.Cin + 15) / 16), block(256);
  if (d.upsample) {
    const size_t lds = (size_t)wb3::ZSIZE * 2 + (size_t)4 * 3 * 64 * wb3::W * 2;      // (>= the 64 x 144 floats of the epilogue)
    grid.z = (d.Cout + 63) / 64;
    if (d.Win == 32) hipLaunchKernelGGL(conv_wgrad_b3_up_kernel<32>, grid, block, lds, st, d, d.ws, rows);
    else hipLaunchKernelGGL(conv_wgrad_b3_up_kernel<16>, grid, block, lds, st, d, d.ws, rows);
  } else {
    size_t lds = (size_t)wb3::ZSIZE * 2 + (size_t)2 * 3 * nco * wb3::W * 2;
    if (epi > lds) lds = epi;
    hipLaunchKernelGGL(conv_wgrad_b3_kernel<2>, grid, block, lds, st, d, d.ws, rows);        // two rows of load lead
  }
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

}  // namespace pdes
