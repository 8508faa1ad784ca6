// Nearest-x2 upsampling followed by a 3x3 convolution (reference models/codec.py:128-150, :163-188:
// `UpsamplingNearest2d` + `nn.Conv2d(3,1,1)`) as FOUR 2x2 convolutions on the low-resolution map
// (sub-pixel decomposition), forward and data gradient, on the f32 matrix cores.
//
// Because up[i][j] = z[i>>1][j>>1], the 9 taps of the 3x3 kernel collapse, for the output pixel
// (2y+dy, 2x+dx), onto the 2x2 low-res neighbourhood rows {y-1,y} (dy=0) or {y,y+1} (dy=1) (same
// for columns) with summed weights
//     Weff[dy][dx][a][b] = sum_{ky in R(dy,a)} sum_{kx in R(dx,b)} W[ky][kx],
//     R(0,0)={0}, R(0,1)={1,2}, R(1,0)={0,1}, R(1,1)={2}.
// That is 16 multiply-adds per low-res pixel instead of 36 (2.25x fewer MFMAs) and the operand tile
// is the plain 3x3 halo tile of the LOW-res map (4x less staging than the upsampled tile).
//
// Forward: M-tiles are low-res pixels; every (M-tile, N-tile) keeps 4 accumulators (one per output
// parity); a 3x3-position tap feeds the 1, 2 or 4 parities that use it.  The epilogue interleaves
// the dx parities into float4 stores of the hi-res rows 2y and 2y+1.
// Data gradient: dz[y][x] = sum_p sum_{a,b} Weff_p[a][b] . G_p[y-a-dy+1][x-b-dx+1] with the parity
// sub-images G_p[Y][X] = G[2Y+dy][2X+dx]; the K loop runs over (16-channel chunk, parity) pairs,
// each staging one de-interleaved sub-image tile and using its 4 taps.  The epilogue is the usual
// one (ReLU mask, gamma, T accumulate, BatchNorm gradient sums) at low resolution.
//
// Everything else (LDS layout, BN/ReLU on the way into LDS, replicated fp64 statistics, packed
// weight images with rolling register prefetch) follows conv_mfma.hip.
#include "pdes_common.h"
#include "../../include/pdes_hip.h"
#include "pack_kernels.h"

namespace pdes {

typedef float v4f __attribute__((ext_vector_type(4)));

struct BnU { float mean, invstd, gamma, beta; };
__device__ __forceinline__ BnU bn_coef_u(const pdes_conv_desc& d, int c, bool publish = false) {
  BnU o;
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

template <int TWG, int MT>
struct UpGeo {                                 // 3x3 halo tile of the low-res map (see TileGeo in conv_mfma.hip)
  static constexpr int TH = MT / TWG, TW = 16 * TWG;
  static constexpr int ROWS = TH + 2;
  static constexpr int COL0 = 4;
  static constexpr int LDW = ((COL0 + TW + 1 + 3) / 4) * 4;
  static constexpr int CS = ((ROWS * LDW - 16 + 31) / 32) * 32 + 16;
  static constexpr int KC = 16;
  static constexpr int NV4 = KC * ROWS * (TW / 4);
  static constexpr int NPV = (NV4 + 255) / 256;
  static constexpr int NH = KC * ROWS * 2;
  static constexpr int NPH = (NH + 255) / 256;
};

enum { UP_FWD = 0, UP_BWD = 1 };

// grid: (low-res tiles, B, ceil(N-tiles/4)); 4 waves, one N-tile each
template <int TWG, int MT, int MODE>
__global__ __launch_bounds__(256) void conv_up_mfma_kernel(pdes_conv_desc d, const float* __restrict__ wm,
                                                          int nt_total) {
  using G = UpGeo<TWG, MT>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int nt_w = blockIdx.z * 4 + wave;            // this wave's N-tile
  const int ntp = (nt_total + 7) & ~7;
  const int Hl = d.Hin, Wl = d.Win, HWl = Hl * Wl;   // low-res map (input of the layer)
  const int Hh = d.Hout, Wh = d.Wout, HWh = Hh * Wh; // hi-res map (output of the layer)
  const int kC = MODE == UP_FWD ? d.Cin : d.Cout;
  const int kpad = (kC + 15) & ~15;
  const int nchunk = kpad / 16;                      // channel chunks
  const int nvc = MODE == UP_FWD ? nchunk : nchunk * 4;   // staged (chunk, parity) tiles
  float* tile = smem + (MODE == UP_FWD ? 4 * kpad : 0);
  float4* cf4 = reinterpret_cast<float4*>(smem);

  const int tiles_x = Wl / G::TW;
  const int oy0 = (blockIdx.x / tiles_x) * G::TH, ox0 = (blockIdx.x % tiles_x) * G::TW;
  const bool halo_live = tiles_x > 1;

  if (MODE == UP_FWD) {
    for (int c = tid; c < kpad; c += 256) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < d.Cin) { const BnU k = bn_coef_u(d, c, (blockIdx.x | blockIdx.y | blockIdx.z) == 0); v = make_float4(k.mean, k.gamma * k.invstd, k.beta, 0.f); }
      cf4[c] = v;
    }
  }

  // ---- staging geometry (per thread, independent of the chunk)
  const float* kbase = MODE == UP_FWD ? d.x + (size_t)b * d.x_ctot * HWl
                                      : d.g + ((size_t)b * d.g_ctot + d.g_coff) * HWh;
  const int HWk = MODE == UP_FWD ? HWl : HWh;
  int vg[G::NPV], vl[G::NPV], hg[G::NPH], hl[G::NPH];
  unsigned vrow = 0, hval = 0;
#pragma unroll
  for (int i = 0; i < G::NPV; ++i) {
    const int e = tid + 256 * i;
    const int rem = e % (G::ROWS * (G::TW / 4));
    const int r = rem / (G::TW / 4), j = rem % (G::TW / 4);
    const int Y = oy0 - 1 + r, X = ox0 + 4 * j;                 // low-res coordinates
    const bool ok = e < G::NV4 && Y >= 0 && Y < Hl;
    const int Yc = min(max(Y, 0), Hl - 1);
    vg[i] = MODE == UP_FWD ? Yc * Wl + X : (2 * Yc) * Wh + 2 * X;       // BWD: + dy*Wh at issue time
    vl[i] = e < G::NV4 ? (e / (G::ROWS * (G::TW / 4))) * G::CS + r * G::LDW + G::COL0 + 4 * j : -1;
    if (ok) vrow |= 1u << i;
  }
#pragma unroll
  for (int i = 0; i < G::NPH; ++i) {
    const int e = tid + 256 * i;
    const int rem = e % (G::ROWS * 2);
    const int r = rem / 2, h = rem % 2;
    const int Y = oy0 - 1 + r, X = h == 0 ? ox0 - 1 : ox0 + G::TW;
    const bool ok = e < G::NH && Y >= 0 && Y < Hl && X >= 0 && X < Wl;
    const int Yc = min(max(Y, 0), Hl - 1), Xc = min(max(X, 0), Wl - 1);
    hg[i] = MODE == UP_FWD ? Yc * Wl + Xc : (2 * Yc) * Wh + 2 * Xc;     // BWD: + dy*Wh + dx at issue time
    hl[i] = e < G::NH ? (e / (G::ROWS * 2)) * G::CS + r * G::LDW + (h == 0 ? G::COL0 - 1 : G::COL0 + G::TW) : -1;
    if (ok) hval |= 1u << i;
  }

  // Two register stages of RAW loads (clamped addresses, nothing consumes a load before its commit): the
  // loop body below is straight-line so that the in-order vmcnt waits are exact (see conv_mfma.hip).
  constexpr int NPW = MODE == UP_BWD ? G::NPV : 1;
  struct Stage {
    float4 pv[G::NPV]; float4 pw[NPW]; float ph[G::NPH];
  };
  Stage sA, sB;
  auto issue = [&](int vc, Stage& st) __attribute__((always_inline)) {
    const int chunk = MODE == UP_FWD ? vc : vc >> 2, p = vc & 3;
    const int dy = MODE == UP_FWD ? 0 : p >> 1, dx = MODE == UP_FWD ? 0 : p & 1;
    const float* src = kbase + (size_t)chunk * 16 * HWk + (MODE == UP_BWD ? dy * Wh : 0);
    const int cmax = kC - chunk * 16 - 1;
#pragma unroll
    for (int i = 0; i < G::NPV; ++i) {
      const int ch = min((tid + 256 * i) / (G::ROWS * (G::TW / 4)), cmax);
      const float* q = src + ch * HWk + vg[i];
      st.pv[i] = *reinterpret_cast<const float4*>(q);
      if (MODE == UP_BWD) st.pw[i] = *reinterpret_cast<const float4*>(q + 4);      // 8 hi-res columns -> 4 of one parity
    }
    if (halo_live) {
#pragma unroll
      for (int i = 0; i < G::NPH; ++i) {
        const int ch = min((tid + 256 * i) / (G::ROWS * 2), cmax);
        st.ph[i] = src[ch * HWk + hg[i] + (MODE == UP_BWD ? dx : 0)];
      }
    }
  };
  auto commit = [&](int vc, int buf, const Stage& st) __attribute__((always_inline)) {
    const int chunk = MODE == UP_FWD ? vc : vc >> 2, p = vc & 3;
    const int dx = p & 1;
    float* t = tile + buf * (G::KC * G::CS);
    const int crem = kC - chunk * 16;
#pragma unroll
    for (int i = 0; i < G::NPV; ++i) {
      if (vl[i] >= 0) {
        const int ch = (tid + 256 * i) / (G::ROWS * (G::TW / 4));
        const bool ok = ((vrow >> i) & 1u) && ch < crem;
        float4 z;
        if (MODE == UP_FWD) {
          const float4 k = cf4[chunk * 16 + ch];
          z = st.pv[i];
          z.x = ok ? fmaxf(0.f, (z.x - k.x) * k.y + k.z) : 0.f;
          z.y = ok ? fmaxf(0.f, (z.y - k.x) * k.y + k.z) : 0.f;
          z.z = ok ? fmaxf(0.f, (z.z - k.x) * k.y + k.z) : 0.f;
          z.w = ok ? fmaxf(0.f, (z.w - k.x) * k.y + k.z) : 0.f;
        } else {
          z = dx ? make_float4(st.pv[i].y, st.pv[i].w, st.pw[i].y, st.pw[i].w)
                 : make_float4(st.pv[i].x, st.pv[i].z, st.pw[i].x, st.pw[i].z);
          if (!ok) z = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        *reinterpret_cast<float4*>(t + vl[i]) = z;
      }
    }
#pragma unroll
    for (int i = 0; i < G::NPH; ++i) {
      if (hl[i] >= 0) {
        const int ch = (tid + 256 * i) / (G::ROWS * 2);
        const bool ok = halo_live && ((hval >> i) & 1u) && ch < crem;
        float z = halo_live ? st.ph[i] : 0.f;
        if (MODE == UP_FWD) {
          const float4 k = cf4[chunk * 16 + ch];
          z = ok ? fmaxf(0.f, (z - k.x) * k.y + k.z) : 0.f;
        } else {
          if (!ok) z = 0.f;
        }
        t[hl[i]] = z;
      }
    }
  };

  constexpr int NACC = MODE == UP_FWD ? 4 : 1;            // accumulators per M-tile (output parities)
  v4f acc[MT][NACC];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[mt][q] = (v4f){0.f, 0.f, 0.f, 0.f};

  // B operand image: [(kstep*16 + q)*ntp + nt][64], q = parity*4 + a*2 + b.  FWD uses all 16 per
  // k-step, BWD the 4 of the staged parity.  Two register sets: while one feeds the matrix pipe the
  // next k-step streams into the other (every wave index is < ntp: the image is padded to 8 N-tiles).
  constexpr int NB = MODE == UP_FWD ? 16 : 4;
  const int ksteps = kpad / 4;
  float bA[NB], bB[NB];
  auto load_b = [&](int kstep, int p, float (&dst)[NB]) __attribute__((always_inline)) {
    const float* wp = wm + ((size_t)(min(kstep, ksteps - 1) * 16 + (MODE == UP_FWD ? 0 : p * 4)) * ntp + nt_w) * 64 + lane;
#pragma unroll
    for (int t = 0; t < NB; ++t) dst[t] = wp[(size_t)t * ntp * 64];
  };
  const int a_lane = (lane >> 4) * G::CS + (lane & 15);
  auto mfma_kstep = [&](const float* tk, int dy, int dx, const float (&bw)[NB]) __attribute__((always_inline)) {
    if (MODE == UP_FWD) {
#pragma unroll
      for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const float a = tk[((mt / TWG) + ty) * G::LDW + (G::COL0 - 1) + (mt % TWG) * 16 + tx];
#pragma unroll
            for (int ddy = 0; ddy < 2; ++ddy)
#pragma unroll
              for (int ddx = 0; ddx < 2; ++ddx) {
                const int ia = ty - ddy, ib = tx - ddx;           // position inside the 2x2 effective kernel
                if (ia < 0 || ia > 1 || ib < 0 || ib > 1) continue;
                const int pp = ddy * 2 + ddx;
                acc[mt][MODE == UP_FWD ? pp : 0] =
                    __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[(MODE == UP_FWD ? pp * 4 : 0) + ia * 2 + ib],
                                                         acc[mt][MODE == UP_FWD ? pp : 0], 0, 0, 0);
              }
          }
        }
    } else {
#pragma unroll
      for (int ia = 0; ia < 2; ++ia)
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
          const int ro = 2 - ia - dy, co = 2 - ib - dx;          // tile offsets of G_p[y-a-dy+1][x-b-dx+1]
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const float a = tk[((mt / TWG) + ro) * G::LDW + (G::COL0 - 1) + (mt % TWG) * 16 + co];
            acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[MODE == UP_FWD ? 0 : ia * 2 + ib], acc[mt][0], 0, 0, 0);
          }
        }
    }
  };

  load_b(0, 0, bA);
  issue(0, sA);
  if (nvc > 1) issue(1, sB);
  __syncthreads();                 // cf4 visible
  commit(0, 0, sA);
  __syncthreads();

  auto step = [&](int vc, Stage& sfree, const Stage& snext) __attribute__((always_inline)) {
    const int buf = vc & 1;
    const int chunk = MODE == UP_FWD ? vc : vc >> 2, p = vc & 3;
    const int dy = p >> 1, dx = p & 1;
    const int vn = vc + 1, chunkn = MODE == UP_FWD ? vn : vn >> 2, pn = vn & 3;   // first k-step of the next staged tile
    const float* tb = tile + buf * (G::KC * G::CS) + a_lane;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < 3) load_b(chunk * 4 + s + 1, p, (s & 1) ? bA : bB);
      else load_b(chunkn * 4, pn, bA);
      if (s == 0) issue(min(vc + 2, nvc - 1), sfree);
      if ((chunk * 4 + s) * 4 < kC)              // scalar: skip k-steps that lie entirely in the zero padding
        mfma_kstep(tb + s * 4 * G::CS, dy, dx, (s & 1) ? bB : bA);
    }
    if (vc + 1 < nvc) commit(vc + 1, buf ^ 1, snext);
    __syncthreads();
  };
  {
    int vc = 0;
    for (; vc + 1 < nvc; vc += 2) { step(vc, sA, sB); step(vc + 1, sB, sA); }
    if (vc < nvc) step(vc, sA, sB);
  }

  const int px = (lane >> 4) * 4;
  const int cn = nt_w * 16 + (lane & 15);             // this lane's N channel
  if (MODE == UP_FWD) {
    float s = 0.f, q = 0.f;
    if (nt_w < nt_total && cn < d.Cout) {
      float* ob = d.out + ((size_t)b * d.out_ctot + d.out_coff + cn) * HWh;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int oy = oy0 + mt / TWG, ox = ox0 + (mt % TWG) * 16 + px;       // low-res
#pragma unroll
        for (int ddy = 0; ddy < 2; ++ddy) {
          const v4f v0 = acc[mt][MODE == UP_FWD ? ddy * 2 : 0], v1 = acc[mt][MODE == UP_FWD ? ddy * 2 + 1 : 0];
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
  } else {
    const float* xb = d.x + (size_t)b * d.x_ctot * HWl;
    float* tb2 = d.t_in + (size_t)b * d.x_ctot * HWl;
    float dg = 0.f, db = 0.f, st = 0.f, sx = 0.f;
    const bool live = nt_w < nt_total && cn < d.Cin;
    if (live) {
      const BnU k = bn_coef_u(d, cn);
      const float scale = k.gamma * k.invstd;
      const bool fin = cn >= d.final_c0 && cn < d.final_c1;
      float4 xq[MT], tq[MT];        // loads first, then arithmetic + stores (see conv_mfma.hip)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const size_t idx = (size_t)cn * HWl + (size_t)(oy0 + mt / TWG) * Wl + ox0 + (mt % TWG) * 16 + px;
        xq[mt] = *reinterpret_cast<const float4*>(xb + idx);
        tq[mt] = d.t_accumulate ? *reinterpret_cast<const float4*>(tb2 + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const v4f v = acc[mt][0];
        const size_t idx = (size_t)cn * HWl + (size_t)(oy0 + mt / TWG) * Wl + ox0 + (mt % TWG) * 16 + px;
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
    if (lane < 16 && live) {
      const long long ro = (long long)rep_of_block(d.nrep) * d.rep_stride;
      atomicAdd(&d.bn_grad[ro + 2 * cn], (double)dg);
      atomicAdd(&d.bn_grad[ro + 2 * cn + 1], (double)db);
      if (cn >= d.final_c0 && cn < d.final_c1) {
        atomicAdd(&d.t_stats[ro + 2 * cn], (double)st);
        atomicAdd(&d.t_stats[ro + 2 * cn + 1], (double)sx);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// effective-weight images (pack_kernels.h)
__global__ __launch_bounds__(256) void pack_up_kernel(const pdes_up_pack_item* __restrict__ items) {
  pack_up_item(items[blockIdx.y], blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------- host dispatch
static bool up_shape_ok(const pdes_conv_desc& d) {
  if (!d.upsample || d.ksize != 3 || d.stride != 1 || d.pad != 1 || !d.has_bn) return false;
  if (d.Hout != 2 * d.Hin || d.Wout != 2 * d.Win) return false;
  const int W = d.Win, H = d.Hin;
  if (W % 16 || (W >= 32 && W % 32)) return false;
  return H % (W >= 32 ? 4 : 8) == 0;
}

template <int MODE>
static int launch_up(const pdes_conv_desc& d, const float* wm, hipStream_t st, bool dry = false) {
  const int kC = MODE == UP_FWD ? d.Cin : d.Cout, nC = MODE == UP_FWD ? d.Cout : d.Cin;
  const int kpad = (kC + 15) & ~15, nt_total = (nC + 15) / 16;
  const int W = d.Win, H = d.Hin;
  const int twg = W >= 32 ? 2 : 1;
  const int gz = (nt_total + 3) / 4;
  int mt = 8;
  if ((long long)(W / (16 * twg)) * (H / (8 / twg)) * d.B * gz < 256) mt = 4;
  if (MODE == UP_FWD) mt = 4;                       // 4 parity accumulators per M-tile: keep the register budget
  const int th = mt / twg;
  if (H % th) return PDES_ENOSUP;
  if (dry) return PDES_OK;
  if (d.g_fused) return PDES_ENOSUP;
  dim3 grid((W / (16 * twg)) * (H / th), d.B, gz), block(256);
#define PDES_UP_LAUNCH(TWG_, MT_)                                                                            \
  do {                                                                                                        \
    using G = UpGeo<TWG_, MT_>;                                                                               \
    const size_t lds = ((MODE == UP_FWD ? 4 * (size_t)kpad : 0) + 2 * (size_t)G::KC * G::CS) * sizeof(float); \
    hipLaunchKernelGGL((conv_up_mfma_kernel<TWG_, MT_, MODE>), grid, block, lds, st, d, wm, nt_total);        \
  } while (0)
  if (twg == 2) { if (mt == 8) PDES_UP_LAUNCH(2, 8); else PDES_UP_LAUNCH(2, 4); }
  else { if (mt == 8) PDES_UP_LAUNCH(1, 8); else PDES_UP_LAUNCH(1, 4); }
#undef PDES_UP_LAUNCH
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

int conv_forward_up_mfma(const pdes_conv_desc& d, hipStream_t st, bool dry) {
  if (!d.wu_fwd || !up_shape_ok(d) || d.Cin < 16 || d.nrep != PDES_NREP) return PDES_ENOSUP;
  return launch_up<UP_FWD>(d, d.wu_fwd, st, dry);
}

// dry = true: only report whether this implementation would take the descriptor (nothing is enqueued)
int conv_backward_data_up_mfma(const pdes_conv_desc& d, hipStream_t st, bool dry) {
  if (!d.wu_bwd || !up_shape_ok(d) || d.eval_mode || d.nrep != PDES_NREP) return PDES_ENOSUP;
  return launch_up<UP_BWD>(d, d.wu_bwd, st, dry);
}

}  // namespace pdes

using namespace pdes;

extern "C" int pdes_pack_weights_up(const pdes_up_pack_item* items, int n, int max_elems, void* stream) {
  if (!items || n <= 0 || max_elems <= 0) return PDES_EINVAL;
  int gx = cdiv(max_elems, 256);
  gx = gx > 128 ? 128 : gx;
  hipLaunchKernelGGL(pack_up_kernel, dim3(gx, n), dim3(256), 0, static_cast<hipStream_t>(stream), items);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}
