// Implicit-GEMM convolutions on the f32 matrix cores (v_mfma_f32_16x16x4_f32, exact f32 = an fmaf
// chain) for the 3x3 and 1x1 convolutions of DenseED (reference models/codec.py:43-188), gfx950.
//
// A k x k convolution is k*k pointwise (1x1) contractions over shifted views of one LDS tile:
//     out[pixel][co] += sum_ci z[ci][pixel + tap] * W[co][ci][tap]
// GEMM roles per MFMA (16x16x4): M = 16 consecutive pixels of one image row (A operand, one
// ds_read_b32 per lane from the LDS tile), N = 16 output channels (B operand, one coalesced global
// load per lane from a pre-packed weight image), K = 4 input channels.  Accumulator lane layout:
// col = lane&15 = output channel, rows (lane>>4)*4+r = 4 consecutive pixels -> one float4 store.
//
// Workgroup = 256 threads = 4 waves, output tile = 8 M-tiles (4 rows x 32 px, or 8 rows x 16 px
// for 16-wide maps) of ONE sample -> 256 workgroups for a 32x32 map at batch 32 (one per CU).
// Input channels are processed in chunks of 16: the chunk's (rows+2) x (cols+2) halo tile is
// BatchNorm+ReLU'd (and nearest-x2 upsampled) on the way into LDS, double buffered, the next
// chunk's global loads being issued before the current chunk's MFMAs (register prefetch).
// LDS channel stride is == 16 (mod 32) dwords so the 2 x 16 lanes of a ds_read_b32 group never
// collide.  Waves split the work either by K (dense layers, Cout = 16: each wave takes one k-step
// of every chunk, partial sums are combined through LDS once at the end) or by N (wide layers:
// each wave owns NT_W of the output-channel tiles).
// The epilogue accumulates the fp64 {sum, sum^2} per output channel that the consumer BatchNorms
// need (one double atomic per channel per workgroup).
//
// The same kernel computes the data gradient (MODE_BWD): the "input" is dL/d(out) (raw, no BN),
// the weights are the transposed / tap-flipped image, and the epilogue applies the ReLU mask and
// gamma, accumulates into T, and reduces dgamma / dbeta / {sum T, sum T xhat}.
#include "pdes_common.h"
#include "../../include/pdes_hip.h"

namespace pdes {

typedef float v4f __attribute__((ext_vector_type(4)));

struct BnC { float mean, invstd, gamma, beta; };
__device__ __forceinline__ BnC bn_coef_m(const pdes_conv_desc& d, int c) {
  BnC o;
  if (d.eval_mode) {
    o.mean = d.run_mean[c];
    o.invstd = (float)(1.0 / sqrt((double)d.run_var[c] + (double)d.eps));
  } else {
    const double n = (double)d.B * d.Hin * d.Win;
    const double m = rep_sum(d.x_stats, 2 * c, d.nrep, d.rep_stride) / n;
    double var = rep_sum(d.x_stats, 2 * c + 1, d.nrep, d.rep_stride) / n - m * m;
    var = var < 0.0 ? 0.0 : var;
    o.mean = (float)m;
    o.invstd = (float)(1.0 / sqrt(var + (double)d.eps));
  }
  o.gamma = d.gamma[c];
  o.beta = d.beta[c];
  return o;
}

template <int KS, int TWG>
struct TileGeo {
  static constexpr int TH = 8 / TWG, TW = 16 * TWG;          // output tile (pixels)
  static constexpr int ROWS = TH + KS - 1, COLS = TW + KS - 1;
  static constexpr int LDW = (KS == 5) ? (TWG == 2 ? 38 : 20)
                             : (KS == 3) ? (TWG == 2 ? 40 : 24) : (TWG == 2 ? 36 : 18);
  static constexpr int CS = ROWS * LDW;                       // channel stride in LDS (dwords)
  static constexpr int KC = 16;                               // input channels per chunk
  static constexpr int NELEM = KC * ROWS * COLS;
  static constexpr int NPF = (NELEM + 255) / 256;             // prefetch registers per thread
  static_assert(CS % 32 == 16, "LDS channel stride must be 16 mod 32 dwords");
  static_assert(LDW >= COLS, "row pitch");
};

enum { MODE_FWD = 0, MODE_BWD = 1 };

// K-operand ("input") view of the kernel: FWD reads x (BN+ReLU, optional nearest x2); BWD reads g.
struct KView {
  const float* base;   // sample base pointer (channel 0)
  int C;               // channels
  int H, W;            // stored size
  int Hc, Wc;          // logical (conv-input) size: 2x stored when upsampled
  int up;
};

template <int KS, int TWG, int WAVES_K, int NT_W, int MODE>
__global__ __launch_bounds__(256) void conv_mfma_kernel(pdes_conv_desc d, const float* __restrict__ wm,
                                                       int nt_total) {
  using G = TileGeo<KS, TWG>;
  constexpr int KK = KS * KS;
  constexpr int KSW = 4 / WAVES_K;          // k-steps of a chunk handled by one wave
  constexpr int PADL = (KS - 1) / 2;        // 'same' convolution (pad = (k-1)/2, stride 1)
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave % WAVES_K, wn = wave / WAVES_K;
  const int b = blockIdx.y;
  const int nt_base = (blockIdx.z * (4 / WAVES_K) + wn) * NT_W;   // first N-tile of this wave

  // ---- views: K operand (staged through LDS) and the N/output side
  KView kv;
  int Hout, Wout, Cn;                        // output map size, number of N channels
  if (MODE == MODE_FWD) {
    kv.C = d.Cin; kv.H = d.Hin; kv.W = d.Win; kv.up = d.upsample;
    kv.base = d.x + (size_t)b * d.x_ctot * d.Hin * d.Win;
    Hout = d.Hout; Wout = d.Wout; Cn = d.Cout;
  } else {
    kv.C = d.Cout; kv.H = d.Hout; kv.W = d.Wout; kv.up = 0;
    kv.base = d.g + ((size_t)b * d.g_ctot + d.g_coff) * d.Hout * d.Wout;
    Hout = d.upsample ? 2 * d.Hin : d.Hin; Wout = d.upsample ? 2 * d.Win : d.Win; Cn = d.Cin;
  }
  kv.Hc = kv.up ? 2 * kv.H : kv.H;
  kv.Wc = kv.up ? 2 * kv.W : kv.W;
  const int kpad = (kv.C + 15) & ~15;
  const int nchunk = kpad / 16;
  float* cf = smem;                          // FWD: [kpad][3] mean, scale, beta
  float* tile = smem + ((MODE == MODE_FWD) ? 3 * kpad : 0);

  const int tiles_x = Wout / G::TW;
  const int oy0 = (blockIdx.x / tiles_x) * G::TH, ox0 = (blockIdx.x % tiles_x) * G::TW;

  if (MODE == MODE_FWD) {
    for (int c = tid; c < kpad; c += 256) {
      float m = 0.f, s = 0.f, bt = 0.f;
      if (c < d.Cin) { const BnC k = bn_coef_m(d, c); m = k.mean; s = k.gamma * k.invstd; bt = k.beta; }
      cf[3 * c] = m; cf[3 * c + 1] = s; cf[3 * c + 2] = bt;
    }
  }

  // ---- per-thread staging geometry (independent of the chunk)
  const int HWs = kv.H * kv.W;
  int goff[G::NPF], loff[G::NPF];
  unsigned vmask = 0;
#pragma unroll
  for (int i = 0; i < G::NPF; ++i) {
    const int e = tid + 256 * i;
    const int ch = e / (G::ROWS * G::COLS), rem = e % (G::ROWS * G::COLS);
    const int r = rem / G::COLS, c = rem % G::COLS;
    const int cy = oy0 - PADL + r, cx = ox0 - PADL + c;
    const bool v = (e < G::NELEM) && cy >= 0 && cy < kv.Hc && cx >= 0 && cx < kv.Wc;
    const int sy = kv.up ? (cy >> 1) : cy, sx = kv.up ? (cx >> 1) : cx;
    goff[i] = v ? (ch * HWs + sy * kv.W + sx) : 0;
    loff[i] = (e < G::NELEM) ? (ch * G::CS + r * G::LDW + c) : -1;
    if (v) vmask |= 1u << i;
  }

  float pf[G::NPF];
  auto issue = [&](int chunk) {
    const float* src = kv.base + (size_t)chunk * 16 * HWs;
    const int crem = kv.C - chunk * 16;               // channels available in this chunk
#pragma unroll
    for (int i = 0; i < G::NPF; ++i) {
      const int e = tid + 256 * i;
      const int ch = e / (G::ROWS * G::COLS);
      pf[i] = ((vmask >> i) & 1u) && ch < crem ? src[goff[i]] : 0.f;
    }
  };
  auto commit = [&](int chunk, int buf) {
    float* t = tile + buf * (G::KC * G::CS);
    const int crem = kv.C - chunk * 16;
#pragma unroll
    for (int i = 0; i < G::NPF; ++i) {
      const int e = tid + 256 * i;
      const int ch = e / (G::ROWS * G::COLS);
      float z = pf[i];
      if (MODE == MODE_FWD) {
        const float* k = cf + 3 * (chunk * 16 + ch);
        const bool v = ((vmask >> i) & 1u) && ch < crem;
        z = v ? fmaxf(0.f, (z - k[0]) * k[1] + k[2]) : 0.f;
      }
      if (loff[i] >= 0) t[loff[i]] = z;
    }
  };

  v4f acc[8][NT_W];
#pragma unroll
  for (int mt = 0; mt < 8; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT_W; ++nt) acc[mt][nt] = (v4f){0.f, 0.f, 0.f, 0.f};

  issue(0);
  __syncthreads();                 // cf visible
  commit(0, 0);
  __syncthreads();

  const int a_lane = (lane >> 4) * G::CS + (lane & 15);
  float bnext[KK];
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const int buf = chunk & 1;
    // B operand of this chunk, packed image [(kstep*KK + tap)*nt_total + nt][64].  These loads are
    // issued BEFORE the next chunk's activation prefetch: vmcnt retires loads in order, so the
    // MFMAs (which wait for B) would otherwise also wait for the whole prefetch.
    float bw[KSW][KK][NT_W];
    if (WAVES_K == 4 && chunk > 0) {
#pragma unroll
      for (int t = 0; t < KK; ++t) bw[0][t][0] = bnext[t];     // prefetched during the previous chunk
    } else {
#pragma unroll
      for (int s = 0; s < KSW; ++s) {
        const int kstep = chunk * 4 + (WAVES_K == 4 ? wk : s);
#pragma unroll
        for (int t = 0; t < KK; ++t)
#pragma unroll
          for (int nt = 0; nt < NT_W; ++nt) {
            const int ntg = nt_base + nt;
            bw[s][t][nt] = ntg < nt_total ? wm[((size_t)(kstep * KK + t) * nt_total + ntg) * 64 + lane] : 0.f;
          }
      }
    }
    if (WAVES_K == 4 && chunk + 1 < nchunk) {
      const int kstep = (chunk + 1) * 4 + wk;
#pragma unroll
      for (int t = 0; t < KK; ++t)
        bnext[t] = nt_base < nt_total ? wm[((size_t)(kstep * KK + t) * nt_total + nt_base) * 64 + lane] : 0.f;
    }
    if (chunk + 1 < nchunk) issue(chunk + 1);
    const float* tb = tile + buf * (G::KC * G::CS) + a_lane;
#pragma unroll
    for (int s = 0; s < KSW; ++s) {
      const int kstep = chunk * 4 + (WAVES_K == 4 ? wk : s);
      if (kstep * 4 >= kv.C) continue;          // wave-uniform: k-step entirely in the zero padding
      const float* tk = tb + (WAVES_K == 4 ? wk : s) * 4 * G::CS;
#pragma unroll
      for (int ky = 0; ky < KS; ++ky)
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
#pragma unroll
          for (int mt = 0; mt < 8; ++mt) {
            const float a = tk[((mt / TWG) + ky) * G::LDW + (mt % TWG) * 16 + kx];
#pragma unroll
            for (int nt = 0; nt < NT_W; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[s][ky * KS + kx][nt], acc[mt][nt], 0, 0, 0);
          }
        }
    }
    if (chunk + 1 < nchunk) commit(chunk + 1, buf ^ 1);
    __syncthreads();
  }

  // ---- combine the K-split partial sums: wave w ends up owning M-tiles {2w, 2w+1}
  constexpr int MT_OWN = (WAVES_K == 4) ? 2 : 8;
  const int mt0 = (WAVES_K == 4) ? 2 * wave : 0;
  if (WAVES_K == 4) {
    float* red = tile;                         // [4 waves][8 mt][4 r][64 lanes] = 8192 floats
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((wave * 8 + mt) * 4 + r) * 64 + lane] = acc[mt][0][r];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red[((w * 8 + mt0 + j) * 4 + r) * 64 + lane];
        acc[j][0][r] = s;                      // acc[0..1] now hold the wave's own two M-tiles
      }
  }

  const int HWo = Hout * Wout;
  const int px = (lane >> 4) * 4;              // first of the lane's 4 consecutive pixels in the M-tile
  if (MODE == MODE_FWD) {
#pragma unroll
    for (int nt = 0; nt < NT_W; ++nt) {
      const int co = (nt_base + nt) * 16 + (lane & 15);
      float s = 0.f, q = 0.f;
      if (co < d.Cout) {
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
          float* sred = tile + 8192;                 // past the accumulator exchange area
          if (lane < 16) { sred[(wave * 16 + lane) * 2] = s; sred[(wave * 16 + lane) * 2 + 1] = q; }
          __syncthreads();
          if (wave == 0 && lane < 32) {
            const int c = lane >> 1, w = lane & 1;
            const float t = (sred[(0 * 16 + c) * 2 + w] + sred[(1 * 16 + c) * 2 + w]) +
                            (sred[(2 * 16 + c) * 2 + w] + sred[(3 * 16 + c) * 2 + w]);
            const int cc = nt_base * 16 + c;
            if (cc < d.Cout) atomicAdd(&os[2 * (d.out_coff + cc) + w], (double)t);
          }
        } else if (lane < 16 && co < d.Cout) {
          atomicAdd(&os[2 * (d.out_coff + co)], (double)s);
          atomicAdd(&os[2 * (d.out_coff + co) + 1], (double)q);
        }
      }
    }
  } else {
    // data gradient epilogue.  With upsample the 2x2 hi-res results of one stored pixel are summed:
    // horizontally inside the float4, vertically between M-tiles mt and mt+TWG (rows oy, oy+1).
    const int HWi = d.Hin * d.Win;
    const float* xb = d.x + (size_t)b * d.x_ctot * HWi;
    float* tb2 = d.t_in + (size_t)b * d.x_ctot * HWi;
#pragma unroll
    for (int nt = 0; nt < NT_W; ++nt) {
      const int ci = (nt_base + nt) * 16 + (lane & 15);
      float dg = 0.f, db = 0.f, st = 0.f, sx = 0.f;
      if (ci < d.Cin) {
        const BnC k = bn_coef_m(d, ci);
        const float scale = k.gamma * k.invstd;
        const bool fin = ci >= d.final_c0 && ci < d.final_c1;
        if (!d.upsample) {
#pragma unroll
          for (int j = 0; j < MT_OWN; ++j) {
            const int mt = mt0 + j;
            const v4f v = acc[j][nt];
            const size_t idx = (size_t)ci * HWi + (size_t)(oy0 + mt / TWG) * d.Win + ox0 + (mt % TWG) * 16 + px;
            const float4 xv = *reinterpret_cast<const float4*>(xb + idx);
            float4 tv = d.t_accumulate ? *reinterpret_cast<const float4*>(tb2 + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
            float ts[4] = {tv.x, tv.y, tv.z, tv.w};
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
        } else {
          // rows come in pairs (mt, mt+TWG) -> requires TH even and M-tiles ordered row-major
#pragma unroll
          for (int j = 0; j < MT_OWN; ++j) {
            const int mt = mt0 + j;
            if ((mt / TWG) & 1) continue;                     // odd rows are folded into the even row above
            const v4f v0 = acc[j][nt], v1 = acc[(j + TWG < 8) ? j + TWG : j][nt];
            const float lo = (v0[0] + v0[1]) + (v1[0] + v1[1]);
            const float hi = (v0[2] + v0[3]) + (v1[2] + v1[3]);
            const int iy = (oy0 + mt / TWG) >> 1, ix = (ox0 + (mt % TWG) * 16 + px) >> 1;
            const size_t idx = (size_t)ci * HWi + (size_t)iy * d.Win + ix;
            const float2 xv = *reinterpret_cast<const float2*>(xb + idx);
            float2 tv = d.t_accumulate ? *reinterpret_cast<const float2*>(tb2 + idx) : make_float2(0.f, 0.f);
            const float xs[2] = {xv.x, xv.y}, vv[2] = {lo, hi};
            float ts[2] = {tv.x, tv.y};
#pragma unroll
            for (int r = 0; r < 2; ++r) {
              const float y = (xs[r] - k.mean) * scale + k.beta;
              const float xh = (xs[r] - k.mean) * k.invstd;
              const float dyv = (y > 0.f) ? vv[r] : 0.f;
              db += dyv; dg += dyv * xh;
              ts[r] += k.gamma * dyv;
              if (fin) { st += ts[r]; sx += ts[r] * xh; }
            }
            *reinterpret_cast<float2*>(tb2 + idx) = make_float2(ts[0], ts[1]);
          }
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
// weight images for the MFMA kernels, rebuilt from the live weights every step (one launch for the
// whole network): forward  [(kstep*KK + tap)*NT + nt][kq*16 + n] = W[n + 16 nt][4 kstep + kq][tap]
//                 backward [(kstep*KK + tap)*NT + nt][kq*16 + n] = W[4 kstep + kq][n + 16 nt][KK-1-tap]
__global__ __launch_bounds__(256) void pack_mfma_kernel(const pdes_mfma_pack_item* __restrict__ items) {
  const pdes_mfma_pack_item it = items[blockIdx.y];
  const int ntf = (it.Cout + 15) / 16, ksf = ((it.Cin + 15) / 16) * 4;
  const int totf = ksf * it.kk * ntf * 64;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < totf; i += gridDim.x * 256) {
    const int l = i & 63, nt = (i >> 6) % ntf, t = ((i >> 6) / ntf) % it.kk, ks = (i >> 6) / (ntf * it.kk);
    const int co = nt * 16 + (l & 15), ci = 4 * ks + (l >> 4);
    it.wm_fwd[i] = (co < it.Cout && ci < it.Cin) ? it.w[((size_t)co * it.Cin + ci) * it.kk + t] : 0.f;
  }
  if (!it.wm_bwd) return;
  const int ntb = (it.Cin + 15) / 16, ksb = ((it.Cout + 15) / 16) * 4;
  const int totb = ksb * it.kk * ntb * 64;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < totb; i += gridDim.x * 256) {
    const int l = i & 63, nt = (i >> 6) % ntb, t = ((i >> 6) / ntb) % it.kk, ks = (i >> 6) / (ntb * it.kk);
    const int ci = nt * 16 + (l & 15), co = 4 * ks + (l >> 4);
    it.wm_bwd[i] = (co < it.Cout && ci < it.Cin) ? it.w[((size_t)co * it.Cin + ci) * it.kk + (it.kk - 1 - t)] : 0.f;
  }
}

// ------------------------------------------------------------------------------- host dispatch
static bool mfma_shape_ok(const pdes_conv_desc& d, bool bwd) {
  if (!(d.ksize == 5 || d.ksize == 3 || d.ksize == 1) || d.stride != 1 || d.pad != (d.ksize - 1) / 2) return false;
  if (!d.has_bn) return false;
  const int W = bwd ? (d.upsample ? 2 * d.Win : d.Win) : d.Wout;
  const int H = bwd ? (d.upsample ? 2 * d.Hin : d.Hin) : d.Hout;
  if (W % 16 || (W >= 32 ? (W % 32 || H % 4) : (H % 8))) return false;
  return true;
}

template <int KS, int MODE>
static int launch_mfma(const pdes_conv_desc& d, const float* wm, hipStream_t st) {
  const bool bwd = MODE == MODE_BWD;
  const int W = bwd ? (d.upsample ? 2 * d.Win : d.Win) : d.Wout;
  const int H = bwd ? (d.upsample ? 2 * d.Hin : d.Hin) : d.Hout;
  const int kC = bwd ? d.Cout : d.Cin, nC = bwd ? d.Cin : d.Cout;
  const int kpad = (kC + 15) & ~15;
  const int nt_total = (nC + 15) / 16;
  const int twg = W >= 32 ? 2 : 1;
  const int tiles = (W / (16 * twg)) * (H / (8 / twg));
  dim3 grid(tiles, d.B), block(256);
  const int cs = twg == 2 ? TileGeo<KS, 2>::CS : TileGeo<KS, 1>::CS;
  size_t lds_f = (size_t)(bwd ? 0 : 3 * kpad) + (size_t)2 * 16 * cs;
  const size_t red_f = (size_t)(bwd ? 0 : 3 * kpad) + 8192 + 128;
#define PDES_MFMA_LAUNCH(TWG_, WK_, NTW_)                                                                    \
  hipLaunchKernelGGL((conv_mfma_kernel<KS, TWG_, WK_, NTW_, MODE>), grid, block,                              \
                     ((WK_) == 4 && red_f > lds_f ? red_f : lds_f) * sizeof(float), st, d, wm, nt_total)
  if (nt_total == 1) {
    if (bwd && d.upsample) return PDES_ENOSUP;          // K-split waves do not own both rows of a pair
    if (twg == 2) PDES_MFMA_LAUNCH(2, 4, 1); else PDES_MFMA_LAUNCH(1, 4, 1);
  } else if (kpad <= 16 || nt_total <= 4) {
    // cheap operand staging (one chunk) or few N-tiles: one N-tile per wave, split N over blockIdx.z
    grid.z = (nt_total + 3) / 4;
    if (twg == 2) PDES_MFMA_LAUNCH(2, 1, 1); else PDES_MFMA_LAUNCH(1, 1, 1);
  } else {
    grid.z = (nt_total + 7) / 8;
    if (twg == 2) PDES_MFMA_LAUNCH(2, 1, 2); else PDES_MFMA_LAUNCH(1, 1, 2);
  }
#undef PDES_MFMA_LAUNCH
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}

// returns PDES_ENOSUP when the shape is not covered (caller falls back to the direct kernels)
int conv_forward_mfma(const pdes_conv_desc& d, hipStream_t st) {
  if (d.nrep != PDES_NREP) return PDES_EINVAL;
  if (!d.wm_fwd || !mfma_shape_ok(d, false) || d.Cin < 16) return PDES_ENOSUP;
  if (d.ksize == 5) return d.upsample ? PDES_ENOSUP : launch_mfma<5, MODE_FWD>(d, d.wm_fwd, st);
  return d.ksize == 3 ? launch_mfma<3, MODE_FWD>(d, d.wm_fwd, st) : launch_mfma<1, MODE_FWD>(d, d.wm_fwd, st);
}

int conv_backward_data_mfma(const pdes_conv_desc& d, hipStream_t st) {
  if (!d.wm_bwd || !mfma_shape_ok(d, true) || d.eval_mode) return PDES_ENOSUP;
  if (d.upsample && d.ksize != 3) return PDES_ENOSUP;
  if (d.ksize == 5) return launch_mfma<5, MODE_BWD>(d, d.wm_bwd, st);
  return d.ksize == 3 ? launch_mfma<3, MODE_BWD>(d, d.wm_bwd, st) : launch_mfma<1, MODE_BWD>(d, d.wm_bwd, st);
}

}  // namespace pdes

using namespace pdes;

extern "C" int pdes_pack_weights_mfma(const pdes_mfma_pack_item* items, int n, int max_elems, void* stream) {
  if (!items || n <= 0 || max_elems <= 0) return PDES_EINVAL;
  int gx = cdiv(max_elems, 256);
  gx = gx > 128 ? 128 : gx;
  hipLaunchKernelGGL(pack_mfma_kernel, dim3(gx, n), dim3(256), 0, static_cast<hipStream_t>(stream), items);
  PDES_LAUNCH_CHECK();
  return PDES_OK;
}
