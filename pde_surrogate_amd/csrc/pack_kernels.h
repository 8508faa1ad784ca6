// Weight re-packing device routines shared by the per-layout entry points and the merged launch
// (pdes_pack_all): the packed images are rebuilt from the live weights once per step.
//   direct   (Cout,Cin,kk) -> w_fwd (Cin,kk,cout_pad), w_bwd (Cout,kk,cin_pad)          (conv_direct.hip)
//   MFMA     [(kstep*KK + tap)*NT + nt][kq*16 + n], NT = N-tiles padded to 8                (conv_mfma.hip)
//   sub-pixel effective 2x2 kernels of nearest-x2 + 3x3, same MFMA image shape            (conv_mfma_up.hip)
// `bx`, `nbx`: this block's index and the block count along x (grid-stride loops).
#pragma once
#include "pdes_common.h"
#include "../../include/pdes_hip.h"

namespace pdes {

__device__ __forceinline__ void pack_direct_item(const pdes_pack_item& it, int bx, int nbx) {
  const int total = it.Cout * it.Cin * it.kk;
  for (int i = bx * 256 + threadIdx.x; i < total; i += nbx * 256) {
    const int t = i % it.kk, ci = (i / it.kk) % it.Cin, co = i / (it.kk * it.Cin);
    const float v = it.w[i];
    it.w_fwd[((size_t)ci * it.kk + t) * it.cout_pad + co] = v;
    it.w_bwd[((size_t)co * it.kk + t) * it.cin_pad + ci] = v;
  }
}

// (the zero tiles that pad N to a multiple of 8 are written once, at allocation: only the real tiles are rebuilt)
__device__ __forceinline__ void pack_mfma_item(const pdes_mfma_pack_item& it, int bx, int nbx) {
  const int ntrf = (it.Cout + 15) / 16, ntf = (ntrf + 7) & ~7, ksf = ((it.Cin + 15) / 16) * 4;
  const int totf = ksf * it.kk * ntrf * 64;
  for (int i = bx * 256 + threadIdx.x; i < totf; i += nbx * 256) {
    const int l = i & 63, nt = (i >> 6) % ntrf, t = ((i >> 6) / ntrf) % it.kk, ks = (i >> 6) / (ntrf * it.kk);
    const int co = nt * 16 + (l & 15), ci = 4 * ks + (l >> 4);
    it.wm_fwd[((size_t)(ks * it.kk + t) * ntf + nt) * 64 + l] =
        (co < it.Cout && ci < it.Cin) ? it.w[((size_t)co * it.Cin + ci) * it.kk + t] : 0.f;
  }
  if (!it.wm_bwd) return;
  const int ntrb = (it.Cin + 15) / 16, ntb = (ntrb + 7) & ~7, ksb = ((it.Cout + 15) / 16) * 4;
  const int totb = ksb * it.kk * ntrb * 64;
  for (int i = bx * 256 + threadIdx.x; i < totb; i += nbx * 256) {
    const int l = i & 63, nt = (i >> 6) % ntrb, t = ((i >> 6) / ntrb) % it.kk, ks = (i >> 6) / (ntrb * it.kk);
    const int ci = nt * 16 + (l & 15), co = 4 * ks + (l >> 4);
    it.wm_bwd[((size_t)(ks * it.kk + t) * ntb + nt) * 64 + l] =
        (co < it.Cout && ci < it.Cin) ? it.w[((size_t)co * it.Cin + ci) * it.kk + (it.kk - 1 - t)] : 0.f;
  }
}

// R(d, i): the 3x3 taps that land on position i of the 2x2 kernel of parity d.  Rows (and, likewise, columns): parity 0:
// position 0 <- {0}, 1 <- {1, 2}; parity 1: position 0 <- {0, 1}, 1 <- {2}.  Branch free: the nine loads are
// unconditional and independent (with data-dependent loop bounds every tap was a serial round trip to L2, and the
// packing launch -- on the serial chain at the start of every step -- spent 12 us in these images).
__device__ __forceinline__ int weff_mask(int d, int i) { return d == 0 ? (i == 0 ? 1 : 6) : (i == 0 ? 3 : 4); }
__device__ __forceinline__ float weff(const float* w9, int dy, int dx, int a, int b) {
  const int rm = weff_mask(dy, a), cm = weff_mask(dx, b);
  float v[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) v[t] = w9[t];
  float s = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) s += ((rm >> ky) & (cm >> kx) & 1) ? v[ky * 3 + kx] : 0.f;
  return s;
}

__device__ __forceinline__ void pack_up_item(const pdes_up_pack_item& it, int bx, int nbx) {
  const int ntf = (((it.Cout + 15) / 16) + 7) & ~7, ksf = ((it.Cin + 15) / 16) * 4;
  const int totf = ksf * 16 * ntf * 64;
  for (int i = bx * 256 + threadIdx.x; i < totf; i += nbx * 256) {
    const int l = i & 63, nt = (i >> 6) % ntf, q = ((i >> 6) / ntf) % 16, ks = (i >> 6) / (ntf * 16);
    const int co = nt * 16 + (l & 15), ci = 4 * ks + (l >> 4);
    const int p = q >> 2, a = (q >> 1) & 1, b = q & 1;
    const float v = weff(it.w + ((size_t)min(co, it.Cout - 1) * it.Cin + min(ci, it.Cin - 1)) * 9, p >> 1, p & 1, a, b);
    it.wu_fwd[i] = (co < it.Cout && ci < it.Cin) ? v : 0.f;       // clamped address + select: no branch around the loads
  }
  const int ntb = (((it.Cin + 15) / 16) + 7) & ~7, ksb = ((it.Cout + 15) / 16) * 4;
  const int totb = ksb * 16 * ntb * 64;
  for (int i = bx * 256 + threadIdx.x; i < totb; i += nbx * 256) {
    const int l = i & 63, nt = (i >> 6) % ntb, q = ((i >> 6) / ntb) % 16, ks = (i >> 6) / (ntb * 16);
    const int ci = nt * 16 + (l & 15), co = 4 * ks + (l >> 4);
    const int p = q >> 2, a = (q >> 1) & 1, b = q & 1;
    const float v = weff(it.w + ((size_t)min(co, it.Cout - 1) * it.Cin + min(ci, it.Cin - 1)) * 9, p >> 1, p & 1, a, b);
    it.wu_bwd[i] = (co < it.Cout && ci < it.Cin) ? v : 0.f;
  }
}

typedef unsigned int u32;
// Three-way bf16 split of a PAIR of fp32 values with the hardware conversion (v_cvt_pk_bf16_f32, round to nearest
// even): each returned word holds the two bf16 terms (x0 in the low half, x1 in the high half) of one plane.
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32 cvt_pk_bf16(float a, float b) {
  const v2bf t = __builtin_convertvector((v2f){a, b}, v2bf);
  return *reinterpret_cast<const u32*>(&t);
}
__device__ __forceinline__ void split3_pair(float x0, float x1, u32& h, u32& m, u32& l) {
  h = cvt_pk_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = cvt_pk_bf16(r0, r1);
  const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
  l = cvt_pk_bf16(s0, s1);
}

// split weight images of the wide 3x3 layers (conv_mfma_b3.hip)
__device__ __forceinline__ void pack_b3_item(const pdes_b3_pack_item& it, int bx, int nbx) {
  for (int dir = 0; dir < 2; ++dir) {
    unsigned short* img = dir == 0 ? it.wb_fwd : it.wb_bwd;
    if (!img) continue;
    const int kC = dir == 0 ? it.Cin : it.Cout, nC = dir == 0 ? it.Cout : it.Cin;
    const int ntp = (((nC + 15) / 16) + 7) & ~7, nch = (kC + 31) / 32;
    const int total = nch * 9 * ntp * 64;                 // one thread per (chunk, tap, N-tile, lane)
    for (int e = bx * 256 + threadIdx.x; e < total; e += nbx * 256) {
      const int l = e & 63, nt = (e >> 6) % ntp, t = ((e >> 6) / ntp) % 9, ch = (e >> 6) / (ntp * 9);
      const int n = nt * 16 + (l & 15), k0 = ch * 32 + 8 * (l >> 4);
      u32 hw[4], mw[4], lw[4];
      float xv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        const int nc = min(n, nC - 1), kc = min(k, kC - 1);   // clamped address + select: eight independent loads
        const float x = dir == 0 ? it.w[((size_t)nc * it.Cin + kc) * 9 + t] : it.w[((size_t)kc * it.Cin + nc) * 9 + (8 - t)];
        xv[j] = (n < nC && k < kC) ? x : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) split3_pair(xv[2 * j], xv[2 * j + 1], hw[j], mw[j], lw[j]);
      unsigned short* q = img + (((size_t)(ch * 9 + t) * ntp + nt) * 3 * 64 + l) * 8;
      *reinterpret_cast<uint4*>(q) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
      *reinterpret_cast<uint4*>(q + 64 * 8) = make_uint4(mw[0], mw[1], mw[2], mw[3]);
      *reinterpret_cast<uint4*>(q + 2 * 64 * 8) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    }
  }
}

// split effective-weight image of the forward, rebuilt from the live weights every step:
//   [(chunk*16 + j)*NT + nt][plane][lane][e] = split_plane( Weff_p[a][b][co = 16 nt + (lane & 15)][ci = 32 chunk + 8 (lane >> 4) + e] )
// j walks the (tile position, parity) pairs in the order of the kernel's loop; NT = N-tiles rounded up to 4.
// ... and of the data gradient (conv_mfma_b3.hip, B3_UPBWD): K = (parity, output channel), N = input channels,
//   [((p * NCH + chunk) * 4 + a * 2 + b) * NT + nt][plane][lane][e] =
//       split_plane( Weff_p[a][b][co = 32 chunk + 8 (lane >> 4) + e][ci = 16 nt + (lane & 15)] ),   NT rounded up to 8
__device__ __forceinline__ void pack_b3up_bwd(const pdes_b3up_pack_item& it, int bx, int nbx) {
  const int ntp = (((it.Cin + 15) / 16) + 7) & ~7, nch = (it.Cout + 31) / 32;
  const int total = 4 * nch * 4 * ntp * 64;
  for (int e = bx * 256 + threadIdx.x; e < total; e += nbx * 256) {
    const int l = e & 63, nt = (e >> 6) % ntp, t = ((e >> 6) / ntp) & 3, vc = (e >> 6) / (ntp * 4);
    const int pp = vc / nch, ch = vc % nch;
    const int n = nt * 16 + (l & 15), k0 = ch * 32 + 8 * (l >> 4);
    u32 hw[4], mw[4], lw[4];
    float xv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = k0 + q;
      const float v = weff(it.w + ((size_t)min(k, it.Cout - 1) * it.Cin + min(n, it.Cin - 1)) * 9, pp >> 1, pp & 1, t >> 1, t & 1);
      xv[q] = (n < it.Cin && k < it.Cout) ? v : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) split3_pair(xv[2 * q], xv[2 * q + 1], hw[q], mw[q], lw[q]);
    unsigned short* dst = it.wbu_bwd + (((size_t)(vc * 4 + t) * ntp + nt) * 3 * 64 + l) * 8;
    *reinterpret_cast<uint4*>(dst) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
    *reinterpret_cast<uint4*>(dst + 64 * 8) = make_uint4(mw[0], mw[1], mw[2], mw[3]);
    *reinterpret_cast<uint4*>(dst + 2 * 64 * 8) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
  }
}

__device__ __forceinline__ void pack_b3up_item(const pdes_b3up_pack_item& it, int bx, int nbx) {
  if (it.wbu_bwd) pack_b3up_bwd(it, bx, nbx);
  if (!it.wbu_fwd) return;
  const int ntp = (((it.Cout + 15) / 16) + 3) & ~3, nch = (it.Cin + 31) / 32;
  const int total = nch * 16 * ntp * 64;
  for (int e = bx * 256 + threadIdx.x; e < total; e += nbx * 256) {
    const int l = e & 63, nt = (e >> 6) % ntp, j = ((e >> 6) / ntp) % 16, ch = (e >> 6) / (ntp * 16);
    int pp = 0, ia = 0, ib = 0, cnt = 0;                       // decode j in the kernel's walking order
    for (int ty = 0; ty < 3; ++ty)
      for (int tx = 0; tx < 3; ++tx)
        for (int ddy = 0; ddy < 2; ++ddy)
          for (int ddx = 0; ddx < 2; ++ddx) {
            const int a = ty - ddy, bb = tx - ddx;
            if (a < 0 || a > 1 || bb < 0 || bb > 1) continue;
            if (cnt == j) { pp = ddy * 2 + ddx; ia = a; ib = bb; }
            ++cnt;
          }
    const int n = nt * 16 + (l & 15), k0 = ch * 32 + 8 * (l >> 4);
    u32 hw[4], mw[4], lw[4];
    float xv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = k0 + q;
      const float v = weff(it.w + ((size_t)min(n, it.Cout - 1) * it.Cin + min(k, it.Cin - 1)) * 9, pp >> 1, pp & 1, ia, ib);
      xv[q] = (n < it.Cout && k < it.Cin) ? v : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) split3_pair(xv[2 * q], xv[2 * q + 1], hw[q], mw[q], lw[q]);
    unsigned short* dst = it.wbu_fwd + (((size_t)(ch * 16 + j) * ntp + nt) * 3 * 64 + l) * 8;
    *reinterpret_cast<uint4*>(dst) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
    *reinterpret_cast<uint4*>(dst + 64 * 8) = make_uint4(mw[0], mw[1], mw[2], mw[3]);
    *reinterpret_cast<uint4*>(dst + 2 * 64 * 8) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
  }
}


}  // namespace pdes
