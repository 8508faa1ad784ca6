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

__device__ __forceinline__ void pack_mfma_item(const pdes_mfma_pack_item& it, int bx, int nbx) {
  const int ntf = (((it.Cout + 15) / 16) + 7) & ~7, ksf = ((it.Cin + 15) / 16) * 4;
  const int totf = ksf * it.kk * ntf * 64;
  for (int i = bx * 256 + threadIdx.x; i < totf; i += nbx * 256) {
    const int l = i & 63, nt = (i >> 6) % ntf, t = ((i >> 6) / ntf) % it.kk, ks = (i >> 6) / (ntf * it.kk);
    const int co = nt * 16 + (l & 15), ci = 4 * ks + (l >> 4);
    it.wm_fwd[i] = (co < it.Cout && ci < it.Cin) ? it.w[((size_t)co * it.Cin + ci) * it.kk + t] : 0.f;
  }
  if (!it.wm_bwd) return;
  const int ntb = (((it.Cin + 15) / 16) + 7) & ~7, ksb = ((it.Cout + 15) / 16) * 4;
  const int totb = ksb * it.kk * ntb * 64;
  for (int i = bx * 256 + threadIdx.x; i < totb; i += nbx * 256) {
    const int l = i & 63, nt = (i >> 6) % ntb, t = ((i >> 6) / ntb) % it.kk, ks = (i >> 6) / (ntb * it.kk);
    const int ci = nt * 16 + (l & 15), co = 4 * ks + (l >> 4);
    it.wm_bwd[i] = (co < it.Cout && ci < it.Cin) ? it.w[((size_t)co * it.Cin + ci) * it.kk + (it.kk - 1 - t)] : 0.f;
  }
}

// R(d, i): the 3x3 taps that land on position i of the 2x2 kernel of parity d
__device__ __forceinline__ float weff(const float* w9, int dy, int dx, int a, int b) {
  const int y0 = dy == 0 ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2), y1 = dy == 0 ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2);
  const int x0 = dx == 0 ? (b == 0 ? 0 : 1) : (b == 0 ? 0 : 2), x1 = dx == 0 ? (b == 0 ? 0 : 2) : (b == 0 ? 1 : 2);
  float s = 0.f;
  for (int ky = y0; ky <= y1; ++ky)
    for (int kx = x0; kx <= x1; ++kx) s += w9[ky * 3 + kx];
  return s;
}

__device__ __forceinline__ void pack_up_item(const pdes_up_pack_item& it, int bx, int nbx) {
  const int ntf = (((it.Cout + 15) / 16) + 7) & ~7, ksf = ((it.Cin + 15) / 16) * 4;
  const int totf = ksf * 16 * ntf * 64;
  for (int i = bx * 256 + threadIdx.x; i < totf; i += nbx * 256) {
    const int l = i & 63, nt = (i >> 6) % ntf, q = ((i >> 6) / ntf) % 16, ks = (i >> 6) / (ntf * 16);
    const int co = nt * 16 + (l & 15), ci = 4 * ks + (l >> 4);
    const int p = q >> 2, a = (q >> 1) & 1, b = q & 1;
    it.wu_fwd[i] = (co < it.Cout && ci < it.Cin) ? weff(it.w + ((size_t)co * it.Cin + ci) * 9, p >> 1, p & 1, a, b) : 0.f;
  }
  const int ntb = (((it.Cin + 15) / 16) + 7) & ~7, ksb = ((it.Cout + 15) / 16) * 4;
  const int totb = ksb * 16 * ntb * 64;
  for (int i = bx * 256 + threadIdx.x; i < totb; i += nbx * 256) {
    const int l = i & 63, nt = (i >> 6) % ntb, q = ((i >> 6) / ntb) % 16, ks = (i >> 6) / (ntb * 16);
    const int ci = nt * 16 + (l & 15), co = 4 * ks + (l >> 4);
    const int p = q >> 2, a = (q >> 1) & 1, b = q & 1;
    it.wu_bwd[i] = (co < it.Cout && ci < it.Cin) ? weff(it.w + ((size_t)co * it.Cin + ci) * 9, p >> 1, p & 1, a, b) : 0.f;
  }
}

}  // namespace pdes
