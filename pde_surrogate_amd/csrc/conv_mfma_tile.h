// Tile geometry and BatchNorm coefficients shared by the f32 matrix-core convolution kernels (conv_mfma.hip,
// conv_mfma_mirror.hip).
#pragma once
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

}  // namespace pdes
