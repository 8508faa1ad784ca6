// BatchNorm-backward "finalize" applied on operand load (gfx950 kernels of this library).
//
// The backward pass keeps, per activation buffer, the accumulator T_c = sum over consumer BNs of
// gamma * dz * relu-mask (include/pdes_hip.h).  The gradient wrt the raw activation is
//     g = invstd * (T - mean(T) - xhat * mean(T xhat)),   xhat = (x - mean) * invstd
// pdes_bn_backward_finalize rewrites T -> g in place (3 passes over the buffer, one launch per
// layer on the critical chain).  A kernel that consumes g can instead read T and x and apply the
// same expression while staging its operand: d.g_fused = 1 (set by pdes_backward when every consumer
// of the layer's gradient supports it).  Same arithmetic as bn_bwd_finalize_kernel.
#pragma once
#include "pdes_common.h"
#include "../../include/pdes_hip.h"

namespace pdes {

// {mean, invstd, mean(T), mean(T xhat)} of OUTPUT channel `co` of the layer described by d
__device__ __forceinline__ float4 fin_coef(const pdes_conv_desc& d, int co) {
  const int c = d.out_coff + co;
  const double n = (double)d.B * d.Hout * d.Wout;
  const double m = rep_sum(d.fin_xstats, 2 * c, d.nrep, d.rep_stride) / n;
  double var = rep_sum(d.fin_xstats, 2 * c + 1, d.nrep, d.rep_stride) / n - m * m;
  var = var < 0.0 ? 0.0 : var;
  const double m1 = rep_sum(d.fin_tstats, 2 * c, d.nrep, d.rep_stride) / n;
  const double m2 = rep_sum(d.fin_tstats, 2 * c + 1, d.nrep, d.rep_stride) / n;
  return make_float4((float)m, (float)(1.0 / sqrt(var + (double)d.eps)), (float)m1, (float)m2);
}

__device__ __forceinline__ float fin_apply(const float4& k, float t, float x) {
  return k.y * (t - k.z - (x - k.x) * k.y * k.w);
}

}  // namespace pdes
