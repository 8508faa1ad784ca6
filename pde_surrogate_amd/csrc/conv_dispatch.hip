// C-ABI entry points for the convolution family: walk a host array of descriptors and enqueue
// one kernel per descriptor on the caller's stream.  Kernel selection lives here so the Python
// side never needs to know which implementation (MFMA or VALU) serves a shape.
#include <stdlib.h>
#include <vector>
#include "pdes_common.h"
#include "../../include/pdes_hip.h"

namespace pdes {
int conv_forward_direct(const pdes_conv_desc& d, hipStream_t st);
int conv_backward_data_direct(const pdes_conv_desc& d, hipStream_t st);
int conv_backward_weight_direct(const pdes_conv_desc& d, hipStream_t st);
int conv_forward_mfma(const pdes_conv_desc& d, hipStream_t st);        // PDES_ENOSUP: shape not covered
int conv_backward_data_mfma(const pdes_conv_desc& d, hipStream_t st);
int conv_backward_weight_mfma(const pdes_conv_desc& d, hipStream_t st);
int conv_forward_up_mfma(const pdes_conv_desc& d, hipStream_t st);        // nearest-x2 + 3x3, sub-pixel form
int conv_backward_data_up_mfma(const pdes_conv_desc& d, hipStream_t st);

// PDES_CONV_IMPL=direct forces the VALU reference kernels (used by the GPU tests to cross-check
// the matrix-core kernels against them); anything else = automatic selection.
static bool force_direct() {
  const char* e = getenv("PDES_CONV_IMPL");
  return e && e[0] == 'd';
}
}  // namespace pdes

using namespace pdes;

extern "C" int pdes_conv_forward(const pdes_conv_desc* descs, int n, void* stream) {
  if (!descs || n <= 0) return PDES_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int i = 0; i < n; ++i) {
    int rc = force_direct() ? PDES_ENOSUP : conv_forward_up_mfma(descs[i], st);
    if (rc == PDES_ENOSUP && !force_direct()) rc = conv_forward_mfma(descs[i], st);
    if (rc == PDES_ENOSUP) rc = conv_forward_direct(descs[i], st);
    if (rc) return rc;
  }
  return PDES_OK;
}

extern "C" int pdes_conv_backward_weight(const pdes_conv_desc* descs, int n, void* stream) {
  if (!descs || n <= 0) return PDES_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int i = 0; i < n; ++i) {
    int rc = force_direct() ? PDES_ENOSUP : conv_backward_weight_mfma(descs[i], st);
    if (rc == PDES_ENOSUP) rc = conv_backward_weight_direct(descs[i], st);
    if (rc) return rc;
  }
  return PDES_OK;
}

extern "C" int pdes_conv_backward_data(const pdes_conv_desc* descs, int n, void* stream) {
  if (!descs || n <= 0) return PDES_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int i = 0; i < n; ++i) {
    int rc = force_direct() ? PDES_ENOSUP : conv_backward_data_up_mfma(descs[i], st);
    if (rc == PDES_ENOSUP && !force_direct()) rc = conv_backward_data_mfma(descs[i], st);
    if (rc == PDES_ENOSUP) rc = conv_backward_data_direct(descs[i], st);
    if (rc) return rc;
  }
  return PDES_OK;
}

// Events used to fork/join the weight-gradient stream (timing disabled: they only order work).
static hipEvent_t chain_event(size_t i) {
  static std::vector<hipEvent_t> pool;
  while (pool.size() <= i) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    pool.push_back(e);
  }
  return pool[i];
}

extern "C" int pdes_backward(const pdes_conv_desc* descs, int n, void* stream, void* wgrad_stream) {
  if (!descs || n <= 0) return PDES_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipStream_t ws = wgrad_stream ? static_cast<hipStream_t>(wgrad_stream) : st;
  const bool fork = ws != st;
  for (int i = n - 1; i >= 0; --i) {
    const pdes_conv_desc& d = descs[i];
    if (d.fin_tstats) {
      int rc = pdes_bn_backward_finalize(const_cast<float*>(d.g), d.out, d.fin_xstats, d.fin_tstats, d.B, d.g_ctot,
                                         d.g_coff, d.g_coff + d.Cout, d.Hout * d.Wout, d.eps, d.nrep, d.rep_stride, st);
      if (rc) return rc;
    }
    if (fork) {
      hipEvent_t e = chain_event(static_cast<size_t>(i));
      if (!e) return (int)hipErrorOutOfMemory;
      hipError_t he = hipEventRecord(e, st);
      if (he == hipSuccess) he = hipStreamWaitEvent(ws, e, 0);
      if (he != hipSuccess) return (int)he;
    }
    int rc = pdes_conv_backward_weight(&d, 1, ws);
    if (rc) return rc;
    if (d.has_bn) {
      rc = pdes_conv_backward_data(&d, 1, st);
      if (rc) return rc;
    }
  }
  if (fork) {
    hipEvent_t e = chain_event(static_cast<size_t>(n));
    if (!e) return (int)hipErrorOutOfMemory;
    hipError_t he = hipEventRecord(e, ws);
    if (he == hipSuccess) he = hipStreamWaitEvent(st, e, 0);
    if (he != hipSuccess) return (int)he;
  }
  return PDES_OK;
}
