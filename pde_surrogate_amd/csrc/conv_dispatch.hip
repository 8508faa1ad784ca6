// C-ABI entry points for the convolution family: walk a host array of descriptors and enqueue
// one kernel per descriptor on the caller's stream.  Kernel selection lives here so the Python
// side never needs to know which implementation (MFMA or VALU) serves a shape.
#include "pdes_common.h"
#include "../../include/pdes_hip.h"

namespace pdes {
int conv_forward_direct(const pdes_conv_desc& d, hipStream_t st);
int conv_backward_data_direct(const pdes_conv_desc& d, hipStream_t st);
int conv_backward_weight_direct(const pdes_conv_desc& d, hipStream_t st);
}  // namespace pdes

using namespace pdes;

extern "C" int pdes_conv_forward(const pdes_conv_desc* descs, int n, void* stream) {
  if (!descs || n <= 0) return PDES_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int i = 0; i < n; ++i) {
    const int rc = conv_forward_direct(descs[i], st);
    if (rc) return rc;
  }
  return PDES_OK;
}

extern "C" int pdes_conv_backward_weight(const pdes_conv_desc* descs, int n, void* stream) {
  if (!descs || n <= 0) return PDES_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int i = 0; i < n; ++i) {
    const int rc = conv_backward_weight_direct(descs[i], st);
    if (rc) return rc;
  }
  return PDES_OK;
}

extern "C" int pdes_conv_backward_data(const pdes_conv_desc* descs, int n, void* stream) {
  if (!descs || n <= 0) return PDES_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int i = 0; i < n; ++i) {
    const int rc = conv_backward_data_direct(descs[i], st);
    if (rc) return rc;
  }
  return PDES_OK;
}
