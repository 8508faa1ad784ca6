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
int conv_backward_data_mfma(const pdes_conv_desc& d, hipStream_t st, bool dry = false);   // dry: capability query only
int conv_backward_weight_mfma(const pdes_conv_desc& d, hipStream_t st, bool dry = false);
int conv_forward_up_mfma(const pdes_conv_desc& d, hipStream_t st);        // nearest-x2 + 3x3, sub-pixel form
int conv_forward_fewout(const pdes_conv_desc& d, hipStream_t st);         // 5x5 with <= 3 output channels
int conv_forward_b3(const pdes_conv_desc& d, hipStream_t st);             // wide 3x3 layers: bf16 x3 split (conv_mfma_b3.hip)
int conv_backward_data_b3(const pdes_conv_desc& d, hipStream_t st, bool dry = false);
int conv_backward_data_up_mfma(const pdes_conv_desc& d, hipStream_t st, bool dry = false);
int conv_forward_1x1(const pdes_conv_desc& d, hipStream_t st);            // 1x1 layers without an LDS tile (conv_mfma_1x1.hip)
int conv_backward_data_1x1(const pdes_conv_desc& d, hipStream_t st, bool dry = false);

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
    if (rc == PDES_ENOSUP && !force_direct()) rc = conv_forward_fewout(descs[i], st);
    if (rc == PDES_ENOSUP && !force_direct()) rc = conv_forward_b3(descs[i], st);
    if (rc == PDES_ENOSUP && !force_direct()) rc = conv_forward_1x1(descs[i], st);
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
    if (rc == PDES_ENOSUP && descs[i].g_fused) return PDES_EINVAL;      // only the matrix-core kernels finalize on load
    if (rc == PDES_ENOSUP) {
      // the VALU kernel adds straight into dw.  If the caller planned deferred split-K partials for this layer
      // (pdes_conv_wgrad_plan said yes, e.g. before PDES_CONV_IMPL changed), its reduce must then add zeros
      if (descs[i].ws_defer && descs[i].ws && descs[i].ws_bytes > 0) {
        const hipError_t he = hipMemsetAsync(descs[i].ws, 0, (size_t)descs[i].ws_bytes, st);
        if (he != hipSuccess) return (int)he;
      }
      rc = conv_backward_weight_direct(descs[i], st);
    }
    if (rc) return rc;
  }
  return PDES_OK;
}

extern "C" int pdes_conv_backward_data(const pdes_conv_desc* descs, int n, void* stream) {
  if (!descs || n <= 0) return PDES_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int i = 0; i < n; ++i) {
    int rc = force_direct() ? PDES_ENOSUP : conv_backward_data_up_mfma(descs[i], st);
    if (rc == PDES_ENOSUP && !force_direct()) rc = conv_backward_data_b3(descs[i], st);
    if (rc == PDES_ENOSUP && !force_direct()) rc = conv_backward_data_1x1(descs[i], st);
    if (rc == PDES_ENOSUP && !force_direct()) rc = conv_backward_data_mfma(descs[i], st);
    if (rc == PDES_ENOSUP && descs[i].g_fused) return PDES_EINVAL;
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

extern "C" int pdes_backward(const pdes_conv_desc* descs, int n, void* stream, void* wgrad_stream,
                             const pdes_reduce_item* reduce_items, const int* reduce_index) {
  if (!descs || n <= 0) return PDES_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipStream_t ws = wgrad_stream ? static_cast<hipStream_t>(wgrad_stream) : st;
  const bool fork = ws != st;
  const bool have_red = reduce_items && reduce_index;
  auto per_of = [&](int i) { return (long long)descs[i].Cout * descs[i].Cin * descs[i].ksize * descs[i].ksize; };

  // ---- second-stream schedule.  A layer's weight gradient is released (one event) as soon as its output gradient
  // exists.  Measured alternatives, all slower: releasing in batches of 2/4/8 layers, reducing the split-K partials
  // per batch, several early reduces, and holding the heavy layers' weight gradients back until the main chain is
  // among the small dense layers (2.24 vs 2.12 ms per step, same box).
  long long per_total = 0, per_done = 0;
  int n_items = 0, early_lo = -1;          // early_lo: first table row already reduced early (-1: none yet)
  if (have_red)
    for (int i = 0; i < n; ++i)
      if (reduce_index[i] >= 0) { per_total += per_of(i); ++n_items; }
  std::vector<char> enq(n, 0);
  size_t nev = 0;
  auto release = [&](const int* layers, int cnt, bool on_main) -> int {
    if (cnt <= 0) return PDES_OK;
    if (fork && !on_main) {
      hipEvent_t e = chain_event(nev++);
      if (!e) return (int)hipErrorOutOfMemory;
      hipError_t he = hipEventRecord(e, st);
      if (he == hipSuccess) he = hipStreamWaitEvent(ws, e, 0);
      if (he != hipSuccess) return (int)he;
    }
    for (int k = 0; k < cnt; ++k) {
      const int i = layers[k];
      const int rc = pdes_conv_backward_weight(&descs[i], 1, on_main ? st : ws);
      if (rc) return rc;
      enq[i] = 1;
      if (have_red && reduce_index[i] >= 0) per_done += per_of(i);
    }
    // The split-K partials of the last layers (the widest ones: LastTransUp holds ~3/4 of the weights) are reduced
    // on the second stream as soon as >= 60 % of the weights' partials exist; the rest waits for the end.
    if (fork && !on_main && have_red && early_lo < 0 && 5 * per_done >= 3 * per_total) {
      int k0 = n;                                   // longest suffix of layers that are all enqueued
      while (k0 > 0 && enq[k0 - 1]) --k0;
      int first = -1;
      long long mx = 0;
      for (int i = k0; i < n; ++i)
        if (reduce_index[i] >= 0) {
          if (first < 0) first = reduce_index[i];
          mx = per_of(i) > mx ? per_of(i) : mx;
        }
      if (first > 0 && 5 * per_done >= 3 * per_total) {
        long long cover = 0;
        for (int i = k0; i < n; ++i) if (reduce_index[i] >= 0) cover += per_of(i);
        if (5 * cover >= 3 * per_total) {
          const int rc = pdes_wgrad_reduce_all(reduce_items + first, n_items - first, (int)mx, ws);
          if (rc) return rc;
          early_lo = first;
        }
      }
    }
    return PDES_OK;
  };

  // BatchNorm-backward finalize: the in-place kernel by default.  PDES_FUSE_FINALIZE=1 fuses it into the operand
  // load of the layer's two consumers when both run on the matrix-core kernels (bn_fused.h; layers with up to
  // PDES_FUSE_MAXC = 16 output channels).  Same-box A/B of the final kernel set: 2.188 ms per step fused vs 2.151 ms
  // with the separate kernel (x staged next to T costs the consumers more than 20 small launches), so it is opt-in.
  const bool fuse_on = getenv("PDES_FUSE_FINALIZE") && getenv("PDES_FUSE_FINALIZE")[0] == '1';
  const int fuse_maxc = getenv("PDES_FUSE_MAXC") ? atoi(getenv("PDES_FUSE_MAXC")) : 16;
  const int fuse_maxhw = getenv("PDES_FUSE_MAXHW") ? atoi(getenv("PDES_FUSE_MAXHW")) : (1 << 30);   // only maps up to this many pixels
  std::vector<pdes_conv_desc> local(descs, descs + n);
  for (int i = 0; i < n; ++i) {
    pdes_conv_desc& d = local[i];
    d.g_fused = 0;
    if (!fuse_on || force_direct() || !d.fin_tstats || !d.fin_xstats || !d.out) continue;
    if (d.Cout > fuse_maxc || d.Hout * d.Wout > fuse_maxhw) continue;     // wide layers: staging x next to T costs the consumers more than the kernel saves
    if (d.g_ctot != d.out_ctot || d.g_coff != d.out_coff || d.nrep != PDES_NREP) continue;
    const bool w_ok = conv_backward_weight_mfma(d, st, true) == PDES_OK;
    const bool d_ok = !d.has_bn || conv_backward_data_up_mfma(d, st, true) == PDES_OK ||
                      conv_backward_data_mfma(d, st, true) == PDES_OK;
    d.g_fused = (w_ok && d_ok) ? 1 : 0;
  }
  descs = local.data();

  for (int i = n - 1; i >= 0; --i) {
    const pdes_conv_desc& d = descs[i];
    if (d.fin_tstats && !d.g_fused) {
      int rc = pdes_bn_backward_finalize(const_cast<float*>(d.g), d.out, d.fin_xstats, d.fin_tstats, d.B, d.g_ctot,
                                         d.g_coff, d.g_coff + d.Cout, d.Hout * d.Wout, d.eps, d.nrep, d.rep_stride, st);
      if (rc) return rc;
    }
    // the very last weight gradient (first layer) has nothing left to overlap with: it stays on the main stream
    // (saves the event hop; its scratch / dw are disjoint from what the second stream still works on)
    {
      const int rc = release(&i, 1, fork && i == 0);
      if (rc) return rc;
    }
    if (d.has_bn) {
      const int rc = pdes_conv_backward_data(&d, 1, st);
      if (rc) return rc;
    }
  }
  if (fork) {
    hipEvent_t e = chain_event(nev++);
    if (!e) return (int)hipErrorOutOfMemory;
    hipError_t he = hipEventRecord(e, ws);
    if (he == hipSuccess) he = hipStreamWaitEvent(st, e, 0);
    if (he != hipSuccess) return (int)he;
  }
  if (have_red) {                          // the rows that were not reduced early: one launch at the end
    int cnt = 0;
    long long max_n = 0;
    for (int i = 0; i < n; ++i)
      if (reduce_index[i] >= 0 && (early_lo < 0 || reduce_index[i] < early_lo)) {
        ++cnt;
        max_n = per_of(i) > max_n ? per_of(i) : max_n;
      }
    if (cnt) {
      const int rc = pdes_wgrad_reduce_all(reduce_items, cnt, (int)max_n, st);
      if (rc) return rc;
    }
  }
  return PDES_OK;
}
