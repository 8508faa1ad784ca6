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
int conv_backward_data_up_mfma(const pdes_conv_desc& d, hipStream_t st, bool dry = false);

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
  // Weight gradients are released to the second stream in batches of PDES_WGRAD_BATCH layers (default 1:
  // one event per layer; each event record is a barrier packet worth a ~6 us bubble on the main stream,
  // but releasing the work early measured as good as batching it: 2.45 / 2.47 / 2.50 ms per step at
  // batch 1 / 2 / 4).  PDES_WGRAD_REDUCE=batch reduces each batch's split-K partials on the second
  // stream instead of once at the end (measured slower: 40 small launches).
  const int batch_layers = getenv("PDES_WGRAD_BATCH") ? atoi(getenv("PDES_WGRAD_BATCH")) : 1;
  const double batch_flops = getenv("PDES_WGRAD_BATCH_GF") ? 1e9 * atof(getenv("PDES_WGRAD_BATCH_GF")) : 1.5e9;
  const bool reduce_per_batch = getenv("PDES_WGRAD_REDUCE") && getenv("PDES_WGRAD_REDUCE")[0] == 'b';
  // The split-K partials of the last layers (the widest ones: LastTransUp holds ~3/4 of the weights) are reduced
  // on the second stream as soon as those layers are done; only the rest waits for the end of the chain.
  long long per_total = 0, per_done = 0;
  int n_items = 0, early_lo = -1;          // early_lo: first table row already reduced early (-1: none yet)
  if (reduce_items && reduce_index)
    for (int i = 0; i < n; ++i)
      if (reduce_index[i] >= 0) {
        per_total += (long long)descs[i].Cout * descs[i].Cin * descs[i].ksize * descs[i].ksize;
        ++n_items;
      }
  int pend_hi = -1;                       // layers [i, pend_hi] are finalized but their weight gradient is not enqueued
  double pend_flops = 0.0;
  size_t nev = 0;
  auto flush = [&](int lo) -> int {       // enqueue weight gradients (and their split-K reduce) of layers pend_hi .. lo
    if (pend_hi < 0) return PDES_OK;
    // the very last weight gradient (first layer) has nothing left to overlap with: keep it on the main stream
    // and save the event hop (its scratch / dw are disjoint from what the second stream still works on)
    const bool last_on_main = fork && lo == 0 && pend_hi == 0 && !reduce_per_batch;
    if (fork && !last_on_main) {
      hipEvent_t e = chain_event(nev++);
      if (!e) return (int)hipErrorOutOfMemory;
      hipError_t he = hipEventRecord(e, st);
      if (he == hipSuccess) he = hipStreamWaitEvent(ws, e, 0);
      if (he != hipSuccess) return (int)he;
    }
    int r_lo = -1, r_hi = -1;
    long long max_n = 0;
    for (int i = pend_hi; i >= lo; --i) {
      const int rc = pdes_conv_backward_weight(&descs[i], 1, last_on_main ? st : ws);
      if (rc) return rc;
      if (fork && reduce_per_batch && reduce_items && reduce_index && reduce_index[i] >= 0) {
        const int k = reduce_index[i];
        r_lo = r_lo < 0 ? k : (k < r_lo ? k : r_lo);
        r_hi = k > r_hi ? k : r_hi;
        const long long per = (long long)descs[i].Cout * descs[i].Cin * descs[i].ksize * descs[i].ksize;
        max_n = per > max_n ? per : max_n;
      }
    }
    if (r_lo >= 0) {                      // item indices grow with the layer index: the batch is one contiguous slice
      const int rc = pdes_wgrad_reduce_all(reduce_items + r_lo, r_hi - r_lo + 1, (int)max_n, ws);
      if (rc) return rc;
    }
    if (fork && !reduce_per_batch && early_lo < 0 && reduce_items && reduce_index && lo > 0) {
      for (int i = pend_hi; i >= lo; --i)
        if (reduce_index[i] >= 0) per_done += (long long)descs[i].Cout * descs[i].Cin * descs[i].ksize * descs[i].ksize;
      if (5 * per_done >= 3 * per_total) {        // >= 60 % of the partials exist: reduce them now, off the critical path
        int first = -1;
        long long mx = 0;
        for (int i = lo; i < n; ++i)
          if (reduce_index[i] >= 0) {
            if (first < 0) first = reduce_index[i];
            const long long per = (long long)descs[i].Cout * descs[i].Cin * descs[i].ksize * descs[i].ksize;
            mx = per > mx ? per : mx;
          }
        if (first >= 0) {
          const int rc = pdes_wgrad_reduce_all(reduce_items + first, n_items - first, (int)mx, ws);
          if (rc) return rc;
          early_lo = first;
        }
      }
    }
    pend_hi = -1;
    pend_flops = 0.0;
    return PDES_OK;
  };
  // BatchNorm-backward finalize: the in-place kernel by default.  PDES_FUSE_FINALIZE=1 fuses it into the operand
  // load of the layer's two consumers when both run on the matrix-core kernels (bn_fused.h; layers with up to
  // PDES_FUSE_MAXC = 16 output channels).  Same-box A/B of the final kernel set: 2.188 ms per step fused vs 2.151 ms
  // with the separate kernel (x staged next to T costs the consumers more than 20 small launches), so it is opt-in.
  const bool fuse_on = getenv("PDES_FUSE_FINALIZE") && getenv("PDES_FUSE_FINALIZE")[0] == '1';
  const int fuse_maxc = getenv("PDES_FUSE_MAXC") ? atoi(getenv("PDES_FUSE_MAXC")) : 16;
  std::vector<pdes_conv_desc> local(descs, descs + n);
  for (int i = 0; i < n; ++i) {
    pdes_conv_desc& d = local[i];
    d.g_fused = 0;
    if (!fuse_on || force_direct() || !d.fin_tstats || !d.fin_xstats || !d.out) continue;
    if (d.Cout > fuse_maxc) continue;     // wide layers: staging x next to T costs the consumers more than the kernel saves
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
    if (pend_hi < 0) pend_hi = i;
    pend_flops += 2.0 * d.B * d.Hout * d.Wout * (double)d.Cout * d.Cin * d.ksize * d.ksize;
    if (!fork || pend_hi - i + 1 >= batch_layers || pend_flops >= batch_flops || i == 0) {
      const int rc = flush(i);
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
  if ((!fork || !reduce_per_batch) && reduce_items && reduce_index) {   // a single reduce over every layer at the end
    int cnt = 0;
    long long max_n = 0;
    for (int i = 0; i < n; ++i)
      if (reduce_index[i] >= 0 && (early_lo < 0 || reduce_index[i] < early_lo)) {
        ++cnt;
        const long long per = (long long)descs[i].Cout * descs[i].Cin * descs[i].ksize * descs[i].ksize;
        max_n = per > max_n ? per : max_n;
      }
    if (cnt) {
      const int rc = pdes_wgrad_reduce_all(reduce_items, cnt, (int)max_n, st);
      if (rc) return rc;
    }
  }
  return PDES_OK;
}
