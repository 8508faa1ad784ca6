// C-ABI entry points for the convolution family: walk a host array of descriptors and enqueue
// one kernel per descriptor on the caller's stream.  Kernel selection lives here so the Python
// side never needs to know which implementation (MFMA or VALU) serves a shape.
#include <stdlib.h>
#include "pdes_common.h"
#include "pdes_options.h"
#include "../../include/pdes_hip.h"

namespace pdes {
int conv_forward_direct(const pdes_conv_desc& d, hipStream_t st);
int conv_backward_data_direct(const pdes_conv_desc& d, hipStream_t st);
int conv_backward_weight_direct(const pdes_conv_desc& d, hipStream_t st);
bool conv_backward_weight_first_applies(const pdes_conv_desc& d);
bool first_layer_partials(const pdes_conv_desc& d);     // conv_direct.hip: the 7x7 first layer writes per-image partials
bool conv_forward_direct_first7(const pdes_conv_desc& d);               // conv_direct.hip: reads the live weights, no image
int conv_forward_mfma(const pdes_conv_desc& d, hipStream_t st, bool dry = false);        // PDES_ENOSUP: shape not covered
int conv_backward_data_mfma(const pdes_conv_desc& d, hipStream_t st, bool dry = false);   // dry: capability query only
void set_dgrad_stop_event(hipEvent_t e);        // conv_mfma.hip: completion signal for the next finalize-on-load data gradient
bool dgrad_stop_event_pending();
int conv_backward_weight_mfma(const pdes_conv_desc& d, hipStream_t st, bool dry = false);
int conv_forward_up_mfma(const pdes_conv_desc& d, hipStream_t st, bool dry = false);        // nearest-x2 + 3x3, sub-pixel form
int conv_forward_fewout(const pdes_conv_desc& d, hipStream_t st, bool dry = false);         // 5x5 with <= 3 output channels
int conv_forward_b3(const pdes_conv_desc& d, hipStream_t st, bool dry = false);             // wide 3x3 layers: bf16 x3 split (conv_mfma_b3.hip)
int conv_forward_b3_up(const pdes_conv_desc& d, hipStream_t st, bool dry = false);          // nearest-x2 + 3x3, sub-pixel form, bf16 x3 split
int conv_backward_data_b3(const pdes_conv_desc& d, hipStream_t st, bool dry = false);
int conv_backward_data_b3_up(const pdes_conv_desc& d, hipStream_t st, bool dry = false);     // sub-pixel data gradient, bf16 x3 split
int conv_backward_data_up_mfma(const pdes_conv_desc& d, hipStream_t st, bool dry = false);
int conv_forward_small(const pdes_conv_desc& d, hipStream_t st, bool dry = false);          // 3x3 on 8x8 maps (conv_small.hip)
int conv_backward_data_small(const pdes_conv_desc& d, hipStream_t st, bool dry = false);
int conv_backward_weight_small(const pdes_conv_desc& d, hipStream_t st);
bool wgrad_small_applies(const pdes_conv_desc& d);
bool wgrad_small_ready(const pdes_conv_desc& d);
int conv_forward_1x1(const pdes_conv_desc& d, hipStream_t st, bool dry = false);            // 1x1 layers without an LDS tile (conv_mfma_1x1.hip)
int conv_backward_data_1x1(const pdes_conv_desc& d, hipStream_t st, bool dry = false);
int upsample_bilinear_forward(const pdes_conv_desc& d, hipStream_t st);   // PDES_UPSAMPLE_BILINEAR_OP descriptors
int upsample_bilinear_backward(const pdes_conv_desc& d, hipStream_t st);
int channel_mask_forward(const pdes_conv_desc& d, hipStream_t st);        // PDES_OP_CHANNEL_MASK descriptors (Dropout2d)
int channel_mask_backward(const pdes_conv_desc& d, hipStream_t st);
// descriptors that are not convolutions: no weights, their own forward / backward kernels
int bn_backward_finalize_launch(const pdes_context* ctx, float* t, const float* x, const double* x_stats,
                                const double* t_stats, int B, int ctot, int c0, int c1, int HW, float eps, int nrep,
                                long long rep_stride, hipStream_t st, hipEvent_t done, const float* add, const float* coef);
// flow operators of the conditional Glow (flow_ops.hip)
int flow_copy_forward(const pdes_conv_desc& d, hipStream_t st);
int flow_copy_backward(const pdes_conv_desc& d, hipStream_t st);
int flow_bias_scale_forward(const pdes_conv_desc& d, hipStream_t st);
int flow_bias_scale_backward(const pdes_conv_desc& d, hipStream_t st);
int flow_coupling_forward(const pdes_conv_desc& d, hipStream_t st);
int flow_coupling_backward(const pdes_conv_desc& d, hipStream_t st);
int flow_mix_forward(const pdes_conv_desc& d, hipStream_t st);
int flow_mix_backward(const pdes_conv_desc& d, hipStream_t st);
int flow_unsqueeze_forward(const pdes_conv_desc& d, hipStream_t st);
int flow_unsqueeze_backward(const pdes_conv_desc& d, hipStream_t st);
int flow_gauss_forward(const pdes_conv_desc& d, hipStream_t st);
int flow_gauss_backward(const pdes_conv_desc& d, hipStream_t st);
static bool is_resample_op(const pdes_conv_desc& d) { return d.upsample >= PDES_UPSAMPLE_BILINEAR_OP && d.upsample <= PDES_OP_GAUSS; }
static int op_forward(const pdes_conv_desc& d, hipStream_t st) {
  switch (d.upsample) {
    case PDES_OP_CHANNEL_MASK: return channel_mask_forward(d, st);
    case PDES_OP_COPY: return flow_copy_forward(d, st);
    case PDES_OP_BIAS_SCALE: return flow_bias_scale_forward(d, st);
    case PDES_OP_COUPLING: return flow_coupling_forward(d, st);
    case PDES_OP_MIX: return flow_mix_forward(d, st);
    case PDES_OP_UNSQUEEZE: return flow_unsqueeze_forward(d, st);
    case PDES_OP_GAUSS: return flow_gauss_forward(d, st);
    default: return upsample_bilinear_forward(d, st);
  }
}
static int op_backward(const pdes_conv_desc& d, hipStream_t st) {
  switch (d.upsample) {
    case PDES_OP_CHANNEL_MASK: return channel_mask_backward(d, st);
    case PDES_OP_COPY: return flow_copy_backward(d, st);
    case PDES_OP_BIAS_SCALE: return flow_bias_scale_backward(d, st);
    case PDES_OP_COUPLING: return flow_coupling_backward(d, st);
    case PDES_OP_MIX: return flow_mix_backward(d, st);
    case PDES_OP_UNSQUEEZE: return flow_unsqueeze_backward(d, st);
    case PDES_OP_GAUSS: return flow_gauss_backward(d, st);
    default: return upsample_bilinear_backward(d, st);
  }
}

// option PDES_CONV_IMPL=direct forces the VALU reference kernels (used by the GPU tests to cross-check
// the matrix-core kernels against them); anything else = automatic selection.
static bool force_direct() { return opt().conv_direct != 0; }

// PDES_FIN_ONLOAD: does this layer's backward apply the BatchNorm-backward finalize of its output gradient on operand
// load (no finalize launch)?  The dense blocks' layers: 3x3, stride 1, <= 16 output channels (ONE chunk of the data
// gradient's K dimension, one N-tile of the weight gradient), both passes on the matrix-core kernels that implement it
// (capability queries: nothing is enqueued).  `t` receives the descriptor with g_fused = 1.
static bool fin_onload(const pdes_conv_desc& d, pdes_conv_desc* t) {
  if (!opt().fin_onload || force_direct() || is_resample_op(d) || !d.fin_tstats || d.g_fused) return false;
  // the first convolution: no data gradient, its weight-gradient kernel stages the gradient planes itself
  if (opt().fin_onload >= 3 && !d.has_bn && !d.t_in && !d.g_add && d.nrep == PDES_NREP && d.g_ctot == d.out_ctot && d.g_coff == d.out_coff &&
      conv_backward_weight_first_applies(d) && !wgrad_small_applies(d) &&
      conv_backward_weight_mfma(d, nullptr, true) == PDES_ENOSUP) {
    *t = d;
    t->g_fused = 1;
    return true;
  }
  if (d.ksize != 3 || d.stride != 1 || d.upsample || d.Cout > 16 || !d.has_bn || !d.t_in || d.g_add) return false;
  *t = d;
  t->g_fused = 1;
  if (conv_backward_data_small(*t, nullptr, true) == PDES_OK) return wgrad_small_ready(*t);      // the 8 x 8 maps (conv_small.hip)
  return conv_backward_data_mfma(*t, nullptr, true) == PDES_OK && conv_backward_weight_mfma(*t, nullptr, true) == PDES_OK;
}
}  // namespace pdes

using namespace pdes;

extern "C" int pdes_conv_forward(const pdes_context* ctx, const pdes_conv_desc* descs, int n, void* stream) {
  if (!descs || n <= 0) return PDES_EINVAL;
  OptScope scope(ctx);
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int i = 0; i < n; ++i) {
    if (is_resample_op(descs[i])) {
      const int rc = op_forward(descs[i], st);
      if (rc) return rc;
      continue;
    }
    int rc = force_direct() ? PDES_ENOSUP : conv_forward_small(descs[i], st);
    if (rc == PDES_ENOSUP && !force_direct()) rc = conv_forward_b3_up(descs[i], st);
    if (rc == PDES_ENOSUP && !force_direct()) rc = conv_forward_up_mfma(descs[i], st);
    if (rc == PDES_ENOSUP && !force_direct()) rc = conv_forward_fewout(descs[i], st);
    if (rc == PDES_ENOSUP && !force_direct()) rc = conv_forward_b3(descs[i], st);
    if (rc == PDES_ENOSUP && !force_direct()) rc = conv_forward_1x1(descs[i], st);
    if (rc == PDES_ENOSUP && !force_direct()) rc = conv_forward_mfma(descs[i], st);
    if (rc == PDES_ENOSUP) rc = conv_forward_direct(descs[i], st);
    if (rc) return rc;
  }
  return PDES_OK;
}

extern "C" int pdes_conv_backward_weight(const pdes_context* ctx, const pdes_conv_desc* descs, int n, void* stream) {
  if (!descs || n <= 0) return PDES_EINVAL;
  OptScope scope(ctx);
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int i = 0; i < n; ++i) {
    if (is_resample_op(descs[i])) continue;                          // a resampling op has no weights
    int rc = force_direct() ? PDES_ENOSUP : conv_backward_weight_small(descs[i], st);
    if (rc == PDES_ENOSUP && !force_direct()) rc = conv_backward_weight_mfma(descs[i], st);
    if (rc == PDES_ENOSUP && descs[i].g_fused && !conv_backward_weight_first_applies(descs[i]))
      return PDES_EINVAL;                                               // (of the VALU kernels only the first convolution's finalizes on load)
    if (rc == PDES_ENOSUP) {
      // the VALU kernel adds straight into dw.  If the caller planned deferred split-K partials for this layer
      // (pdes_conv_wgrad_plan said yes, e.g. before PDES_CONV_IMPL changed), its reduce must then add zeros
      if (descs[i].ws_defer && descs[i].ws && descs[i].ws_bytes > 0 && !first_layer_partials(descs[i])) {
        const hipError_t he = hipMemsetAsync(descs[i].ws, 0, (size_t)descs[i].ws_bytes, st);
        if (he != hipSuccess) return (int)he;
      }
      rc = conv_backward_weight_direct(descs[i], st);
    }
    if (rc) return rc;
  }
  return PDES_OK;
}

extern "C" int pdes_conv_backward_data(const pdes_context* ctx, const pdes_conv_desc* descs, int n, void* stream) {
  if (!descs || n <= 0) return PDES_EINVAL;
  OptScope scope(ctx);
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int i = 0; i < n; ++i) {
    if (is_resample_op(descs[i])) {
      const int rc = op_backward(descs[i], st);
      if (rc) return rc;
      continue;
    }
    int rc = force_direct() ? PDES_ENOSUP : conv_backward_data_small(descs[i], st);
    if (rc == PDES_ENOSUP && !force_direct()) rc = conv_backward_data_b3_up(descs[i], st);
    if (rc == PDES_ENOSUP && !force_direct()) rc = conv_backward_data_up_mfma(descs[i], st);
    if (rc == PDES_ENOSUP && !force_direct()) rc = conv_backward_data_b3(descs[i], st);
    if (rc == PDES_ENOSUP && !force_direct()) rc = conv_backward_data_1x1(descs[i], st);
    if (rc == PDES_ENOSUP && !force_direct()) rc = conv_backward_data_mfma(descs[i], st);
    if (rc == PDES_ENOSUP && descs[i].g_fused) return PDES_EINVAL;
    if (rc == PDES_ENOSUP) rc = conv_backward_data_direct(descs[i], st);
    if (rc) return rc;
  }
  return PDES_OK;
}


// Which packed weight images would the forward and the data-gradient pass of this descriptor READ under the context's
// options?  The same dispatch chains as pdes_conv_forward / pdes_conv_backward_data, as capability queries (nothing is
// enqueued).  The caller's per-step packing launch can leave out every image no pass reads (the VALU images of a net
// that runs on the matrix cores, the f32 images of the layers on the bf16-split kernels): models/codec.py _pack_weights.
extern "C" int pdes_conv_image_use(const pdes_context* ctx, const pdes_conv_desc* desc, int* mask) {
  if (!desc || !mask) return PDES_EINVAL;
  *mask = 0;
  if (is_resample_op(*desc)) return PDES_OK;
  OptScope scope(ctx);
  pdes_conv_desc d = *desc;
  d.eval_mode = 0;                                  // (the data gradient exists in training mode only; the forward chain does
                                                    //  not look at the mode.  g_fused stays the caller's: it moves layers between kernels)
  int m = 0;
  // a caller switches out_stats between its training (accumulate) and evaluation (NULL) forwards and a kernel may be
  // selected by it (the few-output forward takes no statistics): a descriptor that carries statistics is queried BOTH
  // ways, the masks OR-ed
  for (int pass = 0; pass < (desc->out_stats ? 2 : 1); ++pass) {
    d.out_stats = pass ? nullptr : desc->out_stats;
    int rc = PDES_ENOSUP;
    auto tryf = [&](int r, int bit) { if (rc == PDES_ENOSUP && r != PDES_ENOSUP) { rc = r; m |= bit; } };
    if (!force_direct()) {
      tryf(conv_forward_small(d, nullptr, true), PDES_IMG_MFMA_FWD);
      if (rc == PDES_ENOSUP) tryf(conv_forward_b3_up(d, nullptr, true), PDES_IMG_B3UP_FWD);
      if (rc == PDES_ENOSUP) tryf(conv_forward_up_mfma(d, nullptr, true), PDES_IMG_UP_FWD);
      if (rc == PDES_ENOSUP) tryf(conv_forward_fewout(d, nullptr, true), 0);                  // (reads the live weights)
      if (rc == PDES_ENOSUP) tryf(conv_forward_b3(d, nullptr, true), PDES_IMG_B3_FWD);
      if (rc == PDES_ENOSUP) tryf(conv_forward_1x1(d, nullptr, true), PDES_IMG_MFMA_FWD);
      if (rc == PDES_ENOSUP) tryf(conv_forward_mfma(d, nullptr, true), PDES_IMG_MFMA_FWD);
    }
    if (rc == PDES_ENOSUP && !conv_forward_direct_first7(d)) m |= PDES_IMG_DIRECT_FWD;
  }
  d.out_stats = desc->out_stats;
  {
    int rc = PDES_ENOSUP;
    auto tryb = [&](int r, int bit) { if (rc == PDES_ENOSUP && r != PDES_ENOSUP) { rc = r; m |= bit; } };
    if (!force_direct()) {
      tryb(conv_backward_data_small(d, nullptr, true), PDES_IMG_MFMA_BWD);
      if (rc == PDES_ENOSUP) tryb(conv_backward_data_b3_up(d, nullptr, true), PDES_IMG_B3UP_BWD);
      if (rc == PDES_ENOSUP) tryb(conv_backward_data_up_mfma(d, nullptr, true), PDES_IMG_UP_BWD);
      if (rc == PDES_ENOSUP) tryb(conv_backward_data_b3(d, nullptr, true), PDES_IMG_B3_BWD);
      if (rc == PDES_ENOSUP) tryb(conv_backward_data_1x1(d, nullptr, true), PDES_IMG_MFMA_BWD);
      if (rc == PDES_ENOSUP) tryb(conv_backward_data_mfma(d, nullptr, true), PDES_IMG_MFMA_BWD);
    }
    if (rc == PDES_ENOSUP) m |= PDES_IMG_DIRECT_BWD;
  }
  *mask = m;
  return PDES_OK;
}

// The two halves of pdes_backward2 for a range of layers, each on ONE stream and free of events: what the segment graphs
// of step_graph.hip are captured from (also usable eagerly).
extern "C" int pdes_backward_chain(const pdes_context* ctx, const pdes_conv_desc* descs, int lo, int hi, void* stream) {
  if (!descs || lo < 0 || hi <= lo) return PDES_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int i = hi - 1; i >= lo; --i) {
    pdes_conv_desc df;
    OptScope scope(ctx);
    const bool onl = fin_onload(descs[i], &df);
    const pdes_conv_desc& d = onl ? df : descs[i];
    if (d.fin_tstats && !d.g_fused) {
      const int rc = bn_backward_finalize_launch(ctx, const_cast<float*>(d.g), d.out, d.fin_xstats, d.fin_tstats, d.B,
                                                 d.g_ctot, d.g_coff, d.g_coff + d.Cout, d.Hout * d.Wout, d.eps, d.nrep,
                                                 d.rep_stride, st, nullptr, d.g_add, d.fin_coef);
      if (rc) return rc;
    }
    if (d.has_bn || is_resample_op(d) || d.t_in) {
      const int rc = pdes_conv_backward_data(ctx, &d, 1, st);
      if (rc) return rc;
    }
  }
  return PDES_OK;
}

extern "C" int pdes_backward_weights(const pdes_context* ctx, const pdes_conv_desc* descs, int lo, int hi, void* stream) {
  if (!descs || lo < 0 || hi <= lo) return PDES_EINVAL;
  for (int i = hi - 1; i >= lo; --i) {
    if (is_resample_op(descs[i])) continue;
    pdes_conv_desc df;
    OptScope scope(ctx);
    const bool onl = fin_onload(descs[i], &df);
    const int rc = pdes_conv_backward_weight(ctx, onl ? &df : &descs[i], 1, stream);
    if (rc) return rc;
  }
  return PDES_OK;
}

extern "C" int pdes_backward(const pdes_context* ctx, const pdes_conv_desc* descs, int n, void* stream,
                             void* wgrad_stream, const pdes_reduce_item* reduce_items, const int* reduce_index,
                             const pdes_bucket_hook* hook) {
  return pdes_backward2(ctx, descs, n, stream, wgrad_stream, nullptr, reduce_items, reduce_index, hook);
}

extern "C" int pdes_backward2(const pdes_context* ctx, const pdes_conv_desc* descs, int n, void* stream,
                              void* wgrad_stream, void* wgrad_stream_b, const pdes_reduce_item* reduce_items,
                              const int* reduce_index, const pdes_bucket_hook* hook) {
  if (!descs || n <= 0) return PDES_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipStream_t ws = wgrad_stream ? static_cast<hipStream_t>(wgrad_stream) : st;
  const bool fork = ws != st;
  // optional SECOND weight-gradient stream: the weight gradients of successive layers are independent of each other,
  // odd layers go to it; it is joined into `ws` before a split-K reduce and at the end
  hipStream_t wsb = (fork && wgrad_stream_b) ? static_cast<hipStream_t>(wgrad_stream_b) : ws;
  const bool two = wsb != ws;
  // the fork/join events belong to the caller's context (created with it, on its device): one per layer + the join
  const Context* cx = reinterpret_cast<const Context*>(ctx);
  if (fork && (!cx || (int)cx->events.size() < n + 4)) return PDES_EINVAL;
  OptScope scope(ctx);
  const bool have_red = reduce_items && reduce_index;
#ifdef PDES_TIMING_KNOBS
  // Component-timing build only (tools/ab_timing.py; wrong numbers, right launch structure -- never shipped): bits of the
  // environment variable PDES_TIMING, read per call: 1 no weight-gradient kernels (fork events kept), 2 no fork events
  // either, 4 every finalize as ONE workgroup (launch + completion signal kept, no work), 8 no finalize launch at all
  // (forks by hipEventRecord), 16 no data-gradient kernels, 32 neither weight-gradient kernel nor fork for the layers that
  // finalize on load (the dense layers), 64 the same for the finalize-on-load layers on maps of <= 256 pixels only
  const int timing = getenv("PDES_TIMING") ? atoi(getenv("PDES_TIMING")) : 0;
#else
  const int timing = 0;
#endif
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
  size_t nev = 0;
  // every slot of the context's event table is taken through take(): nullptr (-> PDES_EINVAL) when the table is used up.
  // At most one event per layer (its fork: on the finalize's or the previous data gradient's completion signal, or
  // recorded by release()) + the joins; a slot taken for a launch that turned out not to carry signals is handed back
  auto take = [&]() -> hipEvent_t { return (cx && nev < cx->events.size()) ? cx->events[nev++] : nullptr; };
  // layers are released in the order n-1 .. 0, so the enqueued ones always form the suffix [i, n)
  // `signalled`: the fork event already completes with the finalize kernel just launched (its completion signal)
  bool b_dirty = false;                     // work on the second side stream that `ws` has not waited for yet
  auto join_b = [&]() -> int {              // ws waits for everything enqueued on wsb so far
    if (!two || !b_dirty) return PDES_OK;
    hipEvent_t e = take();
    if (!e) return PDES_EINVAL;
    hipError_t he = hipEventRecord(e, wsb);
    if (he == hipSuccess) he = hipStreamWaitEvent(ws, e, 0);
    b_dirty = false;
    return he == hipSuccess ? PDES_OK : (int)he;
  };
  auto release = [&](int i, bool on_main, hipEvent_t signalled) -> int {
    hipStream_t wsi = (two && (i & 1)) ? wsb : ws;
    if (fork && !on_main && !(timing & 2)) {
      hipEvent_t e = signalled;
      hipError_t he = hipSuccess;
      if (!e) {
        e = take();
        if (!e) return PDES_EINVAL;
        he = hipEventRecord(e, st);
      }
      if (he == hipSuccess) he = hipStreamWaitEvent(wsi, e, 0);
      if (he != hipSuccess) return (int)he;
      if (wsi != ws) b_dirty = true;
    }
    if (!(timing & 1)) {
      pdes_conv_desc df;
      const bool onl = fin_onload(descs[i], &df);
      const int rc = pdes_conv_backward_weight(ctx, onl ? &df : &descs[i], 1, on_main ? st : wsi);
      if (rc) return rc;
    }
    if (have_red && reduce_index[i] >= 0) per_done += per_of(i);
    // The split-K partials of the last layers (the widest ones: LastTransUp holds ~3/4 of the weights) are reduced
    // on the second stream as soon as >= 60 % of the weights' partials exist; the rest waits for the end.
    if (fork && !on_main && have_red && early_lo < 0 && 5 * per_done >= 3 * per_total) {
      int first = -1;
      long long mx = 0;
      for (int k = i; k < n; ++k)
        if (reduce_index[k] >= 0) {
          if (first < 0) first = reduce_index[k];
          mx = per_of(k) > mx ? per_of(k) : mx;
        }
      if (first > 0) {
        const int rcj = join_b();
        if (rcj) return rcj;
        const int rc2 = pdes_wgrad_reduce_all(reduce_items + first, n_items - first, (int)mx, ws);
        if (rc2) return rc2;
        early_lo = first;
        // the gradients of layers [i, n) are final on the weight-gradient stream from here on: the caller's hook
        // (e.g. the first bucket of the data-parallel all-reduce) is enqueued behind them, beside the rest of the
        // backward pass on the main stream
        if (hook && hook->fn) {
          const int rc3 = hook->fn(hook->user, i, ws);
          if (rc3) return rc3;
        }
      }
    }
    return PDES_OK;
  };

  const bool use_signal = opt().fork_signal != 0;
  // PDES_WGRAD_HOLD: a layer whose weight gradient is at least that many MFLOP is released behind its DATA gradient
  // (one hipEventRecord on the main stream) instead of behind its finalize.  Both kernels of such a layer fill the chip
  // by themselves: side by side the data gradient -- the one the chain waits for -- takes twice as long (196 -> 98 at
  // 32 x 32, B = 32: 172 us beside its weight gradient, 78 us alone), behind it the weight gradient overlaps the
  // narrow layers that follow, which leave most of the chip idle.
  const long long hold_mflop = opt().wgrad_hold;
  auto held = [&](int i) {
    const pdes_conv_desc& d = descs[i];
    if (!fork || hold_mflop <= 0 || i == 0 || is_resample_op(d) || !(d.has_bn || d.t_in)) return false;
    const long long mflop = 2LL * d.B * d.Hout * d.Wout * d.Cout * d.Cin * d.ksize * d.ksize / 1000000LL;
    return mflop >= hold_mflop;
  };
  hipEvent_t carried = nullptr;      // completion signal of the data gradient just launched, for the next layer's fork
  for (int i = n - 1; i >= 0; --i) {
    // finalize on load (PDES_FIN_ONLOAD): no finalize launch for this layer; its weight gradient is released by an event
    // recorded behind the kernel that completed its output gradient's accumulator (the previous data gradient)
    pdes_conv_desc df;
    const bool onl = fin_onload(descs[i], &df);
    const pdes_conv_desc& d = onl ? df : descs[i];
    hipEvent_t signalled = onl ? carried : nullptr;     // (set by the previous iteration's data-gradient launch)
    const bool hold = held(i);
    if (d.fin_tstats && !d.g_fused) {
      // this layer's weight gradient is released by the completion of ITS finalize kernel: the fork event rides on
      // that kernel's completion signal, no barrier packet sits between the finalize and the data gradient
      if (fork && use_signal && i != 0 && !is_resample_op(d) && !hold && !(timing & (2 | 8))) {
        signalled = take();
        if (!signalled) return PDES_EINVAL;
      }
      int rc = PDES_OK;
      if (timing & 4)              // (one channel of one image: a single workgroup)
        rc = bn_backward_finalize_launch(ctx, const_cast<float*>(d.g), d.out, d.fin_xstats, d.fin_tstats, 1,
                                         d.g_ctot, d.g_coff, d.g_coff + 1, 256, d.eps, d.nrep,
                                         d.rep_stride, st, signalled, d.g_add, d.fin_coef);
      else if (!(timing & 8))
        rc = bn_backward_finalize_launch(ctx, const_cast<float*>(d.g), d.out, d.fin_xstats, d.fin_tstats, d.B,
                                         d.g_ctot, d.g_coff, d.g_coff + d.Cout, d.Hout * d.Wout, d.eps, d.nrep,
                                         d.rep_stride, st, signalled, d.g_add, d.fin_coef);
      if (rc) return rc;
    }
    // the very last weight gradient (first layer) has nothing left to overlap with: it stays on the main stream
    // (saves the event hop; its scratch / dw are disjoint from what the second stream still works on)
    if (((timing & 32) && onl) || ((timing & 64) && onl && d.Hout * d.Wout <= 256)) {      // (64: the small maps only)
      // (component timing: neither the weight-gradient kernel nor the fork of a finalize-on-load layer -- the upper bound of
      //  what folding those weight gradients into the data-gradient launches could win)
    } else if (!is_resample_op(d) && !hold) {
      const int rc = release(i, fork && i == 0, signalled);
      if (rc) return rc;
    }
    // (a convolution without a BatchNorm in front has a data gradient only when the caller gave it somewhere to go)
    if ((d.has_bn || is_resample_op(d) || d.t_in) && !(timing & 16)) {
      // the NEXT layer (i - 1) finalizes on load: its weight gradient is released by THIS data gradient's completion.
      // Where this launch can carry a completion signal (a finalize-on-load data gradient itself: inside a dense block),
      // the fork event rides on it, as it rides on the finalize kernel elsewhere -- no barrier packet on the chain
      pdes_conv_desc dn;
      hipEvent_t se = nullptr;
      if (fork && use_signal && opt().fin_onload >= 2 && onl && i >= 2 && !(timing & (2 | 8 | 32)) && !((timing & 64) && descs[i - 1].Hout * descs[i - 1].Wout <= 256) && fin_onload(descs[i - 1], &dn) &&
          !held(i - 1)) {
        se = take();
        if (!se) return PDES_EINVAL;
        set_dgrad_stop_event(se);
      }
      const int rc = pdes_conv_backward_data(ctx, &d, 1, st);
      if (se && dgrad_stop_event_pending()) {       // (the kernel family that took the launch does not carry signals:
        set_dgrad_stop_event(nullptr);              //  the slot goes back, release() records the same event instead)
        se = nullptr;
        --nev;
      }
      carried = se;
      if (rc) return rc;
    } else {
      carried = nullptr;
    }
    if (hold) {                                           // released by an event recorded behind the data gradient
      const int rc = release(i, false, nullptr);
      if (rc) return rc;
    }
  }
  if (fork) {
    const int rcj = join_b();
    if (rcj) return rcj;
    hipEvent_t e = take();
    if (!e) return PDES_EINVAL;
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
