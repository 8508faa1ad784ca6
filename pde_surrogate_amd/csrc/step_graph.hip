// The training step as a handful of LINEAR hipGraphs replayed on the step's streams.
//
// Why (tools/archive/proto/launch_cost.hip, round 3): one eager launch costs the host 3.0-3.5 us (4.1 with a completion-signal
// event, 9.1 with an event record + cross-stream wait) and leaves ~3.0 us between dependent kernels on the GPU; a
// linear graph of 120 kernel nodes replays for 8 us of host time with ~1.7 us between kernels -- while a graph WITH
// forks runs 37 % slower than the same launches issued eagerly on two streams (the runtime serialises branches).  So
// the step is cut into linear pieces -- the forward pass + loss, per segment of the backward pass the finalize ->
// data-gradient chain and, separately, that segment's weight gradients -- and the fork / join between the main stream
// and the weight-gradient streams stays what it is in the eager step: an event between two launches, now one per
// segment instead of one per layer.
//
// The pieces are captured from the ordinary enqueue-only entry points (pdes_conv_forward, pdes_backward_chain,
// pdes_backward_weights, pdes_darcy_loss, ...): pdes_graph_begin / pdes_graph_end bracket any sequence of them on a
// capture stream.  pdes_program_run replays a caller-built list of {launch graph, record event, wait event, hook}
// operations in one call.  The caller owns graphs and program; the events are the context's.
#include "pdes_common.h"
#include "pdes_options.h"
#include "../../include/pdes_hip.h"

namespace pdes {
struct Graph {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  int nodes = 0;
};
}  // namespace pdes

using namespace pdes;

extern "C" int pdes_graph_begin(void* stream) {
  if (!stream) return PDES_EINVAL;                       // the legacy default stream cannot capture
  const hipError_t he = hipStreamBeginCapture(static_cast<hipStream_t>(stream), hipStreamCaptureModeThreadLocal);
  return he == hipSuccess ? PDES_OK : (int)he;
}

extern "C" int pdes_graph_end(void* stream, pdes_graph** out) {
  if (!stream || !out) return PDES_EINVAL;
  Graph* g = new (std::nothrow) Graph();
  if (!g) return (int)hipErrorOutOfMemory;
  hipError_t he = hipStreamEndCapture(static_cast<hipStream_t>(stream), &g->graph);
  if (he == hipSuccess && !g->graph) he = hipErrorStreamCaptureInvalidated;
  if (he == hipSuccess) {
    size_t n = 0;
    if (hipGraphGetNodes(g->graph, nullptr, &n) == hipSuccess) g->nodes = (int)n;
    he = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
  }
  if (he != hipSuccess) {
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return (int)he;
  }
  *out = reinterpret_cast<pdes_graph*>(g);
  return PDES_OK;
}

extern "C" int pdes_graph_nodes(const pdes_graph* g) { return g ? reinterpret_cast<const Graph*>(g)->nodes : PDES_EINVAL; }

extern "C" int pdes_graph_launch(pdes_graph* g, void* stream) {
  if (!g) return PDES_EINVAL;
  const hipError_t he = hipGraphLaunch(reinterpret_cast<Graph*>(g)->exec, static_cast<hipStream_t>(stream));
  return he == hipSuccess ? PDES_OK : (int)he;
}

extern "C" int pdes_graph_destroy(pdes_graph* g) {
  if (!g) return PDES_EINVAL;
  Graph* p = reinterpret_cast<Graph*>(g);
  if (p->exec) (void)hipGraphExecDestroy(p->exec);
  if (p->graph) (void)hipGraphDestroy(p->graph);
  delete p;
  return PDES_OK;
}

extern "C" int pdes_program_run(const pdes_context* ctx, pdes_graph* const* graphs, int n_graphs, void* const* streams,
                                int n_streams, const pdes_op* ops, int n_ops, const pdes_bucket_hook* hook) {
  if (!ctx || !graphs || !streams || !ops || n_ops <= 0) return PDES_EINVAL;
  const Context* cx = reinterpret_cast<const Context*>(ctx);
  for (int k = 0; k < n_ops; ++k) {
    const pdes_op& o = ops[k];
    if (o.stream < 0 || o.stream >= n_streams) return PDES_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(streams[o.stream]);
    hipError_t he = hipSuccess;
    switch (o.kind) {
      case PDES_OP_LAUNCH:
        if (o.arg < 0 || o.arg >= n_graphs || !graphs[o.arg]) return PDES_EINVAL;
        he = hipGraphLaunch(reinterpret_cast<Graph*>(graphs[o.arg])->exec, st);
        break;
      case PDES_OP_RECORD:
        if (o.arg < 0 || o.arg >= (int)cx->events.size()) return PDES_EINVAL;
        he = hipEventRecord(cx->events[o.arg], st);
        break;
      case PDES_OP_WAIT:
        if (o.arg < 0 || o.arg >= (int)cx->events.size()) return PDES_EINVAL;
        he = hipStreamWaitEvent(st, cx->events[o.arg], 0);
        break;
      case PDES_OP_HOOK:
        if (hook && hook->fn) {
          const int rc = hook->fn(hook->user, o.arg, streams[o.stream]);
          if (rc) return rc;
        }
        break;
      default:
        return PDES_EINVAL;
    }
    if (he != hipSuccess) return (int)he;
  }
  return PDES_OK;
}
