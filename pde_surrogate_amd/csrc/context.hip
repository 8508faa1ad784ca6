// pdes_context: the only state the library keeps, owned by the caller (created / destroyed explicitly).
#include <stdlib.h>
#include <string.h>
#include "pdes_common.h"
#include "pdes_options.h"
#include "../../include/pdes_hip.h"

namespace pdes {

static const Options kDefaults{};
static thread_local const Options* tl_opt = nullptr;

const Options& opt() { return tl_opt ? *tl_opt : kDefaults; }

OptScope::OptScope(const void* ctx) : prev(tl_opt) {
  tl_opt = ctx ? &static_cast<const Context*>(ctx)->opt : &kDefaults;
}
OptScope::~OptScope() { tl_opt = prev; }

struct Knob { const char* name; int Options::*field; };
static const Knob kKnobs[] = {
    {"PDES_MFMA_B3", &Options::mfma_b3},       {"PDES_B3_TAIL", &Options::b3_tail},     {"PDES_MFMA_1X1", &Options::mfma_1x1},
    {"PDES_MFMA_SMALL", &Options::mfma_small}, {"PDES_WGRAD_WGS", &Options::wgrad_wgs}, {"PDES_LOSS_NT", &Options::loss_nt},
    {"PDES_FORK_SIGNAL", &Options::fork_signal}, {"PDES_WGRAD_HOLD", &Options::wgrad_hold},
    {"PDES_FIN_ONLOAD", &Options::fin_onload}, {"PDES_DG_TILEPIPE", &Options::dg_tilepipe},
    {"PDES_DG_TILEPIPE16", &Options::dg_tilepipe16}, {"PDES_MFMA_MT2", &Options::mfma_mt2},
    {"PDES_BAND_FIXED", &Options::band_fixed}, {"PDES_XCD_MAP", &Options::xcd_map},
};

static int set_knob(Options& o, const char* key, const char* value) {
  if (!strcmp(key, "PDES_CONV_IMPL")) {             // "direct" | anything else = automatic
    o.conv_direct = value && value[0] == 'd';
    return PDES_OK;
  }
  for (const Knob& k : kKnobs)
    if (!strcmp(key, k.name)) {
      o.*(k.field) = value ? atoi(value) : kDefaults.*(k.field);
      return PDES_OK;
    }
  return PDES_ENOSUP;
}

}  // namespace pdes

using namespace pdes;

extern "C" int pdes_context_create(pdes_context** out, int n_events) {
  if (!out || n_events < 0 || n_events > 4096) return PDES_EINVAL;
  Context* c = new (std::nothrow) Context();
  if (!c) return (int)hipErrorOutOfMemory;
  hipError_t he = hipGetDevice(&c->device);
  for (int i = 0; i < n_events && he == hipSuccess; ++i) {
    hipEvent_t e = nullptr;
    // order-only events between streams of ONE device: no system-scope fence (cache write-back for the host) when
    // they are recorded -- it delayed the kernel behind every fork by ~0.75 us (27 forks per step)
    he = hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence);
    if (he == hipSuccess) c->events.push_back(e);
  }
  if (he != hipSuccess) {
    for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
    delete c;
    return (int)he;
  }
  *out = reinterpret_cast<pdes_context*>(c);
  return PDES_OK;
}

extern "C" int pdes_context_destroy(pdes_context* ctx) {
  if (!ctx) return PDES_EINVAL;
  Context* c = reinterpret_cast<Context*>(ctx);
  for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
  delete c;
  return PDES_OK;
}

extern "C" int pdes_context_set_option(pdes_context* ctx, const char* key, const char* value) {
  if (!ctx || !key) return PDES_EINVAL;
  return set_knob(reinterpret_cast<Context*>(ctx)->opt, key, value);
}

extern "C" int pdes_context_load_env(pdes_context* ctx) {
  if (!ctx) return PDES_EINVAL;
  Options& o = reinterpret_cast<Context*>(ctx)->opt;
  if (const char* e = getenv("PDES_CONV_IMPL")) set_knob(o, "PDES_CONV_IMPL", e);
  for (const Knob& k : kKnobs)
    if (const char* e = getenv(k.name)) set_knob(o, k.name, e);
  return PDES_OK;
}

extern "C" int pdes_context_device(const pdes_context* ctx) {
  return ctx ? reinterpret_cast<const Context*>(ctx)->device : PDES_EINVAL;
}

// sizeof of the structures that cross the boundary (the ctypes mirrors are checked against it: tests/test_cabi.py)
extern "C" int pdes_sizeof(int which) {
  switch (which) {
    case 0: return (int)sizeof(pdes_conv_desc);
    case 1: return (int)sizeof(pdes_pack_item);
    case 2: return (int)sizeof(pdes_mfma_pack_item);
    case 3: return (int)sizeof(pdes_up_pack_item);
    case 4: return (int)sizeof(pdes_b3_pack_item);
    case 5: return (int)sizeof(pdes_b3up_pack_item);
    case 6: return (int)sizeof(pdes_reduce_item);
    case 7: return (int)sizeof(pdes_bn_item);
    case 8: return (int)sizeof(pdes_op);
    default: return -1;
  }
}
