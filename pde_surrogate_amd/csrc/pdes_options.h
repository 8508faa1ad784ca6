// Run-time knobs of the library and the opaque context that carries them (include/pdes_hip.h: pdes_context).
// Nothing in the library reads the environment on its own: pdes_context_load_env() does, once, when the caller
// asks for it; every other read goes through opt(), which resolves to the options of the context the running
// entry point was called with (or to the compiled-in defaults for a NULL context).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>

namespace pdes {

struct Options {
  int conv_direct = 0;          // PDES_CONV_IMPL=direct : generic VALU kernels for every convolution (cross-check)
  int fuse_finalize = 0;        // PDES_FUSE_FINALIZE    : BatchNorm-backward finalize on operand load (opt-in, slower)
  int fuse_maxc = 16;           // PDES_FUSE_MAXC
  int fuse_maxhw = 1 << 30;     // PDES_FUSE_MAXHW
  int fin_early = 1;            // PDES_FIN_EARLY        : finalize kernel issues its T/x loads before the statistics chain
  int mfma_ntw = 2;             // PDES_MFMA_NTW
  int mfma_mt = 0;              // PDES_MFMA_MT          : 0 = automatic
  int mfma_ng = 2;              // PDES_MFMA_NG
  int mfma_1x1 = 1;             // PDES_MFMA_1X1         : 0 off, 1 forward + data gradient, 2 forward, 3 data gradient
  int k1_ksplit = 0;            // PDES_1X1_KSPLIT       : 0 = automatic, 2, 4
  int mfma_1x1w = 1;            // PDES_MFMA_1X1W
  int w1x1_spi = 0;             // PDES_1X1W_SPI         : 4 = four splits per image
  int mfma_b3 = 1;              // PDES_MFMA_B3          : bf16 x3 split kernel for the wide 3x3 layer
  int mfma_b3w = 1;             // PDES_MFMA_B3W         : bf16 x3 split kernel for the weight gradient of the wide 3x3 layers
  int mfma_b3wu = 1;            // PDES_MFMA_B3WU        : ... and of the nearest-x2 + 3x3 layer with a 32-wide input (sub-pixel form)
  int b3w_pf = 2;               // PDES_B3W_PF           : rows of lead of the operand loads of the bf16 x3 weight-gradient kernel (1 or 2)
  int mfma_b3u = 1;             // PDES_MFMA_B3U         : bf16 x3 split kernel for the forward of the nearest-x2 + 3x3 layers
  int mfma_small = 1;           // PDES_MFMA_SMALL       : matrix-core kernels for 3x3 convolutions on 8x8 maps (conv_small.hip)
  int mfma_b3ub = 1;            // PDES_MFMA_B3UB        : bf16 x3 split kernel for the data gradient of the nearest-x2 + 3x3 layers
  int b3_mt = 4;                // PDES_B3_MT
  int b3_apipe = 1;             // PDES_B3_APIPE         : A-operand fragments of the next (tap, M-tile) read before this one's MFMAs
  int b3_tail = 1;              // PDES_B3_TAIL          : <= 4 channels of the last 32-channel chunk on the f32 pipe (one MFMA per tap instead of six)
  int few_r = 2;                // PDES_FEW_R
  int wgrad_wgs = 256;          // PDES_WGRAD_WGS        : workgroup target of the split-K weight-gradient plan
  int loss_nt = -1;             // PDES_LOSS_NT          : -1 = by working-set size
  int debug_chain = 0;          // PDES_DEBUG_CHAIN      : TIMING EXPERIMENTS ONLY (wrong gradients): 1 = pdes_backward skips the
                                //                         weight-gradient kernels but keeps the fork events, 2 = skips both
  int fork_signal = 1;          // PDES_FORK_SIGNAL      : fork events ride on the finalize kernel's completion signal (0: hipEventRecord)
};

struct Context {
  Options opt;
  int device = 0;
  std::vector<hipEvent_t> events;     // fork/join events of pdes_backward, created with the context on `device`
};

const Options& opt();                 // options in force on this thread (defaults outside an entry point)

struct OptScope {                     // RAII: an entry point installs its context's options for its duration
  const Options* prev;
  explicit OptScope(const void* ctx);
  ~OptScope();
};

}  // namespace pdes
