// Run-time knobs of the library and the opaque context that carries them (include/pdes_hip.h: pdes_context).
// Nothing in the library reads the environment on its own: pdes_context_load_env() does, once, when the caller
// asks for it; every other read goes through opt(), which resolves to the options of the context the running
// entry point was called with (or to the compiled-in defaults for a NULL context).
// Fifteen options: each selects between equivalent kernels for cross-checks (the GPU tests) or re-tuning on other
// parts; the A/B measurements that settled the defaults, and the knobs that went with them, are in EXPERIMENTS.md.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>

namespace pdes {

struct Options {
  int conv_direct = 0;          // PDES_CONV_IMPL=direct : generic VALU kernels for every convolution (cross-check)
  int mfma_b3 = 31;             // PDES_MFMA_B3   : bit mask of the bf16 x3 split kernels (0 = the exact-f32 pipe everywhere):
                                //                  1 wide 3x3 forward + data gradient, 2 wide 3x3 weight gradient, 4 nearest-x2 forward,
                                //                  8 nearest-x2 data gradient, 16 nearest-x2 weight gradient
  int b3_tail = 1;              // PDES_B3_TAIL   : <= 4 channels of the last 32-channel chunk on one f32 MFMA per tap instead of six bf16 ones
  int mfma_1x1 = 5;             // PDES_MFMA_1X1  : bit mask of the register-operand 1x1 kernels: 1 forward, 2 data gradient, 4 weight gradient
                                //                  (round 5: the data gradient back on the LDS-tiled kernel -- 22.2 vs 25.1 us stand-alone for
                                //                  144 <- 72 at 32 x 32, 1.634 vs 1.640 ms per step same process; round 2 had found the opposite
                                //                  inside the step of the time)
  int mfma_small = 1;           // PDES_MFMA_SMALL: matrix-core kernels for 3x3 convolutions on 8x8 maps (conv_small.hip)
  int wgrad_wgs = 256;          // PDES_WGRAD_WGS : workgroup target of the split-K weight-gradient plan
  int loss_nt = -1;             // PDES_LOSS_NT   : streaming loads / stores in the loss kernel: -1 = by working-set size, 0, 1
  int fork_signal = 1;          // PDES_FORK_SIGNAL: fork events ride on the finalize kernel's completion signal (0: hipEventRecord)
  int fin_onload = 3;           // PDES_FIN_ONLOAD : 1: the backward of the dense blocks' 16-output-channel 3x3 layers applies the BatchNorm-
                                //                  backward finalize of the layer's output gradient on operand load (no finalize launch
                                //                  for them); 2: and the forks inside a dense block ride on the data gradients'
                                //                  completion signals; 3 (default): and the first convolution's weight-gradient kernel
                                //                  finalizes while it stages the gradient planes; 0: one finalize launch per layer
  int dg_tilepipe16 = 48;       // PDES_DG_TILEPIPE16: the same threshold for the other tile shapes (16-wide maps, half-height tiles)
  int dg_tilepipe = 48;         // PDES_DG_TILEPIPE: the dense blocks' data gradient on 32-wide tiles runs M-tile by M-tile (each tile's
                                //                  epilogue behind its MFMAs) from this many input channels on; 0: never
  int mfma_mt2 = 1;             // PDES_MFMA_MT2   : the forward of K-split (16-output-channel) layers whose MT = 4 grid leaves half the CUs
                                //                  empty (16x16 maps at batch 32) runs on tiles of 2 rows x 16 pixels (round 6); 0: MT = 4
  int xcd_map = 1;              // PDES_XCD_MAP    : halo-tiled kernels map their workgroups so that an XCD (own L2) takes whole images
                                //                  instead of every eighth tile of every image (round 6); 0: blockIdx as launched
  int band_fixed = 1;           // PDES_BAND_FIXED : the any-size loss kernel's instantiations with compile-time geometry for the common
                                //                  sizes (48, 65, 66, 96, 100, 128 .. 131, 200, 256; round 6); 0: always the run-time plan (cross-checks)
  int wgrad_hold = 0;           // PDES_WGRAD_HOLD : pdes_backward releases the weight gradient of a layer with >= this many MFLOP
                                //                  (2 B Hout Wout Cout Cin k^2 / 1e6) behind the layer's DATA gradient instead of beside
                                //                  it (0: never): two kernels that each fill the chip gain nothing from running together
};

struct Context {
  Options opt;
  int device = 0;
  std::vector<hipEvent_t> events;     // fork/join events of pdes_backward / pdes_program_run, created with the context on `device`
};

const Options& opt();                 // options in force on this thread (defaults outside an entry point)

struct OptScope {                     // RAII: an entry point installs its context's options for its duration
  const Options* prev;
  explicit OptScope(const void* ctx);
  ~OptScope();
};

}  // namespace pdes
