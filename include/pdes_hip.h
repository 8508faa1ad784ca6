/* pdes_hip.h -- C ABI of libpdes_hip.so, the MI355X (gfx950) kernels behind the mixed-residual
 * training hot path of cics-nd/pde-surrogate.
 *
 * The reference has no FFI: its boundary is the Python call surface of
 *   models/darcy.py, utils/image_gradient.py, models/codec.py, train_codec_mixed_residual.py.
 * Each entry point below names the reference code it replaces (file:line).  INTEGRATION.md shows
 * the ctypes stub a maintainer of the reference would add to call them.
 *
 * Conventions (all entry points):
 *   - every pointer is DEVICE memory, fp32 unless stated, NCHW contiguous, 16-byte aligned;
 *   - the caller (PyTorch) owns every buffer including workspaces; the library never allocates or
 *     frees device memory and never retains a pointer;
 *   - the ONLY state is the opaque `pdes_context` the caller creates and destroys: the run-time options
 *     (kernel-selection knobs) and the order-only hipEvent_t objects pdes_backward uses to fork/join
 *     its second stream, created WITH the context on the then-current device.  There is no process-global
 *     or static state, and the library never reads the environment by itself (pdes_context_load_env does,
 *     when the caller asks).  Entry points that consult options take the context as their first argument;
 *     NULL = compiled-in defaults.  A context may be shared by threads that do not change its options
 *     concurrently; use one context per device;
 *   - enqueue-only on `stream` (a hipStream_t passed as void*): no synchronisation, no hipMalloc,
 *     so every call is hipGraph-capturable and callable from the autograd backward thread;
 *   - returns 0 on success, <0 for an argument the library rejects (PDES_E*), >0 = hipError_t
 *     from the launch.  No C++ exception crosses the ABI.
 */
#ifndef PDES_HIP_H
#define PDES_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define PDES_OK 0
#define PDES_EINVAL (-1) /* null pointer / non-positive size */
#define PDES_ENOSUP (-2) /* shape or option not implemented */
#define PDES_EALIGN (-3) /* pointer not 16-byte aligned */

/* ---------------------------------------------------------------------------------------------
 * Context: options + fork/join events (no reference counterpart -- the reference's knobs are Python
 * arguments; these select among equivalent kernels and exist for A/B measurements and cross-checks).
 *   pdes_context_create   n_events order-only events are created on the CURRENT device (pdes_backward with a
 *                         second stream needs n_layers + 1); returns hipError_t > 0 on failure.
 *   pdes_context_set_option  value = decimal string (NULL = default); PDES_ENOSUP: unknown key.  The sixteen keys
 *                         (csrc/pdes_options.h; each selects between EQUIVALENT kernels, for cross-checks and re-tuning):
 *                           "PDES_CONV_IMPL"   "direct": the generic VALU kernels for every convolution | "auto"
 *                           "PDES_MFMA_B3"     bit mask of the bf16 x3 split kernels, default 31 (0: the exact-f32 pipe everywhere):
 *                                              1 wide 3x3 forward + data gradient, 2 wide 3x3 weight gradient, 4 nearest-x2 forward,
 *                                              8 nearest-x2 data gradient, 16 nearest-x2 weight gradient
 *                           "PDES_B3_TAIL"     1: <= 4 channels of a last 32-channel chunk on one f32 MFMA per tap | 0: a whole bf16 chunk
 *                           "PDES_MFMA_1X1"    bit mask of the register-operand 1x1 kernels, default 5: 1 forward, 2 data gradient,
 *                                              4 weight gradient (0: the LDS-tiled generic kernels)
 *                           "PDES_MFMA_SMALL"  1: matrix-core kernels for 3x3 convolutions on 8x8 maps | 0: VALU kernels
 *                           "PDES_WGRAD_WGS"   workgroup target of the split-K weight-gradient plan (default 256)
 *                           "PDES_LOSS_NT"     streaming accesses in the loss kernel: -1 by working-set size (default), 0, 1
 *                           "PDES_FORK_SIGNAL" 1: pdes_backward's fork events ride on the finalize kernel's completion signal |
 *                                              0: hipEventRecord
 *                           "PDES_FIN_ONLOAD"  1: the backward of the dense blocks' layers (3x3, stride 1, <= 16 output channels) applies
 *                                              the BatchNorm-backward finalize of the layer's output gradient while its data- and
 *                                              weight-gradient kernels stage that operand (no finalize launch for them); 2:
 *                                              and inside a dense block the fork events ride on the data gradients' completion
 *                                              signals; 3 (default): and the first convolution's weight-gradient kernel finalizes
 *                                              while it stages the gradient planes; 0: one pdes_bn_backward_finalize launch per layer
 *                           "PDES_DG_TILEPIPE" the data gradient of the dense blocks' layers (one chunk of <= 16 output channels) runs
 *                                              M-tile by M-tile -- every tile's epilogue behind its own MFMAs instead of one epilogue at
 *                                              the end -- for layers with at least this many input channels on 32-wide tiles
 *                                              (default 48: all of them); 0: never.  "PDES_DG_TILEPIPE16": the same for the other
 *                                              tile shapes (16-wide maps; default 48)
 *                           "PDES_MFMA_MT2"    1 (default): the forward of a 16-output-channel 3x3 layer whose grid of 4-row tiles
 *                                              would leave half the CUs without a workgroup (16x16 maps at batch 32) runs on tiles of
 *                                              2 rows x 16 pixels | 0: 4-row tiles
 *                           "PDES_XCD_MAP"     1 (default): kernels whose workgroups re-read each other's halo rows remap blockIdx so that
 *                                              one XCD (its own L2) takes whole images | 0: blockIdx as launched (A/B, cross-checks)
 *                           "PDES_BAND_FIXED"  1 (default): the any-size loss kernel (row bands, 8 <= n <= 256) runs the instantiation with
 *                                              compile-time geometry where one exists for the field size and the flags are 0
 *                                              (48, 65, 66, 96, 100, 128, 129, 130, 131, 200, 256) | 0: always the run-time plan (cross-checks)
 *                           "PDES_WGRAD_HOLD"  pdes_backward with a second stream: the weight gradient of a layer of at least this
 *                                              many MFLOP (2 B Hout Wout Cout Cin k^2 / 1e6) is released behind the layer's data
 *                                              gradient instead of beside it; 0: never
 *   pdes_context_load_env every key above that is set in the process environment, read ONCE, now.
 *   pdes_context_device   the device the context's events belong to.
 */
typedef struct pdes_context pdes_context;
int pdes_context_create(pdes_context** out, int n_events);
int pdes_context_destroy(pdes_context* ctx);
int pdes_context_set_option(pdes_context* ctx, const char* key, const char* value);
int pdes_context_load_env(pdes_context* ctx);
int pdes_context_device(const pdes_context* ctx);

/* ABI version of this header; bumped on any signature change. */
int pdes_abi_version(void);
/* sizeof of a boundary structure: 0 pdes_conv_desc, 1 pdes_pack_item, 2 pdes_mfma_pack_item, 3 pdes_up_pack_item,
 * 4 pdes_b3_pack_item, 5 pdes_b3up_pack_item, 6 pdes_reduce_item, 7 pdes_bn_item, 8 pdes_op; -1 otherwise (a binding checks
 * its own mirrors against it) */
int pdes_sizeof(int which);
/* number of replicas of the fp64 accumulator arena the kernels are compiled for (see pdes_conv_desc.nrep) */
int pdes_stat_replicas(void);

/* ---------------------------------------------------------------------------------------------
 * Fused Sobel + Darcy mixed-residual loss, forward and (optionally) backward, ONE kernel.
 * Replaces: models/darcy.py:162-176 (conv_constitutive_constraint), :179-191 (…_nonlinear),
 *           :210-224 (conv_continuity_constraint, use_tb=True), :226-233 (conv_boundary_condition),
 *           utils/image_gradient.py:50-92 (SobelFilter.grad_h/grad_v, correct=True, 3x3),
 *           train_codec_mixed_residual.py:228-233 (combination + autograd backward wrt output).
 *
 *   K        (B,1,H,W)  permeability ("input")
 *   y        (B,3,H,W)  network output: u, sigma1, sigma2
 *   grad_y   (B,3,H,W)  OUT, d(total)/dy; NULL = forward only (eval)
 *   partials (rows,4)   OUT workspace: partial sums {sum r1^2+r2^2, sum c^2, dirichlet, neumann}; rows =
 *                       pdes_darcy_loss_partial_rows(B, H, W, flags): one per image for the specialised kernel, one per
 *                       (image, tile) for the tiled kernel (reduced in a fixed order: deterministic)
 *   loss_out (5)        OUT {total, L_const, L_cont, L_dir, L_neu}; NULL = skip the final reduce
 *   total = w_const*L_const + w_cont*L_cont + w_dir*L_dir + w_neu*L_neu
 *           (the reference's loss is w = (1, 1, weight_bound, weight_bound))
 *   flags & PDES_LOSS_NONLINEAR: sigma + beta1*sqrt(K)*sigma^2 + beta2*K*sigma^3 constitutive law.
 *   flags & PDES_LOSS_NO_TB: conv_continuity_constraint(use_tb=False), models/darcy.py:224 -- rows 0 and H-1 are left
 *           out of the continuity residual (mean over (H-2) W pixels).
 *   flags & PDES_LOSS_UNCORRECTED: the gradients of SobelFilter(correct=False) (utils/image_gradient.py:72-75, :89-92:
 *           no boundary `modifier`).
 *   Any square field, H == W >= 2 (SobelFilter(imsize) holds one imsize x imsize modifier for both axes; the
 *   reference's docstrings use 65 x 65, models/darcy.py:165-167).  H in {16, 32, 64} with correct=True runs the
 *   specialised one-image-per-workgroup kernel (16-byte aligned K / y / grad_y required); every other size or flag
 *   combination the any-size kernels of csrc/darcy_loss_generic.hip: for 8 <= H <= 256 the ROW-BAND kernel (the same
 *   strip / LDS / neighbour-lane structure without the compile-time size, csrc/darcy_band.h; 16-byte accesses when H is a
 *   multiple of 4 and the pointers are 16-byte aligned), otherwise tiles with halos.  Same results, (rows, 4) partials.
 */
#define PDES_LOSS_NONLINEAR 1
#define PDES_LOSS_NO_TB 2
#define PDES_LOSS_UNCORRECTED 4
#define PDES_LOSS_GENERIC 16     /* cross-checks: never the 16 / 32 / 64 specialisation (the any-size kernels at those sizes too) */
#define PDES_LOSS_TILED 8        /* cross-checks: the tile kernel instead of the row-band kernel (implies PDES_LOSS_GENERIC) */
int pdes_darcy_loss(const pdes_context* ctx, const float* K, const float* y, float* grad_y, float* partials, float* loss_out,
                    int B, int H, int W, float w_const, float w_cont, float w_dir, float w_neu,
                    int flags, float beta1, float beta2, void* stream);
/* The same launch with the four term weights {w_const, w_cont, w_dir, w_neu} read from DEVICE memory (4 floats) by the
 * kernels: what the autograd backward of the reference's three loss functions needs -- the upstream gradients of the
 * terms are device scalars there (models/darcy.py:162-233 under `loss.backward()`, train_codec_mixed_residual.py:233), and
 * reading them on the host would stall the backward pass once per loss function. */
int pdes_darcy_loss_dw(const pdes_context* ctx, const float* K, const float* y, float* grad_y, float* partials, float* loss_out,
                       int B, int H, int W, const float* w_dev, int flags, float beta1, float beta2, void* stream);
/* rows of `partials` the calls above write for these arguments (> 0), or PDES_ENOSUP */
int pdes_darcy_loss_partial_rows(int B, int H, int W, int flags);

/* Stand-alone Sobel gradients of `nimg` single-channel images (either output may be NULL); correct = the
 * SobelFilter's flag.  Any square H == W >= 2.
 * Replaces utils/image_gradient.py:50-75 (grad_h) and :77-92 (grad_v), filter_size=3. */
int pdes_sobel_grad(const float* img, float* gh, float* gv, int nimg, int H, int W, int correct,
                    void* stream);

/* img_bar = grad_h^T(gh_bar) + grad_v^T(gv_bar) (either input may be NULL) for the same `correct`.
 * This is what autograd computes for the reference's pad/conv2d/matmul chain. */
int pdes_sobel_grad_adjoint(const float* gh_bar, const float* gv_bar, float* img_bar, int nimg, int H,
                            int W, int correct, void* stream);

/* The same two calls for filter_size = 5 (utils/image_gradient.py:35-41 kernel, :65-67 / :82-84 selection): replicate
 * pad 2, 5x5 cross-correlation, the same boundary `modifier`.  Any square H == W >= 2.  No reference caller
 * passes filter_size = 5; provided for completeness of SobelFilter's signature. */
int pdes_sobel5_grad(const float* img, float* gh, float* gv, int nimg, int H, int W, int correct, void* stream);
int pdes_sobel5_grad_adjoint(const float* gh_bar, const float* gv_bar, float* img_bar, int nimg, int H, int W,
                             int correct, void* stream);

/* ---------------------------------------------------------------------------------------------
 * DenseED / Decoder building blocks (models/codec.py).  One descriptor per convolution; the
 * BatchNorm + ReLU (+ nearest x2 upsampling) that PRECEDES the convolution in the reference
 * (codec.py:66-69 dense layer, :103-150 transition, :163-188 last decoding) is fused into the
 * convolution's operand load, and the batch statistics of the convolution's OUTPUT (needed by the
 * BatchNorms that consume it) are accumulated in its epilogue.  The `torch.cat` of dense blocks
 * (codec.py:73-75) disappears: a block owns one (B, Ctot, H, W) buffer and each layer writes its
 * `growth_rate` channels at `out_coff`.
 *
 * Backward uses one accumulator tensor T per activation buffer (same shape as the buffer):
 *   T_c = sum over consumer BNs j of gamma_jc * dz_jc * 1[BN_j(x)_c > 0]
 * and then dL/dx_c = invstd_c * (T_c - mean(T_c) - xhat_c * mean(T_c xhat_c)) (pdes_bn_backward_finalize),
 * which is exactly the sum of the BatchNorm backward formulas of all consumers (shared batch stats).
 */
#define PDES_UPSAMPLE_NEAREST 1      /* nearest x2 between BN-ReLU and this convolution (fused into its operand load) */
#define PDES_UPSAMPLE_BILINEAR_OP 2 /* this descriptor is NOT a convolution (ksize = 0, Cout = Cin, Hout = 2 Hin): it
                                       writes out = bilinear_x2(relu(bn(x))), align_corners=True (codec.py:33-40); the
                                       convolution that follows reads `out` through an identity BatchNorm whose
                                       statistics {0, n (1 - eps)} this op stores in out_stats (replica 0).  Backward:
                                       pdes_conv_backward_data applies the adjoint resampling + ReLU mask + gamma into
                                       t_in, dgamma / dbeta into bn_grad; there is no weight gradient and `g` must be
                                       dL/d(out) itself (fin_tstats = NULL) */
#define PDES_OP_CHANNEL_MASK 3       /* this descriptor is NOT a convolution (ksize = 0, Cout = Cin, same map size): nn.Dropout2d
                                       after the previous descriptor's convolution (codec.py:70-71, :111-120, :134-150,
                                       :172-173).  out[b, out_coff + c] *= w[b * Cout + c] in place (`w` = B x Cout floats, 0
                                       or 1/(1-p), drawn by the caller), out_stats accumulated here (the convolution before it
                                       passes out_stats = NULL); backward: g[b, g_coff + c] *= w[b * Cout + c] in place, after
                                       the finalize of these channels, which is carried by THIS descriptor's fin_* */
/* ---- flow ops (models/glow_msc.py of the reference; z -> y direction unless flags & PDES_FLOW_FORWARD).  Descriptors
 * with ksize = 0 that are not convolutions; they use the fields appended at the end of pdes_conv_desc (x2, t2, p0, p1,
 * acc, flags).  Buffers without a BatchNorm consumer keep their gradient in the same T tensor (fin_tstats = NULL); a
 * buffer read BOTH through BatchNorms and directly collects the direct consumers' gradients in a second tensor that
 * the producer's finalize adds (g_add).  H*W must be a multiple of 4. */
#define PDES_OP_COPY 4        /* out[:, out_coff + c] = x[:, c], c < Cin, and (Cout > Cin) out[:, out_coff + Cin + c] = x2[:, c],
                                 c < Cout - Cin: torch.cat((y1, cond), 1) of glow_msc.py:321,339 in one launch (torch.cat((x, out), 1),
                                 :43, with Cout = Cin); out_stats accumulated.  Backward (after this descriptor's finalize):
                                 t_in[:, c] (+)= g[:, g_coff + c], t2[:, c] += g[:, g_coff + Cin + c]; NULL targets: input data.
                                 With g_fused = 1 (set by the CALLER for this operator) pdes_backward skips the finalize launch and the
                                 backward kernel applies it while reading g = T */
#define PDES_OP_BIAS_SCALE 5  /* in place: out = (out + p0[c]) * exp(3 p1[c]) on channels [out_coff, out_coff + Cout); p1 = NULL: bias
                                 only (nn.Conv2d(bias=True), glow_msc.py:34-35; Conv2dZeros, :237-252).  out_stats accumulated when
                                 given (then this descriptor, not the convolution before it, carries fin_*).  Backward, in place on g:
                                 g *= exp(3 p1); acc (Cout, 2) += {sum g exp(3 p1), 3 sum g out} = {dbias, dscale} */
#define PDES_OP_COUPLING 6    /* AffineCouplingLayer.reverse (glow_msc.py:336-345): x (C channels), x2 = the coupling net's output h
                                 (2 n2 channels, n2 = C / 2, n1 = C - n2): out[:n1] = x[:n1]; s = sigmoid(h[2k+1] + 2);
                                 out[n1 + k] = x[n1 + k] / s - h[2k]; acc[b] += sum log s.  PDES_FLOW_FORWARD (:317-334):
                                 out[n1 + k] = (x[n1 + k] + h[2k]) * s.  Backward (z -> y only): t_in = dL/dx (written), t2 = dL/dh,
                                 p1 = dL/d(logp) per sample (B floats, nullable).  gamma != NULL: x2 holds the RAW output of the
                                 coupling net's last convolution and this descriptor applies its Conv2dZeros epilogue (:237-252)
                                 h = (x2 + gamma[c]) * exp(3 beta[c]) (beta = NULL: bias only), rewriting x2 with h; backward:
                                 t2 = dL/d(raw) = dL/dh exp(3 beta), bn_grad (2 n2, 2) += {dbias, dscale} (the PDES_OP_BIAS_SCALE
                                 layout) -- no PDES_OP_BIAS_SCALE descriptor between the convolution and the coupling */
#define PDES_OP_MIX 7         /* InvertibleConv1x1[LU].reverse + ActNorm.reverse (glow_msc.py:390-397): out = (W x - p1) / p0 per pixel,
                                 W = x2 (C, C) row-major, p0 / p1 = ActNorm weight / bias.  PDES_FLOW_FORWARD (:383-388): out = W (p0 x + p1)
                                 (the caller passes the inverse matrix).  Backward: t_in = W^T (g / p0) (written);
                                 acc (2 C + C C doubles) += {dL/dp0, dL/dp1, dL/dW} (the log-determinants depend on the parameters
                                 only: pdes_flow_prepare / pdes_flow_param_grads) */
#define PDES_OP_UNSQUEEZE 8   /* Squeeze.reverse, factor 2 (glow_msc.py:422-432): out[c][i H + h][j W + w] = x[4 c + 2 i + j][h][w]
                                 (QUADRANT layout); PDES_FLOW_FORWARD: Squeeze.forward (:410-420), x (Cin, 2H, 2W) -> out (4 Cin, H, W).
                                 Backward: the inverse permutation of g into t_in (written) */
#define PDES_OP_GAUSS 9       /* GaussianDiag.sample + log_prob (glow_msc.py:435-458) of a prior x2 = (mean | log-stddev) (2 n channels,
                                 n = Cout), p0 = eps (B, n, H, W): out = mean + exp(l) eps, l = clamp(log-stddev, -10, log 5);
                                 acc[b] += sum -0.5 (log 2 pi + 2 l + (out - mean)^2 / exp(2 l)).  PDES_FLOW_FORWARD (Split.forward,
                                 :561-573): x = the given latent, acc[b] += its log-probability, out (nullable) = (x - mean) / exp(l).
                                 Backward: t2 = dL/d(prior) (autograd of both uses of the prior; PDES_GAUSS_DETACH_LSD: no gradient to the
                                 log-stddev, the top latent's `.data`, :524), p1 = dL/d(logp) */
#define PDES_FLOW_FORWARD 1
#define PDES_GAUSS_DETACH_LSD 2
#define PDES_MIX_COUPLED 4    /* PDES_OP_MIX, z -> y only: the affine coupling in front of the invertible 1x1 runs in the same launch.
                                 x = the COUPLING's input (C channels), h / h_ctot = the coupling net's output (as x2 of
                                 PDES_OP_COUPLING, gamma / beta / bn_grad = its folded Conv2dZeros epilogue), acc2[b] += sum log s;
                                 the coupling's output u = (x[:n1], x[n1 + k] / s - h[2k]) is the mix's input and is never
                                 stored: out = (W u - p1) / p0.  Backward: t_in = dL/dx (written), th = dL/dh (dL/d(raw) with
                                 gamma), cst = dL/d(logp) per sample, acc as for PDES_OP_MIX with u recomputed from x and h */

typedef struct pdes_conv_desc {
  /* geometry */
  int B, Cin, Cout, Hin, Win, Hout, Wout;
  int ksize, stride, pad, upsample; /* upsample: 0, PDES_UPSAMPLE_NEAREST, or an op: PDES_UPSAMPLE_BILINEAR_OP / PDES_OP_CHANNEL_MASK */
  /* input activation (raw, pre-BN) */
  const float* x;        /* (B, x_ctot, Hin, Win); channels [0, Cin) are read */
  int x_ctot;
  int has_bn;            /* 0: first convolution, input used as is (no BN, no ReLU) */
  int eval_mode;         /* 1: BN uses run_mean/run_var instead of batch statistics */
  float eps;
  const float* gamma;    /* (Cin) BN weight */
  const float* beta;     /* (Cin) BN bias */
  const double* x_stats; /* (x_ctot, 2) {sum x, sum x^2} over (B,Hin,Win), train mode */
  const float* run_mean; /* (Cin) eval mode */
  const float* run_var;  /* (Cin) eval mode */
  /* weights: reference layout and the two packed copies made by pdes_pack_weights */
  const float* w;        /* (Cout, Cin, k, k) */
  const float* w_fwd;    /* (Cin, k*k, cout_pad) */
  const float* w_bwd;    /* (Cout, k*k, cin_pad) */
  int cout_pad, cin_pad; /* multiples of 16 */
  const float* wm_fwd;   /* MFMA image of w for the forward, or NULL (pdes_pack_weights_mfma) */
  const float* wm_bwd;   /* MFMA image of w for the data gradient, or NULL */
  const float* wu_fwd;   /* upsample+3x3 only: effective 2x2 weights (sub-pixel decomposition), or NULL */
  const float* wu_bwd;   /* same for the data gradient, or NULL (pdes_pack_weights_up) */
  const unsigned short* wb_fwd; /* wide 3x3 layers: bf16 hi/mid/lo split image for the forward, or NULL (pdes_pack_weights_b3) */
  const unsigned short* wb_bwd; /* same for the data gradient, or NULL */
  /* output */
  float* out;            /* (B, out_ctot, Hout, Wout); channels [out_coff, out_coff+Cout) written */
  int out_ctot, out_coff;
  double* out_stats;     /* (out_ctot, 2) accumulated for the written channels; NULL: none */
  const double* fin_xstats; /* pdes_backward only: {sum x, sum x^2} and {sum T, sum T xhat} tables of the */
  const double* fin_tstats; /* OUTPUT buffer for its BN-backward finalize; NULL = `g` is already dL/d(out) */
  /* backward */
  const float* g;        /* dL/d(out): (B, g_ctot, Hout, Wout), channels [g_coff, g_coff+Cout) */
  int g_ctot, g_coff;
  int g_fused;           /* 1: `g` still holds the accumulator T of the output buffer and the consuming kernel applies the
                            BN-backward finalize (fin_xstats / fin_tstats, raw activation = `out`) on operand load.  Callers set
                            it for PDES_OP_COPY only and pass 0 for convolutions; pdes_backward sets it itself on the copies of
                            the descriptors it hands to the finalize-on-load kernels (PDES_FIN_ONLOAD: the dense layers, the
                            first convolution's weight gradient, the 8x8-map kernels).  CONTRACT: on that path `g` is LEFT as the
                            accumulator T -- the finalized gradient dL/d(out) of such a layer exists in registers only; a
                            caller that wants it in memory (need_input_grad of the first layer, debugging) runs with
                            PDES_FIN_ONLOAD=0.  {mean, invstd} come from the published table `fin_coef` where the kernel has it
                            at hand (conv_mfma*), and from the fp64 replica sums by the same expression
                            ((float) m, (float)(1 / sqrt(var + eps)), var = E[x^2] - m^2 clamped at 0, all in fp64) elsewhere
                            (conv_bwd_weight_first, conv_small): the table's publisher evaluates that expression too */
  float* t_in;           /* T accumulator of the input buffer (B, x_ctot, Hin, Win) */
  int t_accumulate;      /* 0: T = ..., 1: T += ... */
  int final_c0, final_c1;/* input channels whose T is complete after this call: their
                            {sum T, sum T*xhat} are accumulated into t_stats */
  double* t_stats;       /* (x_ctot, 2) */
  double* bn_grad;       /* (Cin, 2) {dgamma, dbeta} accumulators (fp64) */
  float* dw;             /* (Cout, Cin, k, k) weight gradient, ACCUMULATED (zero it first) */
  float* ws;             /* scratch for split-K partial weight gradients (may be NULL) */
  long long ws_bytes;
  int ws_defer;          /* 1: leave the partials in `ws` for ONE pdes_wgrad_reduce_all at the end */
  /* the fp64 accumulators (x_stats, out_stats, t_stats, bn_grad) exist in `nrep` replicas,
     `rep_stride` doubles apart, to spread same-address atomics; readers sum the replicas */
  int nrep;
  long long rep_stride;
  const unsigned short* wbu_fwd; /* nearest-x2 + 3x3 layers: bf16 hi/mid/lo split image of the effective sub-pixel weights
                                    for the forward (pdes_pack_weights_b3up), or NULL */
  /* appended in ABI 14 (NULL / 0 for the DenseED chains) */
  const float* g_add;    /* pdes_backward: the finalize of this descriptor's output channels ADDS g_add (layout of g): the gradient
                            from consumers that read `out` without a BatchNorm */
  const float* x2;       /* flow ops: second source (see PDES_OP_*) */
  int x2_ctot;
  float* t2;             /* flow ops: gradient of the second source */
  const float* p0;       /* flow ops: per-channel parameters / noise */
  const float* p1;
  double* acc;           /* flow ops: fp64 accumulators (replicated like the statistics: nrep, rep_stride) */
  int flags;             /* PDES_FLOW_FORWARD, PDES_GAUSS_DETACH_LSD */
  /* appended in ABI 15 */
  const unsigned short* wbu_bwd; /* nearest-x2 + 3x3 layers: split image of the effective sub-pixel weights for the DATA gradient
                                    (pdes_pack_weights_b3up), or NULL */
  /* appended in ABI 20 */
  float* coef;           /* (x_ctot, 2) {batch mean, invstd} of the input buffer's channels, or NULL.  The caller zeroes it together
                            with the statistics at the start of every step; a kernel that sums the replicas of a channel
                            publishes the result here (invstd > 0 marks a valid entry), the kernels after it read 8 bytes
                            instead of 2 x nrep doubles.  Train mode only */
  /* appended in ABI 21 */
  const float* fin_coef; /* the same table of the OUTPUT buffer (next to fin_xstats), for the finalize of this layer's output
                            channels; NULL: the finalize sums the replicas of x itself */
  /* appended in ABI 24: PDES_OP_MIX with flags & PDES_MIX_COUPLED (NULL / 0 otherwise) */
  const float* h;        /* the coupling net's output, (B, h_ctot, H, W); rewritten in place with the Conv2dZeros epilogue if gamma */
  int h_ctot;
  float* th;             /* backward: dL/dh (2 n2 channels of h_ctot) */
  double* acc2;          /* forward: the coupling's log-determinant accumulator (B doubles per replica) */
  const float* cst;      /* backward: dL/d(logp) per sample (B floats, nullable) */
} pdes_conv_desc;

/* `descs` is a HOST array; one kernel launch per descriptor, in order.
 * out = conv(relu(bn(x))) for `n` descriptors in order; accumulates out_stats.
 * Replaces nn.BatchNorm2d + nn.ReLU + [UpsamplingNearest2d] + nn.Conv2d + torch.cat
 * (models/codec.py:43-75, :89-160, :163-188, :242-243). */
int pdes_conv_forward(const pdes_context* ctx, const pdes_conv_desc* descs, int n, void* stream);
/* dw += sum_{b,p} g * relu(bn(x)) (autograd of F.conv2d wrt weight). */
int pdes_conv_backward_weight(const pdes_context* ctx, const pdes_conv_desc* descs, int n, void* stream);
/* T_in (+)= gamma * (conv^T(g)) * 1[bn(x) > 0]; also dgamma/dbeta and the finished channels'
 * {sum T, sum T xhat} (autograd of conv2d wrt input, ReLU, BatchNorm wrt gamma/beta). */
int pdes_conv_backward_data(const pdes_context* ctx, const pdes_conv_desc* descs, int n, void* stream);
/* Which packed weight images the forward / data-gradient kernels selected for `desc` under ctx's options READ (a
 * capability query of the two dispatch chains above: nothing is enqueued).  The weight-gradient kernels read no weights.
 * A caller that rebuilds its images every step (pdes_pack_all2) can leave out the ones no pass of any of its
 * descriptors reads -- and must re-query when it changes an option of the context. */
#define PDES_IMG_DIRECT_FWD 1      /* pdes_pack_item.w_fwd   */
#define PDES_IMG_DIRECT_BWD 2      /* pdes_pack_item.w_bwd   */
#define PDES_IMG_MFMA_FWD 4        /* pdes_mfma_pack_item.wm_fwd */
#define PDES_IMG_MFMA_BWD 8        /* ... wm_bwd */
#define PDES_IMG_UP_FWD 16         /* pdes_up_pack_item.wu_fwd */
#define PDES_IMG_UP_BWD 32
#define PDES_IMG_B3_FWD 64         /* pdes_b3_pack_item.wb_fwd */
#define PDES_IMG_B3_BWD 128
#define PDES_IMG_B3UP_FWD 256      /* pdes_b3up_pack_item.wbu_fwd */
#define PDES_IMG_B3UP_BWD 512
int pdes_conv_image_use(const pdes_context* ctx, const pdes_conv_desc* desc, int* mask);
/* In place T -> dL/dx for channels [c0, c1) of a (B, ctot, H, W) buffer (BatchNorm backward wrt
 * its input, summed over every consumer BN). */
int pdes_bn_backward_finalize(const pdes_context* ctx, float* t, const float* x, const double* x_stats, const double* t_stats,
                              int B, int ctot, int c0, int c1, int HW, float eps, int nrep,
                              long long rep_stride, void* stream);

/* Split plan of the matrix-core weight-gradient kernel for `d` (uses d->ws_bytes as the scratch
 * limit): number of pixel splits and floats of scratch it writes.  PDES_ENOSUP: this layer runs on
 * the generic kernel (fp32 atomics straight into dw, no scratch). */
int pdes_conv_wgrad_plan(const pdes_context* ctx, const pdes_conv_desc* d, int* nsplit, long long* floats);
typedef struct pdes_reduce_item { const float* part; float* dw; int n; int nsplit; } pdes_reduce_item;
/* dw[i] += sum_s part[s][i] (fixed order) for every item; items: DEVICE array. */
int pdes_wgrad_reduce_all(const pdes_reduce_item* items, int n, int max_n, void* stream);

/* The whole backward pass of a descriptor chain -- what `loss.backward()` (train_codec_mixed_residual.py:233)
 * runs through autograd for models/codec.py:43-188 in the reference: for i = n-1 .. 0
 *   [pdes_bn_backward_finalize of descs[i]'s output channels, when fin_tstats != NULL]
 *   pdes_conv_backward_weight(descs[i])   -- on `wgrad_stream` when it is not NULL
 *   pdes_conv_backward_data(descs[i])     -- when has_bn
 * followed by the split-K reduce of the deferred weight-gradient partials (pdes_wgrad_reduce_all).
 * The weight gradients have no consumer before that reduce, so with a second stream they overlap
 * the finalize -> data-gradient dependency chain; each layer is released to it as soon as its
 * output gradient exists (one event per layer) and the widest layers' partials are reduced on that stream as soon as
 * they exist; the two streams are joined before returning (everything is ordered on `stream` again).
 * reduce_items: DEVICE table as for pdes_wgrad_reduce_all, in layer order; reduce_index: HOST
 * array, reduce_index[i] = table index of descs[i] or -1 (no deferred scratch).  Both may be NULL
 * (then the caller reduces).
 * With a second stream the context must hold >= n + 4 events (PDES_EINVAL otherwise).
 * hook (nullable): called ONCE, on the host, right after the early split-K reduce has been enqueued on
 * `wgrad_stream`: the weight gradients `dw` of layers [first_layer, n) are final in stream order on
 * `wgrad_stream` from that point.  The data-parallel trainer enqueues the all-reduce of that bucket there
 * (RCCL over xGMI), so the exchange of ~3/4 of the gradient bytes overlaps the rest of the backward pass.
 * A non-zero return aborts pdes_backward with that code.  Not called when nothing was reduced early. */
typedef struct pdes_bucket_hook {
  int (*fn)(void* user, int first_layer, void* wgrad_stream);
  void* user;
} pdes_bucket_hook;
int pdes_backward(const pdes_context* ctx, const pdes_conv_desc* descs, int n, void* stream, void* wgrad_stream,
                  const pdes_reduce_item* reduce_items, const int* reduce_index, const pdes_bucket_hook* hook);
/* The same with an optional SECOND weight-gradient stream (`wgrad_stream_b`, NULL = pdes_backward): the weight
 * gradients of successive layers are independent, odd layers are enqueued on it; it is joined into `wgrad_stream`
 * before every split-K reduce and at the end.  The context must hold >= n + 4 events. */
int pdes_backward2(const pdes_context* ctx, const pdes_conv_desc* descs, int n, void* stream, void* wgrad_stream,
                   void* wgrad_stream_b, const pdes_reduce_item* reduce_items, const int* reduce_index,
                   const pdes_bucket_hook* hook);

/* The two halves of that pass for the layers [lo, hi) of `descs`, each on ONE stream and without events (for i = hi-1 .. lo):
 *   pdes_backward_chain:   [finalize of descs[i]'s output channels] + pdes_conv_backward_data(descs[i])
 *   pdes_backward_weights: pdes_conv_backward_weight(descs[i])
 * The caller orders them (chain of a range before its weights) and reduces the split-K partials.  These are what the
 * segment graphs below are captured from. */
int pdes_backward_chain(const pdes_context* ctx, const pdes_conv_desc* descs, int lo, int hi, void* stream);
int pdes_backward_weights(const pdes_context* ctx, const pdes_conv_desc* descs, int lo, int hi, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The step as LINEAR hipGraphs (csrc/step_graph.hip; no reference counterpart -- the reference launches ~2,000 aten
 * kernels per step one by one).  On this runtime an eager launch costs the host 3-9 us and leaves ~3 us between dependent
 * kernels, a linear graph of 120 kernels replays for 8 us with ~1.7 us between kernels, and a graph with forks is slower
 * than eager launches on two streams: so the step is cut into linear pieces (forward + loss; per backward segment the
 * finalize -> data-gradient chain, and separately its weight gradients) joined by one event per segment.
 *   pdes_graph_begin / pdes_graph_end  bracket ANY sequence of this library's enqueue-only calls on `stream` (not the
 *                         legacy default stream); end returns the instantiated graph.  Nothing executes during capture.
 *   pdes_graph_launch     replay on any stream;  pdes_graph_nodes: kernel / memset nodes captured.
 *   pdes_program_run      replay a list of operations in one call: launch graphs[arg] on streams[stream], record /
 *                         wait the CONTEXT's event number arg on streams[stream], or call the bucket hook with
 *                         first_layer = arg (data parallel: see pdes_backward).  The caller owns graphs and list. */
typedef struct pdes_graph pdes_graph;
#define PDES_OP_LAUNCH 0
#define PDES_OP_RECORD 1
#define PDES_OP_WAIT 2
#define PDES_OP_HOOK 3
typedef struct pdes_op { int kind, arg, stream; } pdes_op;
int pdes_graph_begin(void* stream);
int pdes_graph_end(void* stream, pdes_graph** out);
int pdes_graph_nodes(const pdes_graph* g);
int pdes_graph_launch(pdes_graph* g, void* stream);
int pdes_graph_destroy(pdes_graph* g);
int pdes_program_run(const pdes_context* ctx, pdes_graph* const* graphs, int n_graphs, void* const* streams, int n_streams,
                     const pdes_op* ops, int n_ops, const pdes_bucket_hook* hook);

/* ---------------------------------------------------------------------------------------------
 * Parameter side of the flow (models/glow_msc.py): one launch each for ALL invertible 1x1 convolutions + ActNorms.
 * pdes_flow_prepare: W = P (L * l_mask + I) (U * u_mask + diag(exp(log_s) sign_s)) (InvertibleConv1x1LU.weight, :205-208), or
 *   W = weight (InvertibleConv1x1, :99-157); optionally W^-1 and log|det W| by fp64 Gauss-Jordan (needed for the plain
 *   parameterisation and for the y -> z direction); logdet[layer] = HW (sum log|actnorm weight| - log|det W|) (double):
 *   the layer's contribution to EVERY sample's log p (ActNorm.reverse :90-96, conv1x1.reverse :150-157 / :222-233).
 * pdes_flow_param_grads: from the accumulators of PDES_OP_MIX's backward (acc: 2 C + C C doubles per layer) and
 *   cB = sum_b dL/d(logp_b): ActNorm weight / bias gradients (+ cB HW / weight), and dW chained into dl, du, dlog_s
 *   (- cB HW) or into dweight (- cB HW W^-T); gradients are ACCUMULATED (fp32).  C <= 48 (five-level flows would need 96). */
typedef struct pdes_flow_item {
  int C, HW, lu;                       /* lu = 1: l, u, log_s, p, sign_s; 0: weight */
  const float* l; const float* u; const float* log_s; const float* p; const float* sign_s; const float* weight;
  const float* an_weight; const float* an_bias;    /* ActNorm (C) */
  float* W; float* Winv;               /* (C, C) outputs; Winv nullable when lu = 1 */
  double* logdet;                      /* 1 double, written */
  const double* acc;                   /* backward accumulators of this layer (replicated) */
  float* dl; float* du; float* dlog_s; float* dweight; float* dan_weight; float* dan_bias;
} pdes_flow_item;
int pdes_flow_prepare(const pdes_flow_item* items, int n, int need_inverse, void* stream);        /* items: DEVICE array */
int pdes_flow_param_grads(const pdes_flow_item* items, int n, const float* glogp, int B, int nrep, long long rep_stride,
                          void* stream);
/* logp[b] = sum of the replicas of acc[b] + sum of logdet[0..n_layers) (fp32 out); what generate() returns as log p(y|x) */
int pdes_flow_logp(const double* acc, const double* logdet, int n_layers, float* logp, int B, int nrep, long long rep_stride,
                   void* stream);

/* Table-driven helpers: one launch for the whole network. */
typedef struct pdes_pack_item {  /* one convolution's weights */
  const float* w; float* w_fwd; float* w_bwd;
  int Cout, Cin, kk, cout_pad, cin_pad;
} pdes_pack_item;
/* items: DEVICE array of n entries; total = max elements over items (grid sizing). */
int pdes_pack_weights(const pdes_pack_item* items, int n, int max_elems, void* stream);

typedef struct pdes_mfma_pack_item { /* one convolution's matrix-core weight images */
  const float* w;        /* (Cout, Cin, kk) */
  float* wm_fwd;         /* (ceil(Cin/16)*4, kk, ceil(Cout/16), 64) */
  float* wm_bwd;         /* (ceil(Cout/16)*4, kk, ceil(Cin/16), 64), or NULL */
  int Cout, Cin, kk;
} pdes_mfma_pack_item;
/* items: DEVICE array; rebuilds both images from the live weights (zero padded). */
int pdes_pack_weights_mfma(const pdes_mfma_pack_item* items, int n, int max_elems, void* stream);

typedef struct pdes_up_pack_item { /* one nearest-x2 + 3x3 convolution: effective-weight images */
  const float* w;        /* (Cout, Cin, 3, 3) */
  float* wu_fwd;         /* (ceil(Cin/16)*4, 16, ceil8(ceil(Cout/16)), 64) */
  float* wu_bwd;         /* (ceil(Cout/16)*4, 16, ceil8(ceil(Cin/16)), 64) */
  int Cout, Cin;
} pdes_up_pack_item;
/* items: DEVICE array; see csrc/conv_mfma_up.hip for the image layout. */
int pdes_pack_weights_up(const pdes_up_pack_item* items, int n, int max_elems, void* stream);
typedef struct pdes_b3_pack_item { /* one wide 3x3 convolution: three-way bf16 split weight images (conv_mfma_b3.hip) */
  const float* w; unsigned short* wb_fwd; unsigned short* wb_bwd;   /* either image may be NULL */
  int Cout, Cin;
} pdes_b3_pack_item;
/* fp32 weights -> hi/mid/lo bf16 planes in the B-operand layout of v_mfma_f32_16x16x32_bf16; sizes from
 * pdes_b3_image_elems (bf16 elements; buffers 16-byte aligned).  total = max elements / 24 over the items. */
int pdes_pack_weights_b3(const pdes_b3_pack_item* items, int n, int max_elems, void* stream);
int pdes_b3_image_elems(int Cout, int Cin, long long* fwd_elems, long long* bwd_elems);
typedef struct pdes_b3up_pack_item { /* one nearest-x2 + 3x3 convolution: split images of its effective 2x2 weights (conv_mfma_b3_up.hip,
                                        and conv_mfma_b3.hip's sub-pixel data gradient); either image may be NULL */
  const float* w; unsigned short* wbu_fwd; unsigned short* wbu_bwd; int Cout, Cin;
} pdes_b3up_pack_item;
int pdes_pack_weights_b3up(const pdes_b3up_pack_item* items, int n, int max_elems, void* stream);   /* max_elems = elements / 24 */
int pdes_b3up_image_elems(int Cout, int Cin, long long* fwd_elems, long long* bwd_elems);
/* The four tables above in ONE launch (any of them may be empty: n = 0); max_elems = the largest packed image
 * (work items: elements, or elements / 24 for the bf16 split images).
 * No reference counterpart: the packed images replace the (Cout,Cin,k,k) weight reads of nn.Conv2d
 * (models/codec.py:57-58, 106-146, 170-186) with coalesced 256-B / 1-KiB operand loads. */
int pdes_pack_all(const pdes_pack_item* items, int n, const pdes_mfma_pack_item* mitems, int nm,
                  const pdes_up_pack_item* uitems, int nu, const pdes_b3_pack_item* bitems, int nb,
                  int max_elems, void* stream);
/* The same with a fifth table: the split effective-weight images of the sub-pixel layers (pdes_pack_weights_b3up). */
int pdes_pack_all2(const pdes_pack_item* items, int n, const pdes_mfma_pack_item* mitems, int nm,
                   const pdes_up_pack_item* uitems, int nu, const pdes_b3_pack_item* bitems, int nb,
                   const pdes_b3up_pack_item* buitems, int nbu, int max_elems, void* stream);

typedef struct pdes_bn_item {    /* one BatchNorm layer */
  const double* x_stats;  /* (>=C, 2) batch sums of its input channels */
  const double* bn_grad;  /* (C, 2) {dgamma, dbeta} */
  float* run_mean; float* run_var; /* (C) updated in place */
  float* dgamma; float* dbeta;     /* (C) fp32 gradients, ACCUMULATED */
  long long* num_batches_tracked;  /* scalar, incremented */
  int C; int count;                /* count = B*H*W */
} pdes_bn_item;
/* running_mean/var <- momentum update with batch mean / unbiased var (nn.BatchNorm2d train mode). */
int pdes_bn_update_running(const pdes_bn_item* items, int n, int max_c, float momentum, int nrep,
                           long long rep_stride, void* stream);
/* dgamma/dbeta (fp32) += fp64 accumulators. */
int pdes_bn_param_grads(const pdes_bn_item* items, int n, int max_c, int nrep, long long rep_stride,
                        void* stream);

/* Adam step on flat fp32 buffers, torch.optim.Adam semantics (L2 weight decay added to the
 * gradient, bias correction, eps outside the sqrt).  lr and step live in DEVICE memory so a
 * captured hipGraph can replay with a new learning rate: hyper = {lr, beta1, beta2, eps,
 * weight_decay, 1-beta1^step, sqrt(1-beta2^step)} (7 floats, bias corrections computed by the
 * host in double like torch).  grad_scale multiplies the gradient first
 * (1/world_size after a SUM all-reduce).
 * Replaces optimizer.step() of train_codec_mixed_residual.py:239 (torch.optim.Adam, :151-152). */
int pdes_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                   const float* hyper, float grad_scale, long long n, void* stream);

/* The same step with the 7 hyper-parameters read from HOST memory at call time and passed to the
 * kernel by value: the eager training step needs no host->device copy of them (an asynchronous copy
 * from a reused pinned buffer would race with the host running ahead of the stream).
 * zero_grad != 0: the kernel also clears `grad` after reading it (optimizer.zero_grad() of the NEXT
 * iteration, :225, without a separate fill launch).
 * Returns PDES_EINVAL when a bias correction is not positive (step 0). */
int pdes_adam_step_host(float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                        const float* hyper_host, float grad_scale, int zero_grad, long long n, void* stream);
/* The same, additionally clearing `nclear` doubles at `clear` (the fp64 statistics arena of the step that just ended:
 * its last reader, pdes_step_tail, precedes this call in stream order), so the next forward starts without a fill launch. */
int pdes_adam_step_host2(float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                         const float* hyper_host, float grad_scale, int zero_grad, long long n,
                         double* clear, long long nclear, void* stream);

/* End of a training step in ONE launch: pdes_bn_param_grads, optionally pdes_bn_update_running
 * (update_running != 0; same arithmetic), and -- when `partials` is given -- the reduction of the
 * per-image loss partials of pdes_darcy_loss (called with loss_out = NULL) into terms[5] = {total,
 * const, cont, dirichlet, neumann} (nullable) and terms_accum[5] += terms (fp64, nullable): the
 * per-epoch loss sums of train_codec_mixed_residual.py:240 without a host sync or extra launches.
 * The weights are those given to pdes_darcy_loss; `partials` holds pdes_darcy_loss_partial_rows(B, H, W, flags) rows of a
 * launch whose flags had neither PDES_LOSS_NO_TB nor PDES_LOSS_UNCORRECTED. */
int pdes_step_tail(const pdes_bn_item* items, int n, int max_c, float momentum, int update_running,
                   const float* partials, int B, int H, int W, float w_const, float w_cont, float w_dir,
                   float w_neu, float* terms, double* terms_accum, int nrep, long long rep_stride,
                   void* stream);

/* ---------------------------------------------------------------------------------------------
 * Test-time metrics and the data-driven (maximum-likelihood) loss: the steps either side of the hot loop.
 *
 * pdes_test_metrics: per (image, channel) {sum_hw (output-target)^2, sum_hw target^2} -> per_image (B, C, 2), and,
 * when `accum` is given (2C+1 doubles, zeroed by the caller before the first batch),
 *   accum[c] += sum_b sqrt(err2/t2), accum[C+c] += sum_b err2, accum[2C] += B   (fixed order: deterministic).
 * After the last batch: relative-l2 (NRMSE) = accum[c] / accum[2C]; R^2 = 1 - accum[C+c] / y_variation[c].
 * Replaces train_codec_mixed_residual.py:180-183 (err2_sum, relative_l2, err2 per batch) and :196-197 (the
 * torch.cat(...).mean(0) / .sum(0) at the end) without per-batch host syncs; C <= 64.
 */
int pdes_test_metrics(const float* output, const float* target, float* per_image, double* accum, int B, int C,
                      int HW, void* stream);

/* loss = mean((output - target)^2) over n elements and, when grad_out is given, grad_out = 2 (output - target) / n.
 * partials: workspace of pdes_mse_partials(n) doubles (per-block sums, reduced in a fixed order: deterministic).
 * loss_out (1 float, nullable) and loss_accum (1 double, += loss, nullable) are written by a one-block finalize.
 * Replaces F.mse_loss + loss.backward() wrt output, train_codec_max_likelihood.py:170,203-204. */
int pdes_mse_partials(long long n);
int pdes_mse_loss(const float* output, const float* target, float* grad_out, double* partials, float* loss_out,
                  double* loss_accum, long long n, void* stream);

/* Inner products of the L-BFGS curvature history against up to four vectors, in one bandwidth-bound pass:
 *   partials[s][r][j] = sum over slice s of k of W[r][k] * V[j][k]     (fp64; the caller sums the nsplit slices)
 * W: (rows, ld) row-major with ld >= n a multiple of 4, V: (nv, n) row-major, nv <= 4; rows of both 16-byte aligned.
 * Used by pde_surrogate_amd/lbfgs.py, which restates the two-loop recursion of torch.optim.LBFGS (the optimiser of
 * solve_conv_mixed_residual.py:124) on the Gram matrix of the history. */
int pdes_multi_dot(const float* W, long long ld, int rows, const float* V, int nv, long long n, double* partials,
                   int nsplit, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PDES_HIP_H */
