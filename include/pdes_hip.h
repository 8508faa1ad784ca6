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
 *   - the caller (PyTorch) owns every buffer including workspaces; the library never allocates,
 *     frees or retains a pointer; no global state;
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

/* ABI version of this header; bumped on any signature change. */
int pdes_abi_version(void);

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
 *   partials (B,4)      OUT workspace: per-image sums {sum r1^2+r2^2, sum c^2, dirichlet, neumann}
 *   loss_out (5)        OUT {total, L_const, L_cont, L_dir, L_neu}; NULL = skip the final reduce
 *   total = w_const*L_const + w_cont*L_cont + w_dir*L_dir + w_neu*L_neu
 *           (the reference's loss is w = (1, 1, weight_bound, weight_bound))
 *   nonlinear != 0: sigma + beta1*sqrt(K)*sigma^2 + beta2*K*sigma^3 constitutive law.
 *   H == W in {16, 32, 64}.
 */
int pdes_darcy_loss(const float* K, const float* y, float* grad_y, float* partials, float* loss_out,
                    int B, int H, int W, float w_const, float w_cont, float w_dir, float w_neu,
                    int nonlinear, float beta1, float beta2, void* stream);

/* Stand-alone Sobel gradients of `nimg` single-channel images (either output may be NULL).
 * Replaces utils/image_gradient.py:50-75 (grad_h) and :77-92 (grad_v), filter_size=3. */
int pdes_sobel_grad(const float* img, float* gh, float* gv, int nimg, int H, int W, int correct,
                    void* stream);

/* img_bar = grad_h^T(gh_bar) + grad_v^T(gv_bar) (either input may be NULL); correct=True only.
 * This is what autograd computes for the reference's pad/conv2d/matmul chain. */
int pdes_sobel_grad_adjoint(const float* gh_bar, const float* gv_bar, float* img_bar, int nimg, int H,
                            int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PDES_HIP_H */
