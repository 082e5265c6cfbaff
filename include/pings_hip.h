/* pings_hip.h — C ABI of libpings_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary of the PINGS render + SDF-query hot path
 * (SURVEY.md §8b).  Every entry point is `extern "C"`, takes plain device
 * pointers, sizes and an opaque `hipStream_t` (passed as void*), returns an
 * integer status (0 = ok) and never throws.  All pointers are DEVICE pointers
 * unless a parameter is documented as host.  All floating point is fp32, all
 * tensors are dense row-major (the layout torch hands over).
 *
 * The reference binds this functionality through three CUDA torch extensions
 * whose sources are absent from the reference checkout (empty submodules,
 * /root/reference/.gitmodules:1-9) and through plain PyTorch code; each block
 * below cites the reference call site it replaces.
 */
#ifndef PINGS_HIP_H_
#define PINGS_HIP_H_

#include <stddef.h>
#include <stdint.h>

#if defined(__cplusplus)
#define PINGS_EXTERN_C extern "C"
#else
#define PINGS_EXTERN_C
#endif
#define PINGS_API PINGS_EXTERN_C __attribute__((visibility("default")))

/* status codes */
#define PINGS_OK 0
#define PINGS_ERR_ARG 1      /* bad shape / null pointer / unsupported option */
#define PINGS_ERR_HIP 2      /* a HIP runtime call failed                      */
#define PINGS_ERR_CAPACITY 3 /* caller-provided buffer too small               */

/* ------------------------------------------------------------------ general */

/* ABI version of this header (bumped on any signature change). */
PINGS_API int pings_abi_version(void);
/* Message of the last failing call on this thread ("" if none). Host string. */
PINGS_API const char* pings_last_error(void);

/* --------------------------------------------------------------- fused SSIM
 * Replaces `fused_ssim.fused_ssim(img1, img2, train=...)`
 * (reference call sites utils/mapper.py:1243,1922,1951; arithmetic of the
 * in-tree torch predecessor gaussian_splatting/utils/loss_utils.py:189-219:
 * 11x11 Gaussian window sigma 1.5, zero "same" padding, C1=0.01^2, C2=0.03^2,
 * mean over all elements).  Images are [planes, H, W] with planes = N*C.
 */

/* Number of floats of scratch `partials` needed by pings_ssim_forward. */
PINGS_API size_t pings_ssim_partials_count(int planes, int H, int W);

/* out_mean[1] <- mean SSIM.  If train != 0 the three partial-derivative maps
 * ([planes,H,W] each) are written for the backward pass; otherwise they may be
 * NULL.  Deterministic (two-stage reduction, no atomics). */
PINGS_API int pings_ssim_forward(const float* img1, const float* img2, int planes, int H, int W,
                                 int train, float* out_mean, float* dm_dmu1,
                                 float* dm_dsigma1_sq, float* dm_dsigma12, float* partials,
                                 void* stream);

/* dL_dimg1[planes,H,W] <- gradient of (dL_dmean * mean SSIM) w.r.t. img1.
 * dL_dmean is a 1-element device buffer (the upstream scalar gradient). */
PINGS_API int pings_ssim_backward(const float* img1, const float* img2, int planes, int H, int W,
                                  const float* dL_dmean, const float* dm_dmu1,
                                  const float* dm_dsigma1_sq, const float* dm_dsigma12,
                                  float* dL_dimg1, void* stream);

#endif /* PINGS_HIP_H_ */
