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

/* Per-stage HIP-event timing of everything this library launches (bench.py's roofline
 * figures).  pings_prof_report synchronises the device and writes "name count total_ms"
 * lines into a HOST buffer, then clears the record. */
PINGS_API int pings_prof_enable(int on);
PINGS_API int pings_prof_report(char* buf, size_t cap);

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


/* ------------------------------------------- Gaussian(-surfel) rasteriser
 * Replaces the `diff_gaussian_surfel_rasterization` / `diff_gaussian_rasterization`
 * torch extensions (absent CUDA submodules, /root/reference/.gitmodules:1-6) at the
 * interface the reference calls them through:
 *   GaussianRasterizationSettings(...)  gaussian_renderer/__init__.py:149-166, :185-199
 *   GaussianRasterizer.markVisible      gaussian_renderer/__init__.py:215
 *   GaussianRasterizer.__call__         gaussian_renderer/__init__.py:318-326, :415-423
 *   its autograd backward               triggered by utils/mapper.py:1581
 * Semantics are those of oracle/raster_cpu.py (assumptions listed in DESIGN.md).
 *
 * Memory protocol: three opaque scratch blobs sized by the *_bytes queries and
 * allocated by the caller (torch's caching allocator in the Python wrapper);
 * the same blobs must be handed unchanged to pings_raster_backward.
 */
#define PINGS_RASTER_SURFEL 0
#define PINGS_RASTER_3DGS 1

typedef struct pings_raster_settings {
  int32_t image_height, image_width;
  int32_t mode;           /* PINGS_RASTER_SURFEL | PINGS_RASTER_3DGS                     */
  int32_t front_only;     /* config[4] of the surfel settings (surfel mode only)          */
  double tanfovx, tanfovy;
  double scale_modifier;
  const float* bg;             /* device [3]                                               */
  const float* viewmatrix;     /* device [4,4] row-major = T_cw^T   (cameras.py:214)       */
  const float* projmatrix;     /* device [4,4] = viewmatrix @ P^T   (cameras.py:216)       */
  const float* projmatrix_raw; /* device [4,4] = P^T                (cameras.py:68-70)     */
  const float* prcppoint;      /* device [2] = (cx/W, cy/H) (cameras.py:61); NULL = centre */
} pings_raster_settings;

/* present[N] (uint8) <- 1 where the point lies inside the view frustum. */
PINGS_API int pings_raster_mark_visible(const float* positions, int N,
                                        const pings_raster_settings* s, uint8_t* present,
                                        void* stream);

PINGS_API size_t pings_raster_geom_bytes(int P);
PINGS_API size_t pings_raster_binning_bytes(int64_t num_instances, int image_height,
                                            int image_width);
PINGS_API size_t pings_raster_image_bytes(int image_height, int image_width);

/* Stage 1: per-Gaussian projection, culling, depth sort and tile counting.
 * Writes radii[P] (int32, 0 = culled) and *num_instances (HOST int64: number of
 * (Gaussian, tile) pairs).  Synchronises `stream` once to return that count. */
PINGS_API int pings_raster_preprocess(const pings_raster_settings* s, int P, const float* means3D,
                                      const float* colors, const float* opacities,
                                      const float* scales, const float* rotations,
                                      void* geom_blob, int32_t* radii, int64_t* num_instances,
                                      void* stream);

/* Stage 2: instance binning (stable tile sort) and per-tile alpha blending.
 * Outputs are planar [C,H,W].  out_normal may be NULL in 3DGS mode.  `per_gaussian`
 * is float contributions[P] (surfel: sum of blend weights over pixels) or int32
 * n_touched[P] (3DGS: pixels where the Gaussian is seen with transmittance > 0.5).
 */
PINGS_API int pings_raster_render(const pings_raster_settings* s, int P, int64_t num_instances,
                                  void* geom_blob, void* binning_blob, void* image_blob,
                                  float* out_color, float* out_normal,
                                  float* out_depth, float* out_alpha, void* per_gaussian,
                                  void* stream);

/* Scratch bytes pings_raster_backward needs (an upper bound in the instance count: the
 * gradient rows of the instances that actually blended are a data-dependent subset). */
PINGS_API size_t pings_raster_backward_bytes(int P, int64_t num_instances);

/* Backward of stage 1+2.  dL_d* of the outputs may be NULL (treated as zero).
 * bwd_blob: device scratch of pings_raster_backward_bytes(P, num_instances) bytes.
 * Gradient outputs are overwritten (not accumulated); dL_dmeans2D[P,3] receives the
 * screen-space positional gradient (xy, z = 0) the reference reads through
 * `viewspace_points`; dL_dtau[6] = [d rho, d theta] for T_cw <- SE3_exp(tau) T_cw
 * (utils/campose_utils.py:79-98).  Deterministic: no floating-point atomics. */
PINGS_API int pings_raster_backward(const pings_raster_settings* s, int P, int64_t num_instances,
                                    const float* means3D, const float* colors,
                                    const float* opacities, const float* scales,
                                    const float* rotations, const void* geom_blob,
                                    const void* binning_blob, const void* image_blob,
                                    const float* out_color, const float* out_normal,
                                    const float* out_depth, const float* out_alpha,
                                    const float* dL_dcolor, const float* dL_dnormal,
                                    const float* dL_ddepth, const float* dL_dalpha,
                                    void* bwd_blob, float* dL_dmeans3D, float* dL_dmeans2D,
                                    float* dL_dcolors, float* dL_dopacities, float* dL_dscales,
                                    float* dL_drotations, float* dL_dtau, void* stream);

/* Debug / parity taps (tests only): copies of the sorted instance list and the
 * per-tile ranges out of the binning blob, and per-pixel state out of the image blob. */
PINGS_API int pings_raster_debug_lists(const void* binning_blob, int64_t num_instances,
                                       int image_height, int image_width, uint32_t* point_list,
                                       uint32_t* ranges_xy, void* stream);
PINGS_API int pings_raster_debug_image(const void* image_blob, int image_height, int image_width,
                                       float* final_T, uint32_t* n_contrib, void* stream);

#endif /* PINGS_HIP_H_ */
