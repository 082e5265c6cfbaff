// Image-space loss block of the mapper's photometric iteration (utils/mapper.py:1197-1295), for gfx950.
//
// The reference evaluates, per training iteration and on full frames, a chain of ~40 elementwise / masked-select /
// mean torch kernels (and autograd doubles it):
//   sky loss        alpha[sky].mean()                                                (mapper.py:1198-1218, loss_utils.py:178)
//   RGB L1          |rgb - gt|.mean() over rows [v_min, v_max)                        (:1224-1239, loss_utils.py:17)
//   depth L1        mean over (dmin < gt < dmax) & (alpha > a_min) of |gt - d|  or  |1/gt - 1/d|      (:1251-1267)
//   normal-depth    mean over (|n| > 0) & (|m| > 0) of |m||n| - <n, m>,  n = rendered normal, m = depth normal,
//   consistency     norms detached, optionally n or m detached in the dot product                        (:1273-1295)
// Here it is ONE streaming pass forward (all planes read once: 60 B / pixel; fp64 per-thread sums, fixed-order
// block partials, fixed-order final sum: bitwise reproducible, no atomics) and ONE pass backward that re-reads the
// planes and writes every gradient plane (up to 44 B / pixel).  HBM bound.
#include "common.hpp"

namespace {

constexpr int BLOCK = 256;
constexpr int MAX_BLOCKS = 2048;
constexpr int NSUM = 8;  // l1 sum, l1 count, depth sum, depth count, consistency sum, count, sky sum, count

struct Planes {
  const float* rgb;       // [3,H,W]
  const float* gt_rgb;    // [3,H,W]
  const float* depth;     // [H,W] | null
  const float* gt_depth;  // [H,W] | null
  const float* alpha;     // [H,W] | null
  const float* normal;    // [3,H,W] | null
  const float* dnormal;   // [3,H,W] | null
  const uint8_t* sky;     // [H,W] | null
};

struct Pixel {
  bool in_rows, depth_ok, is_sky;
  float c[3], g[3], d, gd, n[3], m[3], a;
};

__device__ inline void load_pixel(const pings_image_loss_params& p, const Planes& t, int64_t i, int64_t HW, Pixel& x) {
  x.in_rows = i >= (int64_t)p.v_min * p.W && i < (int64_t)p.v_max * p.W;
  if (x.in_rows) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { x.c[k] = t.rgb[k * HW + i]; x.g[k] = t.gt_rgb[k * HW + i]; }
  }
  x.is_sky = t.sky != nullptr && t.sky[i] != 0;
  x.a = t.alpha ? t.alpha[i] : 0.f;
  x.depth_ok = false;
  if (t.depth && t.gt_depth) {
    x.d = t.depth[i]; x.gd = t.gt_depth[i];
    x.depth_ok = x.gd > p.depth_min && x.gd < p.depth_max && (t.alpha == nullptr || x.a > p.min_accu_alpha);
  }
  if (t.normal && t.dnormal) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { x.n[k] = t.normal[k * HW + i]; x.m[k] = t.dnormal[k * HW + i]; }
    if (x.is_sky) {  // both maps are multiplied by the non-sky mask first (:1213-1216): norms become 0 -> not valid
#pragma unroll
      for (int k = 0; k < 3; ++k) x.n[k] = x.m[k] = 0.f;
    }
  }
}

__device__ inline float norm3(const float* v) { return sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

__global__ __launch_bounds__(BLOCK) void loss_fwd_kernel(pings_image_loss_params p, Planes t,
                                                          double* __restrict__ partials) {
  const int64_t HW = (int64_t)p.H * p.W;
  double s[NSUM] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < HW; i += (int64_t)gridDim.x * BLOCK) {
    Pixel x;
    load_pixel(p, t, i, HW, x);
    if (x.in_rows) {
      s[0] += (double)(fabsf(x.c[0] - x.g[0])) + (double)(fabsf(x.c[1] - x.g[1])) + (double)(fabsf(x.c[2] - x.g[2]));
      s[1] += 3.0;
    }
    if (x.depth_ok) {
      s[2] += p.inverse_depth ? (double)fabsf(1.0f / x.gd - 1.0f / x.d) : (double)fabsf(x.gd - x.d);
      s[3] += 1.0;
    }
    if (t.normal && t.dnormal) {
      const float nn = norm3(x.n), mn = norm3(x.m);
      if (nn > 0.f && mn > 0.f) {
        const float dot = x.n[0] * x.m[0] + x.n[1] * x.m[1] + x.n[2] * x.m[2];
        s[4] += (double)(mn * nn - dot);
        s[5] += 1.0;
      }
    }
    if (x.is_sky && t.alpha) { s[6] += (double)x.a; s[7] += 1.0; }
  }
  __shared__ double sh[BLOCK / 64][NSUM];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NSUM; ++k) {
    const double v = wave_sum(s[k]);
    if (lane == 0) sh[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < NSUM) {
    double v = 0;
    for (int w = 0; w < BLOCK / 64; ++w) v += sh[w][threadIdx.x];
    partials[(size_t)blockIdx.x * NSUM + threadIdx.x] = v;
  }
}

// One workgroup: lane k of each group of NSUM sums every (BLOCK/NSUM)-th partial, then the groups are added in order.
__global__ __launch_bounds__(BLOCK) void loss_final_kernel(const double* __restrict__ partials, int nblocks,
                                                            double* __restrict__ sums, float* __restrict__ losses) {
  __shared__ double sh[BLOCK];
  const int k = threadIdx.x % NSUM, grp = threadIdx.x / NSUM;
  double v = 0;
  for (int b = grp; b < nblocks; b += BLOCK / NSUM) v += partials[(size_t)b * NSUM + k];
  sh[threadIdx.x] = v;
  __syncthreads();
  if (threadIdx.x < NSUM) {
    double tot = 0;
    for (int g = 0; g < BLOCK / NSUM; ++g) tot += sh[g * NSUM + threadIdx.x];
    sums[threadIdx.x] = tot;
    sh[threadIdx.x] = tot;
  }
  __syncthreads();
  if (threadIdx.x < 4)  // mean of an empty selection is NaN, as torch's .mean() of an empty tensor
    losses[threadIdx.x] = (float)(sh[2 * threadIdx.x] / sh[2 * threadIdx.x + 1]);
}

struct Grads {
  float* rgb;      // [3,H,W] | null
  float* depth;    // [H,W] | null
  float* alpha;    // [H,W] | null
  float* normal;   // [3,H,W] | null
  float* dnormal;  // [3,H,W] | null
};

__device__ inline float sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

__global__ __launch_bounds__(BLOCK) void loss_bwd_kernel(pings_image_loss_params p, Planes t,
                                                          const double* __restrict__ sums,
                                                          const float* __restrict__ g_losses, Grads o) {
  const int64_t HW = (int64_t)p.H * p.W;
  // upstream gradient of each mean, already divided by its count (a zero count never reaches a valid pixel)
  const float k_l1 = (float)((double)g_losses[0] / sums[1]);
  const float k_d = (float)((double)g_losses[1] / sums[3]);
  const float k_c = (float)((double)g_losses[2] / sums[5]);
  const float k_s = (float)((double)g_losses[3] / sums[7]);
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < HW; i += (int64_t)gridDim.x * BLOCK) {
    Pixel x;
    load_pixel(p, t, i, HW, x);
    if (o.rgb) {
#pragma unroll
      for (int k = 0; k < 3; ++k) o.rgb[k * HW + i] = x.in_rows ? k_l1 * sgn(x.c[k] - x.g[k]) : 0.f;
    }
    if (o.depth) {
      float g = 0.f;
      if (x.depth_ok)
        g = p.inverse_depth ? k_d * sgn(1.0f / x.gd - 1.0f / x.d) / (x.d * x.d) : -k_d * sgn(x.gd - x.d);
      o.depth[i] = g;
    }
    if (o.alpha) o.alpha[i] = (x.is_sky && t.alpha) ? k_s : 0.f;
    if (o.normal || o.dnormal) {
      bool ok = false;
      if (t.normal && t.dnormal) ok = norm3(x.n) > 0.f && norm3(x.m) > 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (o.normal) o.normal[k * HW + i] = (ok && p.consist_mode != 1) ? -k_c * x.m[k] : 0.f;
        if (o.dnormal) o.dnormal[k * HW + i] = (ok && p.consist_mode != 2) ? -k_c * x.n[k] : 0.f;
      }
    }
  }
}

int check(const pings_image_loss_params* p, const float* rgb, const float* gt_rgb) {
  PINGS_ARG_CHECK(p != nullptr, "null params");
  PINGS_ARG_CHECK(p->H > 0 && p->W > 0, "empty image");
  PINGS_ARG_CHECK(p->v_min >= 0 && p->v_max <= p->H, "row window outside the image");
  PINGS_ARG_CHECK(p->consist_mode >= 0 && p->consist_mode <= 2, "consist_mode is 0 (both), 1 (normal fixed) or 2 (depth fixed)");
  PINGS_ARG_CHECK(rgb && gt_rgb, "null colour image");
  return PINGS_OK;
}

int grid_for(int H, int W) {
  const int64_t HW = (int64_t)H * W;
  return (int)(pings::ceil_div<int64_t>(HW, BLOCK) < MAX_BLOCKS ? pings::ceil_div<int64_t>(HW, BLOCK) : MAX_BLOCKS);
}

}  // namespace

PINGS_API size_t pings_image_losses_scratch_bytes(void) { return sizeof(double) * NSUM * MAX_BLOCKS; }

PINGS_API int pings_image_losses_forward(const pings_image_loss_params* p, const float* rgb, const float* gt_rgb,
                                         const float* depth, const float* gt_depth, const float* alpha,
                                         const float* normal, const float* depth_normal, const uint8_t* sky_mask,
                                         void* scratch, double* sums, float* losses, void* stream) {
  if (int rc = check(p, rgb, gt_rgb)) return rc;
  PINGS_ARG_CHECK(scratch && sums && losses, "null output");
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope sc("image_losses_fwd", st);
  const Planes t{rgb, gt_rgb, depth, gt_depth, alpha, normal, depth_normal, sky_mask};
  const int grid = grid_for(p->H, p->W);
  loss_fwd_kernel<<<grid, BLOCK, 0, st>>>(*p, t, reinterpret_cast<double*>(scratch));
  PINGS_LAUNCH_CHECK();
  loss_final_kernel<<<1, BLOCK, 0, st>>>(reinterpret_cast<const double*>(scratch), grid, sums, losses);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}

PINGS_API int pings_image_losses_backward(const pings_image_loss_params* p, const float* rgb, const float* gt_rgb,
                                          const float* depth, const float* gt_depth, const float* alpha,
                                          const float* normal, const float* depth_normal, const uint8_t* sky_mask,
                                          const double* sums, const float* dL_dlosses, float* dL_drgb,
                                          float* dL_ddepth, float* dL_dalpha, float* dL_dnormal,
                                          float* dL_ddepth_normal, void* stream) {
  if (int rc = check(p, rgb, gt_rgb)) return rc;
  PINGS_ARG_CHECK(sums && dL_dlosses, "null sums / upstream gradient");
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope sc("image_losses_bwd", st);
  const Planes t{rgb, gt_rgb, depth, gt_depth, alpha, normal, depth_normal, sky_mask};
  const Grads o{dL_drgb, dL_ddepth, dL_dalpha, dL_dnormal, dL_ddepth_normal};
  loss_bwd_kernel<<<grid_for(p->H, p->W), BLOCK, 0, st>>>(*p, t, sums, dL_dlosses, o);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}
