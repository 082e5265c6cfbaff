// spawn_gaussians on the device: everything around the five decoder MLPs, for gfx950.
//
// Restates gaussian_splatting/gaussian_renderer/__init__.py:469-778 of the reference as four kernels
// (the MLPs themselves are csrc/mlp.hip):
//   gather_kernel    rows `sel` of the map tensors -> dense per-view inputs: position, orientation,
//                    base colour, geo feature (+ horizontal view distance when dist_concat_on),
//                    colour feature (+ view direction in the neural point's frame when view_concat_on)
//                    (:551-597, :672-675, :692-699).  One wave per 2 rows of up to 64+64 floats: the feature
//                    rows are read and written as whole coalesced rows.
//   plan_kernel      keep flag per Gaussian (alpha > 0, optional scale filter, :727-761), then an
//                    exclusive scan (rocPRIM) gives every kept Gaussian its compacted row.
//   forward_kernel   one lane per Gaussian: activations + quaternion algebra (:605-716) written straight
//                    to the compacted rows; `alpha_all` and the (tiled, :724) free mask ride along.
//   backward_kernel  one lane per Gaussian: gradients of the five output tensors (+ alpha_all) back to
//                    the raw MLP outputs; dropped Gaussians only see the alpha_all term.
// All four are HBM-stream bound: per Gaussian 14 floats in, 15 floats out (+ scan traffic), no reuse.
//
// Raw MLP outputs are [n, d*k] row-major, which the reference reinterprets as [n*k, d] (:632,645,665,687,716):
// Gaussian g = i*k + j of neural point i owns columns j*d .. j*d+d-1, i.e. floats g*d .. g*d+d-1.
#include <hipcub/hipcub.hpp>

#include "common.hpp"

namespace {

constexpr float kNormEps = 1e-12f;  // torch.nn.functional.normalize default eps (:646)
constexpr float kThin = 1e-7f;      // surfel thin-dimension scale (:669)
constexpr float kResidual = 0.1f;   // colour residual range (:707)

struct Q4 {
  float w, x, y, z;
};

// R(q)^T v  — the reference's apply_quaternion_rotation (utils/tools.py:743-751): u = -q.xyz,
// t = 2 u x v, v + w t + u x t.  `sgn` = -1 gives that; `sgn` = +1 gives R(q) v (its transpose, which is
// both the view-direction rotation with quat_inverse (:695-696) and the adjoint used in backward).
__device__ inline void rotate(const Q4 q, float sgn, float vx, float vy, float vz, float& ox, float& oy,
                              float& oz) {
  const float ux = sgn * q.x, uy = sgn * q.y, uz = sgn * q.z;
  const float tx = 2.f * (uy * vz - uz * vy);
  const float ty = 2.f * (uz * vx - ux * vz);
  const float tz = 2.f * (ux * vy - uy * vx);
  ox = vx + q.w * tx + (uy * tz - uz * ty);
  oy = vy + q.w * ty + (uz * tx - ux * tz);
  oz = vz + q.w * tz + (ux * ty - uy * tx);
}

__device__ inline float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// ------------------------------------------------------------------ gather
struct GatherArgs {
  int n, Fg, Fc, ldg, ldc;
  int xy_only, view_concat, dist_concat;
  const int64_t* sel;
  const float *position, *orientation, *color, *geo_feature, *color_feature, *cam;
  const uint8_t* free_mask;
  float *pos, *quat, *base, *geo_in, *col_in, *view_dist;
  uint8_t* free_out;
  const int32_t* n_dev;   // rows actually selected (device word; `n` is then the capacity the grid was sized for)
};

// One 16-lane group per row (four rows per wave, no grid-stride loop: a wave that walks its rows one after the other
// pays the sel -> row -> store latency chain once per row).
__global__ __launch_bounds__(256) void gather_kernel(GatherArgs a) {
  const int lane = threadIdx.x & 15;
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int n_rows = a.n_dev ? min(a.n, *a.n_dev) : a.n;
  if (i >= n_rows) return;   // whole groups leave together: the width-16 shuffles below stay inside live groups
  const int64_t src = a.sel ? a.sel[i] : (int64_t)i;
  // small per-point attributes: lanes 0..2 position, 4..7 quaternion, 8..10 colour
  float pv = 0.f;
  if (lane < 3) {
    pv = a.position[src * 3 + lane];
    a.pos[(size_t)i * 3 + lane] = pv;
  } else if (lane >= 4 && lane < 8) {
    pv = a.orientation[src * 4 + (lane - 4)];
    a.quat[(size_t)i * 4 + (lane - 4)] = pv;
  } else if (lane >= 8 && lane < 11 && a.color) {
    a.base[(size_t)i * 3 + (lane - 8)] = a.color[src * 3 + (lane - 8)];
  } else if (lane == 11 && a.free_mask) {
    a.free_out[i] = a.free_mask[src];
  }
  for (int c = lane; c < a.Fg; c += 16) a.geo_in[(size_t)i * a.ldg + c] = a.geo_feature[src * a.Fg + c];
  for (int c = lane; c < a.Fc; c += 16) a.col_in[(size_t)i * a.ldc + c] = a.color_feature[src * a.Fc + c];
  if (a.cam) {
    const float px = __shfl(pv, 0, 16), py = __shfl(pv, 1, 16), pz = __shfl(pv, 2, 16);
    Q4 q;
    q.w = __shfl(pv, 4, 16); q.x = __shfl(pv, 5, 16); q.y = __shfl(pv, 6, 16); q.z = __shfl(pv, 7, 16);
    float vx = px - a.cam[0], vy = py - a.cam[1], vz = pz - a.cam[2];
    if (a.xy_only) vz = 0.f;                                  // before the norm (:592-597)
    const float dist = sqrtf((vx * vx + vy * vy) + vz * vz);
    vx /= dist; vy /= dist; vz /= dist;
    if (lane == 0) {
      if (a.view_dist) a.view_dist[i] = dist;
      if (a.dist_concat) a.geo_in[(size_t)i * a.ldg + a.Fg] = dist;
      if (a.view_concat) {
        float ox, oy, oz;
        rotate(q, +1.f, vx, vy, vz, ox, oy, oz);             // apply_quaternion_rotation(quat_inverse(q), v)
        float* d = a.col_in + (size_t)i * a.ldc + a.Fc;
        d[0] = ox; d[1] = oy; d[2] = oz;
      }
    }
  }
}

// Rows `sel` of the (pre-zeroed) map-sized gradient tensors receive the per-view feature gradients; `sel`
// holds distinct rows (it is the nonzero() of a mask), so this is a plain scatter.  16 lanes per row.
__global__ __launch_bounds__(256) void gather_bwd_kernel(int n, const int64_t* __restrict__ sel,
                                                         const float* __restrict__ d_in, int F, int ld,
                                                         float* __restrict__ d_feature) {
  const int lane = threadIdx.x & 15;
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  if (i >= n) return;
  const int64_t dst = sel ? sel[i] : (int64_t)i;
  for (int c = lane; c < F; c += 16) d_feature[dst * F + c] = d_in[(size_t)i * ld + c];
}

// ------------------------------------------------------------------ per-Gaussian activations
struct SpawnArgs {
  pings_spawn_params p;
  const float *xyz_raw, *rot_raw, *scale_raw, *alpha_raw, *color_raw;
  const float *pos, *quat, *base, *dist_ratio;
  const uint8_t* free_in;
  const int32_t* dest;  // compacted row per Gaussian, -1 = dropped; nullptr = identity
  const int32_t* n_dev; // neural points actually selected (device word); p.n is then only the capacity
  int32_t* nan_flag;    // set to 1 when a spawned rotation is NaN (nullable)
};

__device__ inline int rows_of(const SpawnArgs& a) { return a.n_dev ? min(a.p.n, *a.n_dev) : a.p.n; }

__device__ inline void scales_of(const SpawnArgs& a, int64_t g, int i, float s[3], float e[3]) {
  const int sd = a.p.scale_dim;
  const float dr = a.dist_ratio ? a.dist_ratio[i] : 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (c < sd) {
      e[c] = a.p.unit_scale * expf(a.scale_raw[g * sd + c] + dr);  // unit * res * exp(mlp + dist/z_far) (:661)
      s[c] = fminf(e[c], a.p.max_scale);                            // clamp(max=) (:662)
    } else {
      e[c] = 0.f;
      s[c] = 0.f;
    }
  }
  if (a.p.surfel) s[2] = kThin;
}

__global__ __launch_bounds__(256) void plan_kernel(SpawnArgs a, int64_t nk, int32_t* __restrict__ flag) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nk) return;
  if (g >= (int64_t)rows_of(a) * a.p.k) {   // capacity rows behind the selected ones: never kept, never read
    flag[g] = 0;
    return;
  }
  bool keep = true;
  if (a.p.alpha_filter_on) keep = tanhf(a.alpha_raw[g]) > 0.f;
  if (keep && a.p.scale_filter_on) {
    float s[3], e[3];
    scales_of(a, g, (int)(g / a.p.k), s, e);
    const int dims = a.p.surfel ? 3 : a.p.scale_dim;
    bool any = false;
    for (int c = 0; c < dims; ++c) any = any || (s[c] > a.p.scale_filter_thr);
    keep = any;
  }
  flag[g] = keep ? 1 : 0;
}

__global__ void plan_finish_kernel(int64_t nk, const int32_t* __restrict__ flag, int32_t* __restrict__ dest,
                                   int32_t* __restrict__ count, int32_t* __restrict__ nan_flag) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nk) return;
  const int32_t f = flag[g], d = dest[g];  // dest holds the exclusive scan of flag
  if (g == nk - 1) {
    count[0] = d + f;
    if (nan_flag) *nan_flag = 0;
  }
  dest[g] = f ? d : -1;
}

__global__ __launch_bounds__(256) void forward_kernel(SpawnArgs a, int64_t nk, float* __restrict__ o_xyz,
                                                      float* __restrict__ o_scale, float* __restrict__ o_rot,
                                                      float* __restrict__ o_alpha, float* __restrict__ o_color,
                                                      float* __restrict__ o_alpha_all,
                                                      uint8_t* __restrict__ o_free) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nk) return;
  const int n_rows = rows_of(a);
  if (g >= (int64_t)n_rows * a.p.k) return;
  const int i = (int)(g / a.p.k);
  const float alpha = tanhf(a.alpha_raw[g]);  // (:685)
  o_alpha_all[g] = alpha;                      // pre-filter clone (:721)
  const int64_t d = a.dest ? (int64_t)a.dest[g] : g;
  if (d < 0) return;
  Q4 q;
  q.w = a.quat[(size_t)i * 4 + 0]; q.x = a.quat[(size_t)i * 4 + 1];
  q.y = a.quat[(size_t)i * 4 + 2]; q.z = a.quat[(size_t)i * 4 + 3];

  // position: p + R(q)^T (range * tanh(mlp)) (:609,634-639)
  {
    const float dx = a.p.displacement_range * tanhf(a.xyz_raw[g * 3 + 0]);
    const float dy = a.p.displacement_range * tanhf(a.xyz_raw[g * 3 + 1]);
    const float dz = a.p.displacement_range * tanhf(a.xyz_raw[g * 3 + 2]);
    float ox, oy, oz;
    rotate(q, -1.f, dx, dy, dz, ox, oy, oz);
    o_xyz[d * 3 + 0] = a.pos[(size_t)i * 3 + 0] + ox;
    o_xyz[d * 3 + 1] = a.pos[(size_t)i * 3 + 1] + oy;
    o_xyz[d * 3 + 2] = a.pos[(size_t)i * 3 + 2] + oz;
  }
  // rotation: q (x) nan_to_num(normalize(mlp)) (:644-649), Hamilton product (utils/tools.py:803-823)
  {
    float r0 = a.rot_raw[g * 4 + 0], r1 = a.rot_raw[g * 4 + 1], r2 = a.rot_raw[g * 4 + 2],
          r3 = a.rot_raw[g * 4 + 3];
    const float nrm = fmaxf(sqrtf(((r0 * r0 + r1 * r1) + r2 * r2) + r3 * r3), kNormEps);
    r0 /= nrm; r1 /= nrm; r2 /= nrm; r3 /= nrm;
    if (r0 != r0) r0 = 0.f;
    if (r1 != r1) r1 = 0.f;
    if (r2 != r2) r2 = 0.f;
    if (r3 != r3) r3 = 0.f;
    const float o0 = q.w * r0 - q.x * r1 - q.y * r2 - q.z * r3, o1 = q.w * r1 + q.x * r0 + q.y * r3 - q.z * r2;
    const float o2 = q.w * r2 - q.x * r3 + q.y * r0 + q.z * r1, o3 = q.w * r3 + q.x * r2 - q.y * r1 + q.z * r0;
    o_rot[d * 4 + 0] = o0;
    o_rot[d * 4 + 1] = o1;
    o_rot[d * 4 + 2] = o2;
    o_rot[d * 4 + 3] = o3;
    if (a.nan_flag && (o0 != o0 || o1 != o1 || o2 != o2 || o3 != o3)) *a.nan_flag = 1;   // (:305-306) checked by render
  }
  // scale (:655-670)
  {
    float s[3], e[3];
    scales_of(a, g, i, s, e);
    const int od = a.p.surfel ? 3 : a.p.scale_dim;
    for (int c = 0; c < od; ++c) o_scale[d * od + c] = s[c];
  }
  o_alpha[d] = alpha;
  // colour (:706-716)
  {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float raw = a.color_raw[g * 3 + c];
      float v;
      if (a.p.color_residual) {
        v = a.base[(size_t)i * 3 + c] + kResidual * tanhf(raw);
        v = fminf(fmaxf(v, 0.f), 1.f);
      } else {
        v = sigmoidf(raw);
      }
      o_color[d * 3 + c] = v;
    }
  }
  // the reference tiles the 1-D per-point mask ([n].repeat(1,k).view(-1), :724): Gaussian g gets free[g % n]
  if (o_free && a.free_in) o_free[d] = a.free_in[g % n_rows];
}

__global__ __launch_bounds__(256) void backward_kernel(SpawnArgs a, int64_t nk, const float* __restrict__ g_xyz,
                                                       const float* __restrict__ g_scale,
                                                       const float* __restrict__ g_rot,
                                                       const float* __restrict__ g_alpha,
                                                       const float* __restrict__ g_color,
                                                       const float* __restrict__ g_alpha_all,
                                                       float* __restrict__ d_xyz_raw, float* __restrict__ d_rot_raw,
                                                       float* __restrict__ d_scale_raw,
                                                       float* __restrict__ d_alpha_raw,
                                                       float* __restrict__ d_color_raw) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nk) return;
  const int i = (int)(g / a.p.k);
  const int sd = a.p.scale_dim;
  const int64_t d = a.dest ? (int64_t)a.dest[g] : g;
  const float alpha = tanhf(a.alpha_raw[g]);
  float ga = g_alpha_all ? g_alpha_all[g] : 0.f;
  if (d >= 0 && g_alpha) ga += g_alpha[d];
  d_alpha_raw[g] = ga * (1.f - alpha * alpha);
  if (d < 0) {  // dropped by the filters: no path from the outputs
    d_xyz_raw[g * 3 + 0] = d_xyz_raw[g * 3 + 1] = d_xyz_raw[g * 3 + 2] = 0.f;
    d_rot_raw[g * 4 + 0] = d_rot_raw[g * 4 + 1] = d_rot_raw[g * 4 + 2] = d_rot_raw[g * 4 + 3] = 0.f;
    for (int c = 0; c < sd; ++c) d_scale_raw[g * sd + c] = 0.f;
    d_color_raw[g * 3 + 0] = d_color_raw[g * 3 + 1] = d_color_raw[g * 3 + 2] = 0.f;
    return;
  }
  Q4 q;
  q.w = a.quat[(size_t)i * 4 + 0]; q.x = a.quat[(size_t)i * 4 + 1];
  q.y = a.quat[(size_t)i * 4 + 2]; q.z = a.quat[(size_t)i * 4 + 3];
  // position: out = p + R^T (range tanh(raw))  =>  d raw = range (1 - t^2) (R g)
  {
    float bx = 0.f, by = 0.f, bz = 0.f;
    if (g_xyz) rotate(q, +1.f, g_xyz[d * 3 + 0], g_xyz[d * 3 + 1], g_xyz[d * 3 + 2], bx, by, bz);
    const float t0 = tanhf(a.xyz_raw[g * 3 + 0]), t1 = tanhf(a.xyz_raw[g * 3 + 1]),
                t2 = tanhf(a.xyz_raw[g * 3 + 2]);
    d_xyz_raw[g * 3 + 0] = a.p.displacement_range * (1.f - t0 * t0) * bx;
    d_xyz_raw[g * 3 + 1] = a.p.displacement_range * (1.f - t1 * t1) * by;
    d_xyz_raw[g * 3 + 2] = a.p.displacement_range * (1.f - t2 * t2) * bz;
  }
  // rotation: out = L(q) r, r = raw / max(|raw|, eps)  =>  d r = L(q)^T g = conj(q) (x) g
  {
    float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
    if (g_rot) { o0 = g_rot[d * 4 + 0]; o1 = g_rot[d * 4 + 1]; o2 = g_rot[d * 4 + 2]; o3 = g_rot[d * 4 + 3]; }
    const float e0 = q.w * o0 + q.x * o1 + q.y * o2 + q.z * o3;
    const float e1 = q.w * o1 - q.x * o0 - q.y * o3 + q.z * o2;
    const float e2 = q.w * o2 + q.x * o3 - q.y * o0 - q.z * o1;
    const float e3 = q.w * o3 - q.x * o2 + q.y * o1 - q.z * o0;
    const float r0 = a.rot_raw[g * 4 + 0], r1 = a.rot_raw[g * 4 + 1], r2 = a.rot_raw[g * 4 + 2],
                r3 = a.rot_raw[g * 4 + 3];
    const float nr = sqrtf(((r0 * r0 + r1 * r1) + r2 * r2) + r3 * r3);
    if (nr > kNormEps) {
      const float inv = 1.f / nr;
      const float u0 = r0 * inv, u1 = r1 * inv, u2 = r2 * inv, u3 = r3 * inv;
      const float dot = ((u0 * e0 + u1 * e1) + u2 * e2) + u3 * e3;
      d_rot_raw[g * 4 + 0] = (e0 - u0 * dot) * inv;
      d_rot_raw[g * 4 + 1] = (e1 - u1 * dot) * inv;
      d_rot_raw[g * 4 + 2] = (e2 - u2 * dot) * inv;
      d_rot_raw[g * 4 + 3] = (e3 - u3 * dot) * inv;
    } else {  // clamp_min(eps) active: r = raw / eps
      d_rot_raw[g * 4 + 0] = e0 / kNormEps;
      d_rot_raw[g * 4 + 1] = e1 / kNormEps;
      d_rot_raw[g * 4 + 2] = e2 / kNormEps;
      d_rot_raw[g * 4 + 3] = e3 / kNormEps;
    }
  }
  // scale: out = min(e, max), e = unit exp(raw + dr)  =>  d raw = g e where e <= max (torch clamp mask)
  {
    float s[3], e[3];
    scales_of(a, g, i, s, e);
    const int od = a.p.surfel ? 3 : sd;
    for (int c = 0; c < sd; ++c) {
      const bool live = (!a.p.surfel || c < 2) && c < od && g_scale != nullptr;
      const float go = live ? g_scale[d * od + c] : 0.f;
      d_scale_raw[g * sd + c] = (e[c] <= a.p.max_scale) ? go * e[c] : 0.f;
    }
  }
  // colour
  {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float raw = a.color_raw[g * 3 + c];
      const float go = g_color ? g_color[d * 3 + c] : 0.f;
      float dr;
      if (a.p.color_residual) {
        const float t = tanhf(raw);
        const float pre = a.base[(size_t)i * 3 + c] + kResidual * t;
        dr = (pre >= 0.f && pre <= 1.f) ? go * kResidual * (1.f - t * t) : 0.f;
      } else {
        const float s = sigmoidf(raw);
        dr = go * s * (1.f - s);
      }
      d_color_raw[g * 3 + c] = dr;
    }
  }
}

int check_params(const pings_spawn_params* p) {
  PINGS_ARG_CHECK(p != nullptr, "null params");
  PINGS_ARG_CHECK(p->n >= 0 && p->k > 0, "bad n / k");
  PINGS_ARG_CHECK((int64_t)p->n * p->k < (int64_t)1 << 31, "n*k must fit 31 bits");
  PINGS_ARG_CHECK(p->scale_dim == 2 || p->scale_dim == 3, "scale_dim must be 2 or 3");
  PINGS_ARG_CHECK(p->surfel || p->scale_dim == 3, "3d_gs needs three scale columns per Gaussian");
  return PINGS_OK;
}

size_t scan_temp_bytes(int64_t nk) {
  size_t b = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b, (int32_t*)nullptr, (int32_t*)nullptr, (int)nk);
  return (b + 255) & ~(size_t)255;
}

int grid_rows16(int n) {  // one 16-lane group per row, 16 rows per 256-thread workgroup
  const int blocks = (int)pings::ceil_div<int64_t>((int64_t)n, 16);
  return blocks < 1 ? 1 : blocks;
}

}  // namespace

PINGS_API int pings_spawn_gather(int n, const int64_t* sel, const float* position, const float* orientation,
                                 const float* color, const uint8_t* free_mask, const float* geo_feature, int Fg,
                                 const float* color_feature, int Fc, const float* cam_origin, int xy_only,
                                 int view_concat, int dist_concat, float* pos, float* quat, float* base_color,
                                 uint8_t* free_out, float* geo_in, float* col_in, float* view_dist,
                                 void* stream) {
  return pings_spawn_gather_dyn(n, nullptr, sel, position, orientation, color, free_mask, geo_feature, Fg, color_feature,
                                Fc, cam_origin, xy_only, view_concat, dist_concat, pos, quat, base_color, free_out,
                                geo_in, col_in, view_dist, stream);
}

PINGS_API int pings_spawn_gather_dyn(int n, const int32_t* n_rows_dev, const int64_t* sel, const float* position,
                                     const float* orientation, const float* color, const uint8_t* free_mask,
                                     const float* geo_feature, int Fg, const float* color_feature, int Fc,
                                     const float* cam_origin, int xy_only, int view_concat, int dist_concat,
                                     float* pos, float* quat, float* base_color, uint8_t* free_out, float* geo_in,
                                     float* col_in, float* view_dist, void* stream) {
  PINGS_ARG_CHECK(n >= 0 && Fg > 0 && Fc > 0, "bad sizes");
  if (n == 0) return PINGS_OK;
  PINGS_ARG_CHECK(position && orientation && geo_feature && color_feature, "null map tensor");
  PINGS_ARG_CHECK(pos && quat && geo_in && col_in, "null output");
  PINGS_ARG_CHECK(!color || base_color, "colour given without an output for it");
  PINGS_ARG_CHECK(!free_mask || free_out, "free mask given without an output for it");
  PINGS_ARG_CHECK(cam_origin || (!view_concat && !dist_concat), "view features need cam_origin");
  GatherArgs a;
  a.n = n; a.Fg = Fg; a.Fc = Fc;
  a.ldg = Fg + (dist_concat ? 1 : 0);
  a.ldc = Fc + (view_concat ? 3 : 0);
  a.xy_only = xy_only; a.view_concat = view_concat; a.dist_concat = dist_concat;
  a.sel = sel; a.position = position; a.orientation = orientation; a.color = color;
  a.geo_feature = geo_feature; a.color_feature = color_feature; a.cam = cam_origin; a.free_mask = free_mask;
  a.pos = pos; a.quat = quat; a.base = base_color; a.geo_in = geo_in; a.col_in = col_in;
  a.view_dist = view_dist; a.free_out = free_out; a.n_dev = n_rows_dev;
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope sc("spawn_gather", st);
  gather_kernel<<<grid_rows16(n), 256, 0, st>>>(a);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}

PINGS_API int pings_spawn_gather_backward(int n, const int64_t* sel, const float* dL_dgeo_in, int Fg, int ldg,
                                          const float* dL_dcol_in, int Fc, int ldc, float* dL_dgeo_feature,
                                          float* dL_dcolor_feature, void* stream) {
  PINGS_ARG_CHECK(n >= 0, "bad n");
  if (n == 0) return PINGS_OK;
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope sc("spawn_gather_bwd", st);
  if (dL_dgeo_in) {
    PINGS_ARG_CHECK(dL_dgeo_feature && Fg > 0 && ldg >= Fg, "bad geo gradient arguments");
    gather_bwd_kernel<<<grid_rows16(n), 256, 0, st>>>(n, sel, dL_dgeo_in, Fg, ldg, dL_dgeo_feature);
    PINGS_LAUNCH_CHECK();
  }
  if (dL_dcol_in) {
    PINGS_ARG_CHECK(dL_dcolor_feature && Fc > 0 && ldc >= Fc, "bad colour gradient arguments");
    gather_bwd_kernel<<<grid_rows16(n), 256, 0, st>>>(n, sel, dL_dcol_in, Fc, ldc, dL_dcolor_feature);
    PINGS_LAUNCH_CHECK();
  }
  return PINGS_OK;
}

PINGS_API size_t pings_spawn_plan_scratch_bytes(int64_t num_gaussians) {
  const int64_t nk = num_gaussians > 0 ? num_gaussians : 1;
  return (size_t)nk * sizeof(int32_t) + 256 + scan_temp_bytes(nk);
}

PINGS_API int pings_spawn_plan(const pings_spawn_params* p, const float* alpha_raw, const float* scale_raw,
                               const float* dist_ratio, void* scratch, int32_t* dest, int32_t* count,
                               void* stream) {
  return pings_spawn_plan_dyn(p, nullptr, alpha_raw, scale_raw, dist_ratio, scratch, dest, count, nullptr, stream);
}

PINGS_API int pings_spawn_plan_dyn(const pings_spawn_params* p, const int32_t* n_rows_dev, const float* alpha_raw,
                                   const float* scale_raw, const float* dist_ratio, void* scratch, int32_t* dest,
                                   int32_t* count, int32_t* nan_flag, void* stream) {
  if (int rc = check_params(p)) return rc;
  PINGS_ARG_CHECK(alpha_raw && dest && count && scratch, "null pointer");
  PINGS_ARG_CHECK(!p->scale_filter_on || scale_raw, "scale filter needs the scale MLP output");
  hipStream_t st = pings::as_stream(stream);
  const int64_t nk = (int64_t)p->n * p->k;
  if (nk == 0) {
    PINGS_HIP_CHECK(hipMemsetAsync(count, 0, sizeof(int32_t), st));
    if (nan_flag) PINGS_HIP_CHECK(hipMemsetAsync(nan_flag, 0, sizeof(int32_t), st));
    return PINGS_OK;
  }
  pings::prof::Scope sc("spawn_plan", st);
  SpawnArgs a{};
  a.p = *p; a.alpha_raw = alpha_raw; a.scale_raw = scale_raw; a.dist_ratio = dist_ratio; a.n_dev = n_rows_dev;
  int32_t* flag = reinterpret_cast<int32_t*>(scratch);
  void* temp = reinterpret_cast<char*>(scratch) + (((size_t)nk * sizeof(int32_t) + 255) & ~(size_t)255);
  size_t tb = scan_temp_bytes(nk);
  const int blocks = (int)pings::ceil_div<int64_t>(nk, 256);
  plan_kernel<<<blocks, 256, 0, st>>>(a, nk, flag);
  PINGS_LAUNCH_CHECK();
  PINGS_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(temp, tb, flag, dest, (int)nk, st));
  plan_finish_kernel<<<blocks, 256, 0, st>>>(nk, flag, dest, count, nan_flag);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}

PINGS_API int pings_spawn_forward(const pings_spawn_params* p, const float* xyz_raw, const float* rot_raw,
                                  const float* scale_raw, const float* alpha_raw, const float* color_raw,
                                  const float* pos, const float* quat, const float* base_color,
                                  const float* dist_ratio, const uint8_t* free_in, const int32_t* dest,
                                  float* gaussian_xyz, float* gaussian_scale, float* gaussian_rot,
                                  float* gaussian_alpha, float* gaussian_color, float* alpha_all,
                                  uint8_t* gaussian_free_mask, void* stream) {
  return pings_spawn_forward_dyn(p, nullptr, xyz_raw, rot_raw, scale_raw, alpha_raw, color_raw, pos, quat, base_color,
                                 dist_ratio, free_in, dest, gaussian_xyz, gaussian_scale, gaussian_rot, gaussian_alpha,
                                 gaussian_color, alpha_all, gaussian_free_mask, nullptr, stream);
}

PINGS_API int pings_spawn_forward_dyn(const pings_spawn_params* p, const int32_t* n_rows_dev, const float* xyz_raw,
                                      const float* rot_raw, const float* scale_raw, const float* alpha_raw,
                                      const float* color_raw, const float* pos, const float* quat,
                                      const float* base_color, const float* dist_ratio, const uint8_t* free_in,
                                      const int32_t* dest, float* gaussian_xyz, float* gaussian_scale,
                                      float* gaussian_rot, float* gaussian_alpha, float* gaussian_color,
                                      float* alpha_all, uint8_t* gaussian_free_mask, int32_t* nan_flag,
                                      void* stream) {
  if (int rc = check_params(p)) return rc;
  const int64_t nk = (int64_t)p->n * p->k;
  if (nk == 0) return PINGS_OK;
  PINGS_ARG_CHECK(xyz_raw && rot_raw && scale_raw && alpha_raw && color_raw && pos && quat, "null input");
  PINGS_ARG_CHECK(!p->color_residual || base_color, "colour residual needs the base colour");
  PINGS_ARG_CHECK(gaussian_xyz && gaussian_scale && gaussian_rot && gaussian_alpha && gaussian_color && alpha_all,
                  "null output");
  SpawnArgs a{};
  a.p = *p; a.xyz_raw = xyz_raw; a.rot_raw = rot_raw; a.scale_raw = scale_raw; a.alpha_raw = alpha_raw;
  a.color_raw = color_raw; a.pos = pos; a.quat = quat; a.base = base_color; a.dist_ratio = dist_ratio;
  a.free_in = free_in; a.dest = dest; a.n_dev = n_rows_dev; a.nan_flag = nan_flag;
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope sc("spawn_forward", st);
  forward_kernel<<<(int)pings::ceil_div<int64_t>(nk, 256), 256, 0, st>>>(
      a, nk, gaussian_xyz, gaussian_scale, gaussian_rot, gaussian_alpha, gaussian_color, alpha_all,
      gaussian_free_mask);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}

PINGS_API int pings_spawn_backward(const pings_spawn_params* p, const float* xyz_raw, const float* rot_raw,
                                   const float* scale_raw, const float* alpha_raw, const float* color_raw,
                                   const float* quat, const float* base_color, const float* dist_ratio,
                                   const int32_t* dest, const float* dL_dxyz, const float* dL_dscale,
                                   const float* dL_drot, const float* dL_dalpha, const float* dL_dcolor,
                                   const float* dL_dalpha_all, float* dL_dxyz_raw, float* dL_drot_raw,
                                   float* dL_dscale_raw, float* dL_dalpha_raw, float* dL_dcolor_raw,
                                   void* stream) {
  if (int rc = check_params(p)) return rc;
  const int64_t nk = (int64_t)p->n * p->k;
  if (nk == 0) return PINGS_OK;
  PINGS_ARG_CHECK(xyz_raw && rot_raw && scale_raw && alpha_raw && color_raw && quat, "null input");
  PINGS_ARG_CHECK(!p->color_residual || base_color, "colour residual needs the base colour");
  PINGS_ARG_CHECK(dL_dxyz_raw && dL_drot_raw && dL_dscale_raw && dL_dalpha_raw && dL_dcolor_raw, "null output");
  SpawnArgs a{};
  a.p = *p; a.xyz_raw = xyz_raw; a.rot_raw = rot_raw; a.scale_raw = scale_raw; a.alpha_raw = alpha_raw;
  a.color_raw = color_raw; a.quat = quat; a.base = base_color; a.dist_ratio = dist_ratio; a.dest = dest;
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope sc("spawn_backward", st);
  backward_kernel<<<(int)pings::ceil_div<int64_t>(nk, 256), 256, 0, st>>>(
      a, nk, dL_dxyz, dL_dscale, dL_drot, dL_dalpha, dL_dcolor, dL_dalpha_all, dL_dxyz_raw, dL_drot_raw,
      dL_dscale_raw, dL_dalpha_raw, dL_dcolor_raw);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}
