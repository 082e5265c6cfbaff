// Image-space operators of `render` (gaussian_splatting/gaussian_renderer/__init__.py:330-337), for gfx950.
//
// depth2normal (gaussian_splatting/utils/point_utils.py:83-149): camera-frame normals from the rendered depth map.
// Every pixel is un-projected to P = ((u - cx) d / fx, (v - cy) d / fy, d), the four neighbour differences
// (replicate padding, each multiplied by the neighbour's visibility mask) are crossed pairwise, the four cross
// products are summed and normalised, and the result is multiplied by the centre's mask and — fused here, `render`
// does it right after (:335) — by the detached rendered alpha.  The reference runs ~25 full-frame torch kernels for
// the forward pass and autograd doubles that; here it is one stencil kernel forward and two backward
// (per-centre adjoints of the five stencil points, then a gather of the five contributions every pixel receives:
// no atomics, bitwise reproducible).  HBM bound: forward 8 B in + 12 B out per pixel, backward 20 B in + 60 B
// scratch write + 60 B scratch read + 4 B out.
#include <cstdlib>
#include "common.hpp"

namespace {

struct D2N {
  int H, W;
  float cx, cy, ifx, ify, min_alpha;
  const float* depth;
  const float* alpha;   // nullable: weight (detached rendered alpha) and, with `mask` null, the visibility test
  const uint8_t* mask;  // nullable: explicit visibility mask (depth2normal's second argument)
};

__device__ inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ inline bool visible(const D2N& a, int x, int y) {
  const size_t i = (size_t)y * a.W + x;
  return a.mask ? a.mask[i] != 0 : a.alpha[i] > a.min_alpha;
}

__device__ inline void unproject(const D2N& a, int x, int y, float& px, float& py, float& pz) {
  const float d = a.depth[(size_t)y * a.W + x];
  px = (((float)x - a.cx) * d) * a.ifx;   // (w - cx) * camD, then @ Kinv^T  (point_utils.py:101-113)
  py = (((float)y - a.cy) * d) * a.ify;
  pz = d;
}

struct Stencil {
  float c[3], u[3], l[3], b[3], r[3];  // masked centre and the four masked differences
  float mu, ml, mb, mr, mc;
};

__device__ inline Stencil stencil(const D2N& a, int x, int y) {
  Stencil s;
  const int xl = clampi(x - 1, 0, a.W - 1), xr = clampi(x + 1, 0, a.W - 1);
  const int yu = clampi(y - 1, 0, a.H - 1), yb = clampi(y + 1, 0, a.H - 1);
  s.mc = visible(a, x, y) ? 1.f : 0.f;
  s.mu = visible(a, x, yu) ? 1.f : 0.f;
  s.ml = visible(a, xl, y) ? 1.f : 0.f;
  s.mb = visible(a, x, yb) ? 1.f : 0.f;
  s.mr = visible(a, xr, y) ? 1.f : 0.f;
  float p[3];
  unproject(a, x, y, p[0], p[1], p[2]);
#pragma unroll
  for (int k = 0; k < 3; ++k) s.c[k] = p[k] * s.mc;
  unproject(a, x, yu, p[0], p[1], p[2]);
#pragma unroll
  for (int k = 0; k < 3; ++k) s.u[k] = (p[k] - s.c[k]) * s.mu;
  unproject(a, xl, y, p[0], p[1], p[2]);
#pragma unroll
  for (int k = 0; k < 3; ++k) s.l[k] = (p[k] - s.c[k]) * s.ml;
  unproject(a, x, yb, p[0], p[1], p[2]);
#pragma unroll
  for (int k = 0; k < 3; ++k) s.b[k] = (p[k] - s.c[k]) * s.mb;
  unproject(a, xr, y, p[0], p[1], p[2]);
#pragma unroll
  for (int k = 0; k < 3; ++k) s.r[k] = (p[k] - s.c[k]) * s.mr;
  return s;
}

__device__ inline void cross(const float a[3], const float b[3], float o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

// n = u x l + r x u + b x r + l x b
__device__ inline void normal_sum(const Stencil& s, float n[3]) {
  float t[3];
  cross(s.u, s.l, n);
  cross(s.r, s.u, t); n[0] += t[0]; n[1] += t[1]; n[2] += t[2];
  cross(s.b, s.r, t); n[0] += t[0]; n[1] += t[1]; n[2] += t[2];
  cross(s.l, s.b, t); n[0] += t[0]; n[1] += t[1]; n[2] += t[2];
}

__global__ __launch_bounds__(256) void d2n_fwd_kernel(D2N a, float* __restrict__ out) {
  const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (x >= a.W || y >= a.H) return;
  const Stencil s = stencil(a, x, y);
  float n[3];
  normal_sum(s, n);
  const float nrm = fmaxf(sqrtf((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]), 1e-12f);  // F.normalize eps
  const size_t i = (size_t)y * a.W + x, HW = (size_t)a.H * a.W;
  const float w = s.mc * (a.alpha ? a.alpha[i] : 1.f);
  out[i] = n[0] / nrm * w;
  out[HW + i] = n[1] / nrm * w;
  out[2 * HW + i] = n[2] / nrm * w;
}

// Pass 1: per centre pixel, dL/d(un-projected point) of the centre and of its four stencil neighbours, each already
// chained through the RECEIVING pixel's P = (ax d, ay d, d), i.e. five scalars dL/d(depth) (planar [5][H][W]; the
// first version sent the 15 vector components and moved 250 MB through HBM at 1080p).
__device__ inline void stencil_adjoint(const D2N& a, const float* __restrict__ g_out, int x, int y, float out[5]) {
  const size_t i = (size_t)y * a.W + x, HW = (size_t)a.H * a.W;
  const Stencil s = stencil(a, x, y);
  float n[3];
  normal_sum(s, n);
  const float nn = sqrtf((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
  const float w = s.mc * (a.alpha ? a.alpha[i] : 1.f);
  float g[3] = {g_out[i] * w, g_out[HW + i] * w, g_out[2 * HW + i] * w};
  float gn[3];
  if (nn > 1e-12f) {  // d(n / |n|) = (I - nh nh^T) / |n|
    const float inv = 1.f / nn;
    const float h0 = n[0] * inv, h1 = n[1] * inv, h2 = n[2] * inv;
    const float dot = (h0 * g[0] + h1 * g[1]) + h2 * g[2];
    gn[0] = (g[0] - h0 * dot) * inv;
    gn[1] = (g[1] - h1 * dot) * inv;
    gn[2] = (g[2] - h2 * dot) * inv;
  } else {            // clamp_min(eps) active: n / eps
    gn[0] = g[0] / 1e-12f; gn[1] = g[1] / 1e-12f; gn[2] = g[2] / 1e-12f;
  }
  // n = u x l + r x u + b x r + l x b.  d(a x b) . gn:  wrt a = b x gn,  wrt b = gn x a
  float gu[3], gl[3], gb[3], gr[3], t[3];
  cross(s.l, gn, gu); cross(gn, s.r, t);                      // u: first of (u x l), second of (r x u)
  gu[0] += t[0]; gu[1] += t[1]; gu[2] += t[2];
  cross(gn, s.u, gl); cross(s.b, gn, t);                      // l: second of (u x l), first of (l x b)
  gl[0] += t[0]; gl[1] += t[1]; gl[2] += t[2];
  cross(s.r, gn, gb); cross(gn, s.l, t);                      // b: first of (b x r), second of (l x b)
  gb[0] += t[0]; gb[1] += t[1]; gb[2] += t[2];
  cross(s.u, gn, gr); cross(gn, s.b, t);                      // r: first of (r x u), second of (b x r)
  gr[0] += t[0]; gr[1] += t[1]; gr[2] += t[2];
  // differences: u = (P_u - c) m_u, c = P_c m_c.  At the image border the replicate padding makes a neighbour the
  // pixel itself: its two contributions (+d to the neighbour, -d m_c to the centre) land on the same point and are
  // folded here into d (1 - m_c).  Left separate they are +-1e11 (n = 0 there, so the normalisation divides by its
  // 1e-12 floor) and swallow every other term of that pixel in fp32.
  const bool su = y == 0, sl = x == 0, sb = y == a.H - 1, sr = x == a.W - 1;
  float to_c = 0.f, to_u = 0.f, to_l = 0.f, to_b = 0.f, to_r = 0.f;
  const float ax = ((float)x - a.cx) * a.ifx, ay = ((float)y - a.cy) * a.ify;
  const float axl = ((float)(x - 1) - a.cx) * a.ifx, axr = ((float)(x + 1) - a.cx) * a.ifx;
  const float ayu = ((float)(y - 1) - a.cy) * a.ify, ayb = ((float)(y + 1) - a.cy) * a.ify;
  const float rc[3] = {ax, ay, 1.f}, ru[3] = {ax, ayu, 1.f}, rl[3] = {axl, ay, 1.f}, rb[3] = {ax, ayb, 1.f},
              rr[3] = {axr, ay, 1.f};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float du = gu[k] * s.mu, dl = gl[k] * s.ml, db = gb[k] * s.mb, dr = gr[k] * s.mr;
    const float other = (su ? 0.f : du) + (sl ? 0.f : dl) + (sb ? 0.f : db) + (sr ? 0.f : dr);
    const float self = (su ? du : 0.f) + (sl ? dl : 0.f) + (sb ? db : 0.f) + (sr ? dr : 0.f);
    to_c += (self * (1.f - s.mc) - other * s.mc) * rc[k];  // to the centre's own point
    to_u += (su ? 0.f : du) * ru[k];                        // to the upper neighbour's point
    to_l += (sl ? 0.f : dl) * rl[k];
    to_b += (sb ? 0.f : db) * rb[k];
    to_r += (sr ? 0.f : dr) * rr[k];
  }
  out[0] = to_c; out[1] = to_u; out[2] = to_l; out[3] = to_b; out[4] = to_r;
}

__global__ __launch_bounds__(256) void d2n_bwd_stencil_kernel(D2N a, const float* __restrict__ g_out,
                                                              float* __restrict__ adj) {
  const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (x >= a.W || y >= a.H) return;
  const size_t i = (size_t)y * a.W + x, HW = (size_t)a.H * a.W;
  float o[5];
  stencil_adjoint(a, g_out, x, y, o);
  adj[i] = o[0];
  adj[HW + i] = o[1];
  adj[2 * HW + i] = o[2];
  adj[3 * HW + i] = o[3];
  adj[4 * HW + i] = o[4];
}

// Pass 2: every pixel collects what the stencils that reference it sent (its own centre term, and the up / left /
// bottom / right terms of the pixels whose clamped neighbour it is).
__global__ __launch_bounds__(256) void d2n_bwd_gather_kernel(D2N a, const float* __restrict__ adj,
                                                             float* __restrict__ g_depth) {
  const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (x >= a.W || y >= a.H) return;
  const size_t HW = (size_t)a.H * a.W;
  float v = adj[(size_t)y * a.W + x];
  // pixel (x, y) is the UPPER neighbour of (x, y+1), ... (self-references at the border were folded in pass 1)
  if (y + 1 < a.H) v += adj[HW + (size_t)(y + 1) * a.W + x];
  if (x + 1 < a.W) v += adj[2 * HW + (size_t)y * a.W + x + 1];       // LEFT neighbour of (x+1, y)
  if (y >= 1) v += adj[3 * HW + (size_t)(y - 1) * a.W + x];          // BOTTOM neighbour of (x, y-1)
  if (x >= 1) v += adj[4 * HW + (size_t)y * a.W + x - 1];            // RIGHT neighbour of (x-1, y)
  g_depth[(size_t)y * a.W + x] = v;
}

// Both passes in one (round 4): a workgroup evaluates the stencil adjoints of its 32 x 8 tile plus a one-pixel ring
// (340 centres: 1.33x the arithmetic of pass 1) into LDS and every pixel collects its five terms from there, in the
// order of d2n_bwd_gather_kernel: bit-identical to the two-pass form, without the 5-plane scratch image going through
// HBM twice (1080p: 131 MB of traffic -> 49 MB).  Measured: the SAME 0.045 ms — the pass is bound by the stencil
// arithmetic (five un-projections, eight cross products, the normalisation's adjoint per centre), not by its traffic;
// what the fusion buys is one launch and no 41 MB scratch image.
__global__ __launch_bounds__(256) void d2n_bwd_fused_kernel(D2N a, const float* __restrict__ g_out,
                                                            float* __restrict__ g_depth) {
  constexpr int TW = 34, TH = 10, LD = 35;
  __shared__ float sAdj[5][TH][LD];
  const int x0 = blockIdx.x * 32 - 1, y0 = blockIdx.y * 8 - 1;      // the ring starts one pixel out
  for (int e = threadIdx.x; e < TW * TH; e += 256) {
    const int lx = e % TW, ly = e / TW;
    const int cx = x0 + lx, cy = y0 + ly;
    float o[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (cx >= 0 && cx < a.W && cy >= 0 && cy < a.H) stencil_adjoint(a, g_out, cx, cy, o);
#pragma unroll
    for (int k = 0; k < 5; ++k) sAdj[k][ly][lx] = o[k];
  }
  __syncthreads();
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int x = blockIdx.x * 32 + tx, y = blockIdx.y * 8 + ty;
  if (x >= a.W || y >= a.H) return;
  const int lx = tx + 1, ly = ty + 1;
  float v = sAdj[0][ly][lx];
  if (y + 1 < a.H) v += sAdj[1][ly + 1][lx];      // pixel (x, y) is the UPPER neighbour of (x, y + 1)
  if (x + 1 < a.W) v += sAdj[2][ly][lx + 1];      // the LEFT neighbour of (x + 1, y)
  if (y >= 1) v += sAdj[3][ly - 1][lx];           // the BOTTOM neighbour of (x, y - 1)
  if (x >= 1) v += sAdj[4][ly][lx - 1];           // the RIGHT neighbour of (x - 1, y)
  g_depth[(size_t)y * a.W + x] = v;
}

int fill(D2N& a, const float* depth, const float* alpha, const uint8_t* mask, int H, int W, float cx, float cy,
         float fx, float fy, float min_alpha) {
  PINGS_ARG_CHECK(H > 0 && W > 0 && depth, "bad image");
  PINGS_ARG_CHECK(alpha || mask, "need a visibility mask or the rendered alpha");
  PINGS_ARG_CHECK(fx != 0.f && fy != 0.f, "zero focal length");
  a.H = H; a.W = W; a.cx = cx; a.cy = cy; a.ifx = 1.0f / fx; a.ify = 1.0f / fy; a.min_alpha = min_alpha;
  a.depth = depth; a.alpha = alpha; a.mask = mask;
  return PINGS_OK;
}

}  // namespace

PINGS_API int pings_depth2normal_forward(const float* depth, const float* alpha, const uint8_t* mask, int H, int W,
                                         float cx, float cy, float fx, float fy, float min_alpha, float* normal,
                                         void* stream) {
  D2N a;
  if (int rc = fill(a, depth, alpha, mask, H, W, cx, cy, fx, fy, min_alpha)) return rc;
  PINGS_ARG_CHECK(normal != nullptr, "null output");
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope sc("depth2normal_fwd", st);
  d2n_fwd_kernel<<<dim3(pings::ceil_div(W, 32), pings::ceil_div(H, 8)), 256, 0, st>>>(a, normal);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}

PINGS_API size_t pings_depth2normal_backward_scratch_bytes(int H, int W) { return (size_t)5 * H * W * sizeof(float); }

PINGS_API int pings_depth2normal_backward(const float* depth, const float* alpha, const uint8_t* mask, int H, int W,
                                          float cx, float cy, float fx, float fy, float min_alpha,
                                          const float* dL_dnormal, void* scratch, float* dL_ddepth, void* stream) {
  D2N a;
  if (int rc = fill(a, depth, alpha, mask, H, W, cx, cy, fx, fy, min_alpha)) return rc;
  PINGS_ARG_CHECK(dL_dnormal && scratch && dL_ddepth, "null pointer");
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope sc("depth2normal_bwd", st);
  const dim3 grid(pings::ceil_div(W, 32), pings::ceil_div(H, 8));
  const char* env = getenv("PINGS_D2N_BWD");       // "2pass": the two-kernel form through the scratch image (A/B, tests)
  if (env && env[0] == '2') {
    d2n_bwd_stencil_kernel<<<grid, 256, 0, st>>>(a, dL_dnormal, reinterpret_cast<float*>(scratch));
    PINGS_LAUNCH_CHECK();
    d2n_bwd_gather_kernel<<<grid, 256, 0, st>>>(a, reinterpret_cast<const float*>(scratch), dL_ddepth);
    PINGS_LAUNCH_CHECK();
    return PINGS_OK;
  }
  d2n_bwd_fused_kernel<<<grid, 256, 0, st>>>(a, dL_dnormal, dL_ddepth);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}
