// Per-camera affine exposure correction of the rendered image (gaussian_renderer/__init__.py:449-461):
//   out[c, p] = sum_k M[c, k] * img[k, p] + b[c]        (reference: img.permute(1,2,0).view(-1,3) @ M^T + b)
// The reference runs it as a [H W, 3] x [3, 3] GEMM plus its two backward GEMMs and a bias reduction — at 1080p
// 1.27 ms per training step in library GEMM kernels tiled for real matrices (measured with rocprofv3 on this very
// step).  It is a streaming pass: 12 B/pixel read + 12 written forward; backward 24 read + 12 written and twelve
// sums (dM: 9, db: 3): a thread's own pixels in fp32, then fp64 per wave / workgroup, finished by one workgroup of
// twelve waves in fixed order (bitwise reproducible, no atomics).  HBM-bound.  (First version: fp64 per pixel, 4-byte
// loads and a finishing kernel whose 12 threads each walked 1,024 partials serially: 0.22 ms at 1080p.)
#include "common.hpp"

namespace {

constexpr int EXP_BLOCKS = 1024;

__global__ __launch_bounds__(256) void exposure_fwd_kernel(const float* __restrict__ img, const float* __restrict__ M,
                                                           const float* __restrict__ b, long long HW, int vec,
                                                           float* __restrict__ out) {
  const float m00 = M[0], m01 = M[1], m02 = M[2], m10 = M[3], m11 = M[4], m12 = M[5], m20 = M[6], m21 = M[7], m22 = M[8];
  const float b0 = b[0], b1 = b[1], b2 = b[2];
  const long long n4 = vec ? HW / 4 : 0;   // float4 path: 16-byte aligned planes (HW % 4 == 0, aligned bases)
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long long i = t0; i < n4; i += stride) {
    const float4 r = reinterpret_cast<const float4*>(img)[i];
    const float4 g = reinterpret_cast<const float4*>(img + HW)[i];
    const float4 bl = reinterpret_cast<const float4*>(img + 2 * HW)[i];
    float4 o0, o1, o2;
#define PINGS_EXP(C) \
    o0.C = fmaf(m02, bl.C, fmaf(m01, g.C, fmaf(m00, r.C, b0))); \
    o1.C = fmaf(m12, bl.C, fmaf(m11, g.C, fmaf(m10, r.C, b1))); \
    o2.C = fmaf(m22, bl.C, fmaf(m21, g.C, fmaf(m20, r.C, b2)));
    PINGS_EXP(x) PINGS_EXP(y) PINGS_EXP(z) PINGS_EXP(w)
#undef PINGS_EXP
    reinterpret_cast<float4*>(out)[i] = o0;
    reinterpret_cast<float4*>(out + HW)[i] = o1;
    reinterpret_cast<float4*>(out + 2 * HW)[i] = o2;
  }
  for (long long p = 4 * n4 + t0; p < HW; p += stride) {   // everything, when the planes are not float4-addressable
    const float r = img[p], g = img[HW + p], bl = img[2 * HW + p];
    out[p] = fmaf(m02, bl, fmaf(m01, g, fmaf(m00, r, b0)));
    out[HW + p] = fmaf(m12, bl, fmaf(m11, g, fmaf(m10, r, b1)));
    out[2 * HW + p] = fmaf(m22, bl, fmaf(m21, g, fmaf(m20, r, b2)));
  }
}

__device__ inline double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// g_img[k, p] = sum_c M[c, k] g[c, p];  partial sums of dM[c, k] += g[c, p] img[k, p], db[c] += g[c, p].
// A thread sums its own (<= 8 at 1080p) pixels in fp32, everything across threads is summed in fp64 in fixed order.
__global__ __launch_bounds__(256) void exposure_bwd_kernel(const float* __restrict__ img, const float* __restrict__ M,
                                                           const float* __restrict__ g, long long HW, int vec,
                                                           float* __restrict__ g_img, double* __restrict__ partials) {
  __shared__ double sP[4][12];
  const float m00 = M[0], m01 = M[1], m02 = M[2], m10 = M[3], m11 = M[4], m12 = M[5], m20 = M[6], m21 = M[7], m22 = M[8];
  float acc[12];
#pragma unroll
  for (int q = 0; q < 12; ++q) acc[q] = 0.f;
#define PINGS_EXP_ACC(x0, x1, x2, g0, g1, g2)                                                       \
  acc[0] = fmaf(g0, x0, acc[0]); acc[1] = fmaf(g0, x1, acc[1]); acc[2] = fmaf(g0, x2, acc[2]);      \
  acc[3] = fmaf(g1, x0, acc[3]); acc[4] = fmaf(g1, x1, acc[4]); acc[5] = fmaf(g1, x2, acc[5]);      \
  acc[6] = fmaf(g2, x0, acc[6]); acc[7] = fmaf(g2, x1, acc[7]); acc[8] = fmaf(g2, x2, acc[8]);      \
  acc[9] += g0; acc[10] += g1; acc[11] += g2;
  const long long n4 = vec ? HW / 4 : 0;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long long i = t0; i < n4; i += stride) {
    const float4 x0 = reinterpret_cast<const float4*>(img)[i];
    const float4 x1 = reinterpret_cast<const float4*>(img + HW)[i];
    const float4 x2 = reinterpret_cast<const float4*>(img + 2 * HW)[i];
    const float4 g0 = reinterpret_cast<const float4*>(g)[i];
    const float4 g1 = reinterpret_cast<const float4*>(g + HW)[i];
    const float4 g2 = reinterpret_cast<const float4*>(g + 2 * HW)[i];
    if (g_img) {
      float4 o0, o1, o2;
#define PINGS_EXP_G(C)                                          \
      o0.C = fmaf(m20, g2.C, fmaf(m10, g1.C, m00 * g0.C));      \
      o1.C = fmaf(m21, g2.C, fmaf(m11, g1.C, m01 * g0.C));      \
      o2.C = fmaf(m22, g2.C, fmaf(m12, g1.C, m02 * g0.C));
      PINGS_EXP_G(x) PINGS_EXP_G(y) PINGS_EXP_G(z) PINGS_EXP_G(w)
#undef PINGS_EXP_G
      reinterpret_cast<float4*>(g_img)[i] = o0;
      reinterpret_cast<float4*>(g_img + HW)[i] = o1;
      reinterpret_cast<float4*>(g_img + 2 * HW)[i] = o2;
    }
    PINGS_EXP_ACC(x0.x, x1.x, x2.x, g0.x, g1.x, g2.x)
    PINGS_EXP_ACC(x0.y, x1.y, x2.y, g0.y, g1.y, g2.y)
    PINGS_EXP_ACC(x0.z, x1.z, x2.z, g0.z, g1.z, g2.z)
    PINGS_EXP_ACC(x0.w, x1.w, x2.w, g0.w, g1.w, g2.w)
  }
  for (long long p = 4 * n4 + t0; p < HW; p += stride) {
    const float x0 = img[p], x1 = img[HW + p], x2 = img[2 * HW + p];
    const float g0 = g[p], g1 = g[HW + p], g2 = g[2 * HW + p];
    if (g_img) {
      g_img[p] = fmaf(m20, g2, fmaf(m10, g1, m00 * g0));
      g_img[HW + p] = fmaf(m21, g2, fmaf(m11, g1, m01 * g0));
      g_img[2 * HW + p] = fmaf(m22, g2, fmaf(m12, g1, m02 * g0));
    }
    PINGS_EXP_ACC(x0, x1, x2, g0, g1, g2)
  }
#undef PINGS_EXP_ACC
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 12; ++q) {
    const double s = wave_sum_f64((double)acc[q]);
    if (lane == 0) sP[wave][q] = s;
  }
  __syncthreads();
  if (threadIdx.x < 12)
    partials[(size_t)blockIdx.x * 12 + threadIdx.x] =
        ((sP[0][threadIdx.x] + sP[1][threadIdx.x]) + sP[2][threadIdx.x]) + sP[3][threadIdx.x];
}

// one wave per sum: lane l adds the partials of workgroups l, l + 64, ... in ascending order, then the lanes are
// folded by the fixed butterfly of wave_sum_f64 (bitwise reproducible)
__global__ __launch_bounds__(64 * 12) void exposure_final_kernel(const double* __restrict__ partials, int nblocks,
                                                                 float* __restrict__ gM, float* __restrict__ gb) {
  const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double s = 0.0;
  for (int b = lane; b < nblocks; b += 64) s += partials[(size_t)b * 12 + q];
  s = wave_sum_f64(s);
  if (lane == 0) {
    if (q < 9) gM[q] = (float)s; else gb[q - 9] = (float)s;
  }
}

}  // namespace

PINGS_API int pings_exposure_forward(const float* img, const float* M, const float* b, int64_t HW, float* out,
                                     void* stream) {
  PINGS_ARG_CHECK(HW >= 0, "negative size");
  if (HW == 0) return PINGS_OK;
  PINGS_ARG_CHECK(img && M && b && out, "null pointer");
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope ps("exposure_fwd", st);
  const int vec = (((reinterpret_cast<uintptr_t>(img) | reinterpret_cast<uintptr_t>(out)) & 15) == 0 && HW % 4 == 0) ? 1 : 0;
  const long long want = (HW / (vec ? 4 : 1) + 255) / 256;
  const int nblocks = (int)(want < EXP_BLOCKS ? want : EXP_BLOCKS);
  hipLaunchKernelGGL(exposure_fwd_kernel, dim3(nblocks), dim3(256), 0, st, img, M, b, (long long)HW, vec, out);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}

PINGS_API size_t pings_exposure_backward_scratch_bytes(void) { return sizeof(double) * 12 * EXP_BLOCKS; }

PINGS_API int pings_exposure_backward(const float* img, const float* M, const float* g_out, int64_t HW,
                                      void* scratch, float* g_img, float* g_M, float* g_b, void* stream) {
  PINGS_ARG_CHECK(HW > 0 && img && M && g_out && scratch && g_M && g_b, "null pointer");
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope ps("exposure_bwd", st);
  double* partials = reinterpret_cast<double*>(scratch);
  const int vec = (((reinterpret_cast<uintptr_t>(img) | reinterpret_cast<uintptr_t>(g_out) |
                     reinterpret_cast<uintptr_t>(g_img)) & 15) == 0 && HW % 4 == 0) ? 1 : 0;
  const long long want = (HW / (vec ? 4 : 1) + 255) / 256;
  const int nblocks = (int)(want < EXP_BLOCKS ? want : EXP_BLOCKS);
  hipLaunchKernelGGL(exposure_bwd_kernel, dim3(nblocks), dim3(256), 0, st, img, M, g_out, (long long)HW, vec, g_img,
                     partials);
  PINGS_LAUNCH_CHECK();
  hipLaunchKernelGGL(exposure_final_kernel, dim3(1), dim3(64 * 12), 0, st, partials, nblocks, g_M, g_b);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}
