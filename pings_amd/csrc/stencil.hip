// Finite-difference stencil of `Mapper.get_numerical_gradient` (utils/mapper.py:2319-2370) around the fused SDF query.
//
// The reference builds the shifted query points with six tensor additions and a concat, evaluates `self.sdf` on them and
// forms the gradient with slices, subtractions, divisions and another concat: ~25 torch operators forward and as many
// autograd nodes backward for an op whose arithmetic is 9 flops per sample.  Three streaming kernels instead:
//   stencil_points    x[N,3] -> [x+ex; x-ex; x+ey; x-ey; x+ez; x-ez]  ([6N,3]; one-sided: [x+ex; x+ey; x+ez], [3N,3])
//   stencil_gradient  S[6N] -> g[N,3] = (S+ - S-) * (1 / (2 eps))      (one-sided: (S+ - S(x)) * (1 / eps))
//   stencil_backward  dL/dg[N,3] -> dL/dS[6N] (+ dL/dS(x)[N] one-sided)
// x + e_axis adds eps to one coordinate and +0.0f to the other two (exact), as `x + eps_x` does (:2325-2336).  The
// division follows torch's HIP kernel for tensor / python-float: a multiplication by the fp32 reciprocal of the fp32
// divisor (ATen BinaryDivTrueKernel, CPU-scalar path) — what the unmodified mapper computes on the device.
#include "common.hpp"

namespace {

__global__ __launch_bounds__(256) void stencil_points_kernel(const float* __restrict__ x, long long N, float eps,
                                                             int two_side, float* __restrict__ out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (point, coordinate)
  if (t >= 3 * N) return;
  const long long i = t / 3;
  const int c = (int)(t - 3 * i);
  const float v = x[t];
  if (two_side) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float d = a == c ? eps : 0.0f;
      out[((2 * a) * N + i) * 3 + c] = v + d;
      out[((2 * a + 1) * N + i) * 3 + c] = v - d;
    }
  } else {
#pragma unroll
    for (int a = 0; a < 3; ++a) out[(a * N + i) * 3 + c] = v + (a == c ? eps : 0.0f);
  }
}

__global__ __launch_bounds__(256) void stencil_gradient_kernel(const float* __restrict__ s, const float* __restrict__ s0,
                                                               long long N, float inv, int two_side,
                                                               float* __restrict__ g) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 3 * N) return;
  const long long i = t / 3;
  const int a = (int)(t - 3 * i);
  const float d = two_side ? s[(2 * a) * N + i] - s[(2 * a + 1) * N + i] : s[a * N + i] - s0[i];
  g[t] = d * inv;
}

__global__ __launch_bounds__(256) void stencil_backward_kernel(const float* __restrict__ gg, long long N, float inv,
                                                               int two_side, float* __restrict__ up,
                                                               float* __restrict__ up0) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float v = gg[3 * i + a] * inv;
    if (two_side) {
      up[(2 * a) * N + i] = v;
      up[(2 * a + 1) * N + i] = -v;
    } else {
      up[a * N + i] = v;
      acc -= v;
    }
  }
  if (!two_side && up0) up0[i] = acc;
}

}  // namespace

PINGS_API int pings_stencil_points(const float* x, int64_t N, float eps, int two_side, float* out, void* stream) {
  PINGS_ARG_CHECK(N >= 0, "negative N");
  if (N == 0) return PINGS_OK;
  PINGS_ARG_CHECK(x && out, "null pointer");
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope ps("stencil", st);
  hipLaunchKernelGGL(stencil_points_kernel, dim3((unsigned)((3 * N + 255) / 256)), dim3(256), 0, st, x, (long long)N, eps,
                     two_side, out);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}

PINGS_API int pings_stencil_gradient(const float* sdf_shifted, const float* sdf_x, int64_t N, float eps, int two_side,
                                     float* grad, void* stream) {
  PINGS_ARG_CHECK(N >= 0 && eps != 0.0f, "bad N / eps");
  if (N == 0) return PINGS_OK;
  PINGS_ARG_CHECK(sdf_shifted && grad && (two_side || sdf_x), "null pointer");
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope ps("stencil", st);
  const float div = two_side ? (float)(2.0 * (double)eps) : eps;
  hipLaunchKernelGGL(stencil_gradient_kernel, dim3((unsigned)((3 * N + 255) / 256)), dim3(256), 0, st, sdf_shifted, sdf_x,
                     (long long)N, 1.0f / div, two_side, grad);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}

PINGS_API int pings_stencil_gradient_backward(const float* dL_dgrad, int64_t N, float eps, int two_side,
                                              float* dL_dsdf_shifted, float* dL_dsdf_x, void* stream) {
  PINGS_ARG_CHECK(N >= 0 && eps != 0.0f, "bad N / eps");
  if (N == 0) return PINGS_OK;
  PINGS_ARG_CHECK(dL_dgrad && dL_dsdf_shifted, "null pointer");
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope ps("stencil", st);
  const float div = two_side ? (float)(2.0 * (double)eps) : eps;
  hipLaunchKernelGGL(stencil_backward_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, dL_dgrad, (long long)N,
                     1.0f / div, two_side, dL_dsdf_shifted, dL_dsdf_x);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}
