// Tracker registration step: normal equations of one point-to-implicit-model Gauss-Newton /
// Levenberg-Marquardt iteration, utils/tracker.py:608-689 (`implicit_reg`, adapted there from LocNDF).
//
// With J_i = [p_i x g_i, g_i] (rotation first, then translation) the reference forms  N = J^T (w J)  and
// g = -(J w)^T r  with two GEMMs over the n x 6 Jacobian after materialising the cross products, the concatenation
// and the weighted copy (five n-sized temporaries).  Here one pass reads the 8 floats of a point and accumulates the
// 21 distinct entries of N and the 6 of g in fp64 registers; wave shuffle + LDS reduce per workgroup, per-workgroup
// partials in global memory, and a second tiny kernel sums them in fixed order: bitwise reproducible.  HBM bound:
// 32 B per point.
#include "common.hpp"

namespace {

constexpr int kTerms = 27;    // 21 upper-triangle entries of N, 6 of g
constexpr int kBlocks = 512;

__global__ __launch_bounds__(256) void reg_accumulate_kernel(const float* __restrict__ pts, const float* __restrict__ grad,
                                                             const float* __restrict__ res, const float* __restrict__ wgt,
                                                             long long n, double* __restrict__ partial) {
  __shared__ double red[4][kTerms];
  double acc[kTerms];
#pragma unroll
  for (int k = 0; k < kTerms; ++k) acc[k] = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    const float gx = grad[3 * i], gy = grad[3 * i + 1], gz = grad[3 * i + 2];
    float J[6];
    J[0] = py * gz - pz * gy;   // torch.linalg.cross(points, sdf_grad) in fp32, as the reference
    J[1] = pz * gx - px * gz;
    J[2] = px * gy - py * gx;
    J[3] = gx; J[4] = gy; J[5] = gz;
    const double w = (double)wgt[i], r = (double)res[i];
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const double wa = w * (double)J[a];
#pragma unroll
      for (int b = a; b < 6; ++b) acc[k++] += wa * (double)J[b];
      acc[21 + a] -= wa * r;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < kTerms; ++k) {
    double v = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < kTerms)
    partial[(size_t)blockIdx.x * kTerms + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

__global__ void reg_finish_kernel(const double* __restrict__ partial, int nblocks, float* __restrict__ out) {
  __shared__ double tot[kTerms];
  const int k = threadIdx.x;
  if (k < kTerms) {
    double v = 0.0;
    for (int b = 0; b < nblocks; ++b) v += partial[(size_t)b * kTerms + k];
    tot[k] = v;
  }
  __syncthreads();
  if (k == 0) {
    int t = 0;
    for (int a = 0; a < 6; ++a)
      for (int b = a; b < 6; ++b) {
        out[a * 6 + b] = (float)tot[t];
        out[b * 6 + a] = (float)tot[t];
        ++t;
      }
    for (int a = 0; a < 6; ++a) out[36 + a] = (float)tot[21 + a];
  }
}

// LM damping, the fp64 6x6 solve and the exponential map of one registration step (utils/tracker.py:649-689, :774-783)
// in one single-thread kernel: the reference spends ~30 tiny torch launches and a host sync (linalg.inv checks its
// info word) on it, 50-100 times per frame.  N is damped in fp32 like the reference's `N_mat += lambda diag(N_mat)`,
// then everything is fp64: t = N^-1 g by Gaussian elimination with partial pivoting, R = I + S sin(a) + S^2 (1 - cos a).
// status (optional device word): PINGS_REG_SINGULAR = a pivot is exactly zero or not finite — the case in which the
// reference's torch.linalg.inv raises (utils/tracker.py:668; LAPACK getrf info > 0); PINGS_REG_ILL_CONDITIONED = the
// smallest pivot is below 1e-7 (one fp32 ulp: N arrives rounded to fp32) of the largest entry of the damped matrix,
// i.e. the step is decided by rounding noise; PINGS_REG_NONFINITE = t or the pose came out Inf / NaN.
__global__ void reg_solve_kernel(const float* __restrict__ ng, float lm_lambda, double* __restrict__ T_out,
                                 double* __restrict__ t_out, int32_t* __restrict__ status) {
  double A[6][7];
  double scale = 0.0, min_piv = 1e300;
  int flags = 0;
  for (int r = 0; r < 6; ++r) {
    for (int c = 0; c < 6; ++c) {
      float v = ng[r * 6 + c];
      if (r == c) v += lm_lambda * v;
      A[r][c] = (double)v;
      scale = fmax(scale, fabs((double)v));
    }
    A[r][6] = (double)ng[36 + r];
  }
  for (int k = 0; k < 6; ++k) {
    int piv = k;
    for (int r = k + 1; r < 6; ++r)
      if (fabs(A[r][k]) > fabs(A[piv][k])) piv = r;
    if (piv != k)
      for (int c = 0; c < 7; ++c) { const double t = A[k][c]; A[k][c] = A[piv][c]; A[piv][c] = t; }
    if (A[k][k] == 0.0 || !isfinite(A[k][k])) flags |= PINGS_REG_SINGULAR;
    min_piv = fmin(min_piv, fabs(A[k][k]));
    const double inv = 1.0 / A[k][k];
    for (int r = k + 1; r < 6; ++r) {
      const double f = A[r][k] * inv;
      for (int c = k; c < 7; ++c) A[r][c] -= f * A[k][c];
    }
  }
  double t[6];
  for (int r = 5; r >= 0; --r) {
    double v = A[r][6];
    for (int c = r + 1; c < 6; ++c) v -= A[r][c] * t[c];
    t[r] = v / A[r][r];
  }
  const double angle = sqrt((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]);
  const double ax = t[0] / angle, ay = t[1] / angle, az = t[2] / angle;   // angle = 0 -> NaN, as the reference
  const double S[3][3] = {{0.0, -az, ay}, {az, 0.0, -ax}, {-ay, ax, 0.0}};
  const double sn = sin(angle), cs = 1.0 - cos(angle);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double s2 = 0.0;
      for (int k = 0; k < 3; ++k) s2 += S[r][k] * S[k][c];
      T_out[r * 4 + c] = (r == c ? 1.0 : 0.0) + S[r][c] * sn + s2 * cs;
    }
  for (int r = 0; r < 3; ++r) T_out[r * 4 + 3] = t[3 + r];
  T_out[12] = 0.0; T_out[13] = 0.0; T_out[14] = 0.0; T_out[15] = 1.0;
  if (t_out)
    for (int r = 0; r < 6; ++r) t_out[r] = t[r];
  if (status) {
    if (!(min_piv >= 1e-7 * scale)) flags |= PINGS_REG_ILL_CONDITIONED;   // also when scale is NaN
    bool fin = true;
    for (int r = 0; r < 6; ++r) fin = fin && isfinite(t[r]);
    // (t = 0 exactly makes the axis 0 / 0, as in the reference's expmap: reported as non-finite, not as singular)
    for (int r = 0; r < 12; ++r) fin = fin && isfinite(T_out[r]);
    if (!fin) flags |= PINGS_REG_NONFINITE;
    *status = flags;
  }
}

}  // namespace

PINGS_API int pings_reg_solve_checked(const float* normal_eq, float lm_lambda, double* T_out, double* t_out,
                                      int32_t* status_dev, int32_t* status_host, void* stream) {
  PINGS_ARG_CHECK(normal_eq && T_out, "null pointer");
  PINGS_ARG_CHECK(status_dev || !status_host, "status_host needs status_dev");
  hipStream_t st = pings::as_stream(stream);
  {
    pings::prof::Scope sc("reg_solve", st);
    reg_solve_kernel<<<1, 1, 0, st>>>(normal_eq, lm_lambda, T_out, t_out, status_dev);
    PINGS_LAUNCH_CHECK();
  }
  if (status_host) {   // the reference synchronises here as well (linalg.inv reads its info word)
    const uint32_t* w[1] = {reinterpret_cast<const uint32_t*>(status_dev)};
    uint32_t v = 0;
    const int rc = pings::host_read_words(w, 1, &v, st);
    if (rc != PINGS_OK) return rc;
    *status_host = (int32_t)v;
  }
  return PINGS_OK;
}

PINGS_API int pings_reg_solve(const float* normal_eq, float lm_lambda, double* T_out, double* t_out, void* stream) {
  return pings_reg_solve_checked(normal_eq, lm_lambda, T_out, t_out, nullptr, nullptr, stream);
}

PINGS_API size_t pings_reg_normal_equations_scratch_bytes(void) { return sizeof(double) * kBlocks * kTerms; }

PINGS_API int pings_reg_normal_equations(const float* points, const float* sdf_grad, const float* sdf_residual,
                                         const float* weight, int64_t n, void* scratch, float* out, void* stream) {
  PINGS_ARG_CHECK(n >= 0 && out && scratch, "bad argument");
  PINGS_ARG_CHECK(n == 0 || (points && sdf_grad && sdf_residual && weight), "null input");
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope sc("reg_normal_equations", st);
  int nb = (int)((n + 255) / 256);
  nb = nb < 1 ? 1 : (nb > kBlocks ? kBlocks : nb);
  reg_accumulate_kernel<<<nb, 256, 0, st>>>(points, sdf_grad, sdf_residual, weight, (long long)n,
                                            reinterpret_cast<double*>(scratch));
  PINGS_LAUNCH_CHECK();
  reg_finish_kernel<<<1, 64, 0, st>>>(reinterpret_cast<const double*>(scratch), nb, out);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}
