// First-order backward of the fused neural-point SDF query (training path of Mapper.sdf_mapping,
// utils/mapper.py:822-970): gradients of  S(x) = sum_m w_m * scale * MLP([f_{idx_m}, x - p_{idx_m}])
// (per-neighbour mode) or  S = scale * MLP(sum_m w_m [f, x - p])  (weighted_first, neural_gaussians.py:701)
// with respect to the feature rows and the decoder parameters.
//
//   sdf_gather_kernel   one wave64 per query: rebuilds the MLP input rows from the saved neighbours / weights
//                       (one row per (query, neighbour), or one weighted row per query), the per-row upstream
//                       gradient, and the (destination feature row, source row) pairs of the scatter.
//   pings_mlp_backward  the MFMA decoder backward (csrc/mlp.hip) on those rows: dL/d(input rows) and the decoder
//                       gradients (per-workgroup partials, fixed-order sum).
//   pings_rows::build + gather_sum (csrc/row_scatter.hip)
//                       counting sort of the (destination row, source row) pairs and one pass over the feature-
//                       gradient table that sums every row's pairs in ascending pair id and stores the row once
//                       (zeros where nothing points): the scatter-add of the feature gradient without float atomics,
//                       bitwise reproducible (the reference's index_put / scatter_add backward is not), 5 launches
//                       where the library merge sort of round 1 took 20.
#include "common.hpp"
#include "row_scatter.hpp"

namespace {

constexpr int WPB = 4;        // waves per workgroup
constexpr int MAX_NNK = 16;

__device__ inline void rot_passive(const float* q, float vx, float vy, float vz, float& ox, float& oy,
                                   float& oz) {
  const float w = q[0], x = -q[1], y = -q[2], z = -q[3];
  const float tx = 2.f * (y * vz - z * vy), ty = 2.f * (z * vx - x * vz), tz = 2.f * (x * vy - y * vx);
  ox = vx + w * tx + (y * tz - z * ty);
  oy = vy + w * ty + (z * tx - x * tz);
  oz = vz + w * tz + (x * ty - y * tx);
}

__global__ __launch_bounds__(64 * WPB) void sdf_gather_kernel(
    int F, int weighted_first, float sdf_scale, const float* __restrict__ features,
    const float* __restrict__ points, const float* __restrict__ orientations, int after_pgo,
    const float* __restrict__ queries, long long B, int nnk, const long long* __restrict__ idx,
    const float* __restrict__ wgt, const float* __restrict__ dL_dsdf, float* __restrict__ X,
    float* __restrict__ gY, unsigned* __restrict__ keys, unsigned* __restrict__ vals,
    float* __restrict__ pair_w, unsigned invalid_key) {
  __shared__ long long sIdx[WPB][MAX_NNK];
  __shared__ float sW[WPB][MAX_NNK];
  __shared__ float sN[WPB][MAX_NNK][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int IN = F + 3;
  const long long nwaves = (long long)gridDim.x * WPB;
  for (long long q = (long long)blockIdx.x * WPB + wave; q < B; q += nwaves) {
    const float qx = queries[3 * q], qy = queries[3 * q + 1], qz = queries[3 * q + 2];
    const float gS = dL_dsdf[q] * sdf_scale;
    if (lane < nnk) {
      const long long id = idx[q * nnk + lane];
      const float w = wgt[q * nnk + lane];
      sIdx[wave][lane] = id;
      sW[wave][lane] = w;
      float nx = 0.f, ny = 0.f, nz = 0.f;
      if (id >= 0) {
        const float vx = qx - points[3 * id], vy = qy - points[3 * id + 1], vz = qz - points[3 * id + 2];
        nx = vx; ny = vy; nz = vz;
        if (after_pgo) rot_passive(orientations + 4 * id, vx, vy, vz, nx, ny, nz);
      }
      sN[wave][lane][0] = nx; sN[wave][lane][1] = ny; sN[wave][lane][2] = nz;
      const long long pr = q * nnk + lane;
      keys[pr] = id >= 0 ? (unsigned)id : invalid_key;
      if (weighted_first) {
        vals[pr] = (unsigned)q;        // source row = the query's single weighted row
        pair_w[pr] = w;
      } else {
        vals[pr] = (unsigned)pr;
        pair_w[pr] = 1.0f;
        gY[pr] = gS * w;
      }
    }
    if (weighted_first && lane == 0) gY[q] = gS;
    __builtin_amdgcn_wave_barrier();
    if (weighted_first) {
      for (int i = lane; i < IN; i += 64) {
        float v = 0.f;
        for (int mm = 0; mm < nnk; ++mm) {
          const long long id = sIdx[wave][mm];
          const float e = i < F ? (id >= 0 ? features[id * F + i] : 0.f) : sN[wave][mm][i - F];
          v = fmaf(sW[wave][mm], e, v);
        }
        X[(size_t)q * IN + i] = v;
      }
    } else {
      for (int e = lane; e < nnk * IN; e += 64) {
        const int mm = e / IN, i = e - mm * IN;
        const long long id = sIdx[wave][mm];
        X[((size_t)q * nnk + mm) * IN + i] = i < F ? (id >= 0 ? features[id * F + i] : 0.f) : sN[wave][mm][i - F];
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

struct Scratch {
  float *X, *gX, *gY, *pair_w;
  unsigned *keys, *src_row;
  void* mlp;
  void* rows_plan;
  size_t total;
};

size_t au(size_t v) { return (v + 255) / 256 * 256; }

Scratch carve(void* base, int64_t B, int nnk, int F, int H, int64_t rows) {
  Scratch s;
  const size_t n = (size_t)(B > 0 ? B : 1) * nnk;
  const int IN = F + 3;
  char* p = reinterpret_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t bytes) { char* r = p ? p + off : nullptr; off = au(off + bytes); return r; };
  s.X = (float*)take(n * IN * sizeof(float));
  s.gX = (float*)take(n * IN * sizeof(float));
  s.gY = (float*)take(n * sizeof(float));
  s.pair_w = (float*)take(n * sizeof(float));
  s.keys = (unsigned*)take(n * 4);
  s.src_row = (unsigned*)take(n * 4);
  s.mlp = take(pings_mlp_backward_scratch_bytes(IN, H, 1));
  s.rows_plan = take(pings_rows::carve(nullptr, (int64_t)n, rows).total);
  s.total = off;
  return s;
}

}  // namespace

PINGS_API size_t pings_sdf_backward_scratch_bytes(int64_t B, int nn_k, int feat_dim, int hidden,
                                                  int64_t feature_rows) {
  if (nn_k <= 0 || feat_dim <= 0 || hidden <= 0 || feature_rows <= 0) return 0;
  return carve(nullptr, B, nn_k, feat_dim, hidden, feature_rows).total;
}

PINGS_API int pings_sdf_backward(const pings_sdf_decoder* dec, const float* features,
                                 int64_t feature_rows, const float* points,
                                 const float* orientations, int32_t after_pgo, const float* queries,
                                 int64_t B, int nn_k, const int64_t* idx, const float* w,
                                 const float* dL_dsdf, void* scratch, float* dL_dfeatures,
                                 float* dL_dW1, float* dL_db1, float* dL_dW2, float* dL_db2,
                                 void* stream) {
  PINGS_ARG_CHECK(dec && dec->W1 && dec->b1 && dec->W2 && dec->b2, "null decoder");
  PINGS_ARG_CHECK(dec->hidden > 0 && dec->hidden <= 64 && dec->hidden % 32 == 0, "hidden must be 32 or 64");
  PINGS_ARG_CHECK(dec->feat_dim > 0 && dec->feat_dim <= 61, "feature dim must be <= 61");
  PINGS_ARG_CHECK(nn_k > 0 && nn_k <= MAX_NNK, "nn_k must be in 1..16");
  PINGS_ARG_CHECK(feature_rows > 0 && feature_rows < 0xFFFFFFFFLL, "feature_rows out of range");
  PINGS_ARG_CHECK(scratch && dL_dfeatures && dL_dW1 && dL_db1 && dL_dW2 && dL_db2, "null output");
  PINGS_ARG_CHECK(!after_pgo || orientations, "after_pgo needs orientations");
  hipStream_t st = pings::as_stream(stream);
  const int F = dec->feat_dim, H = dec->hidden, IN = F + 3;
  PINGS_ARG_CHECK(B == 0 || (features && points && queries && idx && w && dL_dsdf), "null pointer");
  PINGS_ARG_CHECK((int64_t)B * nn_k < 0x7FFFFFFFLL, "too many (query, neighbour) pairs");
  Scratch s = carve(scratch, B, nn_k, F, H, feature_rows);
  const long long n = (long long)B * nn_k;               // (query, neighbour) pairs
  const long long nrows = dec->weighted_first ? B : n;    // MLP rows
  const unsigned invalid_key = (unsigned)feature_rows;    // sorts behind every real destination row
  if (B > 0) {
    pings::prof::Scope ps("sdf_bwd_gather", st);
    const long long want = (B + WPB - 1) / WPB;
    const int grid = (int)(want < 8192 ? want : 8192);
    hipLaunchKernelGGL(sdf_gather_kernel, dim3(grid), dim3(64 * WPB), 0, st, F, (int)dec->weighted_first,
                       dec->sdf_scale, features, points, orientations, (int)after_pgo, queries, (long long)B,
                       nn_k, (const long long*)idx, w, dL_dsdf, s.X, s.gY, s.keys, s.src_row, s.pair_w,
                       invalid_key);
    PINGS_LAUNCH_CHECK();
  }
  // decoder backward on the gathered rows (W2 is [1,H] in the decoder struct = [OUT=1,H] of the MLP)
  if (int e = pings_mlp_backward(s.X, s.gY, nrows, IN, H, 1, dec->W1, dec->b1, dec->W2, s.mlp, s.gX, dL_dW1,
                                 dL_db1, dL_dW2, dL_db2, stream))
    return e;
  // scatter-add of the feature-gradient columns of gX into the table; every row of dL_dfeatures is written
  pings::prof::Scope ps("sdf_bwd_scatter", st);
  pings_rows::Plan plan = pings_rows::carve(s.rows_plan, n, feature_rows);
  if (int e = pings_rows::build(plan, s.keys, n, feature_rows, st)) return e;
  if (int e = pings_rows::gather_sum(plan, feature_rows, F, s.gX, IN, s.src_row, s.pair_w, dL_dfeatures, st)) return e;
  return PINGS_OK;
}
