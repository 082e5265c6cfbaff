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
//   radix sort + seg_sum_kernel
//                       every destination's run of source rows is summed in sorted order by a 32-lane group
//                       and stored once: the scatter-add of the feature gradient without atomics, bitwise
//                       reproducible (the reference's index_put / scatter_add backward is not).
#include <hipcub/hipcub.hpp>

#include "common.hpp"

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

// A wave takes 64 consecutive SORTED POSITIONS: each lane tests whether its position heads a run (first position,
// or a key different from its predecessor's), and the wave's two 32-lane halves then work through the heads found
// (ballot), one run each at a time: the half sums the run's source rows (first F of `ld` floats, times the pair
// weight) in sorted order and stores the destination row once.  No list of run starts is built — a compaction
// through one global cursor costs ~12 ns per returning atomic on this part (140 us for 0.5 M runs) — and no thread is
// launched just to find out that it is not a head.  Pair weights are addressed by the ORIGINAL pair id
// (`pair_sorted`).
__global__ __launch_bounds__(256) void seg_sum_kernel(const unsigned* __restrict__ keys,
                                                       const unsigned* __restrict__ pair_sorted, long long n,
                                                       unsigned invalid_key, int F, int ld,
                                                       const unsigned* __restrict__ src_row,
                                                       const float* __restrict__ pair_w,
                                                       const float* __restrict__ rows, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, c = lane & 31, half = lane >> 5;
  const long long base = ((long long)blockIdx.x * blockDim.x + threadIdx.x) & ~63LL;
  const long long mine = base + lane;
  // every lane resolves ITS position (key -> pair -> source row, weight) up front: one dependent-load chain for the
  // whole wave instead of one per summed row; the run loops below fetch these through cross-lane reads
  unsigned kme = invalid_key, srme = 0u;
  float wme = 0.f;
  bool head = false;
  if (mine < n) {
    kme = keys[mine];
    head = kme != invalid_key && (mine == 0 || keys[mine - 1] != kme);
    const unsigned pr = pair_sorted[mine];
    srme = src_row[pr];
    wme = pair_w[pr];
  }
  unsigned long long m = __ballot(head);
  while (m) {
    // the two lowest heads: half 0 takes the first, half 1 the second (if any)
    const int h0 = __ffsll((long long)m) - 1;
    m &= m - 1;
    int h1 = -1;
    if (m) { h1 = __ffsll((long long)m) - 1; m &= m - 1; }
    // run keys and lengths inside this wave's 64 positions (sorted: the lanes holding a key are contiguous from its
    // head); everything wave-uniform, so that the cross-lane reads below run with all lanes enabled
    const unsigned k0 = (unsigned)__builtin_amdgcn_readlane((int)kme, h0);
    const unsigned k1 = h1 >= 0 ? (unsigned)__builtin_amdgcn_readlane((int)kme, h1) : invalid_key;
    const int len0 = __popcll(__ballot(kme == k0)), len1 = h1 >= 0 ? __popcll(__ballot(kme == k1)) : 0;
    const int hp = half == 0 ? h0 : (h1 < 0 ? h0 : h1);
    const int len = half == 0 ? len0 : len1;
    const bool active = half == 0 || h1 >= 0;
    const unsigned k = half == 0 ? k0 : k1;
    float a0 = 0.f, a1 = 0.f;
    const int steps = len0 > len1 ? len0 : len1;
    for (int t = 0; t < steps; ++t) {
      const int q = min(hp + t, 63);
      const size_t r = (size_t)(unsigned)__shfl((int)srme, q, 64) * ld;
      const float w = __shfl(wme, q, 64);   // both cross-lane reads with every lane enabled
      if (t < len) {
        if (c < F) a0 = fmaf(w, rows[r + c], a0);
        if (c + 32 < F) a1 = fmaf(w, rows[r + c + 32], a1);
      }
    }
    if (active && hp + len == 64) {   // the run may continue in the next wave's positions
      for (long long p = base + 64; p < n && keys[p] == k; ++p) {
        const unsigned pr = pair_sorted[p];
        const size_t r = (size_t)src_row[pr] * ld;
        const float w = pair_w[pr];
        if (c < F) a0 = fmaf(w, rows[r + c], a0);
        if (c + 32 < F) a1 = fmaf(w, rows[r + c + 32], a1);
      }
    }
    if (active) {
      if (c < F) out[(size_t)k * F + c] = a0;
      if (c + 32 < F) out[(size_t)k * F + c + 32] = a1;
    }
  }
}

__global__ __launch_bounds__(256) void iota_kernel(unsigned* __restrict__ v, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = (unsigned)i;
}

struct Scratch {
  float *X, *gX, *gY, *pair_w;
  unsigned *keys, *src_row, *pair_id, *keys_s, *pair_s, *run_start, *run_count;
  void* mlp;
  char* temp;
  size_t temp_bytes, total;
};

size_t au(size_t v) { return (v + 255) / 256 * 256; }

Scratch carve(void* base, int64_t B, int nnk, int F, int H) {
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
  s.pair_id = (unsigned*)take(n * 4);
  s.keys_s = (unsigned*)take(n * 4);
  s.pair_s = (unsigned*)take(n * 4);
  s.run_start = (unsigned*)take(n * 4);
  s.run_count = (unsigned*)take(256);
  s.mlp = take(pings_mlp_backward_scratch_bytes(IN, H, 1));
  size_t tb = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tb, (unsigned*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr,
                                           (unsigned*)nullptr, (int)n, 0, 32);
  s.temp_bytes = au(tb) + 256;
  s.temp = take(s.temp_bytes);
  s.total = off;
  return s;
}

}  // namespace

PINGS_API size_t pings_sdf_backward_scratch_bytes(int64_t B, int nn_k, int feat_dim, int hidden) {
  if (nn_k <= 0 || feat_dim <= 0 || hidden <= 0) return 0;
  return carve(nullptr, B, nn_k, feat_dim, hidden).total;
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
  {
    pings::prof::Scope ps("sdf_bwd_memset", st);
    PINGS_HIP_CHECK(hipMemsetAsync(dL_dfeatures, 0, sizeof(float) * (size_t)feature_rows * F, st));
  }
  PINGS_ARG_CHECK(B == 0 || (features && points && queries && idx && w && dL_dsdf), "null pointer");
  PINGS_ARG_CHECK((int64_t)B * nn_k < 0x7FFFFFFFLL, "too many (query, neighbour) pairs");
  Scratch s = carve(scratch, B, nn_k, F, H);
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
  if (B == 0) return PINGS_OK;
  {
    int bits = 1;
    while ((1LL << bits) <= feature_rows) ++bits;  // keys are in [0, feature_rows]
    size_t tb = s.temp_bytes;
    {
      pings::prof::Scope ps("sdf_bwd_sort", st);
      hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s.pair_id, n);
      PINGS_LAUNCH_CHECK();
      PINGS_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(s.temp, tb, s.keys, s.keys_s, s.pair_id, s.pair_s, (int)n,
                                                         0, bits, st));
    }
    pings::prof::Scope ps("sdf_bwd_segsum", st);
    hipLaunchKernelGGL(seg_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s.keys_s, s.pair_s, n,
                       invalid_key, F, IN, s.src_row, s.pair_w, s.gX, dL_dfeatures);
    PINGS_LAUNCH_CHECK();
  }
  return PINGS_OK;
}
