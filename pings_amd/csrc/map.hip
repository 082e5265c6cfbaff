// Neural-point map maintenance on the device, for gfx950.
//
// Restates, bit for bit in every index / mask / timestamp / table entry, what the reference does once per frame in
// plain torch (model/neural_gaussians.py:214-494, utils/tools.py:924-967; semantics pinned by oracle/map_cpu.py
// against vectors generated from the reference):
//   voxel down-sampling   one representative per occupied voxel: the point closest to the voxel centre in 1000
//                         distance bins, ties by index; order = ascending linear voxel id.  Two reductions
//                         (bounds, largest distance), key build, radix sort by voxel id (rocPRIM), reduce-by-key
//                         (min of packed {bin, index}).  The reference does unique + scatter_reduce(amin).
//   update                hash lookup of the representatives, decision "new neural point" (empty slot / hash
//                         collision farther than sqrt(3) voxels / outside the travel-distance window), colour refresh
//                         of existing points, compaction (scan) to the appended rows, hash-table insert.
//                         Duplicate targets (two samples -> one slot / one point) resolve to the LAST sample, the
//                         reference's CPU semantics, through an atomicMax on {sample+1, value} pairs — deterministic.
//   reset_local_map       travel-distance window (with the "fewer than 100 -> take all" rule), local / surrounding
//                         radius masks, compaction, global2local (non-local entries are 1, not -1: the reference's
//                         full_like(bool, -1) quirk), row gathers of the per-point arrays.
//   assign_local_to_global  row scatters.
// All kernels are HBM-stream / gather bound: integer and byte work, nothing here is shaped for the matrix cores.
#include <hipcub/hipcub.hpp>

#include <algorithm>

#include "common.hpp"

namespace {

using i64 = long long;
constexpr int kQuant = 1000;  // distance bins of voxel_down_sample_torch (utils/tools.py:937)

__host__ __device__ inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

// ------------------------------------------------------------------ voxel down-sampling
struct Bounds {
  float pmin[3];  // min of the points per axis
  float gmax[3];  // max of floor(p / voxel) per axis
  float dmax;     // largest distance to the voxel centre
};

__device__ inline float cell_dist(const float* __restrict__ p, float voxel, float g[3]) {
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    g[a] = floorf(p[a] / voxel);                      // torch: points / voxel_size (true fp32 division)
    const float c = (g[a] + 0.5f) * voxel;
    const float d = p[a] - c;
    s = a == 0 ? d * d : s + d * d;                   // ((dx^2 + dy^2) + dz^2)
  }
  return sqrtf(s);                                    // ** 0.5 == sqrt
}

__global__ __launch_bounds__(256) void vds_bounds_kernel(const float* __restrict__ pts, i64 N, float voxel,
                                                         Bounds* __restrict__ partial,
                                                         const float* __restrict__ value = nullptr) {
  __shared__ float red[7][4];
  float v[7] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY};
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (i64)gridDim.x * blockDim.x) {
    float g[3];
    float d = cell_dist(pts + 3 * i, voxel, g);
    if (value) d = value[i];                          // voxel_down_sample_min_value_torch: the caller's value
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      v[a] = fminf(v[a], pts[3 * i + a]);
      v[3 + a] = fmaxf(v[3 + a], g[a]);
    }
    v[6] = fmaxf(v[6], d);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    float x = v[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float y = __shfl_xor(x, off, 64);
      x = k < 3 ? fminf(x, y) : fmaxf(x, y);
    }
    if (lane == 0) red[k][wave] = x;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Bounds b;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      float x = red[k][0];
      for (int w = 1; w < 4; ++w) x = k < 3 ? fminf(x, red[k][w]) : fmaxf(x, red[k][w]);
      if (k < 3) b.pmin[k] = x; else if (k < 6) b.gmax[k - 3] = x; else b.dmax = x;
    }
    partial[blockIdx.x] = b;
  }
}

struct GridInfo {
  i64 off[3];
  i64 vsize;
  float dmax;
};

__global__ void vds_finish_bounds_kernel(const Bounds* __restrict__ partial, int nblocks, float voxel,
                                         GridInfo* __restrict__ info) {
  Bounds b = partial[0];
  for (int i = 1; i < nblocks; ++i) {
    const Bounds c = partial[i];
    for (int a = 0; a < 3; ++a) {
      b.pmin[a] = fminf(b.pmin[a], c.pmin[a]);
      b.gmax[a] = fmaxf(b.gmax[a], c.gmax[a]);
    }
    b.dmax = fmaxf(b.dmax, c.dmax);
  }
  GridInfo gi;
  i64 vs = 0;
  for (int a = 0; a < 3; ++a) {
    gi.off[a] = (i64)floorf(b.pmin[a] / voxel);       // floor(points.min(0) / voxel).long()
    const i64 e = (i64)b.gmax[a] - gi.off[a];
    vs = a == 0 ? e : (e > vs ? e : vs);              // grid.max(): one scalar over all three axes
  }
  gi.vsize = vs;
  gi.dmax = b.dmax;
  *info = gi;
}

__global__ __launch_bounds__(256) void vds_key_kernel(const float* __restrict__ pts, i64 N, float voxel,
                                                      const GridInfo* __restrict__ info, i64* __restrict__ key,
                                                      unsigned long long* __restrict__ val,
                                                      const float* __restrict__ value = nullptr) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const GridInfo gi = *info;
  float g[3];
  float d = cell_dist(pts + 3 * i, voxel, g);
  if (value) d = value[i];
  const i64 gx = (i64)g[0] - gi.off[0], gy = (i64)g[1] - gi.off[1], gz = (i64)g[2] - gi.off[2];
  key[i] = gx + gy * gi.vsize + gz * gi.vsize * gi.vsize;        // the reference's (aliasing) linear voxel id
  // (dist / dist.max() * 999).long(); an all-zero value vector (0 / 0 in the reference, utils/tools.py:996) bins to 0
  const i64 bin = gi.dmax > 0.f ? (i64)(d / gi.dmax * (float)(kQuant - 1)) : 0;
  val[i] = ((unsigned long long)bin << 32) | (unsigned long long)(uint32_t)i;
}

__global__ void vds_unpack_kernel(const unsigned long long* __restrict__ agg, const int* __restrict__ nruns,
                                  i64* __restrict__ out) {
  const i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < (i64)*nruns) out[r] = (i64)(agg[r] & 0xFFFFFFFFull);
}

struct VdsScratch {
  Bounds* partial;
  GridInfo* info;
  i64 *key, *key_sorted, *uniq;
  unsigned long long *val, *val_sorted, *agg;
  int* nruns;
  void* temp;
  size_t temp_bytes, total;
};
constexpr int kBoundBlocks = 512;

VdsScratch carve_vds(void* base, i64 N) {
  char* p = reinterpret_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t bytes) { void* r = p ? p + off : nullptr; off += up256(bytes); return r; };
  VdsScratch s;
  const size_t n = (size_t)(N > 0 ? N : 1);
  s.partial = (Bounds*)take(sizeof(Bounds) * kBoundBlocks);
  s.info = (GridInfo*)take(sizeof(GridInfo));
  s.key = (i64*)take(8 * n);
  s.key_sorted = (i64*)take(8 * n);
  s.uniq = (i64*)take(8 * n);
  s.val = (unsigned long long*)take(8 * n);
  s.val_sorted = (unsigned long long*)take(8 * n);
  s.agg = (unsigned long long*)take(8 * n);
  s.nruns = (int*)take(sizeof(int));
  size_t a = 0, b = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, (i64*)nullptr, (i64*)nullptr, (unsigned long long*)nullptr,
                                           (unsigned long long*)nullptr, (int)n);
  (void)hipcub::DeviceReduce::ReduceByKey(nullptr, b, (i64*)nullptr, (i64*)nullptr, (unsigned long long*)nullptr,
                                          (unsigned long long*)nullptr, (int*)nullptr, hipcub::Min(), (int)n);
  s.temp_bytes = up256(a > b ? a : b);
  s.temp = take(s.temp_bytes);
  s.total = off;
  return s;
}

// ------------------------------------------------------------------ update
__device__ inline i64 hash_slot_of(const float* __restrict__ p, float res, i64 S) {
  const i64 P0 = 73856093, P1 = 19349669, P2 = 83492791;    // neural_gaussians.py:80-82
  const i64 gx = (i64)floorf(p[0] / res), gy = (i64)floorf(p[1] / res), gz = (i64)floorf(p[2] / res);
  const i64 h = (gx * P0 + gy * P1 + gz * P2) % S;          // fmod: sign of the dividend
  return h < 0 ? h + S : h;                                  // table[h] with python wrap-around
}

struct UpdArgs {
  i64 M, Np, S;
  float res, thr, diff_travel;
  int cur_ts, temporal, is_reliable;
  const float *sp, *sc;
  i64* table;
  const float* travel;
  float *neural_points, *orient, *cert, *colors;
  int32_t *ts_create, *ts_update;
  uint8_t *free_mask, *valid_gs, *valid_color;
  i64 *slot, *hidx;
  int32_t *flag, *pos, *winner;
  uint8_t* upd_out;
};

__global__ __launch_bounds__(256) void upd_classify_kernel(UpdArgs a) {
  const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.M) return;
  const float* p = a.sp + 3 * k;
  const i64 slot = hash_slot_of(p, a.res, a.S);
  const i64 h = a.table[slot];
  a.slot[k] = slot;
  a.hidx[k] = h;
  bool upd = true;
  if (a.Np > 0) {
    const i64 hi = h < 0 ? a.Np + h : h;                     // neural_points[-1]: python indexing
    const float dx = a.neural_points[3 * hi] - p[0], dy = a.neural_points[3 * hi + 1] - p[1],
                dz = a.neural_points[3 * hi + 2] - p[2];
    const float d2 = (dx * dx + dy * dy) + dz * dz;
    upd = (h == -1) || (d2 > a.thr);
    if (a.sc && h > -1 && a.valid_color[h] == 0 && a.sc[3 * k] >= 0.0f)
      atomicMax(&a.winner[h], (int32_t)k + 1);               // colour refresh: the last sample wins
    if (a.temporal) {
      const float dt = a.travel[a.cur_ts] - a.travel[a.ts_update[hi]];
      upd = upd || (dt > a.diff_travel);
    }
  }
  a.flag[k] = upd ? 1 : 0;
}

__global__ __launch_bounds__(256) void upd_color_kernel(UpdArgs a) {
  const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.M || a.Np == 0 || !a.sc) return;
  const i64 h = a.hidx[k];
  if (h > -1 && a.winner[h] == (int32_t)k + 1) {             // only set for samples that passed the colour test
    a.colors[3 * h] = a.sc[3 * k];
    a.colors[3 * h + 1] = a.sc[3 * k + 1];
    a.colors[3 * h + 2] = a.sc[3 * k + 2];
    a.valid_color[h] = 1;
  }
}

__global__ __launch_bounds__(256) void upd_commit_kernel(UpdArgs a) {
  const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.M) return;
  const bool upd = a.flag[k] != 0;
  if (a.upd_out) a.upd_out[k] = upd ? 1 : 0;
  i64 value = a.hidx[k];                                     // cur_pt_idx of a sample that is not inserted
  if (upd) {
    const i64 n = a.Np + a.pos[k];
    value = n;
    a.neural_points[3 * n] = a.sp[3 * k];
    a.neural_points[3 * n + 1] = a.sp[3 * k + 1];
    a.neural_points[3 * n + 2] = a.sp[3 * k + 2];
    a.orient[4 * n] = 1.f; a.orient[4 * n + 1] = 0.f; a.orient[4 * n + 2] = 0.f; a.orient[4 * n + 3] = 0.f;
    a.ts_create[n] = a.cur_ts;
    a.ts_update[n] = a.cur_ts;
    a.cert[n] = 0.f;
    a.free_mask[n] = a.is_reliable ? 0 : 1;
    a.valid_gs[n] = 1;
    if (a.sc) {
      a.colors[3 * n] = a.sc[3 * k];
      a.colors[3 * n + 1] = a.sc[3 * k + 1];
      a.colors[3 * n + 2] = a.sc[3 * k + 2];
      a.valid_color[n] = a.sc[3 * k] >= 0.0f ? 1 : 0;
    } else {
      a.valid_color[n] = 1;
    }
  }
  // buffer_pt_index[hash_value] = cur_pt_idx with duplicate slots: the last sample wins.  Every sample offers
  // {k + 1, value}; plain table contents (>= -1, high word 0 or ~0) lose against any offer.
  const i64 packed = (i64)(((unsigned long long)(k + 1) << 32) | (unsigned long long)(uint32_t)value);
  atomicMax(reinterpret_cast<long long*>(&a.table[a.slot[k]]), (long long)packed);
}

__global__ __launch_bounds__(256) void upd_unpack_kernel(UpdArgs a) {
  const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.M) return;
  const i64 slot = a.slot[k];
  const i64 v = a.table[slot];
  const i64 hi = v >> 32;
  if (hi >= 1) a.table[slot] = (i64)(int32_t)(uint32_t)(v & 0xFFFFFFFFll);  // idempotent: same value from every writer
}

// ------------------------------------------------------------------ reset_local_map
constexpr int kCountShards = 256;
struct RstArgs {
  i64 Np;
  const float* pts;
  const int32_t *ts_create, *ts_update;
  const float* travel;
  const float* sensor;
  int cur_ts, temporal, use_mid_ts, use_travel, diff_ts_local, range_2d;
  float diff_travel, local_r2, sur_r2;
  uint8_t* tflag;
  int* tcount;
  int32_t* lflag;
  int32_t* lpos;
  uint8_t *local_mask, *sur_mask;
  i64 *g2l, *lidx;
  i64* nlocal_dev;
};

__global__ __launch_bounds__(256) void rst_time_kernel(RstArgs a) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  bool t = false;
  if (i < a.Np) {
    // ((create + update) / 2).int(): int32 sum, true division in fp32, truncation
    const int32_t ts = a.use_mid_ts ? (int32_t)((float)(a.ts_create[i] + a.ts_update[i]) / 2.0f) : a.ts_create[i];
    if (a.use_travel) t = fabsf(a.travel[a.cur_ts] - a.travel[ts]) < a.diff_travel;
    else t = abs(a.cur_ts - ts) < a.diff_ts_local;
    a.tflag[i] = t ? 1 : 0;
  }
  // count of points inside the window: sharded (tens of thousands of waves adding to one word serialise)
  const unsigned long long b = __ballot(t);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(a.tcount + 1 + (blockIdx.x & (kCountShards - 1)), __popcll(b));
}

__global__ __launch_bounds__(256) void rst_total_kernel(int* __restrict__ tcount) {
  __shared__ int red[256];
  red[threadIdx.x] = tcount[1 + threadIdx.x];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) tcount[0] = red[0];
}

__global__ __launch_bounds__(256) void rst_mask_kernel(RstArgs a) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > a.Np) return;
  if (i == a.Np) {                                             // padding element: always True (:433-436,:456-459)
    a.local_mask[i] = 1;
    a.sur_mask[i] = 1;
    return;
  }
  bool t = true;
  if (a.temporal && *a.tcount >= 100) t = a.tflag[i] != 0;     // < 100 points in the window: take all (:410-411)
  const float dx = a.pts[3 * i] - a.sensor[0], dy = a.pts[3 * i + 1] - a.sensor[1], dz = a.pts[3 * i + 2] - a.sensor[2];
  const float d2 = a.range_2d ? (dx * dx + dy * dy) : ((dx * dx + dy * dy) + dz * dz);
  const bool in_local = d2 < a.local_r2;
  const bool loc = t && in_local;
  const bool sur = t && !in_local && (d2 < a.sur_r2);
  a.local_mask[i] = loc ? 1 : 0;
  a.sur_mask[i] = sur ? 1 : 0;
  a.lflag[i] = loc ? 1 : 0;
}

__global__ __launch_bounds__(256) void rst_index_kernel(RstArgs a) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > a.Np) return;
  if (i == a.Np) {
    const i64 n = a.Np > 0 ? (i64)a.lpos[a.Np - 1] + a.lflag[a.Np - 1] : 0;
    a.g2l[i] = -1;                                             // global2local[-1] = -1
    a.lidx[n] = a.Np;                                          // the padding row rides along with the features
    *a.nlocal_dev = n;
    return;
  }
  if (a.lflag[i]) {
    a.g2l[i] = a.lpos[i];
    a.lidx[a.lpos[i]] = i;
  } else {
    a.g2l[i] = 1;                                              // torch.full_like(bool_mask, -1).long() == 1
  }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const uint32_t* __restrict__ src, i64 row_words,
                                                          const i64* __restrict__ idx, i64 n, uint32_t* __restrict__ dst) {
  const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * row_words) return;
  const i64 r = e / row_words, c = e - r * row_words;
  dst[e] = src[idx[r] * row_words + c];
}
__global__ __launch_bounds__(256) void gather_bytes_kernel(const uint8_t* __restrict__ src, i64 row_bytes,
                                                           const i64* __restrict__ idx, i64 n, uint8_t* __restrict__ dst) {
  const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * row_bytes) return;
  const i64 r = e / row_bytes, c = e - r * row_bytes;
  dst[e] = src[idx[r] * row_bytes + c];
}
__global__ __launch_bounds__(256) void scatter_rows_kernel(const uint32_t* __restrict__ src, i64 row_words,
                                                           const i64* __restrict__ idx, i64 n, uint32_t* __restrict__ dst) {
  const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * row_words) return;
  const i64 r = e / row_words, c = e - r * row_words;
  dst[idx[r] * row_words + c] = src[e];
}
__global__ __launch_bounds__(256) void scatter_bytes_kernel(const uint8_t* __restrict__ src, i64 row_bytes,
                                                            const i64* __restrict__ idx, i64 n, uint8_t* __restrict__ dst) {
  const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * row_bytes) return;
  const i64 r = e / row_bytes, c = e - r * row_bytes;
  dst[idx[r] * row_bytes + c] = src[e];
}

inline unsigned blocks_for(i64 n) { return (unsigned)((n + 255) / 256 > 0 ? (n + 255) / 256 : 1); }

// ------------------------------------------------------------------ loop-closure maintenance
// prune_map (:871-909): prune = |travel[cur] - travel[ts_update]| > diff_travel  and  certainty < threshold
// A timestamp outside [-T, T) is an IndexError in the reference; here it raises the caller's flag word (read with the
// count the caller reads anyway) and the point is kept / left where it is — never an out-of-bounds read.
__global__ __launch_bounds__(256) void prune_mask_kernel(i64 N, const float* __restrict__ travel, i64 T, int cur_ts,
                                                         const int32_t* __restrict__ ts_update,
                                                         const float* __restrict__ cert, float diff_travel, float thre,
                                                         uint8_t* __restrict__ prune, int32_t* __restrict__ oob) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  i64 ts = ts_update[i];
  if (ts < 0) ts += T;                                       // python indexing
  if (ts < 0 || ts >= T) {
    if (oob) atomicOr(oob, 1);
    prune[i] = 0;
    return;
  }
  const float d = fabsf(travel[cur_ts] - travel[ts]);
  prune[i] = (d > diff_travel && cert[i] < thre) ? 1 : 0;
}

// adjust_map (:911-937): p <- R p + t (fp32, the pose cast to the points' dtype first), q <- quat(R) (x) q with the
// rotation quaternion and the product evaluated in the POSE's dtype (the reference promotes: float64 poses give a
// float64 product that is cast back), pose = pose_diff[ts] with ts = ((create + update) / 2).int() or create.
template <typename PT>
__global__ __launch_bounds__(256) void adjust_kernel(i64 N, float* __restrict__ pts, float* __restrict__ quat,
                                                     const int32_t* __restrict__ ts_create,
                                                     const int32_t* __restrict__ ts_update, int use_mid_ts,
                                                     const PT* __restrict__ pose, i64 T, int32_t* __restrict__ oob) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  i64 ts = use_mid_ts ? (i64)(int32_t)((float)(ts_create[i] + ts_update[i]) / 2.0f) : (i64)ts_create[i];
  if (ts < 0) ts += T;                                       // python indexing
  if (ts < 0 || ts >= T) {                                   // the reference's index error: flagged, point not moved
    if (oob) atomicOr(oob, 1);
    return;
  }
  const PT* M = pose + 16 * ts;
  const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float r0 = (float)M[4 * r], r1 = (float)M[4 * r + 1], r2 = (float)M[4 * r + 2], t = (float)M[4 * r + 3];
    pts[3 * i + r] = ((r0 * x + r1 * y) + r2 * z) + t;       // bmm row, then + translation
  }
  // rotmat_to_quat (utils/tools.py:754-773)
  const PT qw = sqrt((PT)1.0 + M[0] + M[5] + M[10]) / (PT)2.0;
  const PT qx = (M[9] - M[6]) / ((PT)4.0 * qw);
  const PT qy = (M[2] - M[8]) / ((PT)4.0 * qw);
  const PT qz = (M[4] - M[1]) / ((PT)4.0 * qw);
  // quat_multiply(diff, q) (utils/tools.py:811-829)
  const PT w2 = (PT)quat[4 * i], x2 = (PT)quat[4 * i + 1], y2 = (PT)quat[4 * i + 2], z2 = (PT)quat[4 * i + 3];
  quat[4 * i] = (float)(qw * w2 - qx * x2 - qy * y2 - qz * z2);
  quat[4 * i + 1] = (float)(qw * x2 + qx * w2 + qy * z2 - qz * y2);
  quat[4 * i + 2] = (float)(qw * y2 - qx * z2 + qy * w2 + qz * x2);
  quat[4 * i + 3] = (float)(qw * z2 + qx * y2 - qy * x2 + qz * w2);
}

// recreate_hash (:939-1010): buffer_pt_index[hash(points[value_j])] = value_j for j = 0..M-1, duplicates: the last j
// wins (the reference's CPU index_put_ outcome), as in upd_commit_kernel / upd_unpack_kernel.
__global__ __launch_bounds__(256) void rehash_commit_kernel(i64 M, const float* __restrict__ pts,
                                                            const i64* __restrict__ sample_idx, float res, i64 S,
                                                            i64* __restrict__ table, i64* __restrict__ slot_of) {
  const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= M) return;
  const i64 v = sample_idx ? sample_idx[j] : j;
  const i64 slot = hash_slot_of(pts + 3 * v, res, S);
  slot_of[j] = slot;
  const i64 packed = (i64)(((unsigned long long)(j + 1) << 32) | (unsigned long long)(uint32_t)v);
  atomicMax(reinterpret_cast<long long*>(&table[slot]), (long long)packed);
}
__global__ __launch_bounds__(256) void rehash_unpack_kernel(i64 M, const i64* __restrict__ slot_of, i64* __restrict__ table) {
  const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= M) return;
  const i64 v = table[slot_of[j]];
  if ((v >> 32) >= 1) table[slot_of[j]] = (i64)(int32_t)(uint32_t)(v & 0xFFFFFFFFll);
}

}  // namespace

PINGS_API int pings_map_prune_mask(int64_t N, const float* travel_dist, int64_t num_travel, int32_t cur_ts,
                                   const int32_t* point_ts_update, const float* point_certainties,
                                   float diff_travel_dist_local, float prune_certainty_thre, uint8_t* prune_mask,
                                   int32_t* out_of_range, void* stream) {
  PINGS_ARG_CHECK(N >= 0, "negative N");
  if (N == 0) return PINGS_OK;
  PINGS_ARG_CHECK(travel_dist && point_ts_update && point_certainties && prune_mask, "bad argument");
  PINGS_ARG_CHECK(num_travel > 0 && cur_ts >= 0 && cur_ts < num_travel, "cur_ts outside travel_dist");
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope sc("map_prune_mask", st);
  prune_mask_kernel<<<blocks_for(N), 256, 0, st>>>(N, travel_dist, num_travel, cur_ts, point_ts_update,
                                                   point_certainties, diff_travel_dist_local, prune_certainty_thre,
                                                   prune_mask, out_of_range);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}

PINGS_API int pings_map_adjust(int64_t N, float* neural_points, float* point_orientations, const int32_t* point_ts_create,
                               const int32_t* point_ts_update, int32_t use_mid_ts, const void* pose_diff,
                               int32_t pose_is_f64, int64_t num_poses, int32_t* out_of_range, void* stream) {
  PINGS_ARG_CHECK(N >= 0 && num_poses > 0, "bad sizes");
  if (N == 0) return PINGS_OK;
  PINGS_ARG_CHECK(neural_points && point_orientations && point_ts_create && pose_diff, "null pointer");
  PINGS_ARG_CHECK(!use_mid_ts || point_ts_update, "use_mid_ts needs point_ts_update");
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope sc("map_adjust", st);
  if (pose_is_f64)
    adjust_kernel<double><<<blocks_for(N), 256, 0, st>>>(N, neural_points, point_orientations, point_ts_create,
                                                         point_ts_update, use_mid_ts, (const double*)pose_diff, num_poses,
                                                         out_of_range);
  else
    adjust_kernel<float><<<blocks_for(N), 256, 0, st>>>(N, neural_points, point_orientations, point_ts_create,
                                                        point_ts_update, use_mid_ts, (const float*)pose_diff, num_poses,
                                                        out_of_range);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}

PINGS_API int pings_map_rehash(const float* neural_points, const int64_t* sample_idx, int64_t M, float resolution,
                               int64_t buffer_size, int64_t* table, int64_t* slot_scratch, void* stream) {
  PINGS_ARG_CHECK(M >= 0 && buffer_size > 0 && table && resolution > 0.f, "bad argument");
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope sc("map_rehash", st);
  PINGS_HIP_CHECK(hipMemsetAsync(table, 0xFF, sizeof(int64_t) * (size_t)buffer_size, st));   // every slot -1
  if (M == 0) return PINGS_OK;
  PINGS_ARG_CHECK(neural_points && slot_scratch && M < ((int64_t)1 << 31), "bad argument");
  rehash_commit_kernel<<<blocks_for(M), 256, 0, st>>>(M, neural_points, (const i64*)sample_idx, resolution, buffer_size,
                                                      (i64*)table, (i64*)slot_scratch);
  PINGS_LAUNCH_CHECK();
  rehash_unpack_kernel<<<blocks_for(M), 256, 0, st>>>(M, (const i64*)slot_scratch, (i64*)table);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}


PINGS_API size_t pings_voxel_downsample_scratch_bytes(int64_t N) { return carve_vds(nullptr, N).total; }

PINGS_API int pings_voxel_downsample(const float* points, int64_t N, float voxel_size, void* scratch,
                                     int64_t* sample_idx, int64_t* count, void* stream) {
  return pings_voxel_downsample_min_value(points, nullptr, N, voxel_size, scratch, sample_idx, count, stream);
}

PINGS_API int pings_voxel_downsample_min_value(const float* points, const float* value, int64_t N, float voxel_size,
                                               void* scratch, int64_t* sample_idx, int64_t* count, void* stream) {
  PINGS_ARG_CHECK(count != nullptr, "null count");
  *count = 0;
  if (N == 0) return PINGS_OK;
  PINGS_ARG_CHECK(N > 0 && N < ((int64_t)1 << 31), "point count out of range");
  PINGS_ARG_CHECK(points && scratch && sample_idx && voxel_size > 0.f, "bad argument");
  hipStream_t st = pings::as_stream(stream);
  VdsScratch s = carve_vds(scratch, N);
  pings::prof::Scope sc("voxel_downsample", st);
  const int nb = (int)std::min<i64>(kBoundBlocks, (N + 255) / 256);
  vds_bounds_kernel<<<nb, 256, 0, st>>>(points, N, voxel_size, s.partial, value);
  PINGS_LAUNCH_CHECK();
  vds_finish_bounds_kernel<<<1, 1, 0, st>>>(s.partial, nb, voxel_size, s.info);
  vds_key_kernel<<<blocks_for(N), 256, 0, st>>>(points, N, voxel_size, s.info, s.key, s.val, value);
  PINGS_LAUNCH_CHECK();
  size_t tb = s.temp_bytes;
  PINGS_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(s.temp, tb, s.key, s.key_sorted, s.val, s.val_sorted, (int)N, 0,
                                                     64, st));
  tb = s.temp_bytes;
  PINGS_HIP_CHECK(hipcub::DeviceReduce::ReduceByKey(s.temp, tb, s.key_sorted, s.uniq, s.val_sorted, s.agg, s.nruns,
                                                    hipcub::Min(), (int)N, st));
  vds_unpack_kernel<<<blocks_for(N), 256, 0, st>>>(s.agg, s.nruns, reinterpret_cast<i64*>(sample_idx));
  PINGS_LAUNCH_CHECK();
  uint32_t runs = 0;
  const uint32_t* src[1] = {reinterpret_cast<const uint32_t*>(s.nruns)};
  if (int e = pings::host_read_words(src, 1, &runs, st)) return e;
  *count = (int64_t)(int)runs;
  return PINGS_OK;
}

PINGS_API size_t pings_map_update_scratch_bytes(int64_t M, int64_t num_points) {
  const size_t m = (size_t)(M > 0 ? M : 1), np = (size_t)(num_points > 0 ? num_points : 1);
  size_t scan = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan, (int32_t*)nullptr, (int32_t*)nullptr, (int)m);
  return up256(8 * m) * 2 + up256(4 * m) * 2 + up256(4 * np) + up256(scan) + 256;
}

PINGS_API int pings_map_update(const float* sample_points, const float* sample_colors, int64_t M, float resolution,
                               int64_t buffer_size, int64_t* table, int64_t num_points, const float* travel_dist,
                               int cur_ts, float diff_travel_dist_local, int is_reliable, float* neural_points,
                               float* point_orientations, int32_t* point_ts_create, int32_t* point_ts_update,
                               float* point_certainties, uint8_t* free_gs_mask, uint8_t* valid_gs_mask,
                               float* point_colors, uint8_t* valid_color_mask, void* scratch, uint8_t* update_mask,
                               int64_t* num_new, void* stream) {
  PINGS_ARG_CHECK(num_new != nullptr, "null num_new");
  *num_new = 0;
  if (M == 0) return PINGS_OK;
  PINGS_ARG_CHECK(M > 0 && M < ((int64_t)1 << 31) && num_points >= 0 && buffer_size > 0, "bad sizes");
  PINGS_ARG_CHECK(sample_points && table && neural_points && point_orientations && point_ts_create &&
                      point_ts_update && point_certainties && free_gs_mask && valid_gs_mask && valid_color_mask &&
                      scratch, "null pointer");
  PINGS_ARG_CHECK(!sample_colors || point_colors, "sample colours without a colour array");
  hipStream_t st = pings::as_stream(stream);
  char* p = reinterpret_cast<char*>(scratch);
  size_t off = 0;
  auto take = [&](size_t bytes) { void* r = p + off; off += up256(bytes); return r; };
  UpdArgs a{};
  a.M = M; a.Np = num_points; a.S = buffer_size;
  a.res = resolution;
  a.thr = (float)(3.0 * (double)resolution * (double)resolution);   // 3 * res**2 (python float) cast to fp32
  a.diff_travel = diff_travel_dist_local;
  a.cur_ts = cur_ts; a.temporal = travel_dist != nullptr; a.is_reliable = is_reliable;
  a.sp = sample_points; a.sc = sample_colors; a.table = reinterpret_cast<i64*>(table); a.travel = travel_dist;
  a.neural_points = neural_points; a.orient = point_orientations; a.cert = point_certainties; a.colors = point_colors;
  a.ts_create = point_ts_create; a.ts_update = point_ts_update;
  a.free_mask = free_gs_mask; a.valid_gs = valid_gs_mask; a.valid_color = valid_color_mask;
  a.slot = (i64*)take(8 * (size_t)M);
  a.hidx = (i64*)take(8 * (size_t)M);
  a.flag = (int32_t*)take(4 * (size_t)M);
  a.pos = (int32_t*)take(4 * (size_t)M);
  a.winner = (int32_t*)take(4 * (size_t)(num_points > 0 ? num_points : 1));
  a.upd_out = update_mask;
  void* temp = p + off;
  size_t tb = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb, (int32_t*)nullptr, (int32_t*)nullptr, (int)M);
  pings::prof::Scope sc("map_update", st);
  if (num_points > 0 && sample_colors)
    PINGS_HIP_CHECK(hipMemsetAsync(a.winner, 0, 4 * (size_t)num_points, st));
  const unsigned nb = blocks_for(M);
  upd_classify_kernel<<<nb, 256, 0, st>>>(a);
  PINGS_LAUNCH_CHECK();
  upd_color_kernel<<<nb, 256, 0, st>>>(a);
  PINGS_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(temp, tb, a.flag, a.pos, (int)M, st));
  upd_commit_kernel<<<nb, 256, 0, st>>>(a);
  upd_unpack_kernel<<<nb, 256, 0, st>>>(a);
  PINGS_LAUNCH_CHECK();
  uint32_t last[2] = {0, 0};
  const uint32_t* src[2] = {reinterpret_cast<const uint32_t*>(a.pos + (M - 1)),
                            reinterpret_cast<const uint32_t*>(a.flag + (M - 1))};
  if (int e = pings::host_read_words(src, 2, last, st)) return e;
  *num_new = (int64_t)(int32_t)last[0] + (int32_t)last[1];
  return PINGS_OK;
}

PINGS_API size_t pings_map_reset_local_scratch_bytes(int64_t num_points) {
  const size_t n = (size_t)(num_points > 0 ? num_points : 1);
  size_t scan = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan, (int32_t*)nullptr, (int32_t*)nullptr, (int)n);
  return up256(n) + up256(4 * n) * 2 + up256(scan) + 4096;
}

PINGS_API int pings_map_reset_local(int64_t num_points, const float* neural_points, const int32_t* point_ts_create,
                                    const int32_t* point_ts_update, const float* travel_dist, int cur_ts,
                                    int use_mid_ts, int use_travel_dist, float diff_travel_dist_local,
                                    int diff_ts_local, const float* sensor_position, int range_filter_2d,
                                    float local_radius, float sorrounding_radius, void* scratch, uint8_t* local_mask,
                                    uint8_t* sorrounding_mask, int64_t* global2local, int64_t* local_idx,
                                    int64_t* num_local, void* stream) {
  PINGS_ARG_CHECK(num_local != nullptr, "null num_local");
  *num_local = 0;
  PINGS_ARG_CHECK(num_points >= 0 && num_points < ((int64_t)1 << 31), "point count out of range");
  PINGS_ARG_CHECK(sensor_position && scratch && local_mask && sorrounding_mask && global2local && local_idx,
                  "null pointer");
  PINGS_ARG_CHECK(num_points == 0 || (neural_points && point_ts_create && point_ts_update), "null map array");
  hipStream_t st = pings::as_stream(stream);
  char* p = reinterpret_cast<char*>(scratch);
  size_t off = 0;
  auto take = [&](size_t bytes) { void* r = p + off; off += up256(bytes); return r; };
  const size_t n = (size_t)(num_points > 0 ? num_points : 1);
  RstArgs a{};
  a.Np = num_points; a.pts = neural_points; a.ts_create = point_ts_create; a.ts_update = point_ts_update;
  a.travel = travel_dist; a.sensor = sensor_position; a.cur_ts = cur_ts;
  a.temporal = (travel_dist != nullptr) || !use_travel_dist ? 1 : 0;
  a.use_mid_ts = use_mid_ts; a.use_travel = use_travel_dist; a.diff_ts_local = diff_ts_local;
  a.range_2d = range_filter_2d; a.diff_travel = diff_travel_dist_local;
  a.local_r2 = (float)((double)local_radius * (double)local_radius);
  a.sur_r2 = (float)((double)sorrounding_radius * (double)sorrounding_radius);
  a.tflag = (uint8_t*)take(n);
  a.lflag = (int32_t*)take(4 * n);
  a.lpos = (int32_t*)take(4 * n);
  a.tcount = (int*)take(sizeof(int) * (1 + kCountShards));  // [0] total, [1..] shards
  a.nlocal_dev = (i64*)take(sizeof(i64));
  void* temp = p + off;
  size_t tb = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb, (int32_t*)nullptr, (int32_t*)nullptr, (int)n);
  a.local_mask = local_mask; a.sur_mask = sorrounding_mask; a.g2l = reinterpret_cast<i64*>(global2local); a.lidx = reinterpret_cast<i64*>(local_idx);
  pings::prof::Scope sc("map_reset_local", st);
  PINGS_HIP_CHECK(hipMemsetAsync(a.tcount, 0, sizeof(int) * (1 + kCountShards), st));
  if (a.temporal && num_points > 0) {
    PINGS_ARG_CHECK(!use_travel_dist || travel_dist, "travel-distance window without travel_dist");
    rst_time_kernel<<<blocks_for(num_points), 256, 0, st>>>(a);
    rst_total_kernel<<<1, 256, 0, st>>>(a.tcount);
    PINGS_LAUNCH_CHECK();
  }
  rst_mask_kernel<<<blocks_for(num_points + 1), 256, 0, st>>>(a);
  PINGS_LAUNCH_CHECK();
  if (num_points > 0)
    PINGS_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(temp, tb, a.lflag, a.lpos, (int)num_points, st));
  rst_index_kernel<<<blocks_for(num_points + 1), 256, 0, st>>>(a);
  PINGS_LAUNCH_CHECK();
  uint32_t nl[2] = {0, 0};
  const uint32_t* src[2] = {reinterpret_cast<const uint32_t*>(a.nlocal_dev),
                            reinterpret_cast<const uint32_t*>(a.nlocal_dev) + 1};
  if (int e = pings::host_read_words(src, 2, nl, st)) return e;
  *num_local = (int64_t)(((unsigned long long)nl[1] << 32) | nl[0]);
  return PINGS_OK;
}

PINGS_API int pings_gather_rows(const void* src, int64_t row_bytes, const int64_t* idx, int64_t n, void* dst,
                                void* stream) {
  PINGS_ARG_CHECK(n >= 0 && row_bytes > 0, "bad sizes");
  if (n == 0) return PINGS_OK;
  PINGS_ARG_CHECK(src && idx && dst, "null pointer");
  hipStream_t st = pings::as_stream(stream);
  if (row_bytes % 4 == 0 && ((uintptr_t)src % 4 == 0) && ((uintptr_t)dst % 4 == 0))
    gather_rows_kernel<<<blocks_for(n * (row_bytes / 4)), 256, 0, st>>>((const uint32_t*)src, row_bytes / 4,
                                                                       (const i64*)idx, n, (uint32_t*)dst);
  else
    gather_bytes_kernel<<<blocks_for(n * row_bytes), 256, 0, st>>>((const uint8_t*)src, row_bytes, (const i64*)idx, n,
                                                                   (uint8_t*)dst);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}

// ---------------------------------------------------------------- gather_local_data (model/neural_gaussians.py:1135-1173)
namespace {
struct MaskRows {
  i64* num;        // [1] rows selected
  uint32_t* out2;  // [2] {rows selected, mask[n - 1]} for the one read-back
  char* temp;
  size_t temp_bytes, total;
};
MaskRows carve_mask_rows(void* blob, i64 n) {
  char* base = reinterpret_cast<char*>(blob);
  size_t off = 0;
  auto take = [&](size_t bytes) { char* q = base ? base + off : nullptr; off = (off + bytes + 255) & ~(size_t)255; return q; };
  MaskRows m;
  m.num = reinterpret_cast<i64*>(take(sizeof(i64)));
  m.out2 = reinterpret_cast<uint32_t*>(take(2 * sizeof(uint32_t)));
  size_t tb = 0;
  (void)hipcub::DeviceSelect::Flagged(nullptr, tb, hipcub::CountingInputIterator<i64>(0), (const uint8_t*)nullptr,
                                      (i64*)nullptr, (i64*)nullptr, (int)(n > 0 ? n : 1));
  m.temp_bytes = tb;
  m.temp = take(tb);
  m.total = off;
  return m;
}
__global__ void mask_tail_kernel(const uint8_t* __restrict__ mask, i64 n, const i64* __restrict__ num,
                                 uint32_t* __restrict__ out2) {
  out2[0] = (uint32_t)*num;
  out2[1] = mask[n - 1] ? 1u : 0u;
}
constexpr int MAX_GATHER_JOBS = 16;
struct GatherJobs {
  const uint8_t* src[MAX_GATHER_JOBS];
  uint8_t* dst[MAX_GATHER_JOBS];
  i64 row_bytes[MAX_GATHER_JOBS];
  i64 rows[MAX_GATHER_JOBS];
};
// blockIdx.y = tensor; rows of 4-byte multiples move as words, the others (bool masks) as bytes
__global__ __launch_bounds__(256) void gather_rows_multi_kernel(GatherJobs j, const i64* __restrict__ idx) {
  const int g = blockIdx.y;
  const i64 rb = j.row_bytes[g], n = j.rows[g];
  const bool words = (rb & 3) == 0 && ((uintptr_t)j.src[g] & 3) == 0 && ((uintptr_t)j.dst[g] & 3) == 0;
  const i64 stride = (i64)gridDim.x * 256;
  if (words) {
    const i64 rw = rb >> 2;
    const uint32_t* s = reinterpret_cast<const uint32_t*>(j.src[g]);
    uint32_t* d = reinterpret_cast<uint32_t*>(j.dst[g]);
    for (i64 e = (i64)blockIdx.x * 256 + threadIdx.x; e < n * rw; e += stride) {
      const i64 r = e / rw, c = e - r * rw;
      d[e] = s[idx[r] * rw + c];
    }
  } else {
    for (i64 e = (i64)blockIdx.x * 256 + threadIdx.x; e < n * rb; e += stride) {
      const i64 r = e / rb, c = e - r * rb;
      j.dst[g][e] = j.src[g][idx[r] * rb + c];
    }
  }
}
}  // namespace

PINGS_API size_t pings_mask_rows_scratch_bytes(int64_t n) { return carve_mask_rows(nullptr, n).total; }

PINGS_API int pings_mask_rows(const uint8_t* mask, int64_t n, void* scratch, int64_t* rows, int64_t* count_and_last,
                              void* stream) {
  PINGS_ARG_CHECK(n >= 0 && n < (int64_t)0x7FFFFFF0 && count_and_last, "bad arguments");
  count_and_last[0] = 0; count_and_last[1] = 0;
  if (n == 0) return PINGS_OK;
  PINGS_ARG_CHECK(mask && scratch && rows, "null pointer");
  hipStream_t st = pings::as_stream(stream);
  MaskRows m = carve_mask_rows(scratch, n);
  size_t tb = m.temp_bytes;
  PINGS_HIP_CHECK(hipcub::DeviceSelect::Flagged(m.temp, tb, hipcub::CountingInputIterator<i64>(0), mask,
                                                reinterpret_cast<i64*>(rows), m.num, (int)n, st));
  mask_tail_kernel<<<1, 1, 0, st>>>(mask, n, m.num, m.out2);
  PINGS_LAUNCH_CHECK();
  const uint32_t* src[2] = {m.out2, m.out2 + 1};
  uint32_t got[2] = {0u, 0u};
  if (int e = pings::host_read_words(src, 2, got, st)) return e;
  count_and_last[0] = (int64_t)got[0];
  count_and_last[1] = (int64_t)got[1];
  return PINGS_OK;
}

PINGS_API int pings_gather_rows_multi(const pings_gather_job* jobs, int njobs, const int64_t* idx, void* stream) {
  PINGS_ARG_CHECK(jobs && njobs > 0 && njobs <= MAX_GATHER_JOBS, "1..16 jobs");
  GatherJobs J;
  i64 most = 0;
  for (int g = 0; g < njobs; ++g) {
    PINGS_ARG_CHECK(jobs[g].row_bytes > 0 && jobs[g].rows >= 0, "bad sizes in job");
    PINGS_ARG_CHECK(jobs[g].rows == 0 || (jobs[g].src && jobs[g].dst), "null pointer in job");
    J.src[g] = reinterpret_cast<const uint8_t*>(jobs[g].src);
    J.dst[g] = reinterpret_cast<uint8_t*>(jobs[g].dst);
    J.row_bytes[g] = jobs[g].row_bytes;
    J.rows[g] = jobs[g].rows;
    const i64 work = jobs[g].rows * ((jobs[g].row_bytes & 3) == 0 ? jobs[g].row_bytes >> 2 : jobs[g].row_bytes);
    if (work > most) most = work;
  }
  if (most == 0) return PINGS_OK;
  PINGS_ARG_CHECK(idx != nullptr, "null index list");
  hipStream_t st = pings::as_stream(stream);
  const i64 nb = (most + 255) / 256;
  gather_rows_multi_kernel<<<dim3((unsigned)(nb < 4096 ? nb : 4096), (unsigned)njobs), 256, 0, st>>>(J, (const i64*)idx);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}

PINGS_API int pings_scatter_rows(const void* src, int64_t row_bytes, const int64_t* idx, int64_t n, void* dst,
                                 void* stream) {
  PINGS_ARG_CHECK(n >= 0 && row_bytes > 0, "bad sizes");
  if (n == 0) return PINGS_OK;
  PINGS_ARG_CHECK(src && idx && dst, "null pointer");
  hipStream_t st = pings::as_stream(stream);
  if (row_bytes % 4 == 0 && ((uintptr_t)src % 4 == 0) && ((uintptr_t)dst % 4 == 0))
    scatter_rows_kernel<<<blocks_for(n * (row_bytes / 4)), 256, 0, st>>>((const uint32_t*)src, row_bytes / 4,
                                                                        (const i64*)idx, n, (uint32_t*)dst);
  else
    scatter_bytes_kernel<<<blocks_for(n * row_bytes), 256, 0, st>>>((const uint8_t*)src, row_bytes, (const i64*)idx,
                                                                    n, (uint8_t*)dst);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}
