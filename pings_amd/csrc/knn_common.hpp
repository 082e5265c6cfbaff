// Shared device code of the neural-point kernels (csrc/knn_sdf.hip, csrc/query_feature.hip, csrc/sdf_train.hip):
// the hash-grid neighbour search of one query by one wave64 (model/neural_gaussians.py:1061-1115 + the masking /
// top-k of query_feature :544-569), wave reductions and the quaternion helpers.  See knn_sdf.hip for the design notes.
#pragma once
#include "common.hpp"

namespace pings_knn {

constexpr int WAVES_PER_BLOCK = 4;
constexpr int MAX_NNK = 16;
constexpr int MAX_IN = 64;  // F + 3 <= 64
constexpr long long P0 = 73856093LL, P1 = 19349669LL, P2 = 83492791LL;  // neural_gaussians.py:80-82
constexpr float INVALID_D2 = 9e3f;                                       // :562

__device__ inline float wave_sum_all(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// wave64 unsigned minimum through DPP, broadcast to every lane (uniform result)
__device__ inline unsigned wave_min_u32_all(unsigned v) {
  const int id = (int)0xFFFFFFFFu;  // identity for lanes a masked DPP step does not write
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp(id, (int)v, 0xb1, 0xf, 0xf, false));
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp(id, (int)v, 0x4e, 0xf, 0xf, false));
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp(id, (int)v, 0x124, 0xf, 0xf, false));
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp(id, (int)v, 0x128, 0xf, 0xf, false));
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp(id, (int)v, 0x142, 0xa, 0xf, false));
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp(id, (int)v, 0x143, 0xc, 0xf, false));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// (cx*P0 + cy*P1 + cz*P2) % S with the dividend's sign, wrapped to [0, S) — exact.  |x| < 2^53 and
// S < 2^31, so the quotient estimate from one fp64 division is off by at most one; an int64
// division (a ~100-instruction software routine on gfx950) is avoided.
__device__ inline long long hash_slot(long long x, long long S, double inv_S) {
  const long long q = (long long)((double)x * inv_S);
  long long r = x - q * S;       // in (-2S, 2S) whatever the rounding of q did
  // reduce to the C remainder (sign of x), then wrap negatives like a python index
  if (r >= S) r -= S;
  if (r <= -S) r += S;
  if (x >= 0 && r < 0) r += S;
  if (x < 0 && r > 0) r -= S;
  if (r < 0) r += S;
  return r;
}

// Lookup of dense-table slot `h`: either the reference's dense int64 table or its compact mirror
// (open addressing, 8-byte {slot+1, value} entries, linear probing; built by pings_knn_compact_build).
__device__ inline long long table_lookup(const pings_knn_map& m, long long h) {
  if (m.compact == nullptr) return m.table[h];
  const uint2* tab = reinterpret_cast<const uint2*>(m.compact);
  const unsigned key = (unsigned)h + 1u;
  unsigned b = (unsigned)h & m.compact_mask;
  for (;;) {
    const uint2 e = tab[b];
    if (e.x == key) return (long long)e.y;
    if (e.x == 0u) return -1;
    b = (b + 1u) & m.compact_mask;
  }
}

// Per-lane constants of the search, loaded once per wave (not per query).
struct LaneCtx {
  int dx[2][3];     // cell offsets of this lane's two candidate cells
  bool has[2];      // candidate index < K
  float cur_td;     // travel_dist[cur_ts]
  double inv_S;
};

__device__ inline LaneCtx make_lane_ctx(const pings_knn_map& m, int lane) {
  LaneCtx c;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int k = lane + 64 * r;
    c.has[r] = k < m.K;
    const int ks = c.has[r] ? k : 0;
    c.dx[r][0] = m.neighbor_dx[3 * ks];
    c.dx[r][1] = m.neighbor_dx[3 * ks + 1];
    c.dx[r][2] = m.neighbor_dx[3 * ks + 2];
  }
  c.cur_td = m.time_filtering ? m.travel_dist[m.cur_ts] : 0.f;
  c.inv_S = 1.0 / (double)m.buffer_size;
  return c;
}

// Search + selection for one query (whole wave).  On return sIdx[i], sD2[i] (i < nn_k) hold the
// neighbours in order; returns the number of valid candidates over all K cells.
//
// The dependent memory chain is what bounds this kernel, so it is kept to three levels and both of
// the lane's candidates walk it together: (1) table probes, (2) everything that depends only on
// the table entry — position, creation time, free / valid flags, local index — issued as one group
// with clamped indices, (3) the travel distance of the creation time.  A candidate is dropped if
// ANY of the reference's tests fails (model/neural_gaussians.py:1088-1105, :544-554), so the tests
// commute and can be evaluated after the loads.
__device__ inline int knn_one_query(const pings_knn_map& m, const LaneCtx& lc, float qx, float qy, float qz,
                                    int lane, long long* sIdx, float* sD2, long long* sGIdx) {
  const long long gx = (long long)floorf(qx / m.resolution);
  const long long gy = (long long)floorf(qy / m.resolution);
  const long long gz = (long long)floorf(qz / m.resolution);

  // ---- level 1: table entries of both candidate cells
  long long tix[2];
  if (m.compact != nullptr) {
    const uint2* tab = reinterpret_cast<const uint2*>(m.compact);
    unsigned key[2], b[2];
    uint2 e[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const long long h = hash_slot((gx + lc.dx[r][0]) * P0 + (gy + lc.dx[r][1]) * P1 + (gz + lc.dx[r][2]) * P2,
                                    m.buffer_size, lc.inv_S);
      key[r] = (unsigned)h + 1u;
      b[r] = (unsigned)h & m.compact_mask;
    }
    e[0] = tab[b[0]];
    e[1] = tab[b[1]];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      while (e[r].x != key[r] && e[r].x != 0u) {  // collision chain (load factor <= 0.25: rare)
        b[r] = (b[r] + 1u) & m.compact_mask;
        e[r] = tab[b[r]];
      }
      tix[r] = (lc.has[r] && e[r].x == key[r]) ? (long long)e[r].y : -1;
    }
  } else {
    long long h[2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
      h[r] = hash_slot((gx + lc.dx[r][0]) * P0 + (gy + lc.dx[r][1]) * P1 + (gz + lc.dx[r][2]) * P2,
                       m.buffer_size, lc.inv_S);
    const long long t0 = m.table[h[0]], t1 = m.table[h[1]];
    tix[0] = lc.has[0] ? t0 : -1;
    tix[1] = lc.has[1] ? t1 : -1;
  }

  // ---- level 2: everything addressed by the table entry, one group of independent loads
  float px[2], py[2], pz[2];
  int ts[2] = {0, 0};
  unsigned char fr[2] = {0, 0}, va[2] = {1, 1};
  long long loc[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const long long is = tix[r] >= 0 ? tix[r] : 0;
    px[r] = m.neural_points[3 * is];
    py[r] = m.neural_points[3 * is + 1];
    pz[r] = m.neural_points[3 * is + 2];
    if (m.time_filtering) ts[r] = m.point_ts_create[is];
    if (m.use_free_mask) fr[r] = m.free_mask[is];
    if (m.use_valid_mask) va[r] = m.valid_mask[is];
    loc[r] = m.global2local ? m.global2local[is] : is;
  }
  // ---- level 3: travel distance at the creation time
  float td[2] = {0.f, 0.f};
  if (m.time_filtering) {
    td[0] = m.travel_dist[ts[0]];
    td[1] = m.travel_dist[ts[1]];
  }

  unsigned key[2];  // fp32 bits of the squared distance (non-negative: ordered as unsigned); ~0 = taken
  long long cidx[2], gidx[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const float sx = px[r] - qx, sy = py[r] - qy, sz = pz[r] - qz;
    const float dd = (sx * sx + sy * sy) + sz * sz;
    bool ok = tix[r] >= 0;
    if (m.time_filtering) ok = ok && (fabsf(lc.cur_td - td[r]) < m.diff_travel_dist_local);
    ok = ok && !(dd > m.max_valid_dist2);
    ok = ok && !(m.use_free_mask && fr[r]);
    ok = ok && !(m.use_valid_mask && !va[r]);
    ok = ok && (loc[r] >= 0);
    cidx[r] = ok ? loc[r] : -1;
    gidx[r] = ok ? tix[r] : -1;
    key[r] = lc.has[r] ? __float_as_uint(ok ? dd : INVALID_D2) : 0xFFFFFFFFu;
  }
  const int count = __popcll(__ballot(cidx[0] >= 0)) + __popcll(__ballot(cidx[1] >= 0));

  // nn_k rounds: wave minimum of the distance bits, then the lowest candidate index among the
  // ties (cells 0..63 live in key[0] of lanes 0..63, cells 64.. in key[1])
  for (int i = 0; i < m.nn_k; ++i) {
    const unsigned best = wave_min_u32_all(min(key[0], key[1]));
    const unsigned long long b0 = __ballot(key[0] == best);
    const unsigned long long b1 = __ballot(key[1] == best);
    const int which = b0 != 0ull ? 0 : 1;
    const int owner = __ffsll((long long)(which ? b1 : b0)) - 1;
    if (lane == owner) {
      sIdx[i] = which ? cidx[1] : cidx[0];
      sGIdx[i] = which ? gidx[1] : gidx[0];
      sD2[i] = __uint_as_float(best);
      if (which) key[1] = 0xFFFFFFFFu; else key[0] = 0xFFFFFFFFu;
    }
  }
  return count;
}

// quaternion [w,x,y,z]: returns R(q)^T v  (utils/tools.py:743-751, the "passive" rotation)
__device__ inline void rot_passive(const float* q, float vx, float vy, float vz, float& ox, float& oy,
                                   float& oz) {
  const float w = q[0], x = -q[1], y = -q[2], z = -q[3];
  const float tx = 2.f * (y * vz - z * vy), ty = 2.f * (z * vx - x * vz), tz = 2.f * (x * vy - y * vx);
  ox = vx + w * tx + (y * tz - z * ty);
  oy = vy + w * ty + (z * tx - x * tz);
  oz = vz + w * tz + (x * ty - y * tx);
}

// inverse of the above: R(q) v
__device__ inline void rot_active(const float* q, float vx, float vy, float vz, float& ox, float& oy,
                                  float& oz) {
  const float w = q[0], x = q[1], y = q[2], z = q[3];
  const float tx = 2.f * (y * vz - z * vy), ty = 2.f * (z * vx - x * vz), tz = 2.f * (x * vy - y * vx);
  ox = vx + w * tx + (y * tz - z * ty);
  oy = vy + w * ty + (z * tx - x * tz);
  oz = vz + w * tz + (x * ty - y * tx);
}


inline int check_map(const pings_knn_map* m) {
  PINGS_ARG_CHECK(m != nullptr, "null map");
  PINGS_ARG_CHECK((m->table || m->compact) && m->buffer_size > 0 && m->neural_points && m->neighbor_dx,
                  "null map pointer");
  PINGS_ARG_CHECK(m->buffer_size < (1LL << 31), "buffer_size must be below 2^31");
  PINGS_ARG_CHECK(!m->compact || ((m->compact_mask & (m->compact_mask + 1u)) == 0u), "compact_mask must be 2^k - 1");
  PINGS_ARG_CHECK(m->K > 0 && m->K <= 128, "K must be in 1..128");
  PINGS_ARG_CHECK(m->nn_k > 0 && m->nn_k <= MAX_NNK && m->nn_k <= m->K, "nn_k must be in 1..16");
  PINGS_ARG_CHECK(!m->time_filtering || (m->point_ts_create && m->travel_dist), "time filtering needs ts / travel_dist");
  PINGS_ARG_CHECK(!m->use_free_mask || m->free_mask, "use_free_mask without mask");
  PINGS_ARG_CHECK(!m->use_valid_mask || m->valid_mask, "use_valid_mask without mask");
  PINGS_ARG_CHECK(m->resolution > 0.f, "resolution must be positive");
  return PINGS_OK;
}

inline unsigned grid_for(long long B) {
  const long long blocks = (B + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  const long long cap = 256LL * 8 * 4;  // 256 CUs x 8 blocks; waves loop over the rest
  return (unsigned)(blocks < cap ? blocks : cap);
}


}  // namespace pings_knn
