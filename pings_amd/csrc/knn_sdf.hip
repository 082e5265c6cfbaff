// Neural-point kNN search and fused SDF decode for gfx950.
//
// One wave64 per query point.  Lane c (and c+64) owns candidate cell c of the K <= 128
// neighbour cells: it hashes the cell with the reference's rule (int64 sum of cell*prime,
// C remainder by the table size, negative slots wrapped — model/neural_gaussians.py:1069-1084),
// gathers the table entry, applies the travel-distance window, the distance gate and the
// free / valid / local-map filters (:1088-1105, :544-554), so all K random table reads and the
// dependent point reads of a query are in flight together.  The nn_k nearest are then picked
// by nn_k wave-wide 64-bit min-reductions on (distance bits << 32 | candidate), i.e. ordered
// by distance with ties broken by candidate order.
//
// pings_sdf_forward continues in the same wave: inverse-distance weights (:644-662), feature
// rows staged in LDS, the one-hidden-layer MLP with lane j owning hidden unit j (weights of
// that unit live in registers across the queries a wave processes), the IDW reduction
// (mapper.py:2279) and, on request, the analytic gradient d sdf / d query through the MLP
// input x - p_j and through the weights.  HBM/L2 traffic per query is dominated by the K
// random 8-B table gathers (64-B sectors): ~10 kB of sector traffic for 2.4 kB of
// algorithmic bytes; the MLP (28 kFLOP) stays on the fp32 VALU — fp32 MFMA
// has the same rate on gfx950 and bf16 would break the 1e-4 parity bound.
//
// Distances are evaluated as ((dx*dx + dy*dy) + dz*dz) without fused multiply-adds
// (-ffp-contract=off), which reproduces the reference's fp32 values bit for bit, so the
// neighbour order is index-exact.
#include "common.hpp"

namespace {

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

__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void knn_search_kernel(
    pings_knn_map m, const float* __restrict__ queries, long long B, long long* __restrict__ idx_out,
    float* __restrict__ d2_out, long long* __restrict__ cnt_out, long long* __restrict__ gidx_out) {
  __shared__ long long sIdx[WAVES_PER_BLOCK][MAX_NNK];
  __shared__ long long sGIdx[WAVES_PER_BLOCK][MAX_NNK];
  __shared__ float sD2[WAVES_PER_BLOCK][MAX_NNK];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const LaneCtx lc = make_lane_ctx(m, lane);
  const long long nwaves = (long long)gridDim.x * WAVES_PER_BLOCK;
  for (long long q = (long long)blockIdx.x * WAVES_PER_BLOCK + wave; q < B; q += nwaves) {
    const float qx = queries[3 * q], qy = queries[3 * q + 1], qz = queries[3 * q + 2];
    const int count = knn_one_query(m, lc, qx, qy, qz, lane, sIdx[wave], sD2[wave], sGIdx[wave]);
    __builtin_amdgcn_wave_barrier();
    if (lane < m.nn_k) {
      idx_out[q * m.nn_k + lane] = sIdx[wave][lane];
      d2_out[q * m.nn_k + lane] = sD2[wave][lane];
      if (gidx_out) gidx_out[q * m.nn_k + lane] = sGIdx[wave][lane];
    }
    if (lane == 0) cnt_out[q] = count;
    __builtin_amdgcn_wave_barrier();
  }
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

template <int IN_PAD>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK, 4) void sdf_forward_kernel(
    pings_knn_map m, pings_sdf_decoder dec, const float* __restrict__ features,
    const float* __restrict__ points, const float* __restrict__ orientations,
    const float* __restrict__ certainties, int after_pgo, const float* __restrict__ queries,
    long long B, float* __restrict__ sdf_out, float* __restrict__ grad_out,
    long long* __restrict__ cnt_out, float* __restrict__ cert_out, long long* __restrict__ idx_out,
    float* __restrict__ w_out, float* __restrict__ std_out) {
  __shared__ long long sIdx[WAVES_PER_BLOCK][MAX_NNK];
  __shared__ long long sGIdx[WAVES_PER_BLOCK][MAX_NNK];
  __shared__ float sD2[WAVES_PER_BLOCK][MAX_NNK];
  __shared__ float sW[WAVES_PER_BLOCK][MAX_NNK];
  __shared__ __attribute__((aligned(16))) float sIn[WAVES_PER_BLOCK][MAX_NNK][IN_PAD];
  __shared__ float sVec[WAVES_PER_BLOCK][MAX_NNK][4];  // x - p_j w.r.t. the searched (global) point: d(d2)/dx
  __shared__ float sS[WAVES_PER_BLOCK][MAX_NNK];       // per-neighbour prediction / projection

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int F = dec.feat_dim, IN = F + 3, Hd = dec.hidden, nnk = m.nn_k;
  // this lane's hidden unit: weights stay in registers across queries
  float w1[IN_PAD];
  float b1 = 0.f, w2 = 0.f;
#pragma unroll
  for (int i = 0; i < IN_PAD; ++i) w1[i] = (lane < Hd && i < IN) ? dec.W1[lane * IN + i] : 0.f;
  // direction-input weights of this unit (columns F..F+2), kept separately for the gradient
  const float w1n0 = lane < Hd ? dec.W1[lane * IN + F] : 0.f;
  const float w1n1 = lane < Hd ? dec.W1[lane * IN + F + 1] : 0.f;
  const float w1n2 = lane < Hd ? dec.W1[lane * IN + F + 2] : 0.f;
  if (lane < Hd) { b1 = dec.b1[lane]; w2 = dec.W2[lane]; }
  const float b2 = dec.b2[0];

  const LaneCtx lc = make_lane_ctx(m, lane);
  const long long nwaves = (long long)gridDim.x * WAVES_PER_BLOCK;
  for (long long q = (long long)blockIdx.x * WAVES_PER_BLOCK + wave; q < B; q += nwaves) {
    const float qx = queries[3 * q], qy = queries[3 * q + 1], qz = queries[3 * q + 2];
    const int count = knn_one_query(m, lc, qx, qy, qz, lane, sIdx[wave], sD2[wave], sGIdx[wave]);
    __builtin_amdgcn_wave_barrier();

    // ---- inverse-distance weights (lane i < nn_k owns neighbour i)
    long long my_idx = -1;
    float u = 0.f, my_d2 = INVALID_D2;
    if (lane < nnk) {
      my_idx = sIdx[wave][lane];
      my_d2 = sD2[wave][lane];
      if (my_idx >= 0) u = 1.0f / (my_d2 + 1e-15f);
    }
    const float U = wave_sum_all(u);
    const float wgt = (my_idx >= 0) ? u / U : 0.f;
    float cert = 0.f;
    if (lane < nnk) {
      sW[wave][lane] = wgt;
      if (idx_out) idx_out[q * nnk + lane] = my_idx;
      if (w_out) w_out[q * nnk + lane] = wgt;
      float vx = 0.f, vy = 0.f, vz = 0.f;
      if (my_idx >= 0) {
        vx = qx - points[3 * my_idx];
        vy = qy - points[3 * my_idx + 1];
        vz = qz - points[3 * my_idx + 2];
        if (certainties) cert = certainties[my_idx] * wgt;
        const long long gi = sGIdx[wave][lane];
        sVec[wave][lane][0] = qx - m.neural_points[3 * gi];
        sVec[wave][lane][1] = qy - m.neural_points[3 * gi + 1];
        sVec[wave][lane][2] = qz - m.neural_points[3 * gi + 2];
      }
      float nx = vx, ny = vy, nz = vz;
      if (after_pgo && my_idx >= 0) rot_passive(orientations + 4 * my_idx, vx, vy, vz, nx, ny, nz);
      sIn[wave][lane][F] = nx; sIn[wave][lane][F + 1] = ny; sIn[wave][lane][F + 2] = nz;
      for (int i = IN; i < IN_PAD; ++i) sIn[wave][lane][i] = 0.f;
    }
    if (cert_out) {
      const float cs = wave_sum_all(cert);
      if (lane == 0) cert_out[q] = cs;
    }
    // ---- feature rows -> LDS (zeros for missing neighbours)
    for (int e = lane; e < nnk * F; e += 64) {
      const int mm = e / F, f = e - mm * F;
      const long long id = sIdx[wave][mm];
      sIn[wave][mm][f] = id >= 0 ? features[id * F + f] : 0.f;
    }
    __builtin_amdgcn_wave_barrier();

    float S = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
    if (dec.weighted_first) {
      // in = sum_m w_m [f_m, n_m]  (neural_gaussians.py:701-705), one MLP evaluation
      // pre_j = b1_j + sum_m w_m (W1[j,:] . in_m)
      float pre = b1;
      for (int mm = 0; mm < nnk; ++mm) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < IN_PAD; ++i) acc = fmaf(w1[i], sIn[wave][mm][i], acc);
        pre = fmaf(sW[wave][mm], acc, pre);
      }
      const float h = fmaxf(pre, 0.f);
      S = dec.sdf_scale * (b2 + wave_sum_all(lane < Hd ? w2 * h : 0.f));
      if (grad_out && count > 0) {
        const float gh = (lane < Hd && pre > 0.f) ? w2 : 0.f;   // dS/dpre_j / scale
        // g_in[i] = sum_j W1[j][i] gh_j ; needed: the 3 direction entries and g_in . in_m
        float gn0 = wave_sum_all(w1n0 * gh), gn1 = wave_sum_all(w1n1 * gh),
              gn2 = wave_sum_all(w1n2 * gh);
        // through the direction input: sum_m w_m R_m gn  (sum_m w_m = 1)
        for (int mm = 0; mm < nnk; ++mm) {
          const long long id = sIdx[wave][mm];
          if (id < 0) continue;
          float a0 = gn0, a1 = gn1, a2 = gn2;
          if (after_pgo) rot_active(orientations + 4 * id, gn0, gn1, gn2, a0, a1, a2);
          const float wm = sW[wave][mm];
          gx = fmaf(wm, a0, gx); gy = fmaf(wm, a1, gy); gz = fmaf(wm, a2, gz);
        }
        // through the weights: (1/U) sum_m (t_m - tbar) du_m/dx, t_m = g_in . in_m
        float tbar = 0.f;
        for (int mm = 0; mm < nnk; ++mm) {
          float acc = 0.f;
#pragma unroll
          for (int i = 0; i < IN_PAD; ++i) acc = fmaf(w1[i], sIn[wave][mm][i], acc);  // W1[j,:] . in_m
          const float t = wave_sum_all(acc * gh);
          sS[wave][mm] = t;
          tbar = fmaf(sW[wave][mm], t, tbar);
        }
        __builtin_amdgcn_wave_barrier();
        for (int mm = 0; mm < nnk; ++mm) {
          if (sIdx[wave][mm] < 0) continue;
          const float um = 1.0f / (sD2[wave][mm] + 1e-15f);
          const float k = (sS[wave][mm] - tbar) * (-2.f * um * um) / U;
          gx = fmaf(k, sVec[wave][mm][0], gx);
          gy = fmaf(k, sVec[wave][mm][1], gy);
          gz = fmaf(k, sVec[wave][mm][2], gz);
        }
        gx *= dec.sdf_scale; gy *= dec.sdf_scale; gz *= dec.sdf_scale;
      }
    } else {
      // per-neighbour MLP, then IDW of the predictions (mapper.py:2279)
      for (int mm = 0; mm < nnk; ++mm) {
        float pre = b1;
#pragma unroll
        for (int i = 0; i < IN_PAD; ++i) pre = fmaf(w1[i], sIn[wave][mm][i], pre);
        const float h = fmaxf(pre, 0.f);
        const float s_m = dec.sdf_scale * (b2 + wave_sum_all(lane < Hd ? w2 * h : 0.f));
        sS[wave][mm] = s_m;
        const float wm = sW[wave][mm];
        S = fmaf(wm, s_m, S);
        if (grad_out && sIdx[wave][mm] >= 0) {
          const float gh = (lane < Hd && pre > 0.f) ? w2 : 0.f;
          float gn0 = wave_sum_all(w1n0 * gh), gn1 = wave_sum_all(w1n1 * gh),
                gn2 = wave_sum_all(w1n2 * gh);
          if (after_pgo) rot_active(orientations + 4 * sIdx[wave][mm], gn0, gn1, gn2, gn0, gn1, gn2);
          const float k = wm * dec.sdf_scale;
          gx = fmaf(k, gn0, gx); gy = fmaf(k, gn1, gy); gz = fmaf(k, gn2, gz);
        }
      }
      if (grad_out && count > 0) {
        __builtin_amdgcn_wave_barrier();
        for (int mm = 0; mm < nnk; ++mm) {
          if (sIdx[wave][mm] < 0) continue;
          const float um = 1.0f / (sD2[wave][mm] + 1e-15f);
          const float k = (sS[wave][mm] - S) * (-2.f * um * um) / U;
          gx = fmaf(k, sVec[wave][mm][0], gx);
          gy = fmaf(k, sVec[wave][mm][1], gy);
          gz = fmaf(k, sVec[wave][mm][2], gz);
        }
      }
    }
    if (lane == 0) {
      sdf_out[q] = S;
      if (grad_out) { grad_out[3 * q] = gx; grad_out[3 * q + 1] = gy; grad_out[3 * q + 2] = gz; }
      if (cnt_out) cnt_out[q] = count;
      if (std_out) {
        // spread of the per-neighbour predictions, sqrt(sum_m w_m (s_m - S)^2) (utils/tracker.py:303-308);
        // stays 0 in weighted_first mode, as the reference's buffer does
        float var = 0.f;
        if (!dec.weighted_first)
          for (int mm = 0; mm < nnk; ++mm) {
            const float d = sS[wave][mm] - S;
            var += sW[wave][mm] * (d * d);
          }
        std_out[q] = sqrtf(var);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ __launch_bounds__(256) void compact_build_kernel(const long long* __restrict__ table, long long S,
                                                            uint2* __restrict__ tab, unsigned mask) {
  const long long nthreads = (long long)gridDim.x * blockDim.x;
  for (long long h = (long long)blockIdx.x * blockDim.x + threadIdx.x; h < S; h += nthreads) {
    const long long v = table[h];
    if (v < 0) continue;
    const unsigned key = (unsigned)h + 1u;
    unsigned b = (unsigned)h & mask;
    for (;;) {
      const unsigned prev = atomicCAS(&tab[b].x, 0u, key);
      if (prev == 0u) {
        tab[b].y = (unsigned)v;
        break;
      }
      b = (b + 1u) & mask;
    }
  }
}

int check_map(const pings_knn_map* m) {
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

unsigned grid_for(long long B) {
  const long long blocks = (B + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  const long long cap = 256LL * 8 * 4;  // 256 CUs x 8 blocks; waves loop over the rest
  return (unsigned)(blocks < cap ? blocks : cap);
}

}  // namespace

PINGS_API size_t pings_knn_compact_entries(int64_t num_points) {
  size_t cap = 1024;
  const size_t want = (size_t)(num_points > 0 ? num_points : 1) * 4;  // load factor <= 0.25
  while (cap < want) cap <<= 1;
  return cap;
}

PINGS_API int pings_knn_compact_build(const int64_t* table, int64_t buffer_size, void* compact,
                                      size_t entries, void* stream) {
  PINGS_ARG_CHECK(table && compact && buffer_size > 0 && buffer_size < (1LL << 31), "bad table");
  PINGS_ARG_CHECK(entries >= 1024 && (entries & (entries - 1)) == 0 && entries <= (1ull << 31),
                  "entries must be a power of two");
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope ps("knn_compact_build", st);
  PINGS_HIP_CHECK(hipMemsetAsync(compact, 0, entries * sizeof(uint2), st));
  hipLaunchKernelGGL(compact_build_kernel, dim3(256 * 16), dim3(256), 0, st, (const long long*)table,
                     (long long)buffer_size, reinterpret_cast<uint2*>(compact), (unsigned)(entries - 1));
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}

PINGS_API int pings_knn_search(const pings_knn_map* m, const float* queries, int64_t B, int64_t* idx,
                               float* d2, int64_t* nn_counts, int64_t* global_idx, void* stream) {
  if (int e = check_map(m)) return e;
  if (B == 0) return PINGS_OK;
  PINGS_ARG_CHECK(B > 0 && queries && idx && d2 && nn_counts, "null pointer");
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope ps("knn_search", st);
  hipLaunchKernelGGL(knn_search_kernel, dim3(grid_for(B)), dim3(64 * WAVES_PER_BLOCK), 0, st, *m, queries,
                     (long long)B, (long long*)idx, d2, (long long*)nn_counts, (long long*)global_idx);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}

PINGS_API int pings_sdf_forward(const pings_knn_map* m, const pings_sdf_decoder* dec,
                                const float* features, const float* points,
                                const float* orientations, const float* certainties,
                                int32_t after_pgo, const float* queries, int64_t B, float* sdf,
                                float* grad_x, int64_t* nn_counts, float* certainty,
                                int64_t* idx_out, float* w_out, float* sdf_std, void* stream) {
  if (int e = check_map(m)) return e;
  PINGS_ARG_CHECK(dec && dec->W1 && dec->b1 && dec->W2 && dec->b2, "null decoder");
  PINGS_ARG_CHECK(dec->hidden > 0 && dec->hidden <= 64, "hidden must be in 1..64");
  PINGS_ARG_CHECK(dec->feat_dim > 0 && dec->feat_dim + 3 <= MAX_IN, "feature dim must be <= 61");
  PINGS_ARG_CHECK(!after_pgo || orientations, "after_pgo needs orientations");
  PINGS_ARG_CHECK(!certainty || certainties, "certainty output needs the certainty table");
  if (B == 0) return PINGS_OK;
  PINGS_ARG_CHECK(B > 0 && features && points && queries && sdf, "null pointer");
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope ps("sdf_forward", st);
  const int in_dim = dec->feat_dim + 3;
#define PINGS_SDF_LAUNCH(PAD)                                                                          \
  hipLaunchKernelGGL(sdf_forward_kernel<PAD>, dim3(grid_for(B)), dim3(64 * WAVES_PER_BLOCK), 0, st, *m, \
                     *dec, features, points, orientations, certainties, (int)after_pgo, queries,       \
                     (long long)B, sdf, grad_x, (long long*)nn_counts, certainty, (long long*)idx_out, w_out, sdf_std)
  if (in_dim <= 12) PINGS_SDF_LAUNCH(12);
  else if (in_dim <= 36) PINGS_SDF_LAUNCH(36);
  else PINGS_SDF_LAUNCH(64);
#undef PINGS_SDF_LAUNCH
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}
