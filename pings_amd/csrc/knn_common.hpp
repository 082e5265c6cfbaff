// Shared device code of the neural-point kernels (csrc/knn_sdf.hip, csrc/query_feature.hip, csrc/sdf_train.hip):
// the hash-grid neighbour search of one query by one wave64 (model/neural_gaussians.py:1061-1115 + the masking /
// top-k of query_feature :544-569), wave reductions and the quaternion helpers.  See knn_sdf.hip for the design notes.
#pragma once
#include "common.hpp"
#include <cstdlib>
#include <mutex>
#include <unordered_map>

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

template <int CTRL>
__device__ inline float dpp_movf(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// Sums v[0..7] over the 64 lanes with a transposed reduce-scatter (two quad_perm exchange steps 8 -> 4 -> 2 values per
// lane, one rotate step inside each 16-lane row, two cross-row swaps): ~30 vector ops and a short dependency chain
// instead of eight 8-step reductions.  Every add is between a pair of partners that both end with the same sum, so
// the result is bitwise the same in whichever slot a value travels.  Afterwards lane l of every row holds the wave
// total of slot 4*(l&1) + 2*((l>>1)&1) + ((l>>2)&1).
__device__ inline int reduce8_slot(int lane) { return 4 * (lane & 1) + 2 * ((lane >> 1) & 1) + ((lane >> 2) & 1); }
__device__ inline float wave_reduce8(const float (&v)[8], int lane) {
  const bool b0 = lane & 1, b1 = lane & 2, m0 = lane & 4;
  float u[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float send = b0 ? v[k] : v[k + 4];
    const float keep = b0 ? v[k + 4] : v[k];
    u[k] = keep + dpp_movf<0xb1>(send);  // quad_perm [1,0,3,2]
  }
  float t[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float send = b1 ? u[k] : u[k + 2];
    const float keep = b1 ? u[k + 2] : u[k];
    t[k] = keep + dpp_movf<0x4e>(send);  // quad_perm [2,3,0,1]
  }
  // lanes {c, c+4, c+8, c+12} of a row (quads m = 0..3) share the slot base and each hold both slots' quad sums;
  // lane c+4m ends with slot base + (m & 1).  The four quad sums are added as (q_m + q_m+2) + (q_m+1 + q_m+3) — the
  // same tree whichever quad ends up holding the slot — so a value's total does not depend on the slot it was put in.
  const float own = m0 ? t[1] : t[0], other = m0 ? t[0] : t[1];
  const float x2 = own + dpp_movf<0x128>(own);      // row_ror:8: the quad two away wants the same slot
  const float y2 = other + dpp_movf<0x128>(other);  // this lane's share of the neighbouring quads' slot
  float r = x2 + dpp_movf<0x124>(y2);               // row_ror:4: a neighbouring quad's pair sum of MY slot
  {
    float a = r, b = r;   // inline asm: this hipcc maps both results of the swap builtins to one register
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    r = a + b;
    a = r; b = r;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    r = a + b;
  }
  return r;
}

// sum over the 8-lane group of a lane (all 8 lanes get it): xor 1, xor 2 inside the quad, then the other quad through
// row_half_mirror (lane i <-> 7 - i of the 8-lane half row, which holds the other quad's total)
__device__ inline float group8_sum(float v) {
  v += dpp_movf<0xb1>(v);   // quad_perm [1,0,3,2]
  v += dpp_movf<0x4e>(v);   // quad_perm [2,3,0,1]
  v += dpp_movf<0x141>(v);  // row_half_mirror
  return v;
}

// row of a 32x32 MFMA accumulator register: D[rowmap32(reg, lane >> 5)][lane & 31]
__device__ inline int rowmap32(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

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

// ---- cell-block index (pings_knn_blocks_build; design notes in knn_blocks.hip) --------------------------------------
// 4x4x4 cells per block.  Block table: open addressing on the packed block coordinate, 32-byte entries; records: the
// registered points of a block, packed in cell order, 32 bytes each with everything the search tests.
struct __attribute__((aligned(32))) BlockEntry {
  unsigned long long key;   // 1 + packed (bx, by, bz), 0 = empty
  unsigned long long mask;  // occupied cells of the block
  unsigned base;            // first record of the block
  unsigned pad[3];
};
struct __attribute__((aligned(32))) BlockRec {
  float x, y, z;   // neural_points[gidx]
  int gidx;        // row of the global point array
  int loc;         // global2local[gidx] as baked (== gidx when no global2local was given)
  float td;        // travel_dist[point_ts_create[gidx]]
  unsigned flags;  // bit 0: free_gs_mask, bit 1: valid_gs_mask
  unsigned pad;
};
constexpr int BLOCK_COORD_LIMIT = 1 << 20;  // |block coordinate| < 2^20 (cells < 2^22)

__device__ inline bool block_coord_ok(int bx, int by, int bz) {
  return bx >= -BLOCK_COORD_LIMIT && bx < BLOCK_COORD_LIMIT && by >= -BLOCK_COORD_LIMIT && by < BLOCK_COORD_LIMIT &&
         bz >= -BLOCK_COORD_LIMIT && bz < BLOCK_COORD_LIMIT;
}
__device__ inline unsigned long long block_key(int bx, int by, int bz) {
  return 1ull + (((unsigned long long)(unsigned)(bx + BLOCK_COORD_LIMIT) << 42) |
                 ((unsigned long long)(unsigned)(by + BLOCK_COORD_LIMIT) << 21) |
                 (unsigned long long)(unsigned)(bz + BLOCK_COORD_LIMIT));
}
__device__ inline unsigned block_slot(unsigned long long key, unsigned mask) {
  return (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 32) & mask;
}
__device__ inline unsigned cell_bit(int cx, int cy, int cz) { return (cx & 3) | ((cy & 3) << 2) | ((cz & 3) << 4); }

// Per-lane constants of the search, loaded once per wave (not per query).
struct LaneCtx {
  int dx[2][3];     // cell offsets of this lane's two candidate cells
  bool has[2];      // candidate index < K
  float cur_td;     // travel_dist[cur_ts]
  double inv_S;
  bool use_blocks;  // the cell-block index is present and was verified exact for this table (uniform)
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
  c.use_blocks = m.blocks != nullptr && m.blocks_ok != nullptr && *m.blocks_ok == 1;
  return c;
}

// A lane's two candidates after every test of the reference (model/neural_gaussians.py:1088-1105, :544-554).
struct Cand {
  unsigned key[2];      // fp32 bits of the squared distance (non-negative: ordered as unsigned); ~0 = no such cell
  long long cidx[2];    // row of the queried tables (local index for a local query), -1 = dropped
  long long gidx[2];    // row of the global point array, -1 = dropped
  float px[2], py[2], pz[2];
};

// Candidates through the reference's table (dense, or its compact mirror).
//
// The dependent memory chain is what bounds this path, so it is kept to three levels and both of
// the lane's candidates walk it together: (1) table probes, (2) everything that depends only on
// the table entry — position, creation time, free / valid flags, local index — issued as one group
// with clamped indices, (3) the travel distance of the creation time.  A candidate is dropped if
// ANY of the reference's tests fails, so the tests commute and can be evaluated after the loads.
__device__ inline void candidates_table(const pings_knn_map& m, const LaneCtx& lc, float qx, float qy, float qz,
                                        Cand& c) {
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
  int ts[2] = {0, 0};
  unsigned char fr[2] = {0, 0}, va[2] = {1, 1};
  long long loc[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const long long is = tix[r] >= 0 ? tix[r] : 0;
    c.px[r] = m.neural_points[3 * is];
    c.py[r] = m.neural_points[3 * is + 1];
    c.pz[r] = m.neural_points[3 * is + 2];
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

#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const float sx = c.px[r] - qx, sy = c.py[r] - qy, sz = c.pz[r] - qz;
    const float dd = (sx * sx + sy * sy) + sz * sz;
    bool ok = tix[r] >= 0;
    if (m.time_filtering) ok = ok && (fabsf(lc.cur_td - td[r]) < m.diff_travel_dist_local);
    ok = ok && !(dd > m.max_valid_dist2);
    ok = ok && !(m.use_free_mask && fr[r]);
    ok = ok && !(m.use_valid_mask && !va[r]);
    ok = ok && (loc[r] >= 0);
    c.cidx[r] = ok ? loc[r] : -1;
    c.gidx[r] = ok ? tix[r] : -1;
    c.key[r] = lc.has[r] ? __float_as_uint(ok ? dd : INVALID_D2) : 0xFFFFFFFFu;
  }
}

// Candidates through the cell-block index: two levels (block entry, record), one 64-B sector per distinct block and
// per live candidate.  Identical results to candidates_table whenever *m.blocks_ok == 1 (knn_blocks.hip).
__device__ inline void candidates_blocks(const pings_knn_map& m, const LaneCtx& lc, float qx, float qy, float qz,
                                         Cand& c) {
  const float fx = floorf(qx / m.resolution), fy = floorf(qy / m.resolution), fz = floorf(qz / m.resolution);
  // queries further out than any indexed cell have no candidates (and must not overflow the int cell coordinate)
  const float lim = (float)(4 * BLOCK_COORD_LIMIT - 256);
  const bool q_in = fabsf(fx) < lim && fabsf(fy) < lim && fabsf(fz) < lim;
  const int gx = q_in ? (int)fx : 0, gy = q_in ? (int)fy : 0, gz = q_in ? (int)fz : 0;
  const uint4* tab = reinterpret_cast<const uint4*>(m.blocks);

  // ---- level 1: block entries
  unsigned long long key[2];
  unsigned slot[2], bit[2];
  uint4 e[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int cx = gx + lc.dx[r][0], cy = gy + lc.dx[r][1], cz = gz + lc.dx[r][2];
    key[r] = block_key(cx >> 2, cy >> 2, cz >> 2);
    bit[r] = cell_bit(cx, cy, cz);
    slot[r] = block_slot(key[r], m.block_mask);
  }
  e[0] = tab[2 * (size_t)slot[0]];
  e[1] = tab[2 * (size_t)slot[1]];
  unsigned base[2];
  bool found[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    unsigned long long k = ((unsigned long long)e[r].y << 32) | e[r].x;
    while (k != key[r] && k != 0ull) {  // collision chain (load factor <= 0.5 in the worst case, ~0.1 on surfaces)
      slot[r] = (slot[r] + 1u) & m.block_mask;
      e[r] = tab[2 * (size_t)slot[r]];
      k = ((unsigned long long)e[r].y << 32) | e[r].x;
    }
    const unsigned long long mask = ((unsigned long long)e[r].w << 32) | e[r].z;
    found[r] = q_in && lc.has[r] && k == key[r] && ((mask >> bit[r]) & 1ull);
    const unsigned before = (unsigned)__popcll(mask & ((1ull << bit[r]) - 1ull));
    base[r] = found[r] ? tab[2 * (size_t)slot[r] + 1].x + before : 0u;
  }
  // ---- level 2: records
  const uint4* recs = reinterpret_cast<const uint4*>(m.block_records);
  uint4 ra[2], rb[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    ra[r] = recs[2 * (size_t)base[r]];
    rb[r] = recs[2 * (size_t)base[r] + 1];
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    c.px[r] = __uint_as_float(ra[r].x);
    c.py[r] = __uint_as_float(ra[r].y);
    c.pz[r] = __uint_as_float(ra[r].z);
    const int gi = (int)ra[r].w, li = m.global2local ? (int)rb[r].x : gi;
    const float td = __uint_as_float(rb[r].y);
    const unsigned fl = rb[r].z;
    const float sx = c.px[r] - qx, sy = c.py[r] - qy, sz = c.pz[r] - qz;
    const float dd = (sx * sx + sy * sy) + sz * sz;
    bool ok = found[r];
    if (m.time_filtering) ok = ok && (fabsf(lc.cur_td - td) < m.diff_travel_dist_local);
    ok = ok && !(dd > m.max_valid_dist2);
    ok = ok && !(m.use_free_mask && (fl & 1u));
    ok = ok && !(m.use_valid_mask && !(fl & 2u));
    ok = ok && (li >= 0);
    c.cidx[r] = ok ? (long long)li : -1;
    c.gidx[r] = ok ? (long long)gi : -1;
    c.key[r] = lc.has[r] ? __float_as_uint(ok ? dd : INVALID_D2) : 0xFFFFFFFFu;
  }
}

// The nn_k nearest of a lane-distributed candidate set (whole wave).  On return sIdx[i], sD2[i], sGIdx[i] (i < nn_k)
// hold the neighbours in order and, when sPos != nullptr, sPos[3 i ..] the global position the distance was measured
// to; returns the number of valid candidates over all K cells.
__device__ inline int select_topk(const pings_knn_map& m, Cand& c, int lane, long long* sIdx, float* sD2,
                                  long long* sGIdx, float* sPos) {
  const int count = __popcll(__ballot(c.cidx[0] >= 0)) + __popcll(__ballot(c.cidx[1] >= 0));
  // nn_k rounds: wave minimum of the distance bits, then the lowest candidate index among the
  // ties (cells 0..63 live in key[0] of lanes 0..63, cells 64.. in key[1])
  for (int i = 0; i < m.nn_k; ++i) {
    const unsigned best = wave_min_u32_all(min(c.key[0], c.key[1]));
    const unsigned long long b0 = __ballot(c.key[0] == best);
    const unsigned long long b1 = __ballot(c.key[1] == best);
    const int which = b0 != 0ull ? 0 : 1;
    const int owner = __ffsll((long long)(which ? b1 : b0)) - 1;
    if (lane == owner) {
      sIdx[i] = which ? c.cidx[1] : c.cidx[0];
      sGIdx[i] = which ? c.gidx[1] : c.gidx[0];
      sD2[i] = __uint_as_float(best);
      if (sPos) {
        sPos[3 * i] = which ? c.px[1] : c.px[0];
        sPos[3 * i + 1] = which ? c.py[1] : c.py[0];
        sPos[3 * i + 2] = which ? c.pz[1] : c.pz[0];
      }
      if (which) c.key[1] = 0xFFFFFFFFu; else c.key[0] = 0xFFFFFFFFu;
    }
  }
  return count;
}

// ---- two queries at once ---------------------------------------------------------------------------------------------
// The per-query search is a chain of dependent loads (block entry -> record) followed by the selection rounds; a wave
// that handles several queries per step (sdf_fwd_mfma.hip) gathers the candidates of TWO of them together, so that the
// block-entry loads of both and then the record loads of both are in flight at once: half the exposed round trips.
struct Cand32 {
  unsigned key[2];
  int cidx[2], gidx[2];
  float px[2], py[2], pz[2];
};

template <int NQ>
__device__ inline void candidates_blocks_multi(const pings_knn_map& m, const LaneCtx& lc, const float (&q)[NQ][3],
                                               Cand32 (&c)[NQ]) {
  const uint4* tab = reinterpret_cast<const uint4*>(m.blocks);
  const uint4* recs = reinterpret_cast<const uint4*>(m.block_records);
  const float lim = (float)(4 * BLOCK_COORD_LIMIT - 256);
  bool q_in[NQ];
  unsigned long long key[NQ][2];
  unsigned slot[NQ][2], bit[NQ][2];
  uint4 e[NQ][2];
#pragma unroll
  for (int u = 0; u < NQ; ++u) {
    const float fx = floorf(q[u][0] / m.resolution), fy = floorf(q[u][1] / m.resolution),
                fz = floorf(q[u][2] / m.resolution);
    q_in[u] = fabsf(fx) < lim && fabsf(fy) < lim && fabsf(fz) < lim;
    const int gx = q_in[u] ? (int)fx : 0, gy = q_in[u] ? (int)fy : 0, gz = q_in[u] ? (int)fz : 0;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int cx = gx + lc.dx[r][0], cy = gy + lc.dx[r][1], cz = gz + lc.dx[r][2];
      key[u][r] = block_key(cx >> 2, cy >> 2, cz >> 2);
      bit[u][r] = cell_bit(cx, cy, cz);
      slot[u][r] = block_slot(key[u][r], m.block_mask);
    }
  }
#pragma unroll
  for (int u = 0; u < NQ; ++u)
#pragma unroll
    for (int r = 0; r < 2; ++r) e[u][r] = tab[2 * (size_t)slot[u][r]];
  unsigned base[NQ][2];
  bool found[NQ][2];
#pragma unroll
  for (int u = 0; u < NQ; ++u)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      unsigned long long k = ((unsigned long long)e[u][r].y << 32) | e[u][r].x;
      while (k != key[u][r] && k != 0ull) {
        slot[u][r] = (slot[u][r] + 1u) & m.block_mask;
        e[u][r] = tab[2 * (size_t)slot[u][r]];
        k = ((unsigned long long)e[u][r].y << 32) | e[u][r].x;
      }
      const unsigned long long mask = ((unsigned long long)e[u][r].w << 32) | e[u][r].z;
      found[u][r] = q_in[u] && lc.has[r] && k == key[u][r] && ((mask >> bit[u][r]) & 1ull);
      const unsigned before = (unsigned)__popcll(mask & ((1ull << bit[u][r]) - 1ull));
      base[u][r] = found[u][r] ? tab[2 * (size_t)slot[u][r] + 1].x + before : 0u;
    }
  uint4 ra[NQ][2], rb[NQ][2];
#pragma unroll
  for (int u = 0; u < NQ; ++u)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      ra[u][r] = recs[2 * (size_t)base[u][r]];
      rb[u][r] = recs[2 * (size_t)base[u][r] + 1];
    }
#pragma unroll
  for (int u = 0; u < NQ; ++u)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const float px = __uint_as_float(ra[u][r].x), py = __uint_as_float(ra[u][r].y), pz = __uint_as_float(ra[u][r].z);
      const int gi = (int)ra[u][r].w, li = m.global2local ? (int)rb[u][r].x : gi;
      const float td = __uint_as_float(rb[u][r].y);
      const unsigned fl = rb[u][r].z;
      const float sx = px - q[u][0], sy = py - q[u][1], sz = pz - q[u][2];
      const float dd = (sx * sx + sy * sy) + sz * sz;
      bool ok = found[u][r];
      if (m.time_filtering) ok = ok && (fabsf(lc.cur_td - td) < m.diff_travel_dist_local);
      ok = ok && !(dd > m.max_valid_dist2);
      ok = ok && !(m.use_free_mask && (fl & 1u));
      ok = ok && !(m.use_valid_mask && !(fl & 2u));
      ok = ok && (li >= 0);
      c[u].px[r] = px; c[u].py[r] = py; c[u].pz[r] = pz;
      c[u].cidx[r] = ok ? li : -1;
      c[u].gidx[r] = ok ? gi : -1;
      c[u].key[r] = lc.has[r] ? __float_as_uint(ok ? dd : INVALID_D2) : 0xFFFFFFFFu;
    }
}

__device__ inline int select_topk32(const pings_knn_map& m, Cand32& c, int lane, long long* sIdx, float* sD2,
                                    long long* sGIdx, float* sPos) {
  const int count = __popcll(__ballot(c.cidx[0] >= 0)) + __popcll(__ballot(c.cidx[1] >= 0));
  for (int i = 0; i < m.nn_k; ++i) {
    const unsigned best = wave_min_u32_all(min(c.key[0], c.key[1]));
    const unsigned long long b0 = __ballot(c.key[0] == best);
    const unsigned long long b1 = __ballot(c.key[1] == best);
    const int which = b0 != 0ull ? 0 : 1;
    const int owner = __ffsll((long long)(which ? b1 : b0)) - 1;
    if (lane == owner) {
      sIdx[i] = (long long)(which ? c.cidx[1] : c.cidx[0]);
      sGIdx[i] = (long long)(which ? c.gidx[1] : c.gidx[0]);
      sD2[i] = __uint_as_float(best);
      if (sPos) {
        sPos[3 * i] = which ? c.px[1] : c.px[0];
        sPos[3 * i + 1] = which ? c.py[1] : c.py[0];
        sPos[3 * i + 2] = which ? c.pz[1] : c.pz[0];
      }
      if (which) c.key[1] = 0xFFFFFFFFu; else c.key[0] = 0xFFFFFFFFu;
    }
  }
  return count;
}

// The selections of two candidate sets in one loop: the two chains of wave minima / ballots are independent, so the
// scheduler interleaves them (a round is ~10 dependent cross-lane steps).
__device__ inline void select_topk32_pair(const pings_knn_map& m, Cand32 (&c)[2], int lane, long long* sIdx0,
                                          float* sD20, long long* sGIdx0, float* sPos0, long long* sIdx1, float* sD21,
                                          long long* sGIdx1, float* sPos1, int& count0, int& count1) {
  count0 = __popcll(__ballot(c[0].cidx[0] >= 0)) + __popcll(__ballot(c[0].cidx[1] >= 0));
  count1 = __popcll(__ballot(c[1].cidx[0] >= 0)) + __popcll(__ballot(c[1].cidx[1] >= 0));
  for (int i = 0; i < m.nn_k; ++i) {
    const unsigned bestA = wave_min_u32_all(min(c[0].key[0], c[0].key[1]));
    const unsigned bestB = wave_min_u32_all(min(c[1].key[0], c[1].key[1]));
    const unsigned long long a0 = __ballot(c[0].key[0] == bestA), a1 = __ballot(c[0].key[1] == bestA);
    const unsigned long long b0 = __ballot(c[1].key[0] == bestB), b1 = __ballot(c[1].key[1] == bestB);
    const int whichA = a0 != 0ull ? 0 : 1, whichB = b0 != 0ull ? 0 : 1;
    const int ownerA = __ffsll((long long)(whichA ? a1 : a0)) - 1;
    const int ownerB = __ffsll((long long)(whichB ? b1 : b0)) - 1;
    if (lane == ownerA) {
      sIdx0[i] = (long long)(whichA ? c[0].cidx[1] : c[0].cidx[0]);
      sGIdx0[i] = (long long)(whichA ? c[0].gidx[1] : c[0].gidx[0]);
      sD20[i] = __uint_as_float(bestA);
      if (sPos0) {
        sPos0[3 * i] = whichA ? c[0].px[1] : c[0].px[0];
        sPos0[3 * i + 1] = whichA ? c[0].py[1] : c[0].py[0];
        sPos0[3 * i + 2] = whichA ? c[0].pz[1] : c[0].pz[0];
      }
      if (whichA) c[0].key[1] = 0xFFFFFFFFu; else c[0].key[0] = 0xFFFFFFFFu;
    }
    if (lane == ownerB) {
      sIdx1[i] = (long long)(whichB ? c[1].cidx[1] : c[1].cidx[0]);
      sGIdx1[i] = (long long)(whichB ? c[1].gidx[1] : c[1].gidx[0]);
      sD21[i] = __uint_as_float(bestB);
      if (sPos1) {
        sPos1[3 * i] = whichB ? c[1].px[1] : c[1].px[0];
        sPos1[3 * i + 1] = whichB ? c[1].py[1] : c[1].py[0];
        sPos1[3 * i + 2] = whichB ? c[1].pz[1] : c[1].pz[0];
      }
      if (whichB) c[1].key[1] = 0xFFFFFFFFu; else c[1].key[0] = 0xFFFFFFFFu;
    }
  }
}

// Search + selection for one query (whole wave), nothing prefetched.
__device__ inline int knn_one_query(const pings_knn_map& m, const LaneCtx& lc, float qx, float qy, float qz,
                                    int lane, long long* sIdx, float* sD2, long long* sGIdx, float* sPos = nullptr) {
  Cand c;
  if (lc.use_blocks) {
    candidates_blocks(m, lc, qx, qy, qz, c);
  } else {
    candidates_table(m, lc, qx, qy, qz, c);
  }
  return select_topk(m, c, lane, sIdx, sD2, sGIdx, sPos);
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


// Row `lane` of W1[H][IN] into this lane's registers THROUGH LDS.  Read straight from global memory the 64 rows of
// a wave are 64 different lines per load instruction: IN instructions x 64 requests per wave, which at one query per
// wave (B = 16,384) cost more than the query.  Here the workgroup copies the matrix with coalesced row reads, then
// every lane reads its row from LDS at an odd stride (conflict-free).  `stage` holds 64 * IN_PAD floats and is free for
// other use after the call; every thread of the workgroup must call (two barriers inside).
template <int IN_PAD>
__device__ inline void load_w1_rows(const float* __restrict__ W1, int IN, int H, float* stage, float (&w1)[IN_PAD],
                                    float& wn0, float& wn1, float& wn2) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int stride = ((IN & 1) || IN + 1 > IN_PAD) ? IN : IN + 1;
  for (int h = wave; h < H; h += nw)
    for (int c = lane; c < IN; c += 64) stage[h * stride + c] = W1[h * IN + c];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < IN_PAD; ++i) w1[i] = (lane < H && i < IN) ? stage[lane * stride + i] : 0.f;
  // the three direction-input weights (columns IN-3 .. IN-1) once more as scalars: indexing w1[] with the runtime
  // feature count would send the array to scratch
  wn0 = lane < H ? stage[lane * stride + IN - 3] : 0.f;
  wn1 = lane < H ? stage[lane * stride + IN - 2] : 0.f;
  wn2 = lane < H ? stage[lane * stride + IN - 1] : 0.f;
  __syncthreads();
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
  PINGS_ARG_CHECK(!m->blocks || (m->block_records && m->blocks_ok && ((m->block_mask & (m->block_mask + 1u)) == 0u)),
                  "cell-block index needs its records, its status word and a 2^k - 1 mask");
  return PINGS_OK;
}

// Workgroups of a wave-per-query kernel: ONE resident round (what the occupancy calculator says fits on the chip at
// once), the waves loop over the remaining queries.  Measured on the fused SDF forward (1M points): B = 16,384 0.066 ->
// 0.047 ms, B = 131,072 0.310 -> 0.268 ms against a fixed 8,192-workgroup grid — the per-wave prologue (decoder weights,
// lane constants) is paid once per resident wave, and a second, nearly empty round is the worst case.
// PINGS_KNN_GRID_CAP overrides (A/B runs).
// The search-only kernels (knn_search, query_feature forward) have next to no prologue and measured the other way
// round (1M points, B = 131,072: 0.105 ms with 8,192 short-lived workgroups, 0.134 ms with one resident round): they
// pass kernel = nullptr and get the fixed 8,192 cap.
inline unsigned grid_for(long long B, const void* kernel = nullptr) {
  const long long blocks = (B + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  if (kernel == nullptr) return (unsigned)(blocks < 8192 ? blocks : 8192);
  static const long long env_cap = [] {
    const char* e = getenv("PINGS_KNN_GRID_CAP");
    return e ? atoll(e) : 0LL;
  }();
  long long cap = env_cap;
  if (cap <= 0) {
    static std::mutex mu;
    static std::unordered_map<const void*, long long> resident;
    std::lock_guard<std::mutex> lock(mu);
    auto it = resident.find(kernel);
    if (it == resident.end()) {
      long long v = 256LL * 4;
      int dev = 0, cus = 0, per = 0;
      if (kernel && hipGetDevice(&dev) == hipSuccess &&
          hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kernel, 64 * WAVES_PER_BLOCK, 0) == hipSuccess &&
          cus > 0 && per > 0)
        v = (long long)cus * per;
      it = resident.emplace(kernel, v).first;
    }
    cap = it->second;
  }
  return (unsigned)(blocks < cap ? blocks : cap);
}

// sdf_fwd_mfma.hip
bool sdf_forward_mfma_supported(const pings_knn_map* m, const pings_sdf_decoder* dec, const float* features);
int sdf_forward_mfma_launch(const pings_knn_map* m, const pings_sdf_decoder* dec, const float* features,
                            const float* points, const float* orientations, const float* certainties,
                            int32_t after_pgo, const float* queries, int64_t B, float* sdf, float* grad_x,
                            int64_t* nn_counts, float* certainty, int64_t* idx_out, float* w_out, float* sdf_std,
                            int64_t* gidx_out, hipStream_t st);

}  // namespace pings_knn
