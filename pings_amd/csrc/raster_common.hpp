// Shared definitions of the Gaussian(-surfel) rasteriser kernels.
//
// HBM layout (all buffers caller-owned, carved out of three opaque blobs):
//   geom blob   (per Gaussian)  rec[P][4] float4  : {mx,my,opacity,pz} {conic_x,conic_y,conic_z,rz}
//                                                    {r,g,b,q} {nx,ny,nz,-}
//                               rect[P]   uint4   : {inst_base, xmin|ymin<<16, xmax|ymax<<16, tiles}
//                               depth keys / ids (+ sorted copies), tiles-in-depth-order, their scan,
//                               radix-sort / scan scratch
//   binning blob (per instance) sorted Gaussian ids, ranges[num_tiles] uint2, per-instance blend-weight sums,
//                               tile keys / ids before sorting, scratch
//   image blob   (per pixel)    final_T[HW] float, n_contrib[HW] uint32
//
// Instances (Gaussian x touched tile) are created in DEPTH order (the P Gaussians are
// radix-sorted by their fp32 view depth first) and then stably radix-sorted by tile id
// only, so the final order inside a tile is (depth, Gaussian index) — the same order a
// 64-bit (tile<<32 | depth) key sort gives, at a quarter of the sort traffic.
#pragma once
#include <cstdlib>

#include "common.hpp"

namespace pings {
namespace raster {

constexpr int TILE = 16;
constexpr int BLOCK = TILE * TILE;  // 256 threads = 4 waves; wave w owns rows 4w..4w+3
constexpr float NEAR_Z = 0.2f;
constexpr float LOWPASS = 0.3f;
constexpr float ALPHA_MAX = 0.99f;
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float T_EPS = 1e-4f;
constexpr float DEPTH_ALPHA_EPS = 1e-10f;
constexpr float DEN_EPS = 1e-6f;
constexpr uint32_t CULLED_KEY = 0xFFFFFFFFu;
constexpr int DS_NB = 1 << 20;        // depth-sort buckets (2^18 left ~1.5k surfels per bucket on a rough fronto-parallel wall at 60 m: 0.43 ms of ranking)
constexpr uint32_t DS_LIMIT = 4096u;  // bucket population beyond which the library sort takes over
constexpr int DS_SHARDS = 64;  // single-address atomics serialise at ~12 ns each on this part: spread them
// header words: [0, 64) max key shards, [64, 128) max ~key shards, 128 kmin, 129 shift, 130 overflow flag
constexpr int DS_KMIN = 2 * DS_SHARDS, DS_SHIFT = DS_KMIN + 1, DS_FLAG = DS_KMIN + 2, DS_HEAD = DS_KMIN + 8;
// occlusion culling of instances (see occl_budget_kernel)
constexpr float OCC_THR = 9.5f;     // > -ln(T_EPS) = 9.21: transmittance bound that guarantees every pixel has stopped
constexpr float OCC_FIX = 4096.f;   // fixed-point scale of the budget (integer atomics: order independent)
constexpr uint16_t OCC_ALL = 0xFFFFu;

constexpr int MODE_SURFEL = 0;
constexpr int MODE_3DGS = 1;
// tile-rectangle rule of preprocess_kernel (raster_fwd.hip)
constexpr int RECT_TIGHT = 0, RECT_3SIGMA = 1, RECT_ELLIPSE = 2;

// 16 floats of per-instance gradient accumulated by the blend backward pass
// (one 64-B row per (tile, Gaussian) instance, summed per Gaussian afterwards).
// G_CONX/Y/Z hold dL/d(cov2D xx, xy, yy) — NOT the gradient w.r.t. the conic (see blend_bwd_kernel).
constexpr int GRAD_ROW = 16;
enum GradSlot {
  G_MX = 0, G_MY = 1, G_CONX = 2, G_CONY = 3, G_CONZ = 4, G_OPAC = 5,
  G_R = 6, G_G = 7, G_B = 8, G_NX = 9, G_NY = 10, G_NZ = 11, G_Q = 12, G_PZ = 13,
  G_ZLO = 14, G_ZHI = 15
};

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base(reinterpret_cast<char*>(p)) {}
  template <typename T>
  T* take(size_t count) {
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off = align_up(off + count * sizeof(T));
    return p;
  }
};

struct FrameSummary {   // the record of a frame's one host read-back (frame_summary_kernel)
  uint32_t total, overflow;
  unsigned long long stats[3];
  int32_t aux[8];
  uint32_t seq;   // written last (system-scope release): the host polls it instead of blocking in the runtime
  uint32_t pad;
};

struct GeomState {
  float4* rec;
  uint4* rect;
  uint32_t *depth_key, *depth_key_sorted, *gidx, *gidx_sorted;
  uint32_t* rank_of;                        // [P] depth rank of each surviving Gaussian (inverse of gidx_sorted)
  uint32_t *tiles_sorted, *offsets_sorted;  // per depth rank: KEPT tiles of the Gaussian and their inclusive scan
  uint32_t* occ_bucket;                     // [occ_nb][num_tiles] fixed-point opacity budget per (rank bucket, tile)
  uint16_t* occ_bsat;                       // [num_tiles] last rank bucket a tile still needs (0xFFFF = all)
  uint32_t* nvalid;                         // [1] Gaussians that survived culling (= ranks with a real depth key)
  uint32_t* ds_head;                        // depth sort: header (key range shards, range, overflow flag), then
  uint32_t *ds_cnt, *ds_fill, *ds_off;      //   [DS_NB + blocks] bucket / per-block culled counts, [DS_NB] fill cursors,
  char* zero_begin; size_t zero_bytes;      // occ_bucket .. stats .. ds_head: the frame's one memset
  size_t ds_words;                          //   [DS_NB + blocks + 1] their exclusive scan; ds_words = memset extent
  uint32_t* ds_idx;                         //   [P] Gaussian ids in bucket order (keys go to depth_key_sorted)
  unsigned long long* stats;                // [2] pairs before occlusion culling, visible Gaussians
  FrameSummary* summary;
  int occ_nb;
  char* temp;
  size_t temp_bytes;
  size_t total;
};

struct BinState {
  // pre-sort, slot order (= depth rank, then tile order inside the Gaussian's rectangle): tile_key[slot],
  // gval[slot] = Gaussian id, slot_val[slot] = slot;  sorted by tile: point_list[i] = slot
  uint32_t *tile_key, *tile_key_sorted, *gval, *slot_val, *point_list;
  uint2* ranges;
  float* inst_w;       // [I+1] per-instance sum of blend weights (0 = instance never blended)
  uint8_t* inst_qmask; // [I+1] 8x8 quadrants of the tile in which the instance blended something (bit q = qx + 2 qy)
  float* inst_wq;      // [I][4] per-(instance, quadrant) sums of the wave-per-quadrant forward kernel, folded into
  uint32_t* inst_cntq; //        inst_w / inst_cnt / inst_qmask by combine_quadrants_kernel
  uint32_t* inst_cnt;  // [I]   per-instance pixel count with transmittance > 0.5 (3DGS)
  uint32_t* tile_order;  // [2][num_tiles] tiles by descending work: [0] by list length (forward), [1] by the largest
                         // per-pixel contributor count (backward) — longest-processing-time-first dispatch order
  uint32_t* tile_work;   // [num_tiles] scratch of the two orderings
  // segmented blend of long tile lists (raster_fwd.hip, "forward of LONG tile lists")
  uint32_t seg_max_units;      // capacity: every list of more than 2 * SEG entries cut into SEG-entry units
  uint32_t *seg_head, *seg_unit_tile, *seg_unit_seg, *seg_tile_unit0;
  float *seg_P, *seg_slab;
  uint16_t* seg_rel;           // [units][4][SEG] pass T's compacted list: offsets (in the segment) of the entries that can
  uint32_t* seg_nrel;          // [units][4]      reach the quadrant, and how many — pass B walks these instead of re-testing
  char* temp;
  size_t temp_bytes;
  size_t total;
};

struct ImageState {
  float* final_T;
  uint32_t* n_contrib;
  size_t total;
};

uint32_t blend_segment_entries();
GeomState carve_geom(void* blob, int P, int num_tiles);
// tile_order[i] = i-th tile in descending `work` (ties in any order: scheduling only, results do not depend on it)
// n_long (optional): receives min(number of tiles with work >= long_thr, long_max) — they lead the order
int launch_tile_order(const uint32_t* work, int num_tiles, uint32_t* order, hipStream_t st, uint32_t* n_long = nullptr,
                      uint32_t long_thr = 0, uint32_t long_max = 0);
int occlusion_buckets(int num_tiles);
BinState carve_binning(void* blob, int64_t I, int num_tiles);
ImageState carve_image(void* blob, int W, int H);

// wave64 sum through DPP; the total ends up in lane 63.
template <int CTRL>
__device__ inline float dpp_mov(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// 4-way lane-dependent select t[(m + K) & 3], m = 2*m1 + m0 (3 v_cndmask).
template <int K>
__device__ inline float sel4(const float (&t)[4], bool m0, bool m1) {
  const float lo = m0 ? t[(1 + K) & 3] : t[K & 3];
  const float hi = m0 ? t[(3 + K) & 3] : t[(2 + K) & 3];
  return m1 ? hi : lo;
}

// Sums v[0..15] over the 64 lanes with a transposed reduce-scatter: two quad_perm exchange
// steps (16 -> 8 -> 4 values per lane), one rotate step inside each 16-lane row (4 -> 1) and
// two cross-row adds.  Afterwards lane l of EVERY row holds the wave total of slot
//   8*(l&1) + 4*((l>>1)&1) + ((l>>2)&3).
__device__ inline float wave_reduce16(const float (&v)[16], int lane) {
  const bool b0 = lane & 1, b1 = lane & 2;
  float u[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float send = b0 ? v[k] : v[k + 8];
    const float keep = b0 ? v[k + 8] : v[k];
    u[k] = keep + dpp_mov<0xb1>(send);  // quad_perm [1,0,3,2]
  }
  float t[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float send = b1 ? u[k] : u[k + 4];
    const float keep = b1 ? u[k + 4] : u[k];
    t[k] = keep + dpp_mov<0x4e>(send);  // quad_perm [2,3,0,1]
  }
  // lanes {c, c+4, c+8, c+12} of a row share the slot base; lane c+4m ends with slot base+m.
  // row_ror:4k delivers the value of lane (l - 4k) mod 16, whose group index is m-k, so a
  // sender with group index m' offers t[(m'+k)&3].
  const bool m0 = lane & 4, m1 = lane & 8;
  float r = sel4<0>(t, m0, m1);
  r += dpp_mov<0x124>(sel4<1>(t, m0, m1));  // row_ror:4
  r += dpp_mov<0x128>(sel4<2>(t, m0, m1));  // row_ror:8
  r += dpp_mov<0x12c>(sel4<3>(t, m0, m1));  // row_ror:12
  // cross-row adds r[l] + r[l ^ 16], then + the same of l ^ 32, on the vector ALU: gfx950's v_permlane16_swap /
  // v_permlane32_swap exchange rows / halves between two registers, no LDS round trip (a ds_bpermute costs ~64 cycles
  // of latency, twice per record here).  Inline asm: this hipcc maps both results of the builtin to one register.
  {
    float a = r, b = r;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    r = a + b;
    a = r; b = r;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    r = a + b;
  }
  return r;
}

__device__ inline float wave_reduce_sum_dpp(float v) {
  // quad_perm [1,0,3,2], [2,3,0,1], row_ror:4, row_ror:8, row_bcast:15, row_bcast:31
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));
  return v;
}

// wave64 integer sum through DPP; the total ends up in lane 63.
__device__ inline uint32_t wave_reduce_sum_u32_dpp(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
  return v;
}

__device__ inline float wave_sum_to_all(float v) {
  v = wave_reduce_sum_dpp(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ---------------------------------------------------------------- sub-tile culling
// A record contributes to a pixel only if alpha = min(0.99, o exp(power)) >= 1/255, i.e.
// q(d) = cx dx^2 + 2 cy dx dy + cz dy^2 <= 2 ln(255 o).  `footprint_misses_rect` is true when the smallest q over
// a closed pixel rectangle exceeds that bound by more than the fp32 error the blend kernels' own evaluation of
// `power` can have (the terms cancel for large anisotropic footprints, so the margin scales with their
// magnitude): no pixel of the rectangle can pass the alpha test, and skipping the record for the wave that owns
// the rectangle leaves every output bit unchanged.
__device__ inline float edge_min_q(float a, float b, float c, float d_fixed, float lo, float hi) {
  // q along the edge {fixed offset d_fixed on one axis, free offset t in [lo, hi] on the other}:
  //   q(t) = a d^2 + 2 b d t + c t^2, minimised at t* = -b d / c, clamped to the edge
  // (1-ulp reciprocal: an error dt in t raises q by c dt^2 ~ 1e-14 c t^2, far below the margin)
  const float t = fminf(fmaxf(-(b * d_fixed) * __builtin_amdgcn_rcpf(c), lo), hi);
  const float q0 = a * d_fixed * d_fixed, q1 = 2.f * b * d_fixed * t, q2 = c * t * t;
  return (q0 + q1 + q2) - 2e-5f * (q0 + fabsf(q1) + q2);
}

__device__ inline bool footprint_misses_rect(float mx, float my, float cx, float cy, float cz, float thr,
                                             float x0, float x1, float y0, float y1) {
  if (mx >= x0 && mx <= x1 && my >= y0 && my <= y1) return false;  // centre inside: q = 0
  const float lx = x0 - mx, hx = x1 - mx, ly = y0 - my, hy = y1 - my;
  float m = edge_min_q(cx, cy, cz, lx, ly, hy);             // edge x = x0
  m = fminf(m, edge_min_q(cx, cy, cz, hx, ly, hy));         // edge x = x1
  m = fminf(m, edge_min_q(cz, cy, cx, ly, lx, hx));         // edge y = y0
  m = fminf(m, edge_min_q(cz, cy, cx, hy, lx, hx));         // edge y = y1
  return m > thr;                                           // NaN -> false (keep)
}

// Bit q = 1 iff the record may touch the 8x8 pixel quadrant q = (qx + 2 qy) of the 16x16 tile at (X0, Y0).
__device__ inline uint32_t quadrant_mask(float mx, float my, float opacity, float cx, float cy, float cz,
                                         float X0, float Y0) {
  const float thr = 2.f * __logf(255.f * opacity) + 2e-3f;
  uint32_t m = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float x0 = X0 + (float)(8 * (q & 1)), y0 = Y0 + (float)(8 * (q >> 1));
    if (!footprint_misses_rect(mx, my, cx, cy, cz, thr, x0, x0 + 7.f, y0, y0 + 7.f)) m |= 1u << q;
  }
  return m;
}

}  // namespace raster
}  // namespace pings
