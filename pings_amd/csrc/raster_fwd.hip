// Gaussian(-surfel) rasteriser, forward pass, for gfx950.
//
// Pipeline (one HIP stream, no host round trip except the 8-byte instance count):
//   preprocess_kernel   per Gaussian: camera transform, EWA projection, cull, tile rect,
//                       64-B blend record, fp32 depth key                     [HBM: 56 B in, 96 B out]
//   radix sort (P)      Gaussians by depth key (hipcub / rocPRIM)
//   gather + scan       tiles-per-Gaussian in depth order -> instance offsets
//   duplicate_kernel    emits (tile id, Gaussian id) instances in depth order  [8 B per instance]
//   radix sort (I)      STABLE sort by tile id only (ceil(log2(tiles)) bits)    [the only multi-pass
//                       traffic over the instance list; 32-bit keys instead of 64-bit]
//   tile_ranges_kernel  [start,end) of every tile in the sorted list
//   blend_fwd_kernel    one 256-thread workgroup (4 waves) per 16x16 tile; 256 records per round
//                       are staged in LDS with coalesced 16-B loads and broadcast-read by the
//                       four waves; front-to-back compositing per pixel, wave-uniform skip of
//                       Gaussians no lane of the wave sees, workgroup-uniform early exit.
//   per-Gaussian sums   `contributions` / `n_touched` are reduced per (tile, Gaussian) instance on
//                       chip (DPP wave sum + LDS), stored once per instance, then summed per
//                       Gaussian: no global atomics, bitwise reproducible.
//
// The arithmetic of preprocess_kernel follows oracle/raster_cpu.py:preprocess op for op
// (this TU is compiled with -ffp-contract=off; IEEE divide / sqrt), so radii, tile
// rectangles and the sort order match the fp32 oracle bit for bit.
#include <hipcub/hipcub.hpp>

#include <chrono>
#include <algorithm>

#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "raster_common.hpp"

namespace pings {
namespace raster {

struct KParams {
  int P, W, H, gx, gy;
  int front_only;
  float occ_amin;    // smallest per-tile alpha bound that is worth an occlusion-budget entry (see preprocess_kernel)
  int rect_rule;     // RECT_TIGHT (default) | RECT_3SIGMA | RECT_ELLIPSE, see preprocess_kernel (PINGS_RASTER_RECT)
  float fx, fy, limx, limy, scale_mod;
  const float* view;
  const float* proj_raw;
  const float* bg;
  const float* prcp;
  const int32_t* live;   // device word: of the first `dyn_rows` Gaussians only rows [0, *live) exist (nullable)
  int dyn_rows;
};

// ---------------------------------------------------------------- blob carving
static size_t sort_temp_bytes(int64_t n) {
  size_t a = 0, b = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                     (uint32_t*)nullptr, (uint32_t*)nullptr, (int)n, 0, 32);
  (void)hipcub::DeviceScan::InclusiveSum(nullptr, b, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                         (int)std::max<int64_t>(n, DS_NB + n / 256 + 2));
  size_t c = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, c, (uint16_t*)nullptr, (uint16_t*)nullptr,
                                           (uint32_t*)nullptr, (uint32_t*)nullptr, (int)n, 0, 16);
  if (c > a) a = c;
  return align_up(a > b ? a : b) + 256;
}

// rank buckets per tile of the occlusion budget: ~2M counters in total, 32..256 per tile
int occlusion_buckets(int num_tiles) {
  int nb = 256;
  while (nb > 32 && (size_t)nb * (size_t)num_tiles > ((size_t)1 << 21)) nb >>= 1;
  return nb;
}

// entries per segment of the segmented blend of long tile lists (0 = off); PINGS_BLEND_SEG overrides (tests, A/B)
uint32_t blend_segment_entries() {
  if (const char* e = getenv("PINGS_BLEND_SEG")) return (uint32_t)atoi(e);
  return 512u;
}

GeomState carve_geom(void* blob, int P, int num_tiles) {
  Carver c(blob);
  GeomState g;
  const size_t n = (size_t)(P > 0 ? P : 1);
  const size_t nt = (size_t)(num_tiles > 0 ? num_tiles : 1);
  g.rec = c.take<float4>(4 * n);
  g.rect = c.take<uint4>(n);
  g.depth_key = c.take<uint32_t>(n);
  g.depth_key_sorted = c.take<uint32_t>(n);
  g.gidx = c.take<uint32_t>(n);
  g.gidx_sorted = c.take<uint32_t>(n);
  g.rank_of = c.take<uint32_t>(n);
  g.tiles_sorted = c.take<uint32_t>(n);
  g.offsets_sorted = c.take<uint32_t>(n);
  g.occ_nb = occlusion_buckets((int)nt);
  // occ_bucket, stats and ds_head start every frame at zero: adjacent, ONE memset (zero_begin .. zero_end)
  g.occ_bucket = c.take<uint32_t>(nt * (size_t)g.occ_nb);
  g.stats = c.take<unsigned long long>(3 * 256);  // sharded {pairs before culling, visible Gaussians, kept pairs}
  const size_t nblk = (n + 255) / 256;                  // workgroups of the per-Gaussian kernels
  g.ds_words = DS_HEAD + (size_t)DS_NB + nblk;          // header, counts (+ per-block culled)
  g.ds_head = c.take<uint32_t>(g.ds_words);
  g.ds_cnt = g.ds_head ? g.ds_head + DS_HEAD : nullptr;
  g.zero_begin = reinterpret_cast<char*>(g.occ_bucket);
  g.zero_bytes = g.ds_head ? (size_t)(reinterpret_cast<char*>(g.ds_head + g.ds_words) - g.zero_begin) : 0;
  g.occ_bsat = c.take<uint16_t>(nt);
  g.nvalid = c.take<uint32_t>(1);
  g.ds_off = c.take<uint32_t>((size_t)DS_NB + nblk + 1);
  g.ds_idx = c.take<uint32_t>(n);
  g.summary = c.take<FrameSummary>(1);            // what the frame's one read-back fetches
  g.temp_bytes = sort_temp_bytes((int64_t)n);
  g.temp = c.take<char>(g.temp_bytes);
  g.total = c.off;
  return g;
}

BinState carve_binning(void* blob, int64_t I, int num_tiles) {
  Carver c(blob);
  BinState b;
  const size_t n = (size_t)(I > 0 ? I : 1);
  b.point_list = c.take<uint32_t>(n);
  // ranges, inst_w, inst_qmask (and inst_cnt, 3DGS) start at zero: adjacent, ONE memset from `ranges`
  b.ranges = c.take<uint2>((size_t)num_tiles);
  b.inst_w = c.take<float>(n + 1);
  b.inst_qmask = c.take<uint8_t>(n + 1);
  b.inst_cnt = c.take<uint32_t>(n);
  b.inst_wq = c.take<float>(4 * n);
  b.inst_cntq = c.take<uint32_t>(4 * n);
  b.tile_key = c.take<uint32_t>(n);
  b.tile_key_sorted = c.take<uint32_t>(n);
  b.gval = c.take<uint32_t>(n);
  b.slot_val = c.take<uint32_t>(n);
  b.tile_order = c.take<uint32_t>(2 * (size_t)num_tiles + 4);   // + the backward pass' long-list tile count
  b.tile_work = c.take<uint32_t>((size_t)num_tiles);
  // units of SEG entries for lists beyond 2 SEG: at most I / SEG + I / (2 SEG) of them
  {
    const size_t sg = (size_t)blend_segment_entries();
    b.seg_max_units = sg ? (uint32_t)(n / sg + n / (2 * sg) + 64) : 1u;
    if (b.seg_max_units > (1u << 20)) b.seg_max_units = 1u << 20;   // forced tiny segments (tests): capacity is checked on device
  }
  b.seg_head = c.take<uint32_t>(4);
  b.seg_unit_tile = c.take<uint32_t>(b.seg_max_units);
  b.seg_unit_seg = c.take<uint32_t>(b.seg_max_units);
  b.seg_tile_unit0 = c.take<uint32_t>((size_t)num_tiles);
  b.seg_P = c.take<float>((size_t)b.seg_max_units * 256);
  b.seg_slab = c.take<float>((size_t)b.seg_max_units * 256 * 10);
  {
    const size_t sg = (size_t)blend_segment_entries();
    const bool reuse = sg > 0 && sg <= 65535;    // 16-bit offsets
    b.seg_rel = c.take<uint16_t>(reuse ? (size_t)b.seg_max_units * 4 * sg : 1);
    b.seg_nrel = c.take<uint32_t>((size_t)b.seg_max_units * 4);
  }
  b.temp_bytes = sort_temp_bytes((int64_t)n);
  b.temp = c.take<char>(b.temp_bytes);
  b.total = c.off;
  return b;
}

ImageState carve_image(void* blob, int W, int H) {
  Carver c(blob);
  ImageState im;
  im.final_T = c.take<float>((size_t)W * H);
  im.n_contrib = c.take<uint32_t>((size_t)W * H);
  im.total = c.off;
  return im;
}

// ---------------------------------------------------------------- kernels
__device__ inline void to_camera(const float* __restrict__ V, float x, float y, float z, float& px,
                                 float& py, float& pz) {
  px = ((V[0] * x + V[4] * y) + V[8] * z) + V[12];
  py = ((V[1] * x + V[5] * y) + V[9] * z) + V[13];
  pz = ((V[2] * x + V[6] * y) + V[10] * z) + V[14];
}

__global__ __launch_bounds__(256) void mark_visible_kernel(const float* __restrict__ pos, int N,
                                                            const float* __restrict__ V,
                                                            const float* __restrict__ Pm,
                                                            uint8_t* __restrict__ present, int depth_only) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float px, py, pz;
  to_camera(V, pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], px, py, pz);
  const float hx = ((Pm[0] * px + Pm[4] * py) + Pm[8] * pz) + Pm[12];
  const float hy = ((Pm[1] * px + Pm[5] * py) + Pm[9] * pz) + Pm[13];
  const float hw = ((Pm[3] * px + Pm[7] * py) + Pm[11] * pz) + Pm[15];
  const float pw = 1.0f / (hw + 1e-7f);
  const float nx = hx * pw, ny = hy * pw;
  present[i] = (pz > NEAR_Z) && (depth_only || ((nx >= -1.3f) && (nx <= 1.3f) && (ny >= -1.3f) && (ny <= 1.3f)));
}

template <int MODE>
__global__ __launch_bounds__(256) void preprocess_kernel(
    KParams p, const float* __restrict__ means3D, const float* __restrict__ colors,
    const float* __restrict__ opacities, const float* __restrict__ scales,
    const float* __restrict__ rotations, float4* __restrict__ rec, uint4* __restrict__ rect,
    uint32_t* __restrict__ depth_key, uint32_t* __restrict__ gidx, int32_t* __restrict__ radii) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= p.P) return;
  const float* V = p.view;
  const float* Pm = p.proj_raw;

  gidx[g] = (uint32_t)g;
  // defaults for a culled Gaussian
  uint32_t key = CULLED_KEY;
  uint4 rc = make_uint4(0u, 0u, 0u, 0u);
  int rad = 0;

  float px, py, pz;
  to_camera(V, means3D[3 * g], means3D[3 * g + 1], means3D[3 * g + 2], px, py, pz);
  bool ok = pz > NEAR_Z;
  // rows of a worst-case sized buffer behind the count the producer left on the device: culled (their contents are
  // never interpreted: every use below is guarded by `ok`)
  if (p.live && g < p.dyn_rows && g >= *p.live) ok = false;

  const float hx = ((Pm[0] * px + Pm[4] * py) + Pm[8] * pz) + Pm[12];
  const float hy = ((Pm[1] * px + Pm[5] * py) + Pm[9] * pz) + Pm[13];
  const float hw = ((Pm[3] * px + Pm[7] * py) + Pm[11] * pz) + Pm[15];
  const float pw = 1.0f / (hw + 1e-7f);
  const float ndx = hx * pw, ndy = hy * pw;
  const float mx = ((ndx + 1.0f) * (float)p.W - 1.0f) * 0.5f;
  const float my = ((ndy + 1.0f) * (float)p.H - 1.0f) * 0.5f;

  const float qr = rotations[4 * g], qx = rotations[4 * g + 1], qy = rotations[4 * g + 2],
              qz = rotations[4 * g + 3];
  const float R00 = 1.0f - 2.0f * (qy * qy + qz * qz), R01 = 2.0f * (qx * qy - qr * qz),
              R02 = 2.0f * (qx * qz + qr * qy);
  const float R10 = 2.0f * (qx * qy + qr * qz), R11 = 1.0f - 2.0f * (qx * qx + qz * qz),
              R12 = 2.0f * (qy * qz - qr * qx);
  const float R20 = 2.0f * (qx * qz - qr * qy), R21 = 2.0f * (qy * qz + qr * qx),
              R22 = 1.0f - 2.0f * (qx * qx + qy * qy);
  const float S0 = p.scale_mod * scales[3 * g], S1 = p.scale_mod * scales[3 * g + 1],
              S2 = p.scale_mod * scales[3 * g + 2];
  const float RS[3][3] = {{R00 * S0, R01 * S1, R02 * S2},
                          {R10 * S0, R11 * S1, R12 * S2},
                          {R20 * S0, R21 * S1, R22 * S2}};
  // Wc[a][b] = V[b][a] = V[4*b + a]
  float Mc[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int k = 0; k < 3; ++k)
      Mc[a][k] = (V[a] * RS[0][k] + V[4 + a] * RS[1][k]) + V[8 + a] * RS[2][k];

  const float tx = fminf(p.limx, fmaxf(-p.limx, px / pz)) * pz;
  const float ty = fminf(p.limy, fmaxf(-p.limy, py / pz)) * pz;
  const float J00 = p.fx / pz;
  const float J02 = -(p.fx * tx) / (pz * pz);
  const float J11 = p.fy / pz;
  const float J12 = -(p.fy * ty) / (pz * pz);
  float T0[3], T1[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    T0[k] = J00 * Mc[0][k] + J02 * Mc[2][k];
    T1[k] = J11 * Mc[1][k] + J12 * Mc[2][k];
  }
  const float cxx = ((T0[0] * T0[0] + T0[1] * T0[1]) + T0[2] * T0[2]) + LOWPASS;
  const float cxy = (T0[0] * T1[0] + T0[1] * T1[1]) + T0[2] * T1[2];
  const float cyy = ((T1[0] * T1[0] + T1[1] * T1[1]) + T1[2] * T1[2]) + LOWPASS;
  const float det = cxx * cyy - cxy * cxy;
  ok = ok && (det != 0.0f);
  const float det_inv = 1.0f / (det != 0.0f ? det : 1.0f);
  const float conic_x = cyy * det_inv, conic_y = -cxy * det_inv, conic_z = cxx * det_inv;
  const float mid = 0.5f * (cxx + cyy);
  const float lam = mid + sqrtf(fmaxf(mid * mid - det, 0.1f));
  const float radius = ceilf(3.0f * sqrtf(lam));
  ok = ok && isfinite(mx) && isfinite(my) && isfinite(radius);

  float nx = 0.f, ny = 0.f, nz = 0.f, q = 0.f, rz = 0.f;
  if (MODE == MODE_SURFEL) {
    nx = (V[0] * R02 + V[4] * R12) + V[8] * R22;
    ny = (V[1] * R02 + V[5] * R12) + V[9] * R22;
    nz = (V[2] * R02 + V[6] * R12) + V[10] * R22;
    q = (nx * px + ny * py) + nz * pz;
    if (p.front_only) {
      ok = ok && !(q >= 0.0f);
    } else if (q > 0.0f) {
      nx = -nx; ny = -ny; nz = -nz; q = -q;
    }
    rz = 3.0f * fmaxf(S0, S1);
  }

  // Tile rectangle.  Three rules (KParams::rect_rule, PINGS_RASTER_RECT):
  //   RECT_3SIGMA   the published 3DGS getRect(): the square of half-width ceil(3 sqrt(lambda_max)) around the centre,
  //                 upper bound (m + r + TILE - 1) / TILE.
  //   RECT_TIGHT    (default) that square INTERSECTED with the bounding box of the region in which the blend kernels'
  //                 own fp32 evaluation of alpha can reach 1/255.  A tile outside that box holds no pixel that passes
  //                 the alpha test, so dropping it changes no output bit: images, per-Gaussian sums and gradients equal
  //                 RECT_3SIGMA's (tests/test_raster.py::test_default_rectangle_is_lossless).  `radii` is the published
  //                 one: a Gaussian whose square touches the image keeps radius > 0 even if no tile is left.
  //   RECT_ELLIPSE  rounds 1-3: bounding box of the ellipse cut at min(3 sigma, alpha = 1/255) — drops the
  //                 alpha < ~0.011 tail the published rule blends (images differ by up to 3.4e-3); opt-in only.
  // Box of RECT_TIGHT: with the conic (cx, cy, cz) the kernels read, q(d) = cx dx^2 + 2 cy dx dy + cz dy^2 >= sx dx^2,
  // sx = cx - cy^2 / cz (Schur complement; = 1 / cov_xx in exact arithmetic), and the kernels' evaluation q^ of q is
  // off by at most 1e-6 S, S = cx dx^2 + 2 |cy dx dy| + cz dy^2 <= 4 (cx / sx) q (seven roundings of relative 6e-8 on
  // each term; 1e-6 leaves a factor two).  A pixel passes only if q^ <= thr = 2 ln(255 o) (+ 2e-3 for the fast exp /
  // log), hence only if dx^2 <= thr / (sx (1 - 4e-6 cx / sx)).  sx is taken 2e-6 cx low (its own three roundings); a
  // footprint so thin that cx / sx > 1e5 (or a non-finite / non-positive sx) keeps the whole square on that axis.
  const float opac = opacities[g];
  const float k2 = fminf(2.0f * logf(255.0f * opac), 9.0f);
  const bool want = ok;                 // survives culling: `radii` of the published rule depends on the square only
  ok = ok && (k2 > 0.0f || p.rect_rule == RECT_3SIGMA);
  if (want) {
    const bool square = p.rect_rule != RECT_ELLIPSE;
    const float ex = square ? radius : sqrtf(k2 * cxx), ey = square ? radius : sqrtf(k2 * cyy);
    const float up = square ? (float)(TILE - 1) : (float)TILE;
    const float fgx = (float)p.gx, fgy = (float)p.gy;
    int xmin = (int)fminf(fmaxf(floorf((mx - ex) / (float)TILE), 0.0f), fgx);
    int xmax = (int)fminf(fmaxf(floorf(((mx + ex) + up) / (float)TILE), 0.0f), fgx);
    int ymin = (int)fminf(fmaxf(floorf((my - ey) / (float)TILE), 0.0f), fgy);
    int ymax = (int)fminf(fmaxf(floorf(((my + ey) + up) / (float)TILE), 0.0f), fgy);
    const bool sq_tiles = (xmax - xmin) * (ymax - ymin) > 0;
    if (p.rect_rule == RECT_TIGHT) {
      if (sq_tiles) rad = (int)radius;
      if (ok) {
        const float thr = 2.0f * logf(255.0f * opac) + 2e-3f;
        const float sx = (conic_x - (conic_y * conic_y) / conic_z) - 2e-6f * conic_x;
        const float sy = (conic_z - (conic_y * conic_y) / conic_x) - 2e-6f * conic_z;
        const float kx = conic_x / sx, ky = conic_z / sy;
        if (sx > 0.0f && kx <= 1e5f) {   // NaN -> keep the square
          const float bx = sqrtf(thr / (sx * (1.0f - 4e-6f * kx))) * 1.000001f + 1e-3f;
          xmin = max(xmin, (int)fminf(fmaxf(floorf((mx - bx) / (float)TILE), 0.0f), fgx));
          xmax = min(xmax, (int)fminf(fmaxf(floorf(((mx + bx) + (float)TILE) / (float)TILE), 0.0f), fgx));
        }
        if (sy > 0.0f && ky <= 1e5f) {
          const float by = sqrtf(thr / (sy * (1.0f - 4e-6f * ky))) * 1.000001f + 1e-3f;
          ymin = max(ymin, (int)fminf(fmaxf(floorf((my - by) / (float)TILE), 0.0f), fgy));
          ymax = min(ymax, (int)fminf(fmaxf(floorf(((my + by) + (float)TILE) / (float)TILE), 0.0f), fgy));
        }
        xmax = max(xmax, xmin);
        ymax = max(ymax, ymin);
      }
    }
    if (ok) {
      const int tiles = (xmax - xmin) * (ymax - ymin);
      if (tiles > 0) {
        key = __float_as_uint(pz);
        // rect.x: the INNER rectangle of the occlusion-budget pass, as four 8-bit margins inside the tile rectangle
        // (left, right, top, bottom; margins that meet = empty).  A tile adds to the budget only if all of its pixels
        // pass the alpha test (tile_min_alpha > 0), i.e. its four corners lie inside the alpha >= 1/255 ellipse, hence
        // inside that ellipse's bounding box [m - e', m + e'], e' = sqrt(thr cov) with the UNCAPPED thr = 2 ln(255 o)
        // (the tile rectangle itself is cut at 3 sigma).  Typically the rectangle shrinks by a tile on every side —
        // half the pairs of a 7 x 7 rectangle.  The bounds are widened by 0.02 tile against rounding; a Gaussian whose
        // minor semi-axis sqrt(thr lambda_min) is below 7.5 px (threshold 50 = 7.07^2: slack again) cannot hold the
        // 15 x 15 pixel square of a tile at all.
        const float thr_u = 2.0f * logf(opac / p.occ_amin);
        const float lam_min = mid - sqrtf(fmaxf(mid * mid - det, 0.0f));
        uint32_t inner = 0xFFFFFFFFu;   // empty
        if (thr_u * lam_min >= 50.0f) {
          const float exu = sqrtf(thr_u * cxx), eyu = sqrtf(thr_u * cyy);
          const int ix0 = max(xmin, (int)ceilf((mx - exu) / (float)TILE - 0.02f));
          const int ix1 = min(xmax, (int)floorf((mx + exu - 15.0f) / (float)TILE + 0.02f) + 1);
          const int iy0 = max(ymin, (int)ceilf((my - eyu) / (float)TILE - 0.02f));
          const int iy1 = min(ymax, (int)floorf((my + eyu - 15.0f) / (float)TILE + 0.02f) + 1);
          if (ix1 > ix0 && iy1 > iy0)
            inner = (uint32_t)min(ix0 - xmin, 255) | ((uint32_t)min(xmax - ix1, 255) << 8) |
                    ((uint32_t)min(iy0 - ymin, 255) << 16) | ((uint32_t)min(ymax - iy1, 255) << 24);
        }
        const uint32_t covers = inner;
        rc = make_uint4(covers, (uint32_t)xmin | ((uint32_t)ymin << 16),
                        (uint32_t)xmax | ((uint32_t)ymax << 16), (uint32_t)tiles);
        rad = (int)radius;
      }
    }
  }
  depth_key[g] = key;
  rect[g] = rc;
  radii[g] = rad;
  if (key != CULLED_KEY) {  // nobody reads the record of a culled Gaussian (two thirds of Metric-1's cloud)
    rec[4 * g + 0] = make_float4(mx, my, opac, pz);
    rec[4 * g + 1] = make_float4(conic_x, conic_y, conic_z, rz);
    rec[4 * g + 2] = make_float4(colors[3 * g], colors[3 * g + 1], colors[3 * g + 2], q);
    rec[4 * g + 3] = make_float4(nx, ny, nz, 0.0f);
  }
}

// Depth ranks -> lanes.  The nearest Gaussians cover the most tiles and sit at neighbouring ranks: dealt out in rank
// order they would all land in the first waves.  The first quarter of the ranks is therefore dealt round-robin,
// one rank per wave at a time (lane l < 16 of wave w: rank l num_waves + w); behind it the footprints are small and
// even, and ranks go out in runs of 16 (quarter q >= 1 of wave w: ranks 16 (q num_waves + w) .. + 15), so that the
// per-rank loads and stores are 64-byte segments instead of one memory transaction per lane.  (Runs of 16 from rank
// 0 on: Metric-1's duplicate pass 0.054 -> 0.168 ms — sixteen screen-filling footprints in one wave; C3 0.078 ->
// 0.037 ms either way.)
__device__ inline int strided_rank(int P) {
  const int num_waves = (int)(gridDim.x * (blockDim.x >> 6));
  const int wave = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  const int lane = (int)(threadIdx.x & 63);
  const int r = lane < 16 ? lane * num_waves + wave : ((lane >> 4) * num_waves + wave) * 16 + (lane & 15);
  return r < P ? r : -1;
}

// value of `v` in lane `src`, `src` wave-uniform: v_readlane_b32 (scalar path) instead of the ds_bpermute_b32 a
// general __shfl compiles to — no LDS round trip, no lgkmcnt wait in the per-rectangle loops below
__device__ inline int lane_value(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ inline uint32_t lane_value(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
__device__ inline float lane_value(float v, int src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}

constexpr int SMALL_RECT = 8;
constexpr int STAT_SHARDS = 256;

// Walks the tile rectangles of the depth-ranked Gaussians: one lane per rank; rectangles of more than SMALL_RECT
// tiles are walked by the whole wave, 64 tiles per step (`coop(src_lane, k, active)` runs in every lane with its
// own tile number k of the source lane's rectangle), the others serially by their lane (`serial(tile)`).
struct RectLane {
  uint32_t g, n;
  int xmin, ymin, wdt;
};
template <typename Serial, typename CoopBegin, typename Coop>
__device__ inline void walk_rects(const RectLane& me, int gx, Serial serial, CoopBegin coop_begin, Coop coop) {
  const int lane = threadIdx.x & 63;
  if (me.n != 0u && me.n <= (uint32_t)SMALL_RECT) {
    int x = 0, y = 0;
    for (uint32_t k = 0; k < me.n; ++k) {
      serial(me.xmin + x, me.ymin + y);
      if (++x == me.wdt) { x = 0; ++y; }
    }
  }
  unsigned long long m = __ballot(me.n > (uint32_t)SMALL_RECT);
  while (m) {
    const int src = __ffsll((long long)m) - 1;
    m &= m - 1;
    const uint32_t n = lane_value(me.n, src);
    const int x0 = lane_value(me.xmin, src), y0 = lane_value(me.ymin, src), wdt = lane_value(me.wdt, src);
    const float inv_w = 1.0f / (float)wdt;
    coop_begin(src);
    for (uint32_t k0 = 0; k0 < n; k0 += 64) {
      const uint32_t k = k0 + (uint32_t)lane;
      const bool act = k < n;
      // k / wdt without an integer division (k < 2^24: the float quotient is off by at most one)
      int yy = (int)((float)k * inv_w);
      int xx = (int)k - yy * wdt;
      if (xx < 0) { --yy; xx += wdt; }
      if (xx >= wdt) { ++yy; xx -= wdt; }
      coop(src, act ? x0 + xx : 0, act ? y0 + yy : 0, act);
    }
  }
}

// ranks from nvalid on are the culled Gaussians (no tiles): their index and rectangle are not even loaded
__device__ inline RectLane rect_lane(int r, const uint32_t* __restrict__ gidx_sorted, const uint4* __restrict__ rect,
                                     uint32_t nvalid) {
  RectLane me{0u, 0u, 0, 0, 1};
  if (r >= 0 && (uint32_t)r < nvalid) {
    me.g = gidx_sorted[r];
    const uint4 rc = rect[me.g];
    me.n = rc.w;
    me.xmin = (int)(rc.y & 0xFFFF);
    me.ymin = (int)(rc.y >> 16);
    me.wdt = max((int)(rc.z & 0xFFFF) - me.xmin, 1);
  }
  return me;
}

// ---------------------------------------------------------------- occlusion culling of instances
// A tile whose every pixel has stopped (transmittance test, T_EPS) ignores the rest of its list.  Before any
// instance is created, a CONSERVATIVE per-tile bound finds a depth rank beyond which that is certain:
//   * for a (Gaussian, tile) pair, a_min = lower bound of the alpha the blend kernel gives ANY pixel of the tile
//     (largest q at the tile corners, fp32 error margins included; 0 if some pixel might be rejected by the
//     alpha >= 1/255 or power <= 0 tests, in particular for the tile that holds the centre);
//   * -ln(1 - a_min) is added (fixed point, integer atomics: order independent, hence reproducible) to the budget
//     of (tile, rank bucket); ranks are the depth order, OCC buckets per tile;
//   * the first bucket at which the running budget reaches OCC_THR > -ln(T_EPS) is the last one the tile needs:
//     whatever alphas the other Gaussians add, every pixel has met T (1 - alpha) < T_EPS by then.
// Instances of later buckets are never created: the sort, the per-instance arrays and the staging of the blend
// kernels shrink to the lists that can matter (Metric-1: 16.9 M -> 1.8 M), and every output stays bit-identical,
// because a culled instance would never have been blended.  The kept list of a tile is a prefix of its full list.
__device__ inline float tile_min_alpha(float mx, float my, float o, float cx, float cy, float cz, float X0, float Y0) {
  const float x1 = X0 + 15.f, y1 = Y0 + 15.f;
  if (mx >= X0 && mx <= x1 && my >= Y0 && my <= y1) return 0.f;
  float qmax = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float dx = ((c & 1) ? x1 : X0) - mx, dy = ((c & 2) ? y1 : Y0) - my;
    const float q0 = cx * dx * dx, q1 = 2.f * cy * dx * dy, q2 = cz * dy * dy;
    qmax = fmaxf(qmax, (q0 + q1 + q2) + 2e-5f * (q0 + fabsf(q1) + q2));
  }
  const float a = fminf(ALPHA_MAX, o * __expf(-0.5f * (qmax + 1e-3f)) * (1.f - 1e-5f));
  if (!(a >= ALPHA_MIN * 1.001f)) return 0.f;
  // the smallest q over the tile must be safely positive, or the kernel's `power <= 0` test might drop a pixel
  const float lx = X0 - mx, hx = x1 - mx, ly = Y0 - my, hy = y1 - my;
  float m = edge_min_q(cx, cy, cz, lx, ly, hy);
  m = fminf(m, edge_min_q(cx, cy, cz, hx, ly, hy));
  m = fminf(m, edge_min_q(cz, cy, cx, ly, lx, hx));
  m = fminf(m, edge_min_q(cz, cy, cx, hy, lx, hx));
  if (!(m > 1e-6f)) return 0.f;
  return a;
}

// ---------------------------------------------------------------- depth order of the preprocessed Gaussians
// A library radix / merge sort of P = 1M (key, index) pairs costs 20 launches and 150 us, more than the preprocess
// pass itself, although only the ~30 % that survive culling need an order.  Here: the key range of the survivors
// (two atomics per wave) -> 2^18 equal-width buckets of the KEY BITS (monotone in depth; positive floats order as
// integers) -> histogram -> exclusive scan -> scatter in arrival order -> every element ranks itself among its
// bucket mates by (key, index), which is the stable order of a full sort, whatever the arrival order was: the result
// is bit-identical to the library sort.  Culled Gaussians fill ranks [nvalid, P) in any order (they own no tiles; the
// per-rank kernels only need a permutation).  Buckets average a handful of elements; one beyond DS_LIMIT (many
// Gaussians at exactly one depth) raises a flag the host reads at its one synchronisation point and the library
// sort redoes the frame.
struct DsRange { uint32_t kmin, shift; };

__device__ inline DsRange ds_range(const uint32_t* __restrict__ head) { return DsRange{head[DS_KMIN], head[DS_SHIFT]}; }

__global__ __launch_bounds__(256) void ds_minmax_kernel(int P, const uint32_t* __restrict__ key, uint32_t* __restrict__ head) {
  __shared__ uint32_t sm[8];
  uint32_t mx = 0u, mn = 0u;  // max key, max ~key over valid entries
  for (int g = blockIdx.x * 256 + threadIdx.x; g < P; g += gridDim.x * 256) {
    const uint32_t k = key[g];
    if (k != CULLED_KEY) { mx = max(mx, k); mn = max(mn, ~k); }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
    mn = max(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sm[wave] = mx; sm[4 + wave] = mn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = max(max(sm[0], sm[1]), max(sm[2], sm[3]));
    mn = max(max(sm[4], sm[5]), max(sm[6], sm[7]));
    if (mx | mn) {
      atomicMax(&head[blockIdx.x % DS_SHARDS], mx);
      atomicMax(&head[DS_SHARDS + blockIdx.x % DS_SHARDS], mn);
    }
  }
}

__global__ __launch_bounds__(64) void ds_range_kernel(uint32_t* __restrict__ head) {
  uint32_t mx = head[threadIdx.x], mn = head[DS_SHARDS + threadIdx.x];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
    mn = max(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
  }
  if (threadIdx.x == 0) {
    const uint32_t kmin = ~mn;
    uint32_t shift = 0u;
    if (mx > kmin)
      while (((mx - kmin) >> shift) >= (uint32_t)DS_NB) ++shift;
    head[DS_KMIN] = kmin;
    head[DS_SHIFT] = shift;
  }
}

// bucket histogram of the survivors; culled Gaussians are counted per workgroup (cnt[DS_NB + block]) so that the one
// exclusive scan over [bucket counts | per-block culled counts] also yields each block's first culled rank.
// The histogram atomics are aggregated per wave: lanes that fall into the same bucket elect a leader (pure ALU
// rounds: ballot of the lanes equal to the first unassigned one), the leaders issue ONE returning atomic per distinct
// bucket, all in flight together, and every lane derives its arrival slot inside the bucket (`pos`) from the leader's
// return value.  A wall of surfels at one depth — a quarter of a million Gaussians in one bucket — would otherwise
// serialise on one address (~12 ns per atomic: 3 ms); the scatter pass needs no atomics at all.
__global__ __launch_bounds__(256) void ds_hist_kernel(int P, const uint32_t* __restrict__ key, const uint32_t* __restrict__ head,
                                                       uint32_t* __restrict__ cnt, uint32_t* __restrict__ pos) {
  __shared__ uint32_t sc[4];
  const int g = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const uint32_t k = g < P ? key[g] : 0u;
  const bool culled = g < P && k == CULLED_KEY;
  const bool active = g < P && !culled;
  const DsRange r = ds_range(head);
  const uint32_t b = active ? (k - r.kmin) >> r.shift : 0xFFFFFFFFu;
  unsigned long long todo = __ballot(active);
  const unsigned long long below = (1ull << lane) - 1ull;
  int my_leader = lane;
  uint32_t my_rank = 0u, group = 0u;
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t lb = (uint32_t)__builtin_amdgcn_readlane((int)b, leader);
    const unsigned long long m = __ballot(b == lb) & todo;
    if (b == lb) {
      my_leader = leader;
      my_rank = (uint32_t)__popcll(m & below);
      group = (uint32_t)__popcll(m);
    }
    todo &= ~m;
  }
  uint32_t base = 0u;
  if (active && my_leader == lane) base = atomicAdd(&cnt[b], group);
  base = (uint32_t)__shfl((int)base, my_leader, 64);
  if (active) pos[g] = base + my_rank;
  const unsigned long long bal = __ballot(culled);
  if (lane == 0) sc[threadIdx.x >> 6] = (uint32_t)__popcll(bal);
  __syncthreads();
  if (threadIdx.x == 0) cnt[DS_NB + blockIdx.x] = (sc[0] + sc[1]) + (sc[2] + sc[3]);
}

__global__ __launch_bounds__(256) void ds_scatter_kernel(int P, const uint32_t* __restrict__ key, const uint32_t* __restrict__ head,
                                                          const uint32_t* __restrict__ off, const uint32_t* __restrict__ pos,
                                                          uint32_t* __restrict__ tmp_key, uint32_t* __restrict__ tmp_idx,
                                                          uint32_t* __restrict__ gidx_sorted) {
  __shared__ uint32_t sc[4];
  const int g = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t k = g < P ? key[g] : 0u;
  const bool culled = g < P && k == CULLED_KEY;
  if (g < P && !culled) {
    const DsRange r = ds_range(head);
    const uint32_t b = (k - r.kmin) >> r.shift;
    const uint32_t slot = off[b] + pos[g];
    tmp_key[slot] = k;
    tmp_idx[slot] = (uint32_t)g;
  }
  // culled Gaussians take ranks [nvalid, P) in index order
  const unsigned long long bal = __ballot(culled);
  if (lane == 0) sc[wave] = (uint32_t)__popcll(bal);
  __syncthreads();
  if (culled) {
    uint32_t before = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) before += sc[w];
    gidx_sorted[off[DS_NB + blockIdx.x] + before] = (uint32_t)g;
  }
}

__global__ __launch_bounds__(256) void ds_rank_kernel(const uint32_t* __restrict__ tmp_key, const uint32_t* __restrict__ tmp_idx,
                                                       uint32_t* __restrict__ head, const uint32_t* __restrict__ off,
                                                       uint32_t* __restrict__ gidx_sorted, uint32_t* __restrict__ rank_of,
                                                       uint32_t* __restrict__ nvalid) {
  const uint32_t n = off[DS_NB];
  const uint32_t s = blockIdx.x * 256u + threadIdx.x;
  if (s == 0u) *nvalid = n;
  if (s >= n) return;
  const uint32_t k = tmp_key[s], id = tmp_idx[s];
  const DsRange r = ds_range(head);
  const uint32_t b = (k - r.kmin) >> r.shift;
  const uint32_t lo = off[b], hi = off[b + 1];
  if (hi - lo > DS_LIMIT) {  // degenerate depth distribution: keep a valid permutation, let the host redo the sort
    gidx_sorted[s] = id;
    rank_of[id] = s;
    if (s == lo) head[DS_FLAG] = 1u;
    return;
  }
  uint32_t rank = 0u;
  uint32_t t = lo;
  // eight bucket mates per round trip: a crowded bucket (a wall of surfels at one depth: hundreds of mates) made this
  // loop one dependent load latency per mate
  for (; t + 8u <= hi; t += 8u) {
    uint32_t kt[8], it[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { kt[u] = tmp_key[t + u]; it[u] = tmp_idx[t + u]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) rank += (kt[u] < k || (kt[u] == k && it[u] < id)) ? 1u : 0u;
  }
  for (; t < hi; ++t) {
    const uint32_t kt = tmp_key[t], it = tmp_idx[t];
    rank += (kt < k || (kt == k && it < id)) ? 1u : 0u;
  }
  gidx_sorted[lo + rank] = id;
  rank_of[id] = lo + rank;
}

__global__ __launch_bounds__(256) void invert_perm_kernel(int P, const uint32_t* __restrict__ gidx_sorted,
                                                           uint32_t* __restrict__ rank_of) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r < P) rank_of[gidx_sorted[r]] = (uint32_t)r;
}

// ranks [0, nvalid) hold the Gaussians with a real depth key (culled ones sort last): first culled rank,
// found by one wave with a 64-ary search (4 dependent rounds for any P < 2^24, instead of 20+)
__global__ __launch_bounds__(64) void count_valid_kernel(int P, const uint32_t* __restrict__ depth_key_sorted,
                                                         uint32_t* __restrict__ nvalid) {
  const int lane = threadIdx.x;
  long long lo = 0, hi = P;  // invariant: keys below lo are valid, keys from hi on are culled
  while (lo < hi) {
    const long long span = hi - lo;
    const long long step = (span + 63) / 64;
    const long long pos = lo + (long long)lane * step;
    const bool valid = pos < hi && depth_key_sorted[pos] < CULLED_KEY;
    const int nv = __popcll(__ballot(valid));  // probes are ascending: the valid ones form a prefix
    if (nv == 0) { hi = lo; break; }
    const long long last_valid = lo + (long long)(nv - 1) * step;
    lo = last_valid + 1;
    hi = min(hi, last_valid + step);
  }
  if (lane == 0) *nvalid = (uint32_t)lo;
}

__device__ inline uint32_t rank_bucket(int r, int nb, uint32_t nvalid) {
  if (r < 0 || nvalid == 0u) return 0u;
  const uint32_t b = (uint32_t)(((unsigned long long)r * (unsigned long long)nb) / (unsigned long long)nvalid);
  return min(b, (uint32_t)nb - 1u);
}

__global__ __launch_bounds__(256) void occl_budget_kernel(int P, int gx, int nb,
                                                          const uint32_t* __restrict__ gidx_sorted,
                                                          const uint4* __restrict__ rect,
                                                          const float4* __restrict__ rec,
                                                          const uint32_t* __restrict__ nvalid, int num_tiles,
                                                          uint32_t* __restrict__ bucket) {
  const int r = strided_rank(P);
  const uint32_t nv = *nvalid;
  RectLane me = rect_lane(r, gidx_sorted, rect, nv);
  if (me.n != 0u) {   // walk only the inner rectangle (preprocess_kernel): tiles outside it cannot be covered
    const uint4 rc = rect[me.g];
    const int x0 = me.xmin + (int)(rc.x & 255u), x1 = (int)(rc.z & 0xFFFF) - (int)((rc.x >> 8) & 255u);
    const int y0 = me.ymin + (int)((rc.x >> 16) & 255u), y1 = (int)(rc.z >> 16) - (int)(rc.x >> 24);
    if (x1 > x0 && y1 > y0) {
      me.xmin = x0; me.ymin = y0; me.wdt = x1 - x0; me.n = (uint32_t)((x1 - x0) * (y1 - y0));
    } else {
      me.n = 0u;
    }
  }
  float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
  if (me.n != 0u) {
    ra = rec[4 * (size_t)me.g + 0];
    rb = rec[4 * (size_t)me.g + 1];
  }
  const uint32_t bk = rank_bucket(r, nb, nv);
  // bucket-major layout: the 64 tiles a wave handles in one step are neighbours in a tile row, so its atomics
  // hit contiguous words (scattered 4-byte atomics run an order of magnitude slower)
  auto add = [&](int tx, int ty, float mx, float my, float o, float cx, float cy, float cz, uint32_t b) {
    const float a = tile_min_alpha(mx, my, o, cx, cy, cz, (float)(tx * TILE), (float)(ty * TILE));
    if (a > 0.f)
      atomicAdd(&bucket[(size_t)b * num_tiles + (ty * gx + tx)], (uint32_t)(-__logf(1.f - a) * OCC_FIX));
  };
  float smx = 0, smy = 0, so = 0, scx = 0, scy = 0, scz = 0;
  uint32_t sb = 0;
  walk_rects(
      me, gx, [&](int tx, int ty) { add(tx, ty, ra.x, ra.y, ra.z, rb.x, rb.y, rb.z, bk); },
      [&](int src) {
        smx = lane_value(ra.x, src); smy = lane_value(ra.y, src); so = lane_value(ra.z, src);
        scx = lane_value(rb.x, src); scy = lane_value(rb.y, src); scz = lane_value(rb.z, src);
        sb = lane_value(bk, src);
      },
      [&](int, int tx, int ty, bool act) { if (act) add(tx, ty, smx, smy, so, scx, scy, scz, sb); });
}

// Running budget over the rank buckets -> last bucket a tile needs.  A 512-thread workgroup takes 64 tiles x 8
// bucket groups: thread (tile, g) loads its group's buckets (coalesced across the 64 tiles, all loads independent),
// the group sums meet in LDS, and the group in which the running sum crosses the threshold finds the bucket.
__global__ __launch_bounds__(512) void occl_scan_kernel(int num_tiles, int nb, const uint32_t* __restrict__ bucket,
                                                        uint16_t* __restrict__ bsat) {
  __shared__ uint32_t sSum[8][64];
  const int t = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int tile = blockIdx.x * 64 + t;
  const int per = nb / 8;  // nb is a power of two >= 32
  const uint32_t thr = (uint32_t)(OCC_THR * OCC_FIX);
  uint32_t v[32];
  uint32_t sum = 0;
#pragma unroll
  for (int u = 0; u < 32; ++u) {
    v[u] = (u < per && tile < num_tiles) ? min(bucket[(size_t)(g * per + u) * num_tiles + tile], thr) : 0u;
    sum += v[u];
  }
  sSum[g][t] = min(sum, thr);
  __syncthreads();
  uint32_t before = 0;
  for (int k = 0; k < g; ++k) before += sSum[k][t];
  const uint32_t total = before + sSum[g][t];
  if (tile < num_tiles) {
    if (g == 7 && total < thr) bsat[tile] = OCC_ALL;  // never saturates: the last group sees the full sum
    if (before < thr && total >= thr) {               // exactly one group crosses the threshold
      uint32_t run = before;
      int res = -1;
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        run += v[u];
        if (res < 0 && run >= thr) res = u;
      }
      bsat[tile] = (uint16_t)(g * per + res);
    }
  }
}

// kept tiles per depth rank (the Gaussian's rank bucket must not exceed the tile's last needed bucket)
__global__ __launch_bounds__(256) void count_kept_kernel(int P, int gx, int nb,
                                                         const uint32_t* __restrict__ gidx_sorted,
                                                         const uint4* __restrict__ rect,
                                                         const uint16_t* __restrict__ bsat,
                                                         const uint32_t* __restrict__ nvalid,
                                                         uint32_t* __restrict__ tiles_sorted,
                                                         unsigned long long* __restrict__ pairs_full) {
  const int r = strided_rank(P);
  const int lane = threadIdx.x & 63;
  const uint32_t nv = *nvalid;
  const RectLane me = rect_lane(r, gidx_sorted, rect, nv);
  const uint32_t bk = rank_bucket(r, nb, nv);
  uint32_t kept = 0, sb = 0, acc = 0;
  int cur = -1;
  walk_rects(
      me, gx, [&](int tx, int ty) { kept += (bk <= (uint32_t)bsat[ty * gx + tx]) ? 1u : 0u; },
      [&](int src) {
        if (cur >= 0 && lane == cur) kept = acc;
        cur = src;
        acc = 0;
        sb = lane_value(bk, src);
      },
      [&](int, int tx, int ty, bool act) {
        acc += (uint32_t)__popcll(__ballot(act && sb <= (uint32_t)bsat[ty * gx + tx]));
      });
  if (cur >= 0 && lane == cur) kept = acc;
  if (r >= 0) tiles_sorted[r] = kept;
  // footprint statistic for the blend kernels' pixels-per-lane choice: all (Gaussian, tile) pairs before culling
  uint32_t tot = me.n;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) tot += (uint32_t)__shfl_xor((int)tot, off, 64);
  const unsigned long long vis = __ballot(me.n != 0u);
  // the kept total once more in 64 bits: the instance offsets are a 32-bit scan, which a degenerate frame (huge
  // footprints with the occlusion bound off) could wrap without anyone noticing
  uint32_t kept_w = r >= 0 ? kept : 0u;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) kept_w += (uint32_t)__shfl_xor((int)kept_w, off, 64);
  if (lane == 0 && tot) {  // sharded: thousands of waves adding to one word serialise
    const int shard = (int)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (STAT_SHARDS - 1));
    atomicAdd(pairs_full + 3 * shard, (unsigned long long)tot);
    atomicAdd(pairs_full + 3 * shard + 1, (unsigned long long)__popcll(vis));
    atomicAdd(pairs_full + 3 * shard + 2, (unsigned long long)kept_w);
  }
}

// Emits the kept (tile id, slot) instances in depth order; gval[slot] = Gaussian id, slot_val[slot] = slot.
template <typename KeyT>
__global__ __launch_bounds__(256) void duplicate_kernel(int P, int gx, int nb,
                                                         const uint32_t* __restrict__ gidx_sorted,
                                                         const uint32_t* __restrict__ offsets_sorted,
                                                         const uint32_t* __restrict__ tiles_sorted,
                                                         const uint4* __restrict__ rect,
                                                         const uint16_t* __restrict__ bsat,
                                                         const uint32_t* __restrict__ nvalid,
                                                         KeyT* __restrict__ tile_key,
                                                         uint32_t* __restrict__ gval,
                                                         uint32_t* __restrict__ slot_val) {
  const int r = strided_rank(P);
  const uint32_t nv = *nvalid;
  const RectLane me = rect_lane(r, gidx_sorted, rect, nv);
  const uint32_t bk = rank_bucket(r, nb, nv);
  uint32_t o = r >= 0 ? offsets_sorted[r] - tiles_sorted[r] : 0u;  // first slot of this Gaussian
  uint32_t sb = 0, so = 0, sg = 0;
  walk_rects(
      me, gx,
      [&](int tx, int ty) {
        const int tile = ty * gx + tx;
        if (bk <= (uint32_t)bsat[tile]) {
          tile_key[o] = (KeyT)tile;
          gval[o] = me.g;
          slot_val[o] = o;
          ++o;
        }
      },
      [&](int src) {
        sb = lane_value(bk, src);
        so = lane_value(o, src);
        sg = lane_value(me.g, src);
      },
      [&](int, int tx, int ty, bool act) {
        const int tile = ty * gx + tx;
        const bool keep = act && sb <= (uint32_t)bsat[tile];
        const unsigned long long bal = __ballot(keep);
        if (keep) {
          const uint32_t pos = so + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
          tile_key[pos] = (KeyT)tile;
          gval[pos] = sg;
          slot_val[pos] = pos;
        }
        so += (uint32_t)__popcll(bal);
      });
}

// ---------------------------------------------------------------- longest-processing-time-first tile dispatch
// A tile's blend time is proportional to its list; the lists are very uneven (Metric-1: 0..374 blended records,
// mean 127), and workgroups are dispatched in grid order: with tiles in image order the long ones that start late
// run on an otherwise idle chip.  Dispatching tiles in descending work order fills the tail with short ones.
// One workgroup: counting sort of the tiles by min(work / 16, 1023), descending.
__global__ __launch_bounds__(1024) void tile_order_kernel(const uint32_t* __restrict__ work, int num_tiles,
                                                          uint32_t* __restrict__ order, uint32_t* __restrict__ n_long,
                                                          uint32_t long_thr, uint32_t long_max) {
  __shared__ uint32_t hist[1024];
  __shared__ uint32_t base[1024];
  const int tid = threadIdx.x;
  hist[tid] = 0u;
  __syncthreads();
  for (int t = tid; t < num_tiles; t += 1024) atomicAdd(&hist[min(work[t] >> 4, 1023u)], 1u);
  __syncthreads();
  // exclusive scan over the bins in DESCENDING bin order (bin 1023 first); 1024 threads, one bin each
  uint32_t v = hist[1023 - tid];
  base[tid] = v;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const uint32_t add = tid >= off ? base[tid - off] : 0u;
    __syncthreads();
    base[tid] += add;
    __syncthreads();
  }
  const uint32_t excl = base[tid] - v;
  __syncthreads();
  hist[1023 - tid] = excl;   // hist[bin] = first output position of the bin
  __syncthreads();
  for (int t = tid; t < num_tiles; t += 1024) order[atomicAdd(&hist[min(work[t] >> 4, 1023u)], 1u)] = (uint32_t)t;
  // tiles with work >= long_thr (a multiple of 16: whole bins) are the first entries of the order: their number
  if (n_long && tid == 0) {
    const uint32_t bin = min(long_thr >> 4, 1023u);
    const uint32_t cnt = bin == 0u ? (uint32_t)num_tiles : base[1023 - bin];   // inclusive count of the bins >= bin
    *n_long = min(cnt, long_max);
  }
}

__global__ __launch_bounds__(256) void range_len_kernel(const uint2* __restrict__ ranges, int num_tiles,
                                                        uint32_t* __restrict__ work) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < num_tiles) work[t] = ranges[t].y - ranges[t].x;
}

int launch_tile_order(const uint32_t* work, int num_tiles, uint32_t* order, hipStream_t st, uint32_t* n_long,
                      uint32_t long_thr, uint32_t long_max) {
  hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, st, work, num_tiles, order, n_long, long_thr, long_max);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}

template <typename KeyT>
__global__ __launch_bounds__(256) void tile_ranges_kernel(int64_t I, const KeyT* __restrict__ key,
                                                           uint2* __restrict__ ranges) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= I) return;
  const uint32_t t = (uint32_t)key[i];
  if (i == 0 || key[i - 1] != t) ranges[t].x = (uint32_t)i;
  if (i == I - 1 || key[i + 1] != t) ranges[t].y = (uint32_t)(i + 1);
}

#ifdef PINGS_BLEND_STATS  // diagnostic build only: loop-efficiency counters of blend_fwd_kernel
__device__ unsigned long long g_blend_stats[8];
#endif

// PPL = pixels per lane (1 or 2).  A 16x16 tile is four 8x8 quadrants.  PPL = 1: wave w owns quadrant w
// (qx = w & 1, qy = w >> 1), lane l the pixel (l & 7, l >> 3) of it.  PPL = 2: wave w owns the 8-pixel-wide
// column half w (quadrants w and w + 2), lane l the two pixels (l & 7, l >> 3) and (l & 7, (l >> 3) + 8), which
// share x: the x-part of the quadratic form, the record unpacking, the loop control and the wave reductions
// are paid once per lane instead of once per pixel.
//
// Sub-tile culling: while a round of up to 256 records is staged in LDS, the staging thread of a record also
// evaluates which quadrants its footprint ellipse (alpha >= 1/255) can reach (quadrant_mask).  Every wave then
// compacts the records that may touch ITS pixels into a private index list (ballot + popcount, order kept) and
// walks only that list: on surfel scenes 40-50 % of the (wave, record) visits of a plain tile walk blend
// nothing, and a skipped record would have contributed exactly zero, so outputs are unchanged bit for bit.
template <int MODE, int PPL>
__global__ __launch_bounds__(BLOCK / PPL) void blend_fwd_kernel(
    KParams p, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ rec, const uint32_t* __restrict__ gval, float* __restrict__ out_color,
    float* __restrict__ out_normal, float* __restrict__ out_depth, float* __restrict__ out_alpha,
    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ inst_w,
    uint32_t* __restrict__ inst_cnt, uint8_t* __restrict__ inst_qmask, int want_qmask) {
  constexpr int NT = BLOCK / PPL;   // threads per workgroup
  constexpr int NWV = NT / 64;      // waves per workgroup
  constexpr int YS = 8;             // row distance of a lane's pixels
  __shared__ float4 sA[BLOCK];  // mx, my, opacity, pz
  __shared__ float4 sB[BLOCK];  // conic, rz
  __shared__ float4 sC[BLOCK];  // rgb, q
  __shared__ float4 sD[BLOCK];  // normal
  __shared__ uint32_t sSlot[BLOCK];
  __shared__ float sAcc[NWV][BLOCK];     // per-wave partial sums of blend weights
  __shared__ uint32_t sCnt[NWV][BLOCK];  // per-wave counts (3DGS n_touched)
  __shared__ uint8_t sQb[NWV][BLOCK];    // per-wave bits: quadrants in which the record blended something
  __shared__ uint8_t sMask[BLOCK];       // quadrant mask of every staged record
  __shared__ uint8_t sList[NWV][BLOCK];  // per-wave compacted record indices (ascending)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x;
  const int tx = tile % p.gx, ty = tile / p.gx;
  const int pix_x = tx * TILE + (PPL == 1 ? 8 * (wave & 1) : 8 * wave) + (lane & 7);
  const int pix_y0 = ty * TILE + (PPL == 1 ? 8 * (wave >> 1) : 0) + (lane >> 3);
  const uint32_t need = PPL == 1 ? (1u << wave) : ((1u << wave) | (4u << wave));
  const float pixf_x = (float)pix_x, pixf_y0 = (float)pix_y0;
  const float tileX0 = (float)(tx * TILE), tileY0 = (float)(ty * TILE);
  const size_t HW = (size_t)p.W * p.H;

  float rx = 0.f, ry[PPL];
#pragma unroll
  for (int k = 0; k < PPL; ++k) ry[k] = 0.f;
  if (MODE == MODE_SURFEL) {
    const float cxp = (p.prcp ? p.prcp[0] : 0.5f) * (float)p.W - 0.5f;
    const float cyp = (p.prcp ? p.prcp[1] : 0.5f) * (float)p.H - 0.5f;
    rx = (pixf_x - cxp) / p.fx;
#pragma unroll
    for (int k = 0; k < PPL; ++k) ry[k] = ((pixf_y0 + (float)(YS * k)) - cyp) / p.fy;
  }

  const uint2 range = ranges[tile];
  const int todo = (int)(range.y - range.x);

  float T[PPL], C0[PPL], C1[PPL], C2[PPL], N0[PPL], N1[PPL], N2[PPL], D[PPL];
  uint32_t last[PPL];
  bool inside[PPL], done[PPL];
  bool all_done = true;
#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    T[k] = 1.0f;
    C0[k] = C1[k] = C2[k] = N0[k] = N1[k] = N2[k] = D[k] = 0.f;
    last[k] = 0;
    inside[k] = pix_x < p.W && (pix_y0 + YS * k) < p.H;
    done[k] = !inside[k];
    all_done = all_done && done[k];
  }

#ifdef PINGS_BLEND_STATS
  unsigned long long st_iter = 0, st_any = 0, st_lanes = 0, st_staged = 0, st_livepix = 0;
#endif
  for (int base = 0; base < todo; base += BLOCK) {
    if (__syncthreads_and(all_done)) break;
    const int n = min(BLOCK, todo - base);
#ifdef PINGS_BLEND_STATS
    st_staged += (unsigned long long)n;
#endif
#pragma unroll
    for (int rr = 0; rr < PPL; ++rr) {
      const int e = tid + rr * NT;
      if (e < n) {
        const uint32_t slot = point_list[range.x + base + e];
        const uint32_t g = gval[slot];
        const float4 ra = rec[4 * (size_t)g + 0];
        const float4 rb = rec[4 * (size_t)g + 1];
        sA[e] = ra;
        sB[e] = rb;
        sC[e] = rec[4 * (size_t)g + 2];
        if (MODE == MODE_SURFEL) sD[e] = rec[4 * (size_t)g + 3];
        sSlot[e] = slot;
        sMask[e] = (uint8_t)quadrant_mask(ra.x, ra.y, ra.z, rb.x, rb.y, rb.z, tileX0, tileY0);
#pragma unroll
        for (int wv = 0; wv < NWV; ++wv) {
          sAcc[wv][e] = 0.f;
          if (PPL == 2) sQb[wv][e] = 0;
          if (MODE == MODE_3DGS) sCnt[wv][e] = 0u;
        }
      }
    }
    __syncthreads();

    // this wave's list: records whose footprint may reach its pixels, in list order
    int cnt = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
      const int e = c0 + lane;
      const bool hit = e < n && (sMask[e] & need) != 0;
      const unsigned long long bal = __ballot(hit);
      if (hit) sList[wave][cnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = (uint8_t)e;
      cnt += __popcll(bal);
    }
    __builtin_amdgcn_wave_barrier();

    // The record of list entry jj+1 is read from LDS while jj is blended; lanes that skip a Gaussian carry
    // w = 0 and the only branches are wave-uniform.
    int j = cnt > 0 ? (int)sList[wave][0] : 0;
    float4 a = sA[j], b = sB[j], c = sC[j], nn = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == MODE_SURFEL) nn = sD[j];
    for (int jj = 0; jj < cnt; ++jj) {
      const int jn = (int)sList[wave][jj + 1 < cnt ? jj + 1 : jj];
      const float4 a_n = sA[jn], b_n = sB[jn], c_n = sC[jn];
      float4 n_n = nn;
      if (MODE == MODE_SURFEL) n_n = sD[jn];
      if (__all(all_done)) break;  // every pixel of this wave is saturated
      // power = -0.5 (cx dx^2 + cz dy^2) - cy dx dy ; the dx-only part is shared by the lane's pixels
      const float dx = a.x - pixf_x;
      const float p0 = -0.5f * (b.x * dx * dx);
      const float pxy = b.y * dx;
      float alpha[PPL], test_T[PPL];
      bool contrib[PPL];
      bool any_c = false;
#pragma unroll
      for (int k = 0; k < PPL; ++k) {
        const float dy = a.y - (pixf_y0 + (float)(YS * k));  // one rounding, as in the oracle
        const float power = (p0 - 0.5f * (b.z * dy * dy)) - pxy * dy;
        alpha[k] = fminf(ALPHA_MAX, a.z * __expf(power));
        const bool valid = !done[k] && (power <= 0.0f) && (alpha[k] >= ALPHA_MIN);
        test_T[k] = T[k] * (1.0f - alpha[k]);
        const bool stop = valid && (test_T[k] < T_EPS);
        contrib[k] = valid && !stop;
        done[k] = done[k] || stop;
        any_c = any_c || contrib[k];
      }
#ifdef PINGS_BLEND_STATS
      st_iter += 1;
      st_any += __any(any_c) ? 1 : 0;
      st_lanes += (unsigned long long)__popcll(__ballot(any_c));
      {
        bool lv = false;
        for (int k = 0; k < PPL; ++k) lv = lv || !done[k];
        st_livepix += (unsigned long long)__popcll(__ballot(lv));
      }
#endif
      if (__any(any_c)) {
        float wsum = 0.f;
        uint32_t touched = 0;
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
          const float w = contrib[k] ? alpha[k] * T[k] : 0.f;
          wsum += w;
          C0[k] = fmaf(c.x, w, C0[k]);
          C1[k] = fmaf(c.y, w, C1[k]);
          C2[k] = fmaf(c.z, w, C2[k]);
          if (MODE == MODE_SURFEL) {
            const float den = (nn.x * rx + nn.y * ry[k]) + nn.z;
            float d = den < -DEN_EPS ? c.w * __builtin_amdgcn_rcpf(den) : a.w;
            d = fminf(fmaxf(d, a.w - b.w), a.w + b.w);
            N0[k] = fmaf(nn.x, w, N0[k]);
            N1[k] = fmaf(nn.y, w, N1[k]);
            N2[k] = fmaf(nn.z, w, N2[k]);
            D[k] = fmaf(d, w, D[k]);
          } else {
            D[k] = fmaf(a.w, w, D[k]);
            touched += (contrib[k] && test_T[k] > 0.5f) ? 1u : 0u;
          }
          T[k] = contrib[k] ? test_T[k] : T[k];
          last[k] = contrib[k] ? (uint32_t)(base + j + 1) : last[k];
        }
        const float s = wave_reduce_sum_dpp(wsum);
        if (lane == 63) sAcc[wave][j] = s;
        if (PPL == 2 && want_qmask) {
          // quadrants in which the record blended something, exactly (the Gaussian-per-lane backward kernel visits
          // exactly these): a lane's pixel k lies in quadrant wave + 2 k.  With PPL 1 a wave IS a quadrant and the
          // mask follows from sAcc; without `want_qmask` the pixel-per-lane backward kernel runs and needs none.
          uint32_t qb = 0;
#pragma unroll
          for (int k = 0; k < PPL; ++k)
            if (__ballot(contrib[k])) qb |= 1u << (wave + 2 * k);
          if (lane == 63) sQb[wave][j] = (uint8_t)qb;
        }
        if (MODE == MODE_3DGS) {
          const uint32_t cn = wave_reduce_sum_u32_dpp(touched);
          if (lane == 63) sCnt[wave][j] = cn;
        }
      }
      all_done = true;
#pragma unroll
      for (int k = 0; k < PPL; ++k) all_done = all_done && done[k];
      a = a_n; b = b_n; c = c_n; nn = n_n;
      j = jn;
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < PPL; ++rr) {
      const int e = tid + rr * NT;
      if (e < n) {
        float v = sAcc[0][e];
        uint32_t cn = (MODE == MODE_3DGS) ? sCnt[0][e] : 0u;
        uint32_t qm = PPL == 1 ? (sAcc[0][e] != 0.f ? 1u : 0u) : (uint32_t)sQb[0][e];  // quadrants that blended
#pragma unroll
        for (int wv = 1; wv < NWV; ++wv) {
          v += sAcc[wv][e];
          qm |= PPL == 1 ? (sAcc[wv][e] != 0.f ? 1u << wv : 0u) : (uint32_t)sQb[wv][e];
          if (MODE == MODE_3DGS) cn += sCnt[wv][e];
        }
        if (v != 0.f) {  // untouched slots stay at their memset zero
          inst_w[sSlot[e]] = v;
          inst_qmask[sSlot[e]] = (uint8_t)qm;
          if (MODE == MODE_3DGS) inst_cnt[sSlot[e]] = cn;
        }
      }
    }
  }

#ifdef PINGS_BLEND_STATS
  if (lane == 0) {
    atomicAdd(&g_blend_stats[0], st_iter);
    atomicAdd(&g_blend_stats[1], st_any);
    atomicAdd(&g_blend_stats[2], st_lanes);
    atomicAdd(&g_blend_stats[4], st_livepix);
    if (wave == 0) atomicAdd(&g_blend_stats[3], st_staged);
  }
#endif
#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    if (!inside[k]) continue;
    const size_t pix_id = (size_t)(pix_y0 + YS * k) * p.W + pix_x;
    const float A = 1.0f - T[k];
    final_T[pix_id] = T[k];
    n_contrib[pix_id] = last[k];
    out_color[pix_id] = C0[k] + T[k] * p.bg[0];
    out_color[HW + pix_id] = C1[k] + T[k] * p.bg[1];
    out_color[2 * HW + pix_id] = C2[k] + T[k] * p.bg[2];
    out_alpha[pix_id] = A;
    if (MODE == MODE_SURFEL) {
      out_normal[pix_id] = N0[k];
      out_normal[HW + pix_id] = N1[k];
      out_normal[2 * HW + pix_id] = N2[k];
      out_depth[pix_id] = D[k] / fmaxf(A, DEPTH_ALPHA_EPS);
    } else {
      out_depth[pix_id] = D[k];
    }
  }
}

// ---------------------------------------------------------------- forward, one independent wave per 8x8 quadrant
// Footprint class 1 (footprints of a few tiles).  In the workgroup-per-tile kernel above the four waves of a tile
// share the staging rounds, and as their culled lists differ in length (a wave visits ~37 % of the staged records
// on a street-like scene) they idle at the round barriers and cannot stop before the slowest one.  Here every wave
// walks the tile list by itself: 64 entries at a time it fetches slot -> Gaussian -> record (the next window is in
// flight while the current one is blended), tests the footprint against ITS quadrant, compacts the survivors into
// its own LDS slots and blends them; it stops as soon as its 64 pixels are done.  No barriers.  Per-instance sums
// are written per quadrant (inst_wq / inst_cntq) and folded by combine_quadrants_kernel, so everything downstream
// sees the same inst_w / inst_cnt / inst_qmask as from the workgroup kernel.  Pixel arithmetic is the same, op for op.
// Workgroups of four waves, wave = quadrant of ONE tile: the four walk the same list, so what one fetched (list entries,
// ids, records) the others find in the CU's L1; as one-wave workgroups the four quadrants of a tile were dealt to four
// XCDs (round-robin dispatch) and each fetched the list through its own L2.
template <int MODE>
__device__ __forceinline__ void blend_fwd_wave_body(
    const KParams& p, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ rec, const uint32_t* __restrict__ gval, float* __restrict__ out_color,
    float* __restrict__ out_normal, float* __restrict__ out_depth, float* __restrict__ out_alpha,
    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ inst_wq,
    uint32_t* __restrict__ inst_cntq, const uint32_t* __restrict__ tile_order,
    const uint32_t* __restrict__ seg_tile_unit0, const unsigned block) {
  __shared__ float4 sA_[4][64], sB_[4][64], sC_[4][64], sD_[4][64];
  __shared__ uint32_t sSlot_[4][64];
  __shared__ int sE_[4][64];
  __shared__ float sW_[4][64];
  __shared__ uint32_t sCnt_[4][64];

  const int lane = threadIdx.x & 63, q = (int)(threadIdx.x >> 6);
  float4 *sA = sA_[q], *sB = sB_[q], *sC = sC_[q], *sD = sD_[q];
  uint32_t *sSlot = sSlot_[q], *sCnt = sCnt_[q];
  int* sE = sE_[q];
  float* sW = sW_[q];
  const int tile = (int)tile_order[block];
  if (seg_tile_unit0 && seg_tile_unit0[tile] != 0xFFFFFFFFu) return;   // a long list: blended in parallel segments
  const int tx = tile % p.gx, ty = tile / p.gx;
  const int pix_x = tx * TILE + 8 * (q & 1) + (lane & 7);
  const int pix_y = ty * TILE + 8 * (q >> 1) + (lane >> 3);
  const float pixf_x = (float)pix_x, pixf_y = (float)pix_y;
  const float qx0 = (float)(tx * TILE + 8 * (q & 1)), qy0 = (float)(ty * TILE + 8 * (q >> 1));
  const size_t HW = (size_t)p.W * p.H;
  float rx = 0.f, ry = 0.f;
  if (MODE == MODE_SURFEL) {
    const float cxp = (p.prcp ? p.prcp[0] : 0.5f) * (float)p.W - 0.5f;
    const float cyp = (p.prcp ? p.prcp[1] : 0.5f) * (float)p.H - 0.5f;
    rx = (pixf_x - cxp) / p.fx;
    ry = (pixf_y - cyp) / p.fy;
  }
  const uint2 range = ranges[tile];
  const int todo = (int)(range.y - range.x);
  const bool inside = pix_x < p.W && pix_y < p.H;
  float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f, D = 0.f;
  uint32_t last = 0;
  bool done = !inside;

  // window being fetched: slot, Gaussian id, first two record quads of list entry base + lane
  // two-stage fetch pipeline (list entry -> Gaussian id | id -> whole record), see blend_fwd_seg_kernel
  uint32_t f_slot = 0, f_g = 0, n_slot = 0, n_g = 0;
  float4 f_a = make_float4(0.f, 0.f, 0.f, 0.f), f_b = f_a, f_c = f_a, f_d = f_a;
  bool f_ok = false, n_ok = false;
  auto fetch_ids = [&](int base) {
    const int e = base + lane;
    n_ok = e < todo;
    if (n_ok) {
      n_slot = point_list[range.x + e];
      n_g = gval[n_slot];
    }
  };
  auto fetch_records = [&]() {
    f_slot = n_slot; f_g = n_g; f_ok = n_ok;
    if (f_ok) {
      f_a = rec[4 * (size_t)f_g + 0];
      f_b = rec[4 * (size_t)f_g + 1];
      f_c = rec[4 * (size_t)f_g + 2];
      if (MODE == MODE_SURFEL) f_d = rec[4 * (size_t)f_g + 3];
    }
  };
  if (todo > 0 && !__all(done)) {
    fetch_ids(0);
    fetch_records();
    if (64 < todo) fetch_ids(64); else n_ok = false;
  }

  for (int base = 0; base < todo; base += 64) {
    if (__all(done)) break;
    const uint32_t slot = f_slot;
    const float4 ra = f_a, rb = f_b, rc = f_c, rd = f_d;
    const bool ok = f_ok;
    if (base + 64 < todo) {                                // in flight while this window is blended
      fetch_records();
      if (base + 128 < todo) fetch_ids(base + 128); else n_ok = false;
    }
    bool rel = false;
    if (ok) {
      const float thr = 2.f * __logf(255.f * ra.z) + 2e-3f;
      rel = !footprint_misses_rect(ra.x, ra.y, rb.x, rb.y, rb.z, thr, qx0, qx0 + 7.f, qy0, qy0 + 7.f);
    }
    const unsigned long long bal = __ballot(rel);
    const int n = __popcll(bal);
    if (rel) {
      const int at = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
      sA[at] = ra;
      sB[at] = rb;
      sC[at] = rc;
      if (MODE == MODE_SURFEL) sD[at] = rd;
      sSlot[at] = slot;
      sE[at] = base + lane;
    }
    sW[lane] = 0.f;
    if (MODE == MODE_3DGS) sCnt[lane] = 0u;
    __builtin_amdgcn_wave_barrier();

    // records in groups of 16, their weight sums reduced together (see blend_fwd_tile_kernel)
    bool stop_all = false;
    for (int j0 = 0; j0 < n && !stop_all; j0 += 16) {
      float ws[16], wc[16];
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) { ws[jj] = 0.f; wc[jj] = 0.f; }
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const int j = j0 + jj;
        if (j >= n || stop_all) continue;   // wave-uniform
        const float4 a = sA[j], b = sB[j], c = sC[j];
        float4 nn = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == MODE_SURFEL) nn = sD[j];
        const float dx = a.x - pixf_x;
        const float p0 = -0.5f * (b.x * dx * dx);
        const float pxy = b.y * dx;
        const float dy = a.y - pixf_y;
        const float power = (p0 - 0.5f * (b.z * dy * dy)) - pxy * dy;
        const float alpha = fminf(ALPHA_MAX, a.z * __expf(power));
        const bool valid = !done && (power <= 0.0f) && (alpha >= ALPHA_MIN);
        const float test_T = T * (1.0f - alpha);
        const bool stop = valid && (test_T < T_EPS);
        const bool contrib = valid && !stop;
        done = done || stop;
        if (__any(contrib)) {
          const float w = contrib ? alpha * T : 0.f;
          C0 = fmaf(c.x, w, C0);
          C1 = fmaf(c.y, w, C1);
          C2 = fmaf(c.z, w, C2);
          if (MODE == MODE_SURFEL) {
            const float den = (nn.x * rx + nn.y * ry) + nn.z;
            float d = den < -DEN_EPS ? c.w * __builtin_amdgcn_rcpf(den) : a.w;
            d = fminf(fmaxf(d, a.w - b.w), a.w + b.w);
            N0 = fmaf(nn.x, w, N0);
            N1 = fmaf(nn.y, w, N1);
            N2 = fmaf(nn.z, w, N2);
            D = fmaf(d, w, D);
          } else {
            D = fmaf(a.w, w, D);
            wc[jj] = (contrib && test_T > 0.5f) ? 1.f : 0.f;
          }
          T = contrib ? test_T : T;
          last = contrib ? (uint32_t)(sE[j] + 1) : last;
          ws[jj] = w;
        }
        stop_all = __all(done);
      }
      const int slot16 = 8 * (lane & 1) + 4 * ((lane >> 1) & 1) + ((lane >> 2) & 3);
      const float tw = wave_reduce16(ws, lane);
      if (lane < 16 && j0 + slot16 < n) sW[j0 + slot16] = tw;
      if (MODE == MODE_3DGS) {
        const float tc = wave_reduce16(wc, lane);
        if (lane < 16 && j0 + slot16 < n) sCnt[j0 + slot16] = (uint32_t)tc;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < n) {
      const float w = sW[lane];
      if (w != 0.f) {  // untouched entries stay at their memset zero
        inst_wq[4 * (size_t)sSlot[lane] + q] = w;
        if (MODE == MODE_3DGS) inst_cntq[4 * (size_t)sSlot[lane] + q] = sCnt[lane];
      }
    }
    __builtin_amdgcn_wave_barrier();
  }

  if (inside) {
    const size_t pix_id = (size_t)pix_y * p.W + pix_x;
    const float A = 1.0f - T;
    final_T[pix_id] = T;
    n_contrib[pix_id] = last;
    out_color[pix_id] = C0 + T * p.bg[0];
    out_color[HW + pix_id] = C1 + T * p.bg[1];
    out_color[2 * HW + pix_id] = C2 + T * p.bg[2];
    out_alpha[pix_id] = A;
    if (MODE == MODE_SURFEL) {
      out_normal[pix_id] = N0;
      out_normal[HW + pix_id] = N1;
      out_normal[2 * HW + pix_id] = N2;
      out_depth[pix_id] = D / fmaxf(A, DEPTH_ALPHA_EPS);
    } else {
      out_depth[pix_id] = D;
    }
  }
}

// ---------------------------------------------------------------- forward of LONG tile lists, in parallel segments
// A wave blends ~8 list entries per microsecond, one after the other: a tile whose list holds 14,000 records (the
// horizon of a street scene: thousands of edge-on surfels behind one another, none of them opaque) keeps ONE wave per
// quadrant busy for 1.7 ms while the rest of the chip has long finished.  Front-to-back compositing is associative:
//     (C, T) of a list = (C_a + T_a C_b, T_a T_b)  for the list split into a | b,
// so lists longer than SEG_THR entries are cut into segments of SEG entries, each blended by its own wave:
//   pass T  every (tile, segment, quadrant) wave walks its segment and multiplies up (1 - alpha) per pixel — alpha
//           evaluation only, the cheap half of the blend, and no termination test (that needs the transmittance in
//           front, which is what this pass is producing);
//   pass B  the wave multiplies the products of the segments in front of it — which is the transmittance its pixels
//           start with, and also tells it whether a pixel has stopped before (the running value is monotone, so it
//           stopped in segment s iff T_in(s) P(s) < T_EPS) — and then blends its segment EXACTLY like the serial
//           kernel does: same alpha arithmetic, same stop rule, same per-instance weight sums, `n_contrib` in global
//           list positions; colour / normal / depth partial sums, T and the last contributor go to a per-segment slab;
//   pass C  one wave per (tile, quadrant) adds the slabs in list order and writes the pixel outputs.
// What differs from the serial kernel is floating-point association only: T_in is a product of per-segment products
// instead of one running product, the colour sum a sum of per-segment sums (relative 1e-6); the stop decision can
// flip for a pixel whose transmittance sits within that rounding of 1e-4.  Lists up to SEG_THR entries never come
// here, so every list-parity test against the oracle is untouched; `PINGS_BLEND_SEG=<entries>` forces a small
// segment size (tests), 0 turns the path off.
constexpr int SEG_SLAB = 10;           // floats per pixel in a segment slab: C0 C1 C2 N0 N1 N2 D T last(bits) touched

// One workgroup: tiles with more than `thr` entries get ceil(L / seg) units; units of a tile are contiguous.
// head[0] = number of units, head[1] = number of long tiles.  unit_tile[u], unit_seg[u]; tile_unit0[tile] (or ~0u).
__global__ __launch_bounds__(1024) void seg_plan_kernel(const uint2* __restrict__ ranges, int num_tiles, uint32_t thr,
                                                        uint32_t seg, uint32_t max_units, uint32_t* __restrict__ head,
                                                        uint32_t* __restrict__ unit_tile, uint32_t* __restrict__ unit_seg,
                                                        uint32_t* __restrict__ tile_unit0) {
  __shared__ uint32_t sScan[1024];
  __shared__ uint32_t sBase;
  const int tid = threadIdx.x;
  if (tid == 0) sBase = 0u;
  __syncthreads();
  for (int t0 = 0; t0 < num_tiles; t0 += 1024) {
    const int t = t0 + tid;
    uint32_t n = 0;
    if (t < num_tiles) {
      const uint32_t L = ranges[t].y - ranges[t].x;
      if (L > thr) n = (L + seg - 1) / seg;
    }
    sScan[tid] = n;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const uint32_t add = tid >= off ? sScan[tid - off] : 0u;
      __syncthreads();
      sScan[tid] += add;
      __syncthreads();
    }
    const uint32_t base = sBase + sScan[tid] - n;
    if (t < num_tiles) {
      const bool fits = n > 0 && base + n <= max_units;   // a tile that does not fit stays with the serial kernel
      tile_unit0[t] = fits ? base : 0xFFFFFFFFu;
      if (fits) {
        for (uint32_t k = 0; k < n; ++k) { unit_tile[base + k] = (uint32_t)t; unit_seg[base + k] = k; }
      } else {
        for (uint32_t k = 0; k < n && base + k < max_units; ++k) { unit_tile[base + k] = 0xFFFFFFFFu; unit_seg[base + k] = 1u; }
      }
    }
    __syncthreads();
    if (tid == 1023) sBase += sScan[1023];
    __syncthreads();
  }
  if (tid == 0) head[0] = sBase < max_units ? sBase : max_units;
}

// PASS: 0 = transmittance products, 1 = blend.  Pass T leaves, per (unit, quadrant), the offsets of the entries that
// passed its footprint test (seg_rel / seg_nrel); with REUSE pass B walks that list in dense 64-entry windows — no
// fetch of the 63 % of the entries that cannot reach the quadrant, no second footprint test, no second compaction,
// 2.7x fewer windows.  The same entries in the same order: bit-identical to the re-testing form (REUSE = false, kept
// for segment sizes beyond 16-bit offsets).
template <int MODE, int PASS, bool REUSE = false>
__device__ __forceinline__ void blend_fwd_seg_body(
    const KParams& p, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ rec, const uint32_t* __restrict__ gval, uint32_t seg, const uint32_t* __restrict__ head,
    const uint32_t* __restrict__ unit_tile, const uint32_t* __restrict__ unit_seg, float* __restrict__ segP,
    float* __restrict__ slab, float* __restrict__ inst_wq, uint32_t* __restrict__ inst_cntq, const unsigned block,
    uint16_t* __restrict__ seg_rel = nullptr, uint32_t* __restrict__ seg_nrel = nullptr) {
  __shared__ float4 sA_[4][64], sB_[4][64], sC_[PASS == 1 ? 4 : 1][64], sD_[PASS == 1 ? 4 : 1][64];
  __shared__ uint32_t sSlot_[PASS == 1 ? 4 : 1][64];
  __shared__ int sE_[PASS == 1 ? 4 : 1][64];
  __shared__ float sW_[PASS == 1 ? 4 : 1][64];
  __shared__ uint32_t sCnt_[PASS == 1 ? 4 : 1][64];

  const int lane = threadIdx.x & 63, q = (int)(threadIdx.x >> 6);   // wave = quadrant (see blend_fwd_wave_kernel)
  constexpr int QB = PASS == 1 ? 1 : 0;
  float4 *sA = sA_[q], *sB = sB_[q], *sC = sC_[QB * q], *sD = sD_[QB * q];
  uint32_t *sSlot = sSlot_[QB * q], *sCnt = sCnt_[QB * q];
  int* sE = sE_[QB * q];
  float* sW = sW_[QB * q];
  const uint32_t unit = block;
  if (unit >= head[0] || unit_tile[unit] == 0xFFFFFFFFu) return;
  const int tile = (int)unit_tile[unit];
  const uint32_t sg = unit_seg[unit];
  const int tx = tile % p.gx, ty = tile / p.gx;
  const int pix_x = tx * TILE + 8 * (q & 1) + (lane & 7);
  const int pix_y = ty * TILE + 8 * (q >> 1) + (lane >> 3);
  const float pixf_x = (float)pix_x, pixf_y = (float)pix_y;
  const float qx0 = (float)(tx * TILE + 8 * (q & 1)), qy0 = (float)(ty * TILE + 8 * (q >> 1));
  float rx = 0.f, ry = 0.f;
  if (MODE == MODE_SURFEL && PASS == 1) {
    const float cxp = (p.prcp ? p.prcp[0] : 0.5f) * (float)p.W - 0.5f;
    const float cyp = (p.prcp ? p.prcp[1] : 0.5f) * (float)p.H - 0.5f;
    rx = (pixf_x - cxp) / p.fx;
    ry = (pixf_y - cyp) / p.fy;
  }
  const uint2 range = ranges[tile];
  const int L = (int)(range.y - range.x);
  const int e_lo = (int)(sg * seg);
  int e_hi = min(L, (int)((sg + 1) * seg));
  const bool inside = pix_x < p.W && pix_y < p.H;
  const size_t my = ((size_t)unit * 4 + q) * 64 + lane;
  // REUSE: the window loop below runs over positions [0, nrel) of pass T's list instead of entries [e_lo, e_hi)
  uint16_t* rel_list = seg_rel ? seg_rel + ((size_t)unit * 4 + q) * seg : nullptr;
  int w_lo = e_lo;
  if (REUSE) { w_lo = 0; e_hi = (int)seg_nrel[(size_t)unit * 4 + q]; }
  int nrel_out = 0;

  float T = 1.0f;
  bool done = !inside;
  if (PASS == 1) {
    // transmittance in front of this segment; stopped before it?
    const size_t first = ((size_t)(unit - sg) * 4 + q) * 64 + lane;
    for (uint32_t s2 = 0; s2 < sg; ++s2) {
      T *= segP[first + (size_t)s2 * 256];
      if (T < T_EPS) { done = true; break; }
    }
  }
  float C0 = 0.f, C1 = 0.f, C2 = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f, D = 0.f;
  uint32_t last = 0;

  // The fetch of a window is a chain of three dependent loads (list entry -> Gaussian id -> record).  It runs as a
  // two-stage pipeline: the ids of window k + 2 and the records of window k + 1 are in flight while window k is
  // blended, so no stage has to cover more than two dependent latencies with one window's work (with the whole chain
  // one window ahead, the short windows of pass T waited for it).
  uint32_t f_slot = 0, f_g = 0, n_slot = 0, n_g = 0;
  int f_e = 0, n_e = 0;                  // list position (entry index in the tile's list) of the fetched record
  float4 f_a = make_float4(0.f, 0.f, 0.f, 0.f), f_b = f_a, f_c = f_a, f_d = f_a;
  bool f_ok = false, n_ok = false;
  auto fetch_ids = [&](int base) {       // stage 1: list entry -> Gaussian id
    const int e = base + lane;
    n_ok = e < e_hi;
    if (n_ok) {
      n_e = REUSE ? e_lo + (int)rel_list[e] : e;
      n_slot = point_list[range.x + n_e];
      n_g = gval[n_slot];
    }
  };
  auto fetch_records = [&]() {           // stage 2: the ids that stage 1 brought -> first two record quads
    f_slot = n_slot; f_g = n_g; f_ok = n_ok; f_e = n_e;
    if (f_ok) {
      f_a = rec[4 * (size_t)f_g + 0];
      f_b = rec[4 * (size_t)f_g + 1];
      if (PASS == 1) {                   // the other half of the 64-byte record: same cache line, and the blend loop
        f_c = rec[4 * (size_t)f_g + 2];  // no longer starts with a load of its own
        if (MODE == MODE_SURFEL) f_d = rec[4 * (size_t)f_g + 3];
      }
    }
  };
  if (w_lo < e_hi && !__all(done)) {
    fetch_ids(w_lo);
    fetch_records();
    if (w_lo + 64 < e_hi) fetch_ids(w_lo + 64); else n_ok = false;
  }

  for (int base = w_lo; base < e_hi; base += 64) {
    if (__all(done)) break;
    const uint32_t slot = f_slot;
    const float4 ra = f_a, rb = f_b, rc = f_c, rd = f_d;
    const bool ok = f_ok;
    const int ent = f_e;
    if (base + 64 < e_hi) {
      fetch_records();
      if (base + 128 < e_hi) fetch_ids(base + 128); else n_ok = false;
    }
    bool rel = ok;                        // REUSE: pass T tested these entries already
    if (!REUSE && ok) {
      const float thr = 2.f * __logf(255.f * ra.z) + 2e-3f;
      rel = !footprint_misses_rect(ra.x, ra.y, rb.x, rb.y, rb.z, thr, qx0, qx0 + 7.f, qy0, qy0 + 7.f);
    }
    const unsigned long long bal = __ballot(rel);
    const int n = __popcll(bal);
    if (rel) {
      const int at = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
      sA[at] = ra;
      sB[at] = rb;
      if (PASS == 1) {
        sC[at] = rc;
        if (MODE == MODE_SURFEL) sD[at] = rd;
        sSlot[at] = slot;
        sE[at] = ent;
      } else if (rel_list) {
        rel_list[nrel_out + at] = (uint16_t)(ent - e_lo);
      }
    }
    nrel_out += n;
    if (PASS == 1) {
      sW[lane] = 0.f;
      if (MODE == MODE_3DGS) sCnt[lane] = 0u;
    }
    __builtin_amdgcn_wave_barrier();

    if (PASS == 0) {
      for (int j = 0; j < n; ++j) {
        const float4 a = sA[j], b = sB[j];
        const float dx = a.x - pixf_x;
        const float p0 = -0.5f * (b.x * dx * dx);
        const float pxy = b.y * dx;
        const float dy = a.y - pixf_y;
        const float power = (p0 - 0.5f * (b.z * dy * dy)) - pxy * dy;
        const float alpha = fminf(ALPHA_MAX, a.z * __expf(power));
        const bool valid = inside && (power <= 0.0f) && (alpha >= ALPHA_MIN);
        T = valid ? T * (1.0f - alpha) : T;
      }
    } else {
      // records in groups of 16, their weight sums reduced together (see blend_fwd_tile_kernel)
      bool stop_all = false;
      for (int j0 = 0; j0 < n && !stop_all; j0 += 16) {
        float ws[16], wc[16];
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) { ws[jj] = 0.f; wc[jj] = 0.f; }
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
          const int j = j0 + jj;
          if (j >= n || stop_all) continue;   // wave-uniform
          const float4 a = sA[j], b = sB[j];
          const float dx = a.x - pixf_x;
          const float p0 = -0.5f * (b.x * dx * dx);
          const float pxy = b.y * dx;
          const float dy = a.y - pixf_y;
          const float power = (p0 - 0.5f * (b.z * dy * dy)) - pxy * dy;
          const float alpha = fminf(ALPHA_MAX, a.z * __expf(power));
          const float4 c = sC[j];
          float4 nn = make_float4(0.f, 0.f, 0.f, 0.f);
          if (MODE == MODE_SURFEL) nn = sD[j];
          const bool valid = !done && (power <= 0.0f) && (alpha >= ALPHA_MIN);
          const float test_T = T * (1.0f - alpha);
          const bool stop = valid && (test_T < T_EPS);
          const bool contrib = valid && !stop;
          done = done || stop;
          if (__any(contrib)) {
            const float w = contrib ? alpha * T : 0.f;
            C0 = fmaf(c.x, w, C0);
            C1 = fmaf(c.y, w, C1);
            C2 = fmaf(c.z, w, C2);
            if (MODE == MODE_SURFEL) {
              const float den = (nn.x * rx + nn.y * ry) + nn.z;
              float d = den < -DEN_EPS ? c.w * __builtin_amdgcn_rcpf(den) : a.w;
              d = fminf(fmaxf(d, a.w - b.w), a.w + b.w);
              N0 = fmaf(nn.x, w, N0);
              N1 = fmaf(nn.y, w, N1);
              N2 = fmaf(nn.z, w, N2);
              D = fmaf(d, w, D);
            } else {
              D = fmaf(a.w, w, D);
              wc[jj] = (contrib && test_T > 0.5f) ? 1.f : 0.f;
            }
            T = contrib ? test_T : T;
            last = contrib ? (uint32_t)(sE[j] + 1) : last;
            ws[jj] = w;
          }
          stop_all = __all(done);
        }
        const int slot16 = 8 * (lane & 1) + 4 * ((lane >> 1) & 1) + ((lane >> 2) & 3);
        const float tw = wave_reduce16(ws, lane);
        if (lane < 16 && j0 + slot16 < n) sW[j0 + slot16] = tw;
        if (MODE == MODE_3DGS) {
          const float tc = wave_reduce16(wc, lane);
          if (lane < 16 && j0 + slot16 < n) sCnt[j0 + slot16] = (uint32_t)tc;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (PASS == 1 && lane < n) {
      const float w = sW[lane];
      if (w != 0.f) {
        inst_wq[4 * (size_t)sSlot[lane] + q] = w;
        if (MODE == MODE_3DGS) inst_cntq[4 * (size_t)sSlot[lane] + q] = sCnt[lane];
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (PASS == 0) {
    segP[my] = T;
    if (seg_nrel && lane == 0) seg_nrel[(size_t)unit * 4 + q] = (uint32_t)nrel_out;
  } else {
    float* o = slab + ((size_t)unit * 4 + q) * 64 * SEG_SLAB + lane;
    o[0] = C0; o[64] = C1; o[128] = C2; o[192] = N0; o[256] = N1; o[320] = N2; o[384] = D;
    o[448] = T;                                     // transmittance behind this segment (its pixels' running value)
    o[512] = __uint_as_float(last);
    o[576] = done ? 1.f : 0.f;
  }
}

template <int MODE, int PASS, bool REUSE>
__global__ __launch_bounds__(256) void blend_fwd_seg_kernel(
    KParams p, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ rec, const uint32_t* __restrict__ gval, uint32_t seg, const uint32_t* __restrict__ head,
    const uint32_t* __restrict__ unit_tile, const uint32_t* __restrict__ unit_seg, float* __restrict__ segP,
    float* __restrict__ slab, float* __restrict__ inst_wq, uint32_t* __restrict__ inst_cntq,
    uint16_t* __restrict__ seg_rel, uint32_t* __restrict__ seg_nrel) {
  blend_fwd_seg_body<MODE, PASS, REUSE>(p, ranges, point_list, rec, gval, seg, head, unit_tile, unit_seg, segP, slab,
                                        inst_wq, inst_cntq, blockIdx.x, seg_rel, seg_nrel);
}

template <int MODE>
__global__ __launch_bounds__(256) void blend_fwd_wave_kernel(
    KParams p, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ rec, const uint32_t* __restrict__ gval, float* __restrict__ out_color,
    float* __restrict__ out_normal, float* __restrict__ out_depth, float* __restrict__ out_alpha,
    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ inst_wq,
    uint32_t* __restrict__ inst_cntq, const uint32_t* __restrict__ tile_order,
    const uint32_t* __restrict__ seg_tile_unit0) {
  blend_fwd_wave_body<MODE>(p, ranges, point_list, rec, gval, out_color, out_normal, out_depth, out_alpha, final_T,
                            n_contrib, inst_wq, inst_cntq, tile_order, seg_tile_unit0, blockIdx.x);
}

// The short-list tiles and pass T of the segmented tiles in ONE launch (workgroups [0, num_tiles) are tiles, the rest
// segment units): the two touch disjoint tiles and both spend a third of their time in s_waitcnt, so together they
// fill what each leaves idle.  (A side stream does the same with two launches — 0.775 -> 0.719 ms on C3 — but its
// cross-queue event wait once took 11 ms per frame on one box of the pool; one launch needs no such wait.)
template <int MODE>
__global__ __launch_bounds__(256) void blend_fwd_wave_segT_kernel(
    KParams p, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ rec, const uint32_t* __restrict__ gval, float* __restrict__ out_color,
    float* __restrict__ out_normal, float* __restrict__ out_depth, float* __restrict__ out_alpha,
    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ inst_wq,
    uint32_t* __restrict__ inst_cntq, const uint32_t* __restrict__ tile_order,
    const uint32_t* __restrict__ seg_tile_unit0, unsigned num_tiles, uint32_t seg, const uint32_t* __restrict__ head,
    const uint32_t* __restrict__ unit_tile, const uint32_t* __restrict__ unit_seg, float* __restrict__ segP,
    uint16_t* __restrict__ seg_rel, uint32_t* __restrict__ seg_nrel) {
  if (blockIdx.x < num_tiles)
    blend_fwd_wave_body<MODE>(p, ranges, point_list, rec, gval, out_color, out_normal, out_depth, out_alpha, final_T,
                              n_contrib, inst_wq, inst_cntq, tile_order, seg_tile_unit0, blockIdx.x);
  else
    blend_fwd_seg_body<MODE, 0>(p, ranges, point_list, rec, gval, seg, head, unit_tile, unit_seg, segP, nullptr, inst_wq,
                                inst_cntq, blockIdx.x - num_tiles, seg_rel, seg_nrel);
}

// PASS C: first-segment waves add their tile's slabs in list order and write the pixel outputs.
template <int MODE>
__global__ __launch_bounds__(256) void blend_fwd_seg_combine_kernel(
    KParams p, const uint2* __restrict__ ranges, uint32_t seg, const uint32_t* __restrict__ head,
    const uint32_t* __restrict__ unit_tile, const uint32_t* __restrict__ unit_seg, const float* __restrict__ slab,
    float* __restrict__ out_color, float* __restrict__ out_normal, float* __restrict__ out_depth,
    float* __restrict__ out_alpha, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib) {
  const int lane = threadIdx.x & 63;
  const uint32_t unit = blockIdx.x;
  const int q = (int)(threadIdx.x >> 6);
  if (unit >= head[0] || unit_seg[unit] != 0u || unit_tile[unit] == 0xFFFFFFFFu) return;
  const int tile = (int)unit_tile[unit];
  const int tx = tile % p.gx, ty = tile / p.gx;
  const int pix_x = tx * TILE + 8 * (q & 1) + (lane & 7);
  const int pix_y = ty * TILE + 8 * (q >> 1) + (lane >> 3);
  if (pix_x >= p.W || pix_y >= p.H) return;
  const uint32_t L = ranges[tile].y - ranges[tile].x;
  const uint32_t nseg = (L + seg - 1) / seg;
  float C0 = 0.f, C1 = 0.f, C2 = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f, D = 0.f, T = 1.0f;
  uint32_t last = 0;
  for (uint32_t s2 = 0; s2 < nseg; ++s2) {
    const float* o = slab + ((size_t)(unit + s2) * 4 + q) * 64 * SEG_SLAB + lane;
    C0 += o[0]; C1 += o[64]; C2 += o[128]; N0 += o[192]; N1 += o[256]; N2 += o[320]; D += o[384];
    const uint32_t l2 = __float_as_uint(o[512]);
    if (l2 != 0u) { last = l2; T = o[448]; }          // the last segment that blended something holds T and n_contrib
    if (o[576] != 0.f) break;                        // the pixel stopped inside (or before) this segment
  }
  const size_t HW = (size_t)p.W * p.H;
  const size_t pix_id = (size_t)pix_y * p.W + pix_x;
  const float A = 1.0f - T;
  final_T[pix_id] = T;
  n_contrib[pix_id] = last;
  out_color[pix_id] = C0 + T * p.bg[0];
  out_color[HW + pix_id] = C1 + T * p.bg[1];
  out_color[2 * HW + pix_id] = C2 + T * p.bg[2];
  out_alpha[pix_id] = A;
  if (MODE == MODE_SURFEL) {
    out_normal[pix_id] = N0;
    out_normal[HW + pix_id] = N1;
    out_normal[2 * HW + pix_id] = N2;
    out_depth[pix_id] = D / fmaxf(A, DEPTH_ALPHA_EPS);
  } else {
    out_depth[pix_id] = D;
  }
}

// ---------------------------------------------------------------- forward, one independent wave per 16x16 tile
// Footprint class 2 (footprints of many tiles: almost every record of a tile's list reaches all four quadrants, so
// sub-tile culling buys nothing).  Same window pipeline as the quadrant kernel, but the wave owns the whole tile with
// FOUR pixels per lane (k & 1 -> x half, k >> 1 -> y half): the record fetch from LDS, the footprint test and the
// wave reduction of the blend weights are paid once per 256 pixels instead of once per 64, and the per-instance sum
// is final — no per-quadrant partials, no 16-byte-per-instance memset, no combine pass.  Pixel arithmetic is the
// quadrant kernel's, op for op (images bit-identical); the per-instance weight sum adds the same terms in another
// order.  inst_qmask is NOT produced: only the pixel-per-lane backward (which this class uses) may follow.
template <int MODE>
__global__ __launch_bounds__(64) void blend_fwd_tile_kernel(
    KParams p, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ rec, const uint32_t* __restrict__ gval, float* __restrict__ out_color,
    float* __restrict__ out_normal, float* __restrict__ out_depth, float* __restrict__ out_alpha,
    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ inst_w,
    uint32_t* __restrict__ inst_cnt, const uint32_t* __restrict__ tile_order) {
  constexpr int PPL = 4;
  __shared__ float4 sA[64], sB[64], sC[64], sD[64];
  __shared__ uint32_t sSlot[64];
  __shared__ int sE[64];
  __shared__ float sW[64];
  __shared__ uint32_t sCnt[64];

  const int lane = threadIdx.x;
  const int tile = (int)tile_order[blockIdx.x];
  const int tx = tile % p.gx, ty = tile / p.gx;
  const float qx0 = (float)(tx * TILE), qy0 = (float)(ty * TILE);
  const size_t HW = (size_t)p.W * p.H;
  int pix_x[2], pix_y[2];
  float pixf_x[2], pixf_y[2], rx[2] = {0.f, 0.f}, ry[2] = {0.f, 0.f};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    pix_x[h] = tx * TILE + 8 * h + (lane & 7);
    pix_y[h] = ty * TILE + 8 * h + (lane >> 3);
    pixf_x[h] = (float)pix_x[h];
    pixf_y[h] = (float)pix_y[h];
  }
  if (MODE == MODE_SURFEL) {
    const float cxp = (p.prcp ? p.prcp[0] : 0.5f) * (float)p.W - 0.5f;
    const float cyp = (p.prcp ? p.prcp[1] : 0.5f) * (float)p.H - 0.5f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      rx[h] = (pixf_x[h] - cxp) / p.fx;
      ry[h] = (pixf_y[h] - cyp) / p.fy;
    }
  }
  const uint2 range = ranges[tile];
  const int todo = (int)(range.y - range.x);
  float T[PPL], C0[PPL], C1[PPL], C2[PPL], N0[PPL], N1[PPL], N2[PPL], D[PPL];
  uint32_t last[PPL];
  bool done[PPL], inside[PPL];
  bool all_done_lane = true;
#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    T[k] = 1.0f;
    C0[k] = C1[k] = C2[k] = N0[k] = N1[k] = N2[k] = D[k] = 0.f;
    last[k] = 0;
    inside[k] = pix_x[k & 1] < p.W && pix_y[k >> 1] < p.H;
    done[k] = !inside[k];
    all_done_lane = all_done_lane && done[k];
  }

  // two-stage fetch pipeline (list entry -> Gaussian id | id -> whole record), see blend_fwd_seg_kernel
  uint32_t f_slot = 0, f_g = 0, n_slot = 0, n_g = 0;
  float4 f_a = make_float4(0.f, 0.f, 0.f, 0.f), f_b = f_a, f_c = f_a, f_d = f_a;
  bool f_ok = false, n_ok = false;
  auto fetch_ids = [&](int base) {
    const int e = base + lane;
    n_ok = e < todo;
    if (n_ok) {
      n_slot = point_list[range.x + e];
      n_g = gval[n_slot];
    }
  };
  auto fetch_records = [&]() {
    f_slot = n_slot; f_g = n_g; f_ok = n_ok;
    if (f_ok) {
      f_a = rec[4 * (size_t)f_g + 0];
      f_b = rec[4 * (size_t)f_g + 1];
      f_c = rec[4 * (size_t)f_g + 2];
      if (MODE == MODE_SURFEL) f_d = rec[4 * (size_t)f_g + 3];
    }
  };
  if (todo > 0 && !__all(all_done_lane)) {
    fetch_ids(0);
    fetch_records();
    if (64 < todo) fetch_ids(64); else n_ok = false;
  }

  for (int base = 0; base < todo; base += 64) {
    if (__all(all_done_lane)) break;
    const uint32_t slot = f_slot;
    const float4 ra = f_a, rb = f_b, rc = f_c, rd = f_d;
    const bool ok = f_ok;
    if (base + 64 < todo) {                                // in flight while this window is blended
      fetch_records();
      if (base + 128 < todo) fetch_ids(base + 128); else n_ok = false;
    }
    bool rel = false;
    if (ok) {
      const float thr = 2.f * __logf(255.f * ra.z) + 2e-3f;
      rel = !footprint_misses_rect(ra.x, ra.y, rb.x, rb.y, rb.z, thr, qx0, qx0 + 15.f, qy0, qy0 + 15.f);
    }
    const unsigned long long bal = __ballot(rel);
    const int n = __popcll(bal);
    if (rel) {
      const int at = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
      sA[at] = ra;
      sB[at] = rb;
      sC[at] = rc;
      if (MODE == MODE_SURFEL) sD[at] = rd;
      sSlot[at] = slot;
      sE[at] = base + lane;
    }
    sW[lane] = 0.f;
    if (MODE == MODE_3DGS) sCnt[lane] = 0u;
    __builtin_amdgcn_wave_barrier();

    // Records in groups of 16: every record's weight sum over the tile's pixels (and, 3DGS, its count of pixels it is
    // the dominant contributor of) stays in a register until the group is done, then ONE transposed reduce-scatter
    // (raster_common.hpp: wave_reduce16, ~55 vector ops for 16 sums) replaces sixteen 7-step DPP reductions and their
    // dependent chains.  The unrolled group keeps the register indices static.
    bool stop_all = false;
    for (int j0 = 0; j0 < n && !stop_all; j0 += 16) {
      float ws[16], wc[16];
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) { ws[jj] = 0.f; wc[jj] = 0.f; }
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const int j = j0 + jj;
        if (j >= n || stop_all) continue;   // wave-uniform
        const float4 a = sA[j], b = sB[j], c = sC[j];
        float4 nn = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == MODE_SURFEL) nn = sD[j];
        float dxv[2], p0v[2], pxyv[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          dxv[h] = a.x - pixf_x[h];
          p0v[h] = -0.5f * (b.x * dxv[h] * dxv[h]);
          pxyv[h] = b.y * dxv[h];
        }
        float alpha[PPL], test_T[PPL];
        bool contrib[PPL];
        bool any_c = false;
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
          const float dy = a.y - pixf_y[k >> 1];
          const float power = (p0v[k & 1] - 0.5f * (b.z * dy * dy)) - pxyv[k & 1] * dy;
          alpha[k] = fminf(ALPHA_MAX, a.z * __expf(power));
          const bool valid = !done[k] && (power <= 0.0f) && (alpha[k] >= ALPHA_MIN);
          test_T[k] = T[k] * (1.0f - alpha[k]);
          const bool stop = valid && (test_T[k] < T_EPS);
          contrib[k] = valid && !stop;
          done[k] = done[k] || stop;
          any_c = any_c || contrib[k];
        }
        if (__any(any_c)) {
          float wsum = 0.f;
          uint32_t touched = 0;
          const uint32_t e1 = (uint32_t)(sE[j] + 1);
#pragma unroll
          for (int k = 0; k < PPL; ++k) {
            const float w = contrib[k] ? alpha[k] * T[k] : 0.f;
            C0[k] = fmaf(c.x, w, C0[k]);
            C1[k] = fmaf(c.y, w, C1[k]);
            C2[k] = fmaf(c.z, w, C2[k]);
            if (MODE == MODE_SURFEL) {
              const float den = (nn.x * rx[k & 1] + nn.y * ry[k >> 1]) + nn.z;
              float d = den < -DEN_EPS ? c.w * __builtin_amdgcn_rcpf(den) : a.w;
              d = fminf(fmaxf(d, a.w - b.w), a.w + b.w);
              N0[k] = fmaf(nn.x, w, N0[k]);
              N1[k] = fmaf(nn.y, w, N1[k]);
              N2[k] = fmaf(nn.z, w, N2[k]);
              D[k] = fmaf(d, w, D[k]);
            } else {
              D[k] = fmaf(a.w, w, D[k]);
              touched += (contrib[k] && test_T[k] > 0.5f) ? 1u : 0u;
            }
            T[k] = contrib[k] ? test_T[k] : T[k];
            last[k] = contrib[k] ? e1 : last[k];
            wsum += w;
          }
          ws[jj] = wsum;
          if (MODE == MODE_3DGS) wc[jj] = (float)touched;   // <= 4 per lane, <= 256 per record: exact in fp32
        }
        all_done_lane = (done[0] && done[1]) && (done[2] && done[3]);
        stop_all = __all(all_done_lane);
      }
      // lane l of every 16-lane row ends with the total of slot 8*(l&1) + 4*((l>>1)&1) + ((l>>2)&3)
      const int slot16 = 8 * (lane & 1) + 4 * ((lane >> 1) & 1) + ((lane >> 2) & 3);
      const float tw = wave_reduce16(ws, lane);
      if (lane < 16 && j0 + slot16 < n) sW[j0 + slot16] = tw;
      if (MODE == MODE_3DGS) {
        const float tc = wave_reduce16(wc, lane);
        if (lane < 16 && j0 + slot16 < n) sCnt[j0 + slot16] = (uint32_t)tc;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < n) {
      const float w = sW[lane];
      if (w != 0.f) {  // untouched entries stay at their memset zero
        inst_w[sSlot[lane]] = w;
        if (MODE == MODE_3DGS) inst_cnt[sSlot[lane]] = sCnt[lane];
      }
    }
    __builtin_amdgcn_wave_barrier();
  }

#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    if (!inside[k]) continue;
    const size_t pix_id = (size_t)pix_y[k >> 1] * p.W + pix_x[k & 1];
    const float A = 1.0f - T[k];
    final_T[pix_id] = T[k];
    n_contrib[pix_id] = last[k];
    out_color[pix_id] = C0[k] + T[k] * p.bg[0];
    out_color[HW + pix_id] = C1[k] + T[k] * p.bg[1];
    out_color[2 * HW + pix_id] = C2[k] + T[k] * p.bg[2];
    out_alpha[pix_id] = A;
    if (MODE == MODE_SURFEL) {
      out_normal[pix_id] = N0[k];
      out_normal[HW + pix_id] = N1[k];
      out_normal[2 * HW + pix_id] = N2[k];
      out_depth[pix_id] = D[k] / fmaxf(A, DEPTH_ALPHA_EPS);
    } else {
      out_depth[pix_id] = D[k];
    }
  }
}

// inst_w[slot] = sum over the quadrants in fixed order, inst_cnt likewise, inst_qmask = quadrants that blended
template <int MODE>
__global__ __launch_bounds__(256) void combine_quadrants_kernel(int64_t I, const float4* __restrict__ wq,
                                                                const uint4* __restrict__ cq, float* __restrict__ inst_w,
                                                                uint32_t* __restrict__ inst_cnt,
                                                                uint8_t* __restrict__ inst_qmask) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) { inst_w[I] = 0.f; inst_qmask[I] = 0; }  // sentinel entry of the live-instance scans
  if (i >= I) return;
  const float4 w = wq[i];
  inst_w[i] = ((w.x + w.y) + w.z) + w.w;
  inst_qmask[i] = (uint8_t)((w.x != 0.f ? 1u : 0u) | (w.y != 0.f ? 2u : 0u) | (w.z != 0.f ? 4u : 0u) | (w.w != 0.f ? 8u : 0u));
  if (MODE == MODE_3DGS) {
    const uint4 c = cq[i];
    inst_cnt[i] = c.x + c.y + c.z + c.w;
  }
}

__device__ inline uint32_t wave_reduce_sum_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off, 64);
  return v;
}

// Per-Gaussian sum of its per-instance values (contiguous run of `tiles` slots).  One lane per
// depth rank (neighbouring lanes own neighbouring runs); runs longer than SMALL_RUN are summed by
// the whole wave.  Fixed summation order -> bitwise reproducible.
constexpr int SMALL_RUN = 16;
template <typename T>
__global__ __launch_bounds__(256) void per_gaussian_sum_kernel(
    int P, const uint4* __restrict__ rect, const uint32_t* __restrict__ rank_of,
    const uint32_t* __restrict__ offsets_sorted, const uint32_t* __restrict__ tiles_sorted,
    const T* __restrict__ inst, T* __restrict__ out) {
  // one lane per Gaussian in INDEX order (coalesced output; depth order is random in the index, so long runs are
  // spread over the waves anyway); the run of a surviving Gaussian is found through its depth rank
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  uint32_t n = 0, base = 0;
  if (g < P && rect[g].w != 0u) {
    const uint32_t r = rank_of[g];
    n = tiles_sorted[r];
    base = offsets_sorted[r] - n;
  }
  T sum = (T)0;
  if (n <= (uint32_t)SMALL_RUN)
    for (uint32_t k = 0; k < n; ++k) sum += inst[base + k];
  unsigned long long m = __ballot(n > (uint32_t)SMALL_RUN);
  while (m) {
    const int src = __ffsll((long long)m) - 1;
    m &= m - 1;
    const uint32_t nb = lane_value(n, src);
    const uint32_t bb = lane_value(base, src);
    T acc = (T)0, acc1 = (T)0, acc2 = (T)0, acc3 = (T)0;
    uint32_t k = lane;
    for (; k + 192 < nb; k += 256) {  // four independent loads in flight
      acc += inst[bb + k];
      acc1 += inst[bb + k + 64];
      acc2 += inst[bb + k + 128];
      acc3 += inst[bb + k + 192];
    }
    for (; k < nb; k += 64) acc += inst[bb + k];
    acc = (acc + acc1) + (acc2 + acc3);
    if constexpr (sizeof(T) == 4 && !std::is_integral<T>::value) {
      acc = wave_sum_to_all(acc);
    } else {
      acc = (T)wave_reduce_sum_u32((uint32_t)acc);
    }
    if (lane == src) sum = acc;
  }
  if (g < P) out[g] = sum;
}

// One wave folds everything the host reads at the frame's one synchronisation into a 64-byte record: instance total,
// depth-sort overflow flag, the three sharded statistics and up to 8 caller words (counts other kernels of the frame
// left on the device: pings_raster_preprocess_dyn).
struct AuxPtrs {
  const int32_t* p[8];
};
__global__ __launch_bounds__(64) void frame_summary_kernel(const uint32_t* __restrict__ last_offset,
                                                           const uint32_t* __restrict__ overflow,
                                                           const unsigned long long* __restrict__ shards, AuxPtrs aux,
                                                           int aux_words, FrameSummary* __restrict__ out, uint32_t seq) {
  __shared__ unsigned long long part[3][64];
  const int lane = threadIdx.x;
  unsigned long long a0 = 0, a1 = 0, a2 = 0;
  for (int i = lane; i < STAT_SHARDS; i += 64) {
    a0 += shards[3 * i]; a1 += shards[3 * i + 1]; a2 += shards[3 * i + 2];
  }
  part[0][lane] = a0; part[1][lane] = a1; part[2][lane] = a2;
  __syncthreads();
  if (lane < 3) {
    unsigned long long t = 0;
    for (int i = 0; i < 64; ++i) t += part[lane][i];
    out->stats[lane] = t;
  }
  if (lane == 3) out->total = *last_offset;
  if (lane == 4) out->overflow = overflow ? *overflow : 0u;
  if (lane >= 8 && lane < 16) out->aux[lane - 8] = (lane - 8 < aux_words && aux.p[lane - 8]) ? *aux.p[lane - 8] : 0;
  // `out` is pinned HOST memory: every field first, then the sequence number with system-scope release semantics
  __threadfence_system();
  __syncthreads();
  if (lane == 0) __hip_atomic_store(&out->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static int make_params(const pings_raster_settings* s, int P, KParams& kp) {
  PINGS_ARG_CHECK(s != nullptr, "null settings");
  PINGS_ARG_CHECK(s->image_height > 0 && s->image_width > 0, "empty image");
  PINGS_ARG_CHECK(s->image_height < 65536 * TILE / 16 && s->image_width < 65536, "image too large");
  PINGS_ARG_CHECK(s->mode == PINGS_RASTER_SURFEL || s->mode == PINGS_RASTER_3DGS, "unknown mode");
  PINGS_ARG_CHECK(s->viewmatrix && s->projmatrix_raw && s->bg, "null camera pointer");
  PINGS_ARG_CHECK(s->tanfovx > 0 && s->tanfovy > 0, "non-positive tanfov");
  kp.P = P;
  kp.W = s->image_width;
  kp.H = s->image_height;
  kp.gx = ceil_div(kp.W, TILE);
  kp.gy = ceil_div(kp.H, TILE);
  kp.front_only = s->front_only;
  kp.rect_rule = RECT_TIGHT;
  // Tiles a Gaussian covers with less than this alpha everywhere are left out of the occlusion budget: fewer entries
  // is still a lower bound of the opacity in front (conservative: the kept lists can only grow, results unchanged),
  // and the faint rim of every footprint was most of the budget pass's atomics.  Metric-1 sweep (r03): 1/255 -> 0.15
  // takes occl_budget 0.116 -> 0.053 ms and the step 1.136 -> 1.072 ms for 7.8 % more instances (0.2: 17 % more for
  // 0.005 ms); C2 / C3 unchanged.  PINGS_OCC_AMIN overrides (1/255 = every covered tile, the round-2 behaviour).
  kp.occ_amin = 0.15f;
  if (const char* e = getenv("PINGS_OCC_AMIN")) kp.occ_amin = fminf(fmaxf((float)atof(e), 1.0f / 255.0f), 0.99f);
  if (const char* e = getenv("PINGS_RASTER_RECT")) kp.rect_rule = e[0] == '3' ? RECT_3SIGMA : e[0] == 'e' ? RECT_ELLIPSE : RECT_TIGHT;
  kp.fx = (float)((double)kp.W / (2.0 * s->tanfovx));
  kp.fy = (float)((double)kp.H / (2.0 * s->tanfovy));
  kp.limx = (float)(1.3 * s->tanfovx);
  kp.limy = (float)(1.3 * s->tanfovy);
  kp.scale_mod = (float)s->scale_modifier;
  kp.view = s->viewmatrix;
  kp.proj_raw = s->projmatrix_raw;
  kp.bg = s->bg;
  kp.prcp = s->prcppoint;
  kp.live = nullptr;
  kp.dyn_rows = 0;
  return PINGS_OK;
}

static int tile_bits(int num_tiles) {
  int b = 1;
  while ((1 << b) < num_tiles) ++b;
  return b;
}

}  // namespace raster
}  // namespace pings

using namespace pings::raster;

PINGS_API int pings_raster_mark_visible(const float* positions, int N,
                                        const pings_raster_settings* s, uint8_t* present,
                                        void* stream) {
  PINGS_ARG_CHECK(s && s->viewmatrix && s->projmatrix_raw, "null settings / matrices");
  if (N == 0) return PINGS_OK;
  PINGS_ARG_CHECK(N > 0 && positions && present, "null pointer");
  // PINGS_MARK_VISIBLE=depth: the variant in which upstream's frustum test stays commented out (assumption 2)
  int depth_only = 0;
  if (const char* e = getenv("PINGS_MARK_VISIBLE")) depth_only = e[0] == 'd';
  hipLaunchKernelGGL(mark_visible_kernel, dim3(pings::ceil_div(N, 256)), dim3(256), 0,
                     pings::as_stream(stream), positions, N, s->viewmatrix, s->projmatrix_raw,
                     present, depth_only);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}

PINGS_API size_t pings_raster_geom_bytes(int P, int image_height, int image_width) {
  const int nt = pings::ceil_div(image_width, TILE) * pings::ceil_div(image_height, TILE);
  return carve_geom(nullptr, P, nt).total;
}

PINGS_API size_t pings_raster_binning_bytes(int64_t num_instances, int image_height,
                                            int image_width) {
  const int nt = pings::ceil_div(image_width, TILE) * pings::ceil_div(image_height, TILE);
  return carve_binning(nullptr, num_instances, nt).total;
}

PINGS_API size_t pings_raster_image_bytes(int image_height, int image_width) {
  return carve_image(nullptr, image_width, image_height).total;
}

PINGS_API int pings_raster_preprocess(const pings_raster_settings* s, int P, const float* means3D,
                                      const float* colors, const float* opacities,
                                      const float* scales, const float* rotations,
                                      void* geom_blob, int32_t* radii, int64_t* num_instances,
                                      int32_t* footprint_class, void* stream) {
  return pings_raster_preprocess_dyn(s, P, means3D, colors, opacities, scales, rotations, geom_blob, radii, nullptr, 0,
                                     nullptr, 0, nullptr, num_instances, footprint_class, stream);
}

PINGS_API int pings_raster_preprocess_dyn(const pings_raster_settings* s, int P, const float* means3D,
                                          const float* colors, const float* opacities, const float* scales,
                                          const float* rotations, void* geom_blob, int32_t* radii,
                                          const int32_t* live_rows_dev, int dyn_rows,
                                          const int32_t* const* aux_dev, int aux_words, int32_t* aux_host,
                                          int64_t* num_instances, int32_t* footprint_class, void* stream) {
  KParams kp;
  if (int e = make_params(s, P, kp)) return e;
  PINGS_ARG_CHECK(num_instances != nullptr && footprint_class != nullptr, "null output");
  *num_instances = 0;
  *footprint_class = 1;
  PINGS_ARG_CHECK(aux_words >= 0 && aux_words <= 8 && (aux_words == 0 || (aux_dev && aux_host)), "0..8 aux words");
  PINGS_ARG_CHECK(!live_rows_dev || (dyn_rows >= 0 && dyn_rows <= P), "dyn_rows must lie in 0..P");
  PINGS_ARG_CHECK(P > 0 || aux_words == 0, "aux words need a non-empty frame");
  if (P == 0) return PINGS_OK;
  kp.live = live_rows_dev;
  kp.dyn_rows = live_rows_dev ? dyn_rows : 0;
  PINGS_ARG_CHECK(P > 0 && means3D && colors && opacities && scales && rotations && geom_blob && radii,
                  "null pointer");
  hipStream_t st = pings::as_stream(stream);
  const int num_tiles = kp.gx * kp.gy;
  GeomState gs = carve_geom(geom_blob, P, num_tiles);
  const dim3 grid(pings::ceil_div(P, 256)), block(256);
  {
    pings::prof::Scope ps("preprocess", st);
    if (s->mode == PINGS_RASTER_SURFEL)
      hipLaunchKernelGGL(preprocess_kernel<MODE_SURFEL>, grid, block, 0, st, kp, means3D, colors,
                         opacities, scales, rotations, gs.rec, gs.rect, gs.depth_key, gs.gidx, radii);
    else
      hipLaunchKernelGGL(preprocess_kernel<MODE_3DGS>, grid, block, 0, st, kp, means3D, colors,
                         opacities, scales, rotations, gs.rec, gs.rect, gs.depth_key, gs.gidx, radii);
    PINGS_LAUNCH_CHECK();
  }
  // PINGS_RASTER_OCCLUSION=0 keeps every (Gaussian, tile) instance (A/B runs, list-parity tests);
  // PINGS_DEPTH_SORT=library forces the rocPRIM sort (A/B runs, parity tests of the bucket sort)
  bool occlusion = true;
  if (const char* e = getenv("PINGS_RASTER_OCCLUSION")) occlusion = atoi(e) != 0;
  bool library_sort = false;
  if (const char* e = getenv("PINGS_DEPTH_SORT")) library_sort = e[0] == 'l';
  // The record of the frame's one read-back is written by the summary kernel straight into pinned, device-mapped host
  // memory and the host POLLS its sequence number: no copy, and no blocking wait inside the runtime.  (Three pageable
  // copies, each a wait of its own, in round 2.  A blocking hipStreamSynchronize wakes through an interrupt; on one box
  // of this pool every wait that outlasted the runtime's spin phase — any frame of a million Gaussians — returned only
  // on a 60 Hz tick: 16 ms per frame, 1.15 -> 11.7 ms per headline step.  Polling a host word does not depend on it.)
  static thread_local FrameSummary* host_sum = nullptr;
  static thread_local FrameSummary* host_sum_dev = nullptr;
  static thread_local uint32_t frame_seq = 0;
  if (!host_sum) {
    PINGS_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&host_sum), sizeof(FrameSummary), hipHostMallocMapped | hipHostMallocPortable));
    PINGS_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&host_sum_dev), host_sum, 0));
    host_sum->seq = 0;
  }
  AuxPtrs aux;
  for (int i = 0; i < 8; ++i) aux.p[i] = i < aux_words ? aux_dev[i] : nullptr;
  uint32_t total = 0;
  // the occlusion budget, the frame statistics and the depth-sort header in one clear (adjacent in the blob); a retry
  // (depth-bucket overflow) clears again what it reuses
  PINGS_HIP_CHECK(hipMemsetAsync(gs.zero_begin, 0, gs.zero_bytes, st));
  for (int attempt = 0;; ++attempt) {
    size_t tb = gs.temp_bytes;
    if (library_sort) {
      pings::prof::Scope ps("depth_sort", st);
      PINGS_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(gs.temp, tb, gs.depth_key, gs.depth_key_sorted,
                                                         gs.gidx, gs.gidx_sorted, P, 0, 32, st));
      hipLaunchKernelGGL(count_valid_kernel, dim3(1), dim3(64), 0, st, P, gs.depth_key_sorted, gs.nvalid);
      PINGS_LAUNCH_CHECK();
      hipLaunchKernelGGL(invert_perm_kernel, grid, block, 0, st, P, gs.gidx_sorted, gs.rank_of);
      PINGS_LAUNCH_CHECK();
    } else {
      pings::prof::Scope ps("depth_sort", st);
      if (attempt > 0) PINGS_HIP_CHECK(hipMemsetAsync(gs.ds_head, 0, sizeof(uint32_t) * gs.ds_words, st));
      hipLaunchKernelGGL(ds_minmax_kernel, dim3(std::min(pings::ceil_div(P, 256), 512)), block, 0, st, P,
                         gs.depth_key, gs.ds_head);
      PINGS_LAUNCH_CHECK();
      hipLaunchKernelGGL(ds_range_kernel, dim3(1), dim3(64), 0, st, gs.ds_head);
      PINGS_LAUNCH_CHECK();
      // arrival slots of the survivors live in rank_of until ds_rank_kernel overwrites it with the final ranks
      hipLaunchKernelGGL(ds_hist_kernel, grid, block, 0, st, P, gs.depth_key, gs.ds_head, gs.ds_cnt, gs.rank_of);
      PINGS_LAUNCH_CHECK();
      PINGS_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(gs.temp, tb, gs.ds_cnt, gs.ds_off, DS_NB + (int)grid.x + 1, st));
      hipLaunchKernelGGL(ds_scatter_kernel, grid, block, 0, st, P, gs.depth_key, gs.ds_head, gs.ds_off, gs.rank_of,
                         gs.depth_key_sorted, gs.ds_idx, gs.gidx_sorted);
      PINGS_LAUNCH_CHECK();
      hipLaunchKernelGGL(ds_rank_kernel, grid, block, 0, st, gs.depth_key_sorted, gs.ds_idx, gs.ds_head, gs.ds_off,
                         gs.gidx_sorted, gs.rank_of, gs.nvalid);
      PINGS_LAUNCH_CHECK();
    }
    if (occlusion) {
      {
        pings::prof::Scope ps("occl_setup", st);
        if (attempt > 0)
          PINGS_HIP_CHECK(hipMemsetAsync(gs.occ_bucket, 0, sizeof(uint32_t) * (size_t)num_tiles * gs.occ_nb, st));
      }
      {
        pings::prof::Scope ps("occl_budget", st);
        hipLaunchKernelGGL(occl_budget_kernel, grid, block, 0, st, P, kp.gx, gs.occ_nb, gs.gidx_sorted, gs.rect,
                           gs.rec, gs.nvalid, num_tiles, gs.occ_bucket);
        PINGS_LAUNCH_CHECK();
      }
      {
        pings::prof::Scope ps("occl_scan", st);
        hipLaunchKernelGGL(occl_scan_kernel, dim3(pings::ceil_div(num_tiles, 64)), dim3(512), 0, st, num_tiles,
                           gs.occ_nb, gs.occ_bucket, gs.occ_bsat);
        PINGS_LAUNCH_CHECK();
      }
    } else {
      // every tile keeps every rank bucket (OCC_ALL); nvalid stays: rect_lane skips the culled ranks with it
      PINGS_HIP_CHECK(hipMemsetAsync(gs.occ_bsat, 0xFF, sizeof(uint16_t) * (size_t)num_tiles, st));
    }
    {
      pings::prof::Scope ps("tile_count_scan", st);
      if (attempt > 0) PINGS_HIP_CHECK(hipMemsetAsync(gs.stats, 0, 3 * STAT_SHARDS * sizeof(unsigned long long), st));
      hipLaunchKernelGGL(count_kept_kernel, grid, block, 0, st, P, kp.gx, gs.occ_nb, gs.gidx_sorted, gs.rect,
                         gs.occ_bsat, gs.nvalid, gs.tiles_sorted, gs.stats);
      PINGS_LAUNCH_CHECK();
      tb = gs.temp_bytes;
      PINGS_HIP_CHECK(hipcub::DeviceScan::InclusiveSum(gs.temp, tb, gs.tiles_sorted, gs.offsets_sorted,
                                                       P, st));
    }
    const uint32_t seq = ++frame_seq ? frame_seq : ++frame_seq;   // never 0
    hipLaunchKernelGGL(frame_summary_kernel, dim3(1), dim3(64), 0, st, gs.offsets_sorted + (P - 1),
                       library_sort ? (const uint32_t*)nullptr : gs.ds_head + DS_FLAG, gs.stats, aux, aux_words,
                       host_sum_dev, seq);
    PINGS_LAUNCH_CHECK();
    {
      const auto t_start = std::chrono::steady_clock::now();
      unsigned spins = 0;
      while (__atomic_load_n(&host_sum->seq, __ATOMIC_ACQUIRE) != seq) {
        if ((++spins & 0x3FFu) == 0) {
          // every 1024 polls: has the stream failed, or has this taken absurdly long?  Then let the runtime decide.
          const hipError_t q = hipStreamQuery(st);
          if (q == hipErrorNotReady) (void)hipGetLastError();   // "not ready" must not linger as the thread's last error
          else if (q != hipSuccess) PINGS_HIP_CHECK(q);
          if (std::chrono::steady_clock::now() - t_start > std::chrono::seconds(20)) {
            PINGS_HIP_CHECK(hipStreamSynchronize(st));
            PINGS_ARG_CHECK(__atomic_load_n(&host_sum->seq, __ATOMIC_ACQUIRE) == seq, "frame summary never arrived");
          }
        }
        __builtin_ia32_pause();
      }
    }
    total = host_sum->total;
    if (host_sum->overflow == 0) break;
    library_sort = true;  // a depth bucket overflowed: redo the frame with the library sort
  }
  *num_instances = (int64_t)total;
  const unsigned long long stats[3] = {host_sum->stats[0], host_sum->stats[1], host_sum->stats[2]};
  for (int i = 0; i < aux_words; ++i) aux_host[i] = host_sum->aux[i];
  PINGS_ARG_CHECK(stats[2] == (unsigned long long)total && stats[2] < 0x7FFFFFFFull,
                  "more than 2^31 - 1 (Gaussian, tile) instances in this frame");
  // Footprints of many tiles keep most lanes of a wave busy: two pixels per lane then amortise the per-record
  // work; small footprints leave lanes idle and one pixel per lane (four 8x8 waves with their own culled lists)
  // wins (measured: 52 tiles per Gaussian -> PPL 2 is 6 % faster, 5.6 tiles per Gaussian -> PPL 1 is 19 % faster).
  *footprint_class = (stats[1] > 0 && stats[0] > 16ull * stats[1]) ? 2 : 1;
  return PINGS_OK;
}

PINGS_API int pings_raster_render(const pings_raster_settings* s, int P, int64_t I,
                                  void* geom_blob, void* binning_blob, void* image_blob,
                                  float* out_color, float* out_normal,
                                  float* out_depth, float* out_alpha, void* per_gaussian,
                                  int footprint_class, void* stream) {
  KParams kp;
  if (int e = make_params(s, P, kp)) return e;
  PINGS_ARG_CHECK(out_color && out_depth && out_alpha && image_blob && binning_blob, "null pointer");
  PINGS_ARG_CHECK(s->mode == PINGS_RASTER_3DGS || out_normal, "surfel mode needs out_normal");
  PINGS_ARG_CHECK(I >= 0 && I < (int64_t)0x7FFFFFFF, "instance count out of range");
  PINGS_ARG_CHECK(P == 0 || (geom_blob && per_gaussian), "null pointer");
  hipStream_t st = pings::as_stream(stream);
  const int num_tiles = kp.gx * kp.gy;
  GeomState gs = carve_geom(geom_blob, P, num_tiles);
  BinState bs = carve_binning(binning_blob, I, num_tiles);
  ImageState im = carve_image(image_blob, kp.W, kp.H);

  {
    // tile ranges, per-instance weights and quadrant masks (3DGS: contributor counts too) in one clear
    const char* z0 = reinterpret_cast<const char*>(bs.ranges);
    const char* z1 = I > 0 ? (s->mode == PINGS_RASTER_3DGS ? reinterpret_cast<const char*>(bs.inst_cnt + I)
                                                             : reinterpret_cast<const char*>(bs.inst_qmask + I + 1))
                           : reinterpret_cast<const char*>(bs.ranges + num_tiles);
    PINGS_HIP_CHECK(hipMemsetAsync(bs.ranges, 0, (size_t)(z1 - z0), st));
  }
  if (I > 0) {
    // tile ids fit 16 bits for every image up to 4096x4096: a 2-byte key cuts the sort traffic by a quarter
    const bool k16 = num_tiles <= 65536;
    uint16_t* key16 = reinterpret_cast<uint16_t*>(bs.tile_key);
    uint16_t* key16s = reinterpret_cast<uint16_t*>(bs.tile_key_sorted);
    {
      pings::prof::Scope ps("duplicate", st);
      if (k16)
        hipLaunchKernelGGL(duplicate_kernel<uint16_t>, dim3(pings::ceil_div(P, 256)), dim3(256), 0, st, P,
                           kp.gx, gs.occ_nb, gs.gidx_sorted, gs.offsets_sorted, gs.tiles_sorted, gs.rect,
                           gs.occ_bsat, gs.nvalid, key16, bs.gval, bs.slot_val);
      else
        hipLaunchKernelGGL(duplicate_kernel<uint32_t>, dim3(pings::ceil_div(P, 256)), dim3(256), 0, st, P,
                           kp.gx, gs.occ_nb, gs.gidx_sorted, gs.offsets_sorted, gs.tiles_sorted, gs.rect,
                           gs.occ_bsat, gs.nvalid, bs.tile_key, bs.gval, bs.slot_val);
      PINGS_LAUNCH_CHECK();
    }
    {
      pings::prof::Scope ps("tile_sort", st);
      size_t tb = bs.temp_bytes;
      if (k16)
        PINGS_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(bs.temp, tb, key16, key16s, bs.slot_val,
                                                           bs.point_list, (int)I, 0, tile_bits(num_tiles),
                                                           st));
      else
        PINGS_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(bs.temp, tb, bs.tile_key, bs.tile_key_sorted,
                                                           bs.slot_val, bs.point_list, (int)I, 0,
                                                           tile_bits(num_tiles), st));
    }
    {
      pings::prof::Scope ps("tile_ranges", st);
      const dim3 rg((unsigned)pings::ceil_div<int64_t>(I, 256));
      if (k16)
        hipLaunchKernelGGL(tile_ranges_kernel<uint16_t>, rg, dim3(256), 0, st, I, (const uint16_t*)key16s,
                           bs.ranges);
      else
        hipLaunchKernelGGL(tile_ranges_kernel<uint32_t>, rg, dim3(256), 0, st, I,
                           (const uint32_t*)bs.tile_key_sorted, bs.ranges);
      PINGS_LAUNCH_CHECK();
    }
  }
  {
    // forward dispatch order: tiles by descending list length
    pings::prof::Scope ps_o("tile_order", st);
    hipLaunchKernelGGL(range_len_kernel, dim3(pings::ceil_div(num_tiles, 256)), dim3(256), 0, st, bs.ranges, num_tiles,
                       bs.tile_work);
    PINGS_LAUNCH_CHECK();
    if (int e = launch_tile_order(bs.tile_work, num_tiles, bs.tile_order, st)) return e;
  }
  {
  pings::prof::Scope ps_blend("blend_fwd", st);
  // the wave-per-quadrant kernel is the fastest forward on both footprint classes (Metric-1: 0.244 vs 0.265 ms for
  // the workgroup-per-tile kernel with two pixels per lane; street-like scene: 0.86 vs 1.31 ms with one);
  // PINGS_BLEND_PPL = 1 | 2 selects the workgroup-per-tile kernel with that many pixels per lane (A/B runs, tests)
  int ppl = 0;
  if (const char* e = getenv("PINGS_BLEND_PPL")) ppl = atoi(e);
  // (the per-instance sums the workgroup-per-tile and wave-per-tile kernels accumulate into were cleared with `ranges`)
  // exact per-quadrant masks are needed iff the backward pass of this view will run the Gaussian-per-lane kernel
  // (same predicate as pings_raster_backward: footprint class, PINGS_BLEND_BWD override); they only cost something
  // in the 2-pixels-per-lane forward
  int want_qmask = footprint_class != 2;
  if (const char* e = getenv("PINGS_BLEND_BWD")) want_qmask = strcmp(e, "pixel") != 0;
#define PINGS_BLEND_FWD(M, L)                                                                          \
  hipLaunchKernelGGL((blend_fwd_kernel<M, L>), dim3(num_tiles), dim3(BLOCK / L), 0, st, kp, bs.ranges,  \
                     bs.point_list, gs.rec, bs.gval, out_color, out_normal, out_depth, out_alpha,      \
                     im.final_T, im.n_contrib, bs.inst_w, bs.inst_cnt, bs.inst_qmask, want_qmask)
  // long lists in parallel segments (see blend_fwd_seg_kernel): PINGS_BLEND_SEG = entries per segment, 0 = off
  const uint32_t seg = blend_segment_entries();
  const bool seg_on = seg > 0 && I > (int64_t)num_tiles * (seg / 4) && I > 2 * (int64_t)seg;
  // pass B walks pass T's compacted lists (16-bit offsets); PINGS_BLEND_SEG_REUSE=0 re-tests (A/B runs, tests)
  bool seg_reuse = seg <= 65535u;
  if (const char* e = getenv("PINGS_BLEND_SEG_REUSE")) seg_reuse = seg_reuse && atoi(e) != 0;
#define PINGS_BLEND_FWD_WAVE(M)                                                                        \
  do {                                                                                                 \
    if (I > 0) PINGS_HIP_CHECK(hipMemsetAsync(bs.inst_wq, 0, 16 * (size_t)I, st));                     \
    if (I > 0 && M == MODE_3DGS) PINGS_HIP_CHECK(hipMemsetAsync(bs.inst_cntq, 0, 16 * (size_t)I, st)); \
    if (seg_on) {                                                                                      \
      hipLaunchKernelGGL(seg_plan_kernel, dim3(1), dim3(1024), 0, st, bs.ranges, num_tiles, 2u * seg, seg, \
                         bs.seg_max_units, bs.seg_head, bs.seg_unit_tile, bs.seg_unit_seg, bs.seg_tile_unit0); \
      PINGS_LAUNCH_CHECK();                                                                            \
    }                                                                                                  \
    if (!seg_on) {                                                                                     \
      hipLaunchKernelGGL((blend_fwd_wave_kernel<M>), dim3(num_tiles), dim3(256), 0, st, kp, bs.ranges, \
                         bs.point_list, gs.rec, bs.gval, out_color, out_normal, out_depth, out_alpha,  \
                         im.final_T, im.n_contrib, bs.inst_wq, bs.inst_cntq, bs.tile_order,            \
                         (const uint32_t*)nullptr);                                                    \
    } else {                                                                                           \
      /* short-list tiles + pass T of the segments in one launch, then pass B and pass C */            \
      const dim3 gseg(bs.seg_max_units);                                                               \
      hipLaunchKernelGGL((blend_fwd_wave_segT_kernel<M>), dim3((unsigned)num_tiles + bs.seg_max_units), dim3(256), 0, \
                         st, kp, bs.ranges, bs.point_list, gs.rec, bs.gval, out_color, out_normal, out_depth, \
                         out_alpha, im.final_T, im.n_contrib, bs.inst_wq, bs.inst_cntq, bs.tile_order, \
                         bs.seg_tile_unit0, (unsigned)num_tiles, seg, bs.seg_head, bs.seg_unit_tile,   \
                         bs.seg_unit_seg, bs.seg_P, seg_reuse ? bs.seg_rel : (uint16_t*)nullptr,      \
                         seg_reuse ? bs.seg_nrel : (uint32_t*)nullptr);                                \
      if (seg_reuse)                                                                                   \
        hipLaunchKernelGGL((blend_fwd_seg_kernel<M, 1, true>), gseg, dim3(256), 0, st, kp, bs.ranges, bs.point_list, \
                           gs.rec, bs.gval, seg, bs.seg_head, bs.seg_unit_tile, bs.seg_unit_seg, bs.seg_P, bs.seg_slab, \
                           bs.inst_wq, bs.inst_cntq, bs.seg_rel, bs.seg_nrel);                         \
      else                                                                                             \
        hipLaunchKernelGGL((blend_fwd_seg_kernel<M, 1, false>), gseg, dim3(256), 0, st, kp, bs.ranges, bs.point_list, \
                           gs.rec, bs.gval, seg, bs.seg_head, bs.seg_unit_tile, bs.seg_unit_seg, bs.seg_P, bs.seg_slab, \
                           bs.inst_wq, bs.inst_cntq, (uint16_t*)nullptr, (uint32_t*)nullptr);          \
      hipLaunchKernelGGL((blend_fwd_seg_combine_kernel<M>), gseg, dim3(256), 0, st, kp, bs.ranges, seg, bs.seg_head, \
                         bs.seg_unit_tile, bs.seg_unit_seg, bs.seg_slab, out_color, out_normal, out_depth, \
                         out_alpha, im.final_T, im.n_contrib);                                         \
      PINGS_LAUNCH_CHECK();                                                                            \
    }                                                                                                  \
    if (I > 0)                                                                                         \
      hipLaunchKernelGGL((combine_quadrants_kernel<M>), dim3((unsigned)pings::ceil_div<int64_t>(I, 256)), \
                         dim3(256), 0, st, I, reinterpret_cast<const float4*>(bs.inst_wq),              \
                         reinterpret_cast<const uint4*>(bs.inst_cntq), bs.inst_w, bs.inst_cnt, bs.inst_qmask); \
  } while (0)
  // footprint class 2 with the pixel-per-lane backward to follow (no quadrant masks needed): wave per TILE, four pixels
  // per lane (Metric-1: 0.245 -> see DESIGN); PINGS_BLEND_PPL=4 forces it, PINGS_BLEND_PPL=-1 forces the quadrant kernel
  const bool tile_wave = (ppl == 4) || (ppl == 0 && footprint_class == 2 && !want_qmask);
#define PINGS_BLEND_FWD_TILE(M)                                                                        \
  do {                                                                                                 \
    hipLaunchKernelGGL((blend_fwd_tile_kernel<M>), dim3(num_tiles), dim3(64), 0, st, kp, bs.ranges,     \
                       bs.point_list, gs.rec, bs.gval, out_color, out_normal, out_depth, out_alpha,    \
                       im.final_T, im.n_contrib, bs.inst_w, bs.inst_cnt, bs.tile_order);               \
  } while (0)
  if (s->mode == PINGS_RASTER_SURFEL) {
    if (ppl == 1) PINGS_BLEND_FWD(MODE_SURFEL, 1);
    else if (ppl == 2) PINGS_BLEND_FWD(MODE_SURFEL, 2);
    else if (tile_wave) PINGS_BLEND_FWD_TILE(MODE_SURFEL);
    else PINGS_BLEND_FWD_WAVE(MODE_SURFEL);
  } else {
    if (ppl == 1) PINGS_BLEND_FWD(MODE_3DGS, 1);
    else if (ppl == 2) PINGS_BLEND_FWD(MODE_3DGS, 2);
    else if (tile_wave) PINGS_BLEND_FWD_TILE(MODE_3DGS);
    else PINGS_BLEND_FWD_WAVE(MODE_3DGS);
  }
#undef PINGS_BLEND_FWD_TILE
#undef PINGS_BLEND_FWD
#undef PINGS_BLEND_FWD_WAVE
  PINGS_LAUNCH_CHECK();
  }
  if (P > 0) {
    pings::prof::Scope ps("per_gaussian_sum", st);
    const dim3 grid(pings::ceil_div(P, 256)), block(256);
    if (I == 0) {
      PINGS_HIP_CHECK(hipMemsetAsync(per_gaussian, 0, 4 * (size_t)P, st));
    } else if (s->mode == PINGS_RASTER_SURFEL) {
      hipLaunchKernelGGL(per_gaussian_sum_kernel<float>, grid, block, 0, st, P, gs.rect, gs.rank_of,
                         gs.offsets_sorted, gs.tiles_sorted, (const float*)bs.inst_w,
                         reinterpret_cast<float*>(per_gaussian));
    } else {
      hipLaunchKernelGGL(per_gaussian_sum_kernel<uint32_t>, grid, block, 0, st, P, gs.rect, gs.rank_of,
                         gs.offsets_sorted, gs.tiles_sorted, (const uint32_t*)bs.inst_cnt,
                         reinterpret_cast<uint32_t*>(per_gaussian));
    }
    PINGS_LAUNCH_CHECK();
  }
  return PINGS_OK;
}

__global__ void slots_to_ids_kernel(int64_t I, const uint32_t* __restrict__ list, const uint32_t* __restrict__ gval,
                                    uint32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < I) out[i] = gval[list[i]];
}

#ifdef PINGS_BLEND_STATS
PINGS_API int pings_debug_blend_stats(unsigned long long* out8, int reset) {
  PINGS_HIP_CHECK(hipDeviceSynchronize());
  PINGS_HIP_CHECK(hipMemcpyFromSymbol(out8, HIP_SYMBOL(pings::raster::g_blend_stats), 64));
  if (reset) {
    unsigned long long z[8] = {0};
    PINGS_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(pings::raster::g_blend_stats), z, 64));
  }
  return PINGS_OK;
}
#endif

PINGS_API int pings_raster_debug_lists(const void* binning_blob, int64_t I, int image_height,
                                       int image_width, uint32_t* point_list, uint32_t* ranges_xy,
                                       void* stream) {
  PINGS_ARG_CHECK(binning_blob && ranges_xy, "null pointer");
  const int nt = pings::ceil_div(image_width, TILE) * pings::ceil_div(image_height, TILE);
  BinState bs = carve_binning(const_cast<void*>(binning_blob), I, nt);
  hipStream_t st = pings::as_stream(stream);
  if (I > 0 && point_list) {  // sorted list holds instance slots; the tap returns Gaussian ids
    hipLaunchKernelGGL(slots_to_ids_kernel, dim3((unsigned)pings::ceil_div<int64_t>(I, 256)), dim3(256), 0, st, I,
                       bs.point_list, bs.gval, point_list);
    PINGS_LAUNCH_CHECK();
  }
  PINGS_HIP_CHECK(hipMemcpyAsync(ranges_xy, bs.ranges, sizeof(uint2) * (size_t)nt,
                                 hipMemcpyDeviceToDevice, st));
  return PINGS_OK;
}

PINGS_API int pings_raster_debug_image(const void* image_blob, int image_height, int image_width,
                                       float* final_T, uint32_t* n_contrib, void* stream) {
  PINGS_ARG_CHECK(image_blob && final_T && n_contrib, "null pointer");
  ImageState im = carve_image(const_cast<void*>(image_blob), image_width, image_height);
  hipStream_t st = pings::as_stream(stream);
  const size_t n = (size_t)image_height * image_width;
  PINGS_HIP_CHECK(hipMemcpyAsync(final_T, im.final_T, sizeof(float) * n, hipMemcpyDeviceToDevice, st));
  PINGS_HIP_CHECK(hipMemcpyAsync(n_contrib, im.n_contrib, sizeof(uint32_t) * n,
                                 hipMemcpyDeviceToDevice, st));
  return PINGS_OK;
}
