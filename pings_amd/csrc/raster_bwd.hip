// Gaussian(-surfel) rasteriser, backward pass, for gfx950.  No floating-point atomics anywhere.
//
//   live scan             cidx = exclusive scan of (inst_w > 0) over the instance slots: only
//                         instances that blended something in the forward pass get a gradient row.
//   blend_bwd_kernel      one 256-thread workgroup per 16x16 tile walks the tile's sorted list
//                         back to front in batches of 64 records staged in LDS; dead instances are
//                         skipped by the whole workgroup.  Each pixel re-derives alpha /
//                         transmittance and forms the 16 per-(pixel, Gaussian) gradient terms of the
//                         blend record; the wave reduces them with a transposed DPP reduce-scatter
//                         (16 values x 64 lanes -> one 64-B row in 16 lanes, ~55 VALU ops instead of
//                         96 for 16 separate wave sums).  The four waves' rows are summed in LDS in
//                         fixed order and stored ONCE per live instance as a 64-B row; the rows of
//                         one Gaussian are contiguous (slot order = depth rank order).
//   row_chunk_sum_kernel  balanced first level of the per-Gaussian sum: one 16-lane group per
//                         (Gaussian, chunk of <= 64 rows) pair, lane = column, streaming 64-B rows.
//   gaussian_bwd_kernel   per Gaussian (in index order, rows located through its depth rank): adds its chunk partials, then chains
//                         through conic / EWA projection / quaternion / scale / camera transform to
//                         the input gradients and the pose-tangent terms (block-reduced, fixed order).
//   tau_reduce_kernel     final fixed-order reduction of the pose-tangent partials.
//
// Gradient definitions are those torch autograd derives from oracle/raster_cpu.py (clamps have
// zero gradient where active, min(0.99, .) included).
#include <hipcub/hipcub.hpp>

#include <cstdlib>
#include <cstring>

#include "raster_common.hpp"

namespace pings {
namespace raster {

struct BParams {
  int P, W, H, gx, gy;
  int front_only;
  float fx, fy, limx, limy, scale_mod;
  const float* view;
  const float* proj_raw;
  const float* bg;
  const float* prcp;
};

constexpr int BATCH = 64;
// blend_bwd_scan_kernel: tiles whose largest per-pixel contributor count reaches the threshold get four waves per
// quadrant (PINGS_BWD_LONG overrides the threshold, 0 = never); at most LONG_TILES_MAX tiles per frame.  C3 street
// sweep (r03, kernel ms): never 0.72, 256 0.65, 768 0.59, 2048 0.535, 3072 0.53, 4096 0.52, 8192 0.57 — the split
// costs four queue walks and two barriers per chunk, so only the lists that set the kernel's duration should pay it.
constexpr uint32_t LONG_TILES_MAX = 2048;
inline uint32_t long_list_threshold() {
  long v = 3072;
  if (const char* e = getenv("PINGS_BWD_LONG")) v = atol(e);   // read per call: tests switch it
  if (v <= 0) return 0xFFFFFFF0u;
  return (uint32_t)((v + 15) / 16 * 16);
}
constexpr uint32_t DEAD_ROW = 0xFFFFFFFFu;
constexpr int CH = 64;  // rows per first-level chunk of the per-Gaussian sum

// PPL = pixels per lane (same lane -> pixel map as blend_fwd_kernel): the 16 gradient terms of a
// lane's PPL pixels are summed in registers before the wave reduction, which is the expensive part.
template <int MODE, int PPL>
__global__ __launch_bounds__(BLOCK / PPL, PPL == 4 ? 3 : 4) void blend_bwd_kernel(
    BParams p, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ rec, const uint32_t* __restrict__ gval,
    const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ out_depth, const float* __restrict__ dL_dcolor,
    const float* __restrict__ dL_dnormal, const float* __restrict__ dL_ddepth,
    const float* __restrict__ dL_dalpha, const float* __restrict__ inst_w,
    const uint32_t* __restrict__ cidx, float* __restrict__ rows, const uint32_t* __restrict__ tile_order) {
  constexpr int NT = BLOCK / PPL;
  constexpr int NWV = NT / 64;
  __shared__ float4 sA[BATCH], sB[BATCH], sC[BATCH], sD[BATCH];
  __shared__ uint32_t sRow[BATCH];  // compact gradient-row index, DEAD_ROW for dead instances
  __shared__ float4 sG[NWV][BATCH][4];
  __shared__ uint32_t sMax;
  __shared__ uint8_t sList[NWV][BATCH];  // per-wave list of live records that may reach its pixels (ascending)
  __shared__ int sNum[NWV];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int tile = (int)tile_order[blockIdx.x];   // longest-processing-time-first dispatch (raster_fwd.hip)
  const int tx = tile % p.gx, ty = tile / p.gx;
  // lane -> pixel map and per-wave culled record lists as in blend_fwd_kernel (8x8 quadrants)
  // PPL 1: wave = 8x8 quadrant; PPL 2: wave = 8x16 column strip (k -> y half); PPL 4: ONE wave owns the 16x16 tile
  // (k & 1 -> x half, k >> 1 -> y half): no second wave to wait for, and the per-record overhead (LDS record fetch,
  // 16-value wave reduction) is paid once per 256 pixels
  constexpr int NXH = PPL == 4 ? 2 : 1;   // x halves per lane
  const int pix_x = tx * TILE + (PPL == 1 ? 8 * (wave & 1) : (PPL == 2 ? 8 * wave : 0)) + (lane & 7);
  const int pix_y0 = ty * TILE + (PPL == 1 ? 8 * (wave >> 1) : 0) + (lane >> 3);
  const float pixf_x = (float)pix_x, pixf_y0 = (float)pix_y0;
  auto xoff = [](int k) { return PPL == 4 ? 8 * (k & 1) : 0; };
  auto yoff = [](int k) { return PPL == 4 ? 8 * (k >> 1) : 8 * k; };
  const float tileX0 = (float)(tx * TILE), tileY0 = (float)(ty * TILE);
  const size_t HW = (size_t)p.W * p.H;

  float rxv[NXH], ry[PPL];
#pragma unroll
  for (int h = 0; h < NXH; ++h) rxv[h] = 0.f;
#pragma unroll
  for (int k = 0; k < PPL; ++k) ry[k] = 0.f;
  if (MODE == MODE_SURFEL) {
    const float cxp = (p.prcp ? p.prcp[0] : 0.5f) * (float)p.W - 0.5f;
    const float cyp = (p.prcp ? p.prcp[1] : 0.5f) * (float)p.H - 0.5f;
#pragma unroll
    for (int h = 0; h < NXH; ++h) rxv[h] = ((pixf_x + (float)(8 * h)) - cxp) / p.fx;
#pragma unroll
    for (int k = 0; k < PPL; ++k) ry[k] = ((pixf_y0 + (float)yoff(k)) - cyp) / p.fy;
  }

  const uint2 range = ranges[tile];
  uint32_t last[PPL];
  float T[PPL], gC0[PPL], gC1[PPL], gC2[PPL], gN0[PPL], gN1[PPL], gN2[PPL], gD[PPL], coefT[PPL];
  float Bs[PPL];  // blend of s = (upstream gradient) . (record features) over the records behind, normalised
  uint32_t lmax = 0;
#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    last[k] = 0;
    T[k] = 1.f;
    gC0[k] = gC1[k] = gC2[k] = gN0[k] = gN1[k] = gN2[k] = gD[k] = coefT[k] = 0.f;
    Bs[k] = 0.f;
    const int pix_y = pix_y0 + yoff(k), pix_xk = pix_x + xoff(k);
    if (pix_xk < p.W && pix_y < p.H) {
      const size_t pix_id = (size_t)pix_y * p.W + pix_xk;
      last[k] = n_contrib[pix_id];
      const float T_final = final_T[pix_id];
      T[k] = T_final;
      if (dL_dcolor) {
        gC0[k] = dL_dcolor[pix_id];
        gC1[k] = dL_dcolor[HW + pix_id];
        gC2[k] = dL_dcolor[2 * HW + pix_id];
      }
      float gA = dL_dalpha ? dL_dalpha[pix_id] : 0.f;
      const float gDo = dL_ddepth ? dL_ddepth[pix_id] : 0.f;
      if (MODE == MODE_SURFEL) {
        if (dL_dnormal) {
          gN0[k] = dL_dnormal[pix_id];
          gN1[k] = dL_dnormal[HW + pix_id];
          gN2[k] = dL_dnormal[2 * HW + pix_id];
        }
        const float A = 1.0f - T_final;
        if (A > DEPTH_ALPHA_EPS) {
          gD[k] = gDo / A;
          gA -= gDo * out_depth[pix_id] / A;
        }
      } else {
        gD[k] = gDo;
      }
      const float bgdot = (p.bg[0] * gC0[k] + p.bg[1] * gC1[k]) + p.bg[2] * gC2[k];
      coefT[k] = (gA - bgdot) * T_final;
    }
    lmax = max(lmax, last[k]);
  }

  if (tid == 0) sMax = 0u;
  __syncthreads();
  {
    uint32_t m = lmax;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
    if (lane == 0) atomicMax(&sMax, m);
  }
  __syncthreads();
  const int max_last = (int)sMax;

  const int nbatch = ceil_div(max_last, BATCH);
  for (int b = nbatch - 1; b >= 0; --b) {
    const int start = b * BATCH;
    const int n = min(BATCH, max_last - start);
    __syncthreads();  // previous batch's LDS fully consumed
    if (wave == 0) {  // BATCH == 64: the first wave stages the batch and builds every wave's list
      uint32_t qm = 0;
      if (tid < n) {
        const uint32_t slot = point_list[range.x + start + tid];
        const uint32_t g = gval[slot];
        const float4 ra = rec[4 * (size_t)g + 0];
        const float4 rb = rec[4 * (size_t)g + 1];
        sA[tid] = ra;
        sB[tid] = rb;
        sC[tid] = rec[4 * (size_t)g + 2];
        if (MODE == MODE_SURFEL) sD[tid] = rec[4 * (size_t)g + 3];
        const bool live = inst_w[slot] > 0.f;  // blended something in the forward pass
        sRow[tid] = live ? cidx[slot] : DEAD_ROW;
        if (live) qm = quadrant_mask(ra.x, ra.y, ra.z, rb.x, rb.y, rb.z, tileX0, tileY0);
      }
#pragma unroll
      for (int wv = 0; wv < NWV; ++wv) {
        const uint32_t need = PPL == 1 ? (1u << wv) : (PPL == 2 ? ((1u << wv) | (4u << wv)) : 0xFu);
        const bool hit = (qm & need) != 0;
        const unsigned long long bal = __ballot(hit);
        if (hit)
          sList[wv][__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] =
              (uint8_t)tid;
        if (lane == 0) sNum[wv] = __popcll(bal);
      }
    }
    {
      float4* z = &sG[0][0][0];
      const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int e = tid; e < NWV * BATCH * 4; e += NT) z[e] = zero;
    }
    __syncthreads();

    // Record j-1 is read from LDS while j is processed; per-lane conditions are selects, the only
    // branches are workgroup- or wave-uniform.
    // Two-deep software pipeline over the wave's list: the INDEX of record jj-2 and the RECORD jj-1 are fetched
    // from LDS while record jj is processed, so neither LDS latency (list byte -> record address -> 64-B record)
    // sits in an iteration's dependency chain.
    const int cnt = sNum[wave];
    int jcur = cnt > 0 ? (int)sList[wave][cnt - 1] : 0;
    int jnext = cnt > 1 ? (int)sList[wave][cnt - 2] : 0;
    float4 a = sA[jcur], bq = sB[jcur], c = sC[jcur], nn = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == MODE_SURFEL) nn = sD[jcur];
    // (unrolling this loop by two so that a record's reduction overlaps the next record's pixel math: no change, 0.349 ms)
    for (int jj = cnt - 1; jj >= 0; --jj) {
      const int j = jcur;
      const int jn = jnext;
      const int jn2 = (int)sList[wave][jj > 1 ? jj - 2 : 0];
      const float4 a_n = sA[jn], b_n = sB[jn], c_n = sC[jn];
      float4 n_n = nn;
      if (MODE == MODE_SURFEL) n_n = sD[jn];
      const float4 ca = a, cb = bq, cc = c, cn = nn;
      a = a_n; bq = b_n; c = c_n; nn = n_n;
      jcur = jn;
      jnext = jn2;
      const uint32_t idx = (uint32_t)(start + j);
      float dxv[NXH], p0v[NXH], pxyv[NXH];
#pragma unroll
      for (int h = 0; h < NXH; ++h) {
        dxv[h] = ca.x - (pixf_x + (float)(8 * h));
        p0v[h] = -0.5f * (cb.x * dxv[h] * dxv[h]);
        pxyv[h] = cb.y * dxv[h];
      }
      float Gs[PPL], raw[PPL], alpha[PPL], dyv[PPL];
      bool valid[PPL];
      bool any_v = false;
#pragma unroll
      for (int k = 0; k < PPL; ++k) {
        const float dy = ca.y - (pixf_y0 + (float)yoff(k));
        dyv[k] = dy;
        const float p0 = p0v[PPL == 4 ? (k & 1) : 0], pxy = pxyv[PPL == 4 ? (k & 1) : 0];
        const float power = (p0 - 0.5f * (cb.z * dy * dy)) - pxy * dy;
        Gs[k] = __expf(power);
        raw[k] = ca.z * Gs[k];
        alpha[k] = fminf(ALPHA_MAX, raw[k]);
        valid[k] = idx < last[k] && (power <= 0.0f) && (alpha[k] >= ALPHA_MIN);
        any_v = any_v || valid[k];
      }
      if (!__any(any_v)) continue;

      float v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = 0.f;
      const float zlo = ca.w - cb.w, zhi = ca.w + cb.w;
#pragma unroll
      for (int k = 0; k < PPL; ++k) {
        // gradient accumulation only (the alpha / validity decisions were taken above, un-contracted, exactly as in
        // the forward pass): let multiply-adds fuse here — fewer instructions, one rounding less per term
#pragma clang fp contract(fast)
        // (a uniform `if (!__any(valid[k])) continue;` per pixel set here: 0.346 -> 0.399 ms — four more branches per
        // record cost more than the skipped sets save)
        const float dy = dyv[k];
        const float dx = dxv[PPL == 4 ? (k & 1) : 0], rx = rxv[PPL == 4 ? (k & 1) : 0];
        const float one_m = 1.0f - alpha[k];
        const float inv_one_m = __builtin_amdgcn_rcpf(one_m);  // 1-ulp reciprocal: alpha <= 0.99
        const float Tn = T[k] * inv_one_m;
        const float w = valid[k] ? alpha[k] * Tn : 0.f;
        // dL/dalpha needs sum_ch (feature_ch - B_ch) g_ch with B the normalised blend of the records behind.  Both
        // the blend recurrence and the dot product are linear, so ONE scalar per pixel is tracked instead of the
        // eight channels:  s = g . feature,  dLda = s - Bs,  Bs <- Bs + alpha (s - Bs).
        float sdot = (cc.x * gC0[k] + cc.y * gC1[k]) + cc.z * gC2[k];
        const float av = valid[k] ? alpha[k] : 0.f;  // only contributing lanes enter the blend
        v[G_R] = fmaf(gC0[k], w, v[G_R]);
        v[G_G] = fmaf(gC1[k], w, v[G_G]);
        v[G_B] = fmaf(gC2[k], w, v[G_B]);
        if (MODE == MODE_SURFEL) {
          sdot += (cn.x * gN0[k] + cn.y * gN1[k]) + cn.z * gN2[k];
          // per-pixel depth of this surfel
          const float den = (cn.x * rx + cn.y * ry[k]) + cn.z;
          const bool hit = den < -DEN_EPS;
          const float inv_den = __builtin_amdgcn_rcpf(den);
          const float d0 = hit ? cc.w * inv_den : ca.w;
          const float d = fminf(fmaxf(d0, zlo), zhi);
          sdot = fmaf(d, gD[k], sdot);
          const float gd = gD[k] * w;
          const bool lo = d0 < zlo, hi = d0 > zhi;
          const bool mid = !lo && !hi;
          v[G_ZLO] += lo ? gd : 0.f;
          v[G_ZHI] += hi ? gd : 0.f;
          const float gq = (mid && hit) ? gd * inv_den : 0.f;
          v[G_Q] += gq;
          v[G_PZ] += (mid && !hit) ? gd : 0.f;
          const float gden = -gq * d0;  // = -gd * d0 / den on the unclamped ray hit, else 0
          v[G_NX] += fmaf(gden, rx, gN0[k] * w);
          v[G_NY] += fmaf(gden, ry[k], gN1[k] * w);
          v[G_NZ] += gN2[k] * w + gden;
        } else {
          sdot = fmaf(ca.w, gD[k], sdot);
          v[G_PZ] = fmaf(gD[k], w, v[G_PZ]);
        }
        float dLda = sdot - Bs[k];
        Bs[k] = fmaf(av, dLda, Bs[k]);
        dLda = fmaf(dLda, Tn, coefT[k] * inv_one_m);
        // alpha = min(0.99, opacity * G): no gradient through the clamp when it is active
        dLda = (valid[k] && raw[k] <= ALPHA_MAX) ? dLda : 0.f;
        v[G_OPAC] = fmaf(Gs[k], dLda, v[G_OPAC]);
        const float dLp = raw[k] * dLda;  // dL/dpower = G * (opacity * dL/dalpha)
        // q = conic . d.  d power / d mean = -q, and d power / d cov2D = 0.5 q q^T: accumulating the
        // gradient w.r.t. the 2-D COVARIANCE here (instead of w.r.t. the conic, to be pushed through
        // -conic G conic afterwards) keeps every per-pixel term O(1); the conic form sums d d^T terms
        // of order 1e6 that must cancel to O(1) for large anisotropic footprints, which fp32 cannot do.
        const float qx = cb.x * dx + cb.y * dy, qy = cb.y * dx + cb.z * dy;
        v[G_MX] -= dLp * qx;
        v[G_MY] -= dLp * qy;
        v[G_CONX] += 0.5f * qx * qx * dLp;
        v[G_CONY] += qx * qy * dLp;
        v[G_CONZ] += 0.5f * qy * qy * dLp;
        T[k] = valid[k] ? Tn : T[k];
      }
      const float tot = wave_reduce16(v, lane);
      // (one-wave kernel, PPL 4: storing the 64-byte row straight to global memory from here instead of through the
      // LDS batch below was measured 5 % SLOWER: a scattered 16-lane store per record in the hot loop)
      if (lane < 16)
        reinterpret_cast<float*>(&sG[wave][j][0])[((lane & 1) << 3) | ((lane & 2) << 1) | (lane >> 2)] = tot;
    }
    __syncthreads();
    for (int e = tid; e < n * 4; e += NT) {
      const int j = e >> 2, part = e & 3;
      if (sRow[j] != DEAD_ROW) {
        float4 s = sG[0][j][part];
#pragma unroll
        for (int wv = 1; wv < NWV; ++wv) {
          const float4 r1 = sG[wv][j][part];
          s.x += r1.x; s.y += r1.y; s.z += r1.z; s.w += r1.w;
        }
        reinterpret_cast<float4*>(rows)[(size_t)sRow[j] * 4 + part] = s;
      }
    }
  }
}

// ---------------------------------------------------------------- blend backward, Gaussian-per-lane ("wave64 scan")
// The kernel above puts PIXELS on the lanes: the 16 gradient terms of a record then have to be summed over the wave
// for every record (a 16 x 64 reduce-scatter, ~60 of the ~250 instructions of an iteration), and eight blend
// channels are walked per pixel.  Here RECORDS are on the lanes.  One wave owns one 8x8 quadrant of a tile and walks
// the tile list back to front; the records that blended something in this quadrant (the forward pass left a 4-bit
// quadrant mask per instance) are compacted, 64 at a time, onto the lanes — lane 0 the rearmost.  For each of the
// quadrant's 64 pixels (uniform across the wave) every lane evaluates its record's alpha; the transmittance in
// front of each record is the pixel's running transmittance divided by an inclusive PREFIX PRODUCT of (1 - alpha)
// over the lanes, and the colour / normal / depth blended behind it collapses, after the dot product with the
// pixel's upstream gradient, into an exclusive PREFIX SUM of one scalar (w s): two DPP wave scans per pixel replace
// the per-record reduction, and the 16 gradient terms simply accumulate in the lane's registers over the 64
// pixels.  Each lane stores its 64-byte row once per (instance, quadrant); rows of one Gaussian stay contiguous and
// are summed by the same per-Gaussian kernels.  Waves never synchronise with each other; pixels that finished
// before the chunk are skipped for the whole wave.  Bitwise reproducible (fixed pixel order per lane).
// work estimate of a tile for the dispatch order: the largest n_contrib of its 256 pixels (one wave per tile)
__global__ __launch_bounds__(64) void tile_max_contrib_kernel(int W, int H, int gx, const uint32_t* __restrict__ n_contrib,
                                                              uint32_t* __restrict__ work) {
  const int tile = blockIdx.x, lane = threadIdx.x;
  const int tx = tile % gx, ty = tile / gx;
  uint32_t m = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int x = tx * TILE + (lane & 15), y = ty * TILE + 4 * r + (lane >> 4);
    if (x < W && y < H) m = max(m, n_contrib[(size_t)y * W + x]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
  if (lane == 0) work[tile] = m;
}

template <int CTRL, int ROW_MASK>
__device__ inline float dpp_f(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                CTRL, ROW_MASK, 0xf, false));
}

__device__ inline float wave_incl_scan_mul(float x) {
  x *= dpp_f<0x111, 0xf>(1.f, x);  // row_shr:1
  x *= dpp_f<0x112, 0xf>(1.f, x);  // row_shr:2
  x *= dpp_f<0x114, 0xf>(1.f, x);  // row_shr:4
  x *= dpp_f<0x118, 0xf>(1.f, x);  // row_shr:8
  x *= dpp_f<0x142, 0xa>(1.f, x);  // row_bcast:15 -> rows 1, 3
  x *= dpp_f<0x143, 0xc>(1.f, x);  // row_bcast:31 -> rows 2, 3
  return x;
}

__device__ inline float wave_incl_scan_add(float x) {
  x += dpp_f<0x111, 0xf>(0.f, x);
  x += dpp_f<0x112, 0xf>(0.f, x);
  x += dpp_f<0x114, 0xf>(0.f, x);
  x += dpp_f<0x118, 0xf>(0.f, x);
  x += dpp_f<0x142, 0xa>(0.f, x);
  x += dpp_f<0x143, 0xc>(0.f, x);
  return x;
}

// value of lane - 1 (lane 0 gets `first`)
__device__ inline float wave_shr1(float x, float first) { return dpp_f<0x138, 0xf>(first, x); }

// The same scans inside lane GROUPS of G = 16 / 32 / 64 lanes (a 16-lane DPP row, a half wave, the wave): a chunk with
// at most G records puts 64 / G pixels on the wave at once, each on its own group (grp_* below).
template <int G>
__device__ inline float grp_incl_scan_mul(float x) {
  x *= dpp_f<0x111, 0xf>(1.f, x);
  x *= dpp_f<0x112, 0xf>(1.f, x);
  x *= dpp_f<0x114, 0xf>(1.f, x);
  x *= dpp_f<0x118, 0xf>(1.f, x);
  if (G >= 32) x *= dpp_f<0x142, 0xa>(1.f, x);
  if (G >= 64) x *= dpp_f<0x143, 0xc>(1.f, x);
  return x;
}
template <int G>
__device__ inline float grp_incl_scan_add(float x) {
  x += dpp_f<0x111, 0xf>(0.f, x);
  x += dpp_f<0x112, 0xf>(0.f, x);
  x += dpp_f<0x114, 0xf>(0.f, x);
  x += dpp_f<0x118, 0xf>(0.f, x);
  if (G >= 32) x += dpp_f<0x142, 0xa>(0.f, x);
  if (G >= 64) x += dpp_f<0x143, 0xc>(0.f, x);
  return x;
}
// value of the previous lane of the group (the group's first lane gets `first`)
template <int G>
__device__ inline float grp_shr1(float x, float first, int lane) {
  if (G == 16) return dpp_f<0x111, 0xf>(first, x);          // row_shr:1, lanes without a source keep `first`
  const float y = dpp_f<0x138, 0xf>(first, x);               // wave_shr:1
  return (G == 32 && lane == 32) ? first : y;
}

struct PopOp {
  __host__ __device__ uint32_t operator()(const uint8_t& m) const { return (uint32_t)__builtin_popcount((unsigned)m & 15u); }
};

#ifdef PINGS_BWD_STATS
#define STATS_PARAMS , unsigned long long& st_dead, unsigned long long& st_any, unsigned long long& st_exec, unsigned long long& st_valid
#define STATS_ARGS , st_dead, st_any, st_exec, st_valid
#else
#define STATS_PARAMS
#define STATS_ARGS
#endif

// The pixel loop of blend_bwd_scan_kernel for one chunk of `take` <= G records: 64 / G pixels per step, lane l = record
// l % G of pixel pp0 + l / G (records are replicated over the groups by the caller).  Per pixel the arithmetic is the
// one-pixel form's, op for op; a lane's 16 sums run over its group's pixels and the groups are added at the end
// (fixed order: deterministic).
// Pixels [pix0, pix0 + NPIX) of the quadrant: all 64 for a wave that owns the quadrant, 16 (two pixel rows) for each of
// the four waves that share a long-list quadrant.
template <int MODE, int G, int NPIX>
__device__ inline void scan_pixels(float4 (*sPix)[4], int lane, int pix0, int e, int min_e, const float4& a,
                                   const float4& b, const float4& c, const float4& nn, float (&v)[16] STATS_PARAMS) {
  constexpr int NG = 64 / G;
  const int grp = lane / G;
  const float zlo = a.w - b.w, zhi = a.w + b.w;
  for (int pp0 = 0; pp0 < NPIX; pp0 += NG) {
    const int pp = pix0 + pp0 + grp;
    const float4 s0 = sPix[pp][0];
    const int last_p = (int)__float_as_uint(s0.z);
    if (!__any(last_p > min_e)) {                      // these pixels had stopped before the chunk's first record
#ifdef PINGS_BWD_STATS
      st_dead += NG;
#endif
      continue;
    }
    const float4 s1 = sPix[pp][1], s2 = sPix[pp][2], s3 = sPix[pp][3];
    const float T_p = s0.x, R_p = s0.y, coefT = s0.w;
    const float dx = a.x - s2.w;
    const float dy = a.y - s3.x;
    const float p0 = -0.5f * (b.x * dx * dx);
    const float pxy = b.y * dx;
    const float power = (p0 - 0.5f * (b.z * dy * dy)) - pxy * dy;   // the forward pass' op order
    const float Gs = __expf(power);
    const float raw = a.z * Gs;
    const float alpha = fminf(ALPHA_MAX, raw);
    const bool valid = (e < last_p) && (power <= 0.0f) && (alpha >= ALPHA_MIN);
#ifdef PINGS_BWD_STATS
    if (!__any(valid)) st_any += NG; else { st_exec += NG; st_valid += (unsigned)__popcll(__ballot(valid)); }
#endif
    if (!__any(valid)) continue;                          // no record reaches any of these pixels: their state
                                                          // (T / 1, R + 0) and every gradient sum stay as they are
    const float av = valid ? alpha : 0.f;
    const float P_in = grp_incl_scan_mul<G>(1.0f - av);   // product of (1 - alpha) over this and the records behind
    const float P_ex = grp_shr1<G>(P_in, 1.0f, lane);
    const float inv_P = __builtin_amdgcn_rcpf(P_in);
    const float Tn = T_p * inv_P;                        // transmittance in front of this record
    const float inv_one_m = P_ex * inv_P;                // 1 / (1 - alpha)
    const float w = av * Tn;
    float S_in;
    {
    // gradient accumulation only (alpha / validity were decided above in the forward pass' op order): multiply-adds
    // may fuse here
#pragma clang fp contract(fast)
    // s = (upstream gradient) . (features of this record at this pixel)
    float sdot = (c.x * s1.x + c.y * s1.y) + c.z * s1.z;
    v[G_R] = fmaf(s1.x, w, v[G_R]);
    v[G_G] = fmaf(s1.y, w, v[G_G]);
    v[G_B] = fmaf(s1.z, w, v[G_B]);
    if (MODE == MODE_SURFEL) {
      sdot += (nn.x * s2.x + nn.y * s2.y) + nn.z * s2.z;
      const float den = (nn.x * s3.y + nn.y * s3.z) + nn.z;
      const bool hit = den < -DEN_EPS;
      const float inv_den = __builtin_amdgcn_rcpf(den);
      const float d0 = hit ? c.w * inv_den : a.w;
      const float d = fminf(fmaxf(d0, zlo), zhi);
      sdot = fmaf(d, s1.w, sdot);
      const float gd = s1.w * w;
      const bool lo = d0 < zlo, hi = d0 > zhi;
      const bool mid = !lo && !hi;
      v[G_ZLO] += lo ? gd : 0.f;
      v[G_ZHI] += hi ? gd : 0.f;
      const float gq = (mid && hit) ? gd * inv_den : 0.f;
      v[G_Q] += gq;
      v[G_PZ] += (mid && !hit) ? gd : 0.f;
      const float gden = -gq * d0;
      v[G_NX] += fmaf(gden, s3.y, s2.x * w);
      v[G_NY] += fmaf(gden, s3.z, s2.y * w);
      v[G_NZ] += s2.z * w + gden;
    } else {
      sdot = fmaf(a.w, s1.w, sdot);
      v[G_PZ] = fmaf(s1.w, w, v[G_PZ]);
    }
    const float ws = w * sdot;
    S_in = grp_incl_scan_add<G>(ws);                     // w s over this and the records behind (in the chunk)
    const float R_l = R_p + grp_shr1<G>(S_in, 0.f, lane); // blended behind this record, dotted with the gradient
    // dL/dalpha = T s - (R - coefT) / (1 - alpha)
    float dLda = fmaf(Tn, sdot, (coefT - R_l) * inv_one_m);
    dLda = (valid && raw <= ALPHA_MAX) ? dLda : 0.f;     // no gradient through the active 0.99 clamp
    v[G_OPAC] = fmaf(Gs, dLda, v[G_OPAC]);
    const float dLp = raw * dLda;
    const float qx = b.x * dx + b.y * dy, qy = b.y * dx + b.z * dy;
    v[G_MX] -= dLp * qx;
    v[G_MY] -= dLp * qy;
    v[G_CONX] += 0.5f * qx * qx * dLp;
    v[G_CONY] += qx * qy * dLp;
    v[G_CONZ] += 0.5f * qy * qy * dLp;
    }
    if ((lane & (G - 1)) == G - 1) {                     // the group's last lane: pixel state after the whole chunk
      sPix[pp][0].x = Tn;
      sPix[pp][0].y = R_p + S_in;
    }
  }
  if (G < 64) {                                          // the groups hold partial sums of the SAME records: add them
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (G == 16) v[k] += __shfl_xor(v[k], 16, 64);
      v[k] += __shfl_xor(v[k], 32, 64);
    }
  }
}

#ifdef PINGS_BWD_STATS  // diagnostic build only: loop-efficiency counters of blend_bwd_scan_kernel
__device__ unsigned long long g_bwd_stats[8];
#endif

// Workgroups of four waves.  A tile of ordinary length takes one workgroup, wave w = quadrant w.  A LONG tile (largest
// per-pixel contributor count >= the long-list threshold: the first *n_long entries of the descending tile order)
// takes four workgroups, one per quadrant, whose four waves share the quadrant — 16 pixels each, all walking the same
// chunks; a record's four partial rows are added in LDS (wave order: deterministic).  With one wave per quadrant
// throughout, the launch waited for the few waves with the longest lists (C3 street: longest wave 59 chunks, about
// 0.6 of the kernel's 0.70 ms; mean 4.8).  Long tiles come first in the grid; a second launch for them would run
// alone on the chip (measured: slower than no split at all).
template <int MODE>
__global__ __launch_bounds__(256) void blend_bwd_scan_kernel(
    BParams p, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ rec, const uint32_t* __restrict__ gval, const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ out_depth, const float* __restrict__ dL_dcolor,
    const float* __restrict__ dL_dnormal, const float* __restrict__ dL_ddepth, const float* __restrict__ dL_dalpha,
    const uint8_t* __restrict__ qmask, const uint32_t* __restrict__ cidx, float* __restrict__ rows,
    const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ n_long) {
  // pixel state, per quadrant: {T, R, last, coefT} {gC0, gC1, gC2, gD} {gN0, gN1, gN2, px} {py, rx, ry, -}
  // ordinary tile: four quadrants x 64 pixels x 4; long tile: the first quarter + the waves' partial rows of a chunk
  __shared__ float4 sBuf[256 + 1024];
  __shared__ int sQeAll[4][128];        // per wave: queue of relevant list entries, back to front
  __shared__ uint32_t sQsAll[4][128];   // their instance slots

  const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
  int* sQe = sQeAll[wv];
  uint32_t* sQs = sQsAll[wv];
  const uint32_t nl = *n_long;
  const bool lng = blockIdx.x < 4u * nl;
  int tile, q;
  if (lng) {
    tile = (int)tile_order[blockIdx.x >> 2];
    q = (int)(blockIdx.x & 3);
  } else {
    const uint32_t ti = nl + (blockIdx.x - 4u * nl);
    if (ti >= (uint32_t)(p.gx * p.gy)) return;
    tile = (int)tile_order[ti];
    q = wv;
  }
  float4 (*sPix)[4] = reinterpret_cast<float4 (*)[4]>(sBuf + (lng ? 0 : 256 * wv));
  float (*sV)[16][64] = reinterpret_cast<float (*)[16][64]>(sBuf + 256);   // [wave][term][record]: 16 KB
  const int tx = tile % p.gx, ty = tile / p.gx;
  const int pix_x = tx * TILE + 8 * (q & 1) + (lane & 7);
  const int pix_y = ty * TILE + 8 * (q >> 1) + (lane >> 3);
  const size_t HW = (size_t)p.W * p.H;

  uint32_t last = 0;
  {
    float T = 1.f, gC0 = 0.f, gC1 = 0.f, gC2 = 0.f, gN0 = 0.f, gN1 = 0.f, gN2 = 0.f, gD = 0.f, coefT = 0.f;
    float rx = 0.f, ry = 0.f;
    if (MODE == MODE_SURFEL) {
      const float cxp = (p.prcp ? p.prcp[0] : 0.5f) * (float)p.W - 0.5f;
      const float cyp = (p.prcp ? p.prcp[1] : 0.5f) * (float)p.H - 0.5f;
      rx = ((float)pix_x - cxp) / p.fx;
      ry = ((float)pix_y - cyp) / p.fy;
    }
    if (pix_x < p.W && pix_y < p.H) {
      const size_t pix_id = (size_t)pix_y * p.W + pix_x;
      last = n_contrib[pix_id];
      const float T_final = final_T[pix_id];
      T = T_final;
      if (dL_dcolor) {
        gC0 = dL_dcolor[pix_id];
        gC1 = dL_dcolor[HW + pix_id];
        gC2 = dL_dcolor[2 * HW + pix_id];
      }
      float gA = dL_dalpha ? dL_dalpha[pix_id] : 0.f;
      const float gDo = dL_ddepth ? dL_ddepth[pix_id] : 0.f;
      if (MODE == MODE_SURFEL) {
        if (dL_dnormal) {
          gN0 = dL_dnormal[pix_id];
          gN1 = dL_dnormal[HW + pix_id];
          gN2 = dL_dnormal[2 * HW + pix_id];
        }
        const float A = 1.0f - T_final;
        if (A > DEPTH_ALPHA_EPS) {
          gD = gDo / A;
          gA -= gDo * out_depth[pix_id] / A;
        }
      } else {
        gD = gDo;
      }
      const float bgdot = (p.bg[0] * gC0 + p.bg[1] * gC1) + p.bg[2] * gC2;
      coefT = (gA - bgdot) * T_final;
    }
    sPix[lane][0] = make_float4(T, 0.f, __uint_as_float(last), coefT);
    sPix[lane][1] = make_float4(gC0, gC1, gC2, gD);
    sPix[lane][2] = make_float4(gN0, gN1, gN2, (float)pix_x);
    sPix[lane][3] = make_float4((float)pix_y, rx, ry, 0.f);
  }
  uint32_t m = last;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
  if (lng) __syncthreads();   // every wave wrote the whole quadrant's state (identical values) before any updates it
  int pos = (int)m;  // list entries [0, pos) can matter to this quadrant
  if (pos == 0) return;
  const uint2 range = ranges[tile];
  const uint32_t below = (1u << q) - 1u;
  int nq = 0;
#ifdef PINGS_BWD_STATS
  unsigned long long st_chunks = 0, st_dead = 0, st_any = 0, st_exec = 0, st_valid = 0, st_take = 0;
#endif

  while (true) {
    // ---- queue the next relevant entries (entries behind `pos` are done)
    while (nq < 64 && pos > 0) {
      const int e = pos - 1 - lane;
      uint32_t slot = 0;
      bool rel = false;
      if (e >= 0) {
        slot = point_list[range.x + e];
        rel = ((qmask[slot] >> q) & 1u) != 0u;
      }
      const unsigned long long bal = __ballot(rel);
      if (rel) {
        const int at = nq + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        sQe[at] = e;
        sQs[at] = slot;
      }
      nq += __popcll(bal);
      pos -= 64;
    }
    if (nq == 0) break;
    const int take = nq < 64 ? nq : 64;
    // records replicated over the lane groups of scan_pixels: group size 16 / 32 / 64 by the chunk's record count
    const int gl = take <= 16 ? (lane & 15) : (take <= 32 ? (lane & 31) : lane);
    const bool act = gl < take;
    const int e = act ? sQe[gl] : 0x7FFFFFFF;        // an idle lane is never `valid`
    const uint32_t slot = act ? sQs[gl] : 0u;
    const int rest = nq - take;                        // < 64: shift the remainder to the queue's front
    const int me = lane < rest ? sQe[take + lane] : 0;
    const uint32_t ms = lane < rest ? sQs[take + lane] : 0u;
    if (lane < rest) { sQe[lane] = me; sQs[lane] = ms; }
    nq = rest;
    const int min_e = __shfl(e, take - 1, 64);         // frontmost record of the chunk

    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, c = a, nn = a;
    uint32_t row = 0;
    if (act) {
      const uint32_t g = gval[slot];
      a = rec[4 * (size_t)g + 0];
      b = rec[4 * (size_t)g + 1];
      c = rec[4 * (size_t)g + 2];
      if (MODE == MODE_SURFEL) nn = rec[4 * (size_t)g + 3];
      row = cidx[slot] + (uint32_t)__builtin_popcount((unsigned)qmask[slot] & below);
    }
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = 0.f;

#ifdef PINGS_BWD_STATS
    ++st_chunks; st_take += (unsigned)take;
#endif
    // Chunks that hold at most 16 / 32 records (short lists: a quadrant of a mapping view sees ~20 relevant records)
    // put four / two pixels on the wave at once: lane l works on record l % G for pixel pp0 + l / G.  Measured before
    // this split on the bench's mapping view: 21 records per chunk, 6.9 valid lanes of 64 per pixel iteration.
    if (lng) {
      if (take <= 16) scan_pixels<MODE, 16, 16>(sPix, lane, 16 * wv, e, min_e, a, b, c, nn, v STATS_ARGS);
      else if (take <= 32) scan_pixels<MODE, 32, 16>(sPix, lane, 16 * wv, e, min_e, a, b, c, nn, v STATS_ARGS);
      else scan_pixels<MODE, 64, 16>(sPix, lane, 16 * wv, e, min_e, a, b, c, nn, v STATS_ARGS);
    } else {
      if (take <= 16) scan_pixels<MODE, 16, 64>(sPix, lane, 0, e, min_e, a, b, c, nn, v STATS_ARGS);
      else if (take <= 32) scan_pixels<MODE, 32, 64>(sPix, lane, 0, e, min_e, a, b, c, nn, v STATS_ARGS);
      else scan_pixels<MODE, 64, 64>(sPix, lane, 0, e, min_e, a, b, c, nn, v STATS_ARGS);
    }
    float4* dst = reinterpret_cast<float4*>(rows) + (size_t)row * 4;
    if (!lng) {
      if (act && gl == lane) {                         // every group holds the same sums: the first one stores
        dst[0] = make_float4(v[0], v[1], v[2], v[3]);
        dst[1] = make_float4(v[4], v[5], v[6], v[7]);
        dst[2] = make_float4(v[8], v[9], v[10], v[11]);
        dst[3] = make_float4(v[12], v[13], v[14], v[15]);
      }
    } else {
      // the four waves' sums over their 16 pixels each meet in LDS; wave w adds and stores quarter w of the row
      if (gl == lane) {
#pragma unroll
        for (int k = 0; k < 16; ++k) sV[wv][k][lane] = v[k];
      }
      __syncthreads();
      if (act && gl == lane) {
        float o4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = 4 * wv + u;
          o4[u] = ((sV[0][k][lane] + sV[1][k][lane]) + sV[2][k][lane]) + sV[3][k][lane];
        }
        dst[wv] = make_float4(o4[0], o4[1], o4[2], o4[3]);
      }
      __syncthreads();                                 // the next chunk overwrites sV
    }
  }
#ifdef PINGS_BWD_STATS
  if (lane == 0) {
    atomicAdd(&g_bwd_stats[0], st_chunks); atomicAdd(&g_bwd_stats[1], st_dead); atomicAdd(&g_bwd_stats[2], st_any);
    atomicAdd(&g_bwd_stats[3], st_exec); atomicAdd(&g_bwd_stats[4], st_valid); atomicAdd(&g_bwd_stats[5], st_take);
  }
#endif
}

// ---------------------------------------------------------------- balanced per-Gaussian row sums
struct LiveOp {
  __host__ __device__ uint32_t operator()(const float& w) const { return w > 0.f ? 1u : 0u; }
};

// per depth rank: compact row range of the Gaussian and its number of <= CH-row chunks
__global__ __launch_bounds__(256) void row_ranges_kernel(int P, const uint32_t* __restrict__ offsets_sorted,
                                                          const uint32_t* __restrict__ tiles_sorted,
                                                          const uint32_t* __restrict__ cidx,
                                                          uint32_t* __restrict__ cbeg,
                                                          uint32_t* __restrict__ nch) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > P) return;
  if (r == P) { nch[P] = 0u; return; }
  const uint32_t se = offsets_sorted[r], sb = se - tiles_sorted[r];
  const uint32_t cb = cidx[sb], ce = cidx[se];
  cbeg[r] = cb;
  nch[r] = (ce - cb + (uint32_t)CH - 1u) / (uint32_t)CH;
}

__global__ __launch_bounds__(256) void pair_owner_kernel(int P, const uint32_t* __restrict__ pair_off,
                                                          uint32_t* __restrict__ pair_owner) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= P) return;
  const uint32_t a = pair_off[r], b = pair_off[r + 1];
  for (uint32_t q = a; q < b; ++q) pair_owner[q] = (uint32_t)r;
}

// one 16-lane group per (Gaussian, chunk) pair; lane = column of the 16-float row
__global__ __launch_bounds__(256) void row_chunk_sum_kernel(int P, const uint32_t* __restrict__ pair_off,
                                                             const uint32_t* __restrict__ pair_owner,
                                                             const uint32_t* __restrict__ cbeg,
                                                             const uint32_t* __restrict__ offsets_sorted,
                                                             const uint32_t* __restrict__ cidx,
                                                             const float* __restrict__ rows,
                                                             float* __restrict__ partials) {
  const uint32_t q = (uint32_t)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 4);
  const int c = threadIdx.x & 15;
  if (q >= pair_off[P]) return;
  const uint32_t r = pair_owner[q];
  const uint32_t j = q - pair_off[r];
  const uint32_t row_end = cidx[offsets_sorted[r]];
  uint32_t row = cbeg[r] + j * (uint32_t)CH;
  const uint32_t stop = min(row + (uint32_t)CH, row_end);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (; row + 4 <= stop; row += 4) {
    a0 += rows[(size_t)row * 16 + c];
    a1 += rows[(size_t)(row + 1) * 16 + c];
    a2 += rows[(size_t)(row + 2) * 16 + c];
    a3 += rows[(size_t)(row + 3) * 16 + c];
  }
  for (; row < stop; ++row) a0 += rows[(size_t)row * 16 + c];
  partials[(size_t)q * 16 + c] = (a0 + a1) + (a2 + a3);
}

// ---------------------------------------------------------------- per-Gaussian chain
template <int MODE>
__global__ __launch_bounds__(256, 4) void gaussian_bwd_kernel(
    BParams p, const float* __restrict__ means3D, const float* __restrict__ scales,
    const float* __restrict__ rotations, const uint4* __restrict__ rect, const uint32_t* __restrict__ rank_of,
    const uint32_t* __restrict__ tiles_sorted, const uint32_t* __restrict__ pair_off,
    const float* __restrict__ partials, float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dmeans2D, float* __restrict__ dL_dcolors,
    float* __restrict__ dL_dopacities, float* __restrict__ dL_dscales,
    float* __restrict__ dL_drotations, float* __restrict__ tau_partials) {
  __shared__ float sTau[256 / 64][6];
  // one thread per Gaussian in INDEX order: parameters are read and gradients written coalesced; only the two
  // words that locate the Gaussian's chunk partials go through its depth rank
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  float tau[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  if (g < p.P) {
    uint32_t n_tiles = 0u, qa = 0u, qb = 0u;
    if (rect[g].w != 0u) {  // survived culling: has a depth rank
      const uint32_t rank = rank_of[g];
      n_tiles = tiles_sorted[rank];
      qa = pair_off[rank];
      qb = pair_off[rank + 1];
    }
    float G[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) G[k] = 0.f;
    const float4* prow = reinterpret_cast<const float4*>(partials);
    for (uint32_t k = qa; k < qb; ++k) {
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const float4 r = prow[(size_t)k * 4 + q4];
        G[q4 * 4 + 0] += r.x;
        G[q4 * 4 + 1] += r.y;
        G[q4 * 4 + 2] += r.z;
        G[q4 * 4 + 3] += r.w;
      }
    }

    float gm[3] = {0.f, 0.f, 0.f}, gs[3] = {0.f, 0.f, 0.f}, gq4[4] = {0.f, 0.f, 0.f, 0.f};
    float g2d[2] = {0.f, 0.f};
    if (n_tiles > 0 && qb > qa) {
      const float* V = p.view;
      const float* Pm = p.proj_raw;
      const float x = means3D[3 * g], y = means3D[3 * g + 1], z = means3D[3 * g + 2];
      const float px = ((V[0] * x + V[4] * y) + V[8] * z) + V[12];
      const float py = ((V[1] * x + V[5] * y) + V[9] * z) + V[13];
      const float pz = ((V[2] * x + V[6] * y) + V[10] * z) + V[14];
      float gp[3] = {0.f, 0.f, 0.f};
      // ---- 2-D mean
      {
        const float hx = ((Pm[0] * px + Pm[4] * py) + Pm[8] * pz) + Pm[12];
        const float hy = ((Pm[1] * px + Pm[5] * py) + Pm[9] * pz) + Pm[13];
        const float hw = ((Pm[3] * px + Pm[7] * py) + Pm[11] * pz) + Pm[15];
        const float pw = 1.0f / (hw + 1e-7f);
        const float gndx = 0.5f * (float)p.W * G[G_MX];
        const float gndy = 0.5f * (float)p.H * G[G_MY];
        g2d[0] = gndx;
        g2d[1] = gndy;
        const float kx = hx * pw * pw, ky = hy * pw * pw;
#pragma unroll
        for (int i = 0; i < 3; ++i)
          gp[i] = gndx * (Pm[4 * i + 0] * pw - kx * Pm[4 * i + 3]) +
                  gndy * (Pm[4 * i + 1] * pw - ky * Pm[4 * i + 3]);
      }
      // ---- covariance
      const float qr = rotations[4 * g], qx = rotations[4 * g + 1], qy = rotations[4 * g + 2],
                  qz = rotations[4 * g + 3];
      float R[3][3];
      R[0][0] = 1.0f - 2.0f * (qy * qy + qz * qz); R[0][1] = 2.0f * (qx * qy - qr * qz); R[0][2] = 2.0f * (qx * qz + qr * qy);
      R[1][0] = 2.0f * (qx * qy + qr * qz); R[1][1] = 1.0f - 2.0f * (qx * qx + qz * qz); R[1][2] = 2.0f * (qy * qz - qr * qx);
      R[2][0] = 2.0f * (qx * qz - qr * qy); R[2][1] = 2.0f * (qy * qz + qr * qx); R[2][2] = 1.0f - 2.0f * (qx * qx + qy * qy);
      const float S[3] = {p.scale_mod * scales[3 * g], p.scale_mod * scales[3 * g + 1],
                          p.scale_mod * scales[3 * g + 2]};
      float RS[3][3], Mc[3][3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) RS[i][k] = R[i][k] * S[k];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int k = 0; k < 3; ++k)
          Mc[a][k] = (V[a] * RS[0][k] + V[4 + a] * RS[1][k]) + V[8 + a] * RS[2][k];
      const float txz = px / pz, tyz = py / pz;
      const bool clx = (txz < -p.limx) || (txz > p.limx), cly = (tyz < -p.limy) || (tyz > p.limy);
      const float tx = fminf(p.limx, fmaxf(-p.limx, txz)) * pz;
      const float ty = fminf(p.limy, fmaxf(-p.limy, tyz)) * pz;
      const float iz = 1.0f / pz, iz2 = iz * iz;
      const float J00 = p.fx * iz, J02 = -(p.fx * tx) * iz2, J11 = p.fy * iz, J12 = -(p.fy * ty) * iz2;
      float T0[3], T1[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        T0[k] = J00 * Mc[0][k] + J02 * Mc[2][k];
        T1[k] = J11 * Mc[1][k] + J12 * Mc[2][k];
      }
      const float ca = ((T0[0] * T0[0] + T0[1] * T0[1]) + T0[2] * T0[2]) + LOWPASS;
      const float cb = (T0[0] * T1[0] + T0[1] * T1[1]) + T0[2] * T1[2];
      const float cc = ((T1[0] * T1[0] + T1[1] * T1[1]) + T1[2] * T1[2]) + LOWPASS;
      // the blend pass accumulated dL/d(cov2D) directly: a = cov_xx, b = cov_xy (the single
      // off-diagonal parameter), c = cov_yy
      (void)ca; (void)cb; (void)cc;
      const float ga = G[G_CONX], gb = G[G_CONY], gc = G[G_CONZ];
      float gT0[3], gT1[3], gMc[3][3];
      float gJ00 = 0.f, gJ02 = 0.f, gJ11 = 0.f, gJ12 = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        gT0[k] = 2.0f * ga * T0[k] + gb * T1[k];
        gT1[k] = 2.0f * gc * T1[k] + gb * T0[k];
        gJ00 += gT0[k] * Mc[0][k];
        gJ02 += gT0[k] * Mc[2][k];
        gJ11 += gT1[k] * Mc[1][k];
        gJ12 += gT1[k] * Mc[2][k];
        gMc[0][k] = J00 * gT0[k];
        gMc[1][k] = J11 * gT1[k];
        gMc[2][k] = J02 * gT0[k] + J12 * gT1[k];
      }
      // J -> camera point
      gp[2] += -p.fx * iz2 * gJ00 - p.fy * iz2 * gJ11;
      gp[2] += 2.0f * p.fx * tx * iz2 * iz * gJ02 + 2.0f * p.fy * ty * iz2 * iz * gJ12;
      const float gtx = -p.fx * iz2 * gJ02, gty = -p.fy * iz2 * gJ12;
      if (clx) gp[2] += gtx * tx * iz; else gp[0] += gtx;
      if (cly) gp[2] += gty * ty * iz; else gp[1] += gty;
      // Mc = Wc RS
      float gRS[3][3], gWc[3][3];
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int k = 0; k < 3; ++k)
          gRS[b][k] = (V[4 * b + 0] * gMc[0][k] + V[4 * b + 1] * gMc[1][k]) + V[4 * b + 2] * gMc[2][k];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
          gWc[a][b] = (gMc[a][0] * RS[b][0] + gMc[a][1] * RS[b][1]) + gMc[a][2] * RS[b][2];
      float gR[3][3];
      float gS[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          gS[k] += gRS[i][k] * R[i][k];
          gR[i][k] = gRS[i][k] * S[k];
        }
      float gpz_direct = G[G_PZ];
      if (MODE == MODE_SURFEL) {
        // normal n = sgn * Wc R[:,2], q = n . p
        float nx = (V[0] * R[0][2] + V[4] * R[1][2]) + V[8] * R[2][2];
        float ny = (V[1] * R[0][2] + V[5] * R[1][2]) + V[9] * R[2][2];
        float nz = (V[2] * R[0][2] + V[6] * R[1][2]) + V[10] * R[2][2];
        const float q = (nx * px + ny * py) + nz * pz;
        float sgn = 1.0f;
        if (!p.front_only && q > 0.0f) sgn = -1.0f;
        nx *= sgn; ny *= sgn; nz *= sgn;
        const float gq = G[G_Q];
        const float gn[3] = {G[G_NX] + gq * px, G[G_NY] + gq * py, G[G_NZ] + gq * pz};
        gp[0] += gq * nx;
        gp[1] += gq * ny;
        gp[2] += gq * nz;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          gR[b][2] += sgn * ((V[4 * b + 0] * gn[0] + V[4 * b + 1] * gn[1]) + V[4 * b + 2] * gn[2]);
#pragma unroll
          for (int a = 0; a < 3; ++a) gWc[a][b] += sgn * gn[a] * R[b][2];
        }
        // depth clamp window zlo = pz - rz, zhi = pz + rz, rz = 3 max(S0, S1)
        gpz_direct += G[G_ZLO] + G[G_ZHI];
        const float grz = 3.0f * (G[G_ZHI] - G[G_ZLO]);
        if (S[0] > S[1]) gS[0] += grz;
        else if (S[1] > S[0]) gS[1] += grz;
        else { gS[0] += 0.5f * grz; gS[1] += 0.5f * grz; }
      }
      gp[2] += gpz_direct;
      // quaternion
      gq4[0] = 2.0f * (-qz * gR[0][1] + qy * gR[0][2] + qz * gR[1][0] - qx * gR[1][2] - qy * gR[2][0] + qx * gR[2][1]);
      gq4[1] = 2.0f * (qy * gR[0][1] + qz * gR[0][2] + qy * gR[1][0] - 2.0f * qx * gR[1][1] - qr * gR[1][2] + qz * gR[2][0] + qr * gR[2][1] - 2.0f * qx * gR[2][2]);
      gq4[2] = 2.0f * (-2.0f * qy * gR[0][0] + qx * gR[0][1] + qr * gR[0][2] + qx * gR[1][0] + qz * gR[1][2] - qr * gR[2][0] + qz * gR[2][1] - 2.0f * qy * gR[2][2]);
      gq4[3] = 2.0f * (-2.0f * qz * gR[0][0] - qr * gR[0][1] + qx * gR[0][2] + qr * gR[1][0] - 2.0f * qz * gR[1][1] + qy * gR[1][2] + qx * gR[2][0] + qy * gR[2][1]);
#pragma unroll
      for (int k = 0; k < 3; ++k) gs[k] = p.scale_mod * gS[k];
      // world mean: p = Wc m + t
#pragma unroll
      for (int b = 0; b < 3; ++b)
        gm[b] = (V[4 * b + 0] * gp[0] + V[4 * b + 1] * gp[1]) + V[4 * b + 2] * gp[2];
      // pose tangent at tau = 0: d rho = gp ; d theta = p x gp + sum_b Wc[:,b] x gWc[:,b]
      tau[0] = gp[0]; tau[1] = gp[1]; tau[2] = gp[2];
      float th0 = py * gp[2] - pz * gp[1];
      float th1 = pz * gp[0] - px * gp[2];
      float th2 = px * gp[1] - py * gp[0];
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const float w0 = V[4 * b + 0], w1 = V[4 * b + 1], w2 = V[4 * b + 2];
        th0 += w1 * gWc[2][b] - w2 * gWc[1][b];
        th1 += w2 * gWc[0][b] - w0 * gWc[2][b];
        th2 += w0 * gWc[1][b] - w1 * gWc[0][b];
      }
      tau[3] = th0; tau[4] = th1; tau[5] = th2;
    }
    dL_dmeans3D[3 * g] = gm[0]; dL_dmeans3D[3 * g + 1] = gm[1]; dL_dmeans3D[3 * g + 2] = gm[2];
    dL_dmeans2D[3 * g] = g2d[0]; dL_dmeans2D[3 * g + 1] = g2d[1]; dL_dmeans2D[3 * g + 2] = 0.f;
    dL_dcolors[3 * g] = G[G_R]; dL_dcolors[3 * g + 1] = G[G_G]; dL_dcolors[3 * g + 2] = G[G_B];
    dL_dopacities[g] = G[G_OPAC];
    dL_dscales[3 * g] = gs[0]; dL_dscales[3 * g + 1] = gs[1]; dL_dscales[3 * g + 2] = gs[2];
    dL_drotations[4 * g] = gq4[0]; dL_drotations[4 * g + 1] = gq4[1];
    dL_drotations[4 * g + 2] = gq4[2]; dL_drotations[4 * g + 3] = gq4[3];
  }

  // block reduction of the pose-tangent terms (fixed order -> deterministic)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float s = wave_reduce_sum_dpp(tau[k]);
    if (lane == 63) sTau[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    tau_partials[(size_t)blockIdx.x * 6 + k] = ((sTau[0][k] + sTau[1][k]) + sTau[2][k]) + sTau[3][k];
  }
}

__global__ __launch_bounds__(256) void tau_reduce_kernel(const float* __restrict__ partials, int nblocks,
                                                          float* __restrict__ dL_dtau) {
  __shared__ double sRed[4][6];
  double s[6] = {0, 0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < nblocks; i += 256) {
#pragma unroll
    for (int k = 0; k < 6; ++k) s[k] += (double)partials[(size_t)i * 6 + k];
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s[k] += __shfl_down(s[k], off, 64);
    if ((threadIdx.x & 63) == 0) sRed[threadIdx.x >> 6][k] = s[k];
  }
  __syncthreads();
  if (threadIdx.x < 6)
    dL_dtau[threadIdx.x] = (float)(((sRed[0][threadIdx.x] + sRed[1][threadIdx.x]) + sRed[2][threadIdx.x]) + sRed[3][threadIdx.x]);
}

}  // namespace raster
}  // namespace pings

namespace pings {
namespace raster {

struct BwdState {
  uint32_t* cidx;        // [I+2] exclusive scan of live flags over instance slots (cidx[I] = #live)
  float* rows;           // [I][16] gradient rows of live instances (upper bound; only #live used)
  uint32_t *cbeg, *nch, *pair_off, *pair_owner;
  float* partials;       // [NPmax][16]
  float* tau_partials;   // [ceil(P/256)][6]
  char* temp;
  size_t temp_bytes, np_max, total;
};

static BwdState carve_bwd(void* blob, int P, int64_t I) {
  Carver c(blob);
  BwdState b;
  // gradient rows: one per live (instance, 8x8 quadrant) pair in the Gaussian-per-lane kernel, i.e. at most 4 I
  const size_t ni = (size_t)(I > 0 ? I : 1), n = 4 * ni, np = (size_t)(P > 0 ? P : 1);
  b.np_max = n / CH + np + 1;
  b.cidx = c.take<uint32_t>(ni + 2);
  b.cbeg = c.take<uint32_t>(np + 1);
  b.nch = c.take<uint32_t>(np + 1);
  b.pair_off = c.take<uint32_t>(np + 1);
  b.pair_owner = c.take<uint32_t>(b.np_max);
  b.partials = c.take<float>(b.np_max * GRAD_ROW);
  b.tau_partials = c.take<float>((size_t)ceil_div((int)np, 256) * 6);
  size_t a = 0, d = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, a, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)(ni + 1));
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, d, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)(np + 1));
  b.temp_bytes = align_up(a > d ? a : d) + 256;
  b.temp = c.take<char>(b.temp_bytes);
  b.rows = c.take<float>(n * GRAD_ROW);
  b.total = c.off;
  return b;
}

}  // namespace raster
}  // namespace pings

using namespace pings::raster;

PINGS_API size_t pings_raster_backward_bytes(int P, int64_t num_instances) {
  return carve_bwd(nullptr, P, num_instances).total;
}

PINGS_API int pings_raster_backward(const pings_raster_settings* s, int P, int64_t I,
                                    const float* means3D, const float* colors,
                                    const float* opacities, const float* scales,
                                    const float* rotations, const void* geom_blob,
                                    const void* binning_blob, const void* image_blob,
                                    const float* out_color, const float* out_normal,
                                    const float* out_depth, const float* out_alpha,
                                    const float* dL_dcolor, const float* dL_dnormal,
                                    const float* dL_ddepth, const float* dL_dalpha,
                                    void* bwd_blob, float* dL_dmeans3D, float* dL_dmeans2D,
                                    float* dL_dcolors, float* dL_dopacities, float* dL_dscales,
                                    float* dL_drotations, float* dL_dtau, int footprint_class,
                                    void* stream) {
  PINGS_ARG_CHECK(s != nullptr, "null settings");
  PINGS_ARG_CHECK(s->mode == PINGS_RASTER_SURFEL || s->mode == PINGS_RASTER_3DGS, "unknown mode");
  PINGS_ARG_CHECK(dL_dtau != nullptr, "null dL_dtau");
  hipStream_t st = pings::as_stream(stream);
  if (P == 0) {   // otherwise tau_reduce_kernel writes all six
    PINGS_HIP_CHECK(hipMemsetAsync(dL_dtau, 0, 6 * sizeof(float), st));
    return PINGS_OK;
  }
  PINGS_ARG_CHECK(P > 0 && means3D && colors && opacities && scales && rotations && geom_blob &&
                      binning_blob && image_blob && out_depth && bwd_blob && dL_dmeans3D &&
                      dL_dmeans2D && dL_dcolors && dL_dopacities && dL_dscales && dL_drotations,
                  "null pointer");
  PINGS_ARG_CHECK(I >= 0 && I < (int64_t)0x7FFFFFF0, "instance count out of range");
  (void)out_color; (void)out_normal; (void)out_alpha;
  BParams bp;
  bp.P = P;
  bp.W = s->image_width;
  bp.H = s->image_height;
  bp.gx = pings::ceil_div(bp.W, TILE);
  bp.gy = pings::ceil_div(bp.H, TILE);
  bp.front_only = s->front_only;
  bp.fx = (float)((double)bp.W / (2.0 * s->tanfovx));
  bp.fy = (float)((double)bp.H / (2.0 * s->tanfovy));
  bp.limx = (float)(1.3 * s->tanfovx);
  bp.limy = (float)(1.3 * s->tanfovy);
  bp.scale_mod = (float)s->scale_modifier;
  bp.view = s->viewmatrix;
  bp.proj_raw = s->projmatrix_raw;
  bp.bg = s->bg;
  bp.prcp = s->prcppoint;
  const int num_tiles = bp.gx * bp.gy;
  GeomState gs = carve_geom(const_cast<void*>(geom_blob), P, num_tiles);
  BinState bs = carve_binning(const_cast<void*>(binning_blob), I, num_tiles);
  ImageState im = carve_image(const_cast<void*>(image_blob), bp.W, bp.H);
  BwdState bw = carve_bwd(bwd_blob, P, I);
  const dim3 gridP(pings::ceil_div(P + 1, 256)), block(256);
  // Blend backward kernel: Gaussian-per-lane wave scans when footprints are small (lanes of the pixel-per-lane kernel
  // would idle: 2.25x faster on a street-like surfel scene), pixel-per-lane with two pixels per lane when they are
  // large (chunks of the scan kernel would stay half empty and every instance would need four rows: 25 % faster on
  // the Metric-1 cloud).  `footprint_class` comes from pings_raster_preprocess; PINGS_BLEND_BWD=pixel|scan overrides.
  bool scan_mode = footprint_class != 2;
  if (const char* e = getenv("PINGS_BLEND_BWD")) scan_mode = strcmp(e, "pixel") != 0;

  {
    pings::prof::Scope ps("live_scan", st);
    if (I > 0 && scan_mode) {
      // one row per (instance, quadrant it blended in): inst_qmask has I+1 entries, the last one zero
      hipcub::TransformInputIterator<uint32_t, PopOp, const uint8_t*> cnt(bs.inst_qmask, PopOp());
      size_t tb = bw.temp_bytes;
      PINGS_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(bw.temp, tb, cnt, bw.cidx, (int)(I + 1), st));
    } else if (I > 0) {
      // inst_w has I+1 entries, the last one zero: cidx[I] = number of live instances
      hipcub::TransformInputIterator<uint32_t, LiveOp, const float*> flags(bs.inst_w, LiveOp());
      size_t tb = bw.temp_bytes;
      PINGS_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(bw.temp, tb, flags, bw.cidx, (int)(I + 1), st));
    } else {
      PINGS_HIP_CHECK(hipMemsetAsync(bw.cidx, 0, 2 * sizeof(uint32_t), st));
    }
    hipLaunchKernelGGL(row_ranges_kernel, gridP, block, 0, st, P, gs.offsets_sorted, gs.tiles_sorted,
                       bw.cidx, bw.cbeg, bw.nch);
    PINGS_LAUNCH_CHECK();
    size_t tb = bw.temp_bytes;
    PINGS_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(bw.temp, tb, bw.nch, bw.pair_off, P + 1, st));
    hipLaunchKernelGGL(pair_owner_kernel, gridP, block, 0, st, P, bw.pair_off, bw.pair_owner);
    PINGS_LAUNCH_CHECK();
  }
  if (I > 0) {
    // backward dispatch order: tiles by descending largest per-pixel contributor count (the records a tile walks).
    // (On a side stream next to the row scans above — two chains of small launches that share nothing — the step came
    // out 0.008 - 0.025 ms SLOWER on Metric-1 / C2: the fork and join cost more than the 16 us they could hide.)
    pings::prof::Scope ps("tile_order", st);
    hipLaunchKernelGGL(tile_max_contrib_kernel, dim3(num_tiles), dim3(64), 0, st, bp.W, bp.H, bp.gx, im.n_contrib,
                       bs.tile_work);
    PINGS_LAUNCH_CHECK();
    if (int e = launch_tile_order(bs.tile_work, num_tiles, bs.tile_order + num_tiles, st,
                                  bs.tile_order + 2 * (size_t)num_tiles, long_list_threshold(), LONG_TILES_MAX))
      return e;
  }
  if (I > 0 && scan_mode) {
    pings::prof::Scope ps("blend_bwd", st);
    // one workgroup per ordinary tile, four per long tile (their number lives on the device: the grid is sized for
    // the cap, workgroups past the last tile leave at once)
    const uint32_t* order = bs.tile_order + num_tiles;
    const uint32_t* n_long = bs.tile_order + 2 * (size_t)num_tiles;
    const unsigned grid_s = (unsigned)(num_tiles + 3 * std::min<long long>(num_tiles, LONG_TILES_MAX));
#define PINGS_BWD_SCAN(M)                                                                                             \
  hipLaunchKernelGGL((blend_bwd_scan_kernel<M>), dim3(grid_s), dim3(256), 0, st, bp, bs.ranges,                       \
                     bs.point_list, gs.rec, bs.gval, im.final_T, im.n_contrib, out_depth, dL_dcolor, dL_dnormal,     \
                     dL_ddepth, dL_dalpha, bs.inst_qmask, bw.cidx, bw.rows, order, n_long)
    if (s->mode == PINGS_RASTER_SURFEL) PINGS_BWD_SCAN(MODE_SURFEL);
    else PINGS_BWD_SCAN(MODE_3DGS);
#undef PINGS_BWD_SCAN
    PINGS_LAUNCH_CHECK();
  } else if (I > 0) {
    pings::prof::Scope ps("blend_bwd", st);
    int ppl = footprint_class == 2 ? 4 : 1;   // class 2: one wave per tile, four pixels per lane (0.40 -> 0.35 ms on Metric-1 vs two)
    if (const char* e = getenv("PINGS_BLEND_BWD_PPL")) ppl = atoi(e);
#define PINGS_BLEND_BWD(M, L)                                                                            \
  hipLaunchKernelGGL((blend_bwd_kernel<M, L>), dim3(num_tiles), dim3(BLOCK / L), 0, st, bp, bs.ranges,    \
                     bs.point_list, gs.rec, bs.gval, im.final_T, im.n_contrib, out_depth, dL_dcolor,     \
                     dL_dnormal, dL_ddepth, dL_dalpha, bs.inst_w, bw.cidx, bw.rows, bs.tile_order + num_tiles)
    if (s->mode == PINGS_RASTER_SURFEL) {
      if (ppl == 1) PINGS_BLEND_BWD(MODE_SURFEL, 1);
      else if (ppl == 4) PINGS_BLEND_BWD(MODE_SURFEL, 4);
      else PINGS_BLEND_BWD(MODE_SURFEL, 2);
    } else {
      if (ppl == 1) PINGS_BLEND_BWD(MODE_3DGS, 1);
      else if (ppl == 4) PINGS_BLEND_BWD(MODE_3DGS, 4);
      else PINGS_BLEND_BWD(MODE_3DGS, 2);
    }
#undef PINGS_BLEND_BWD
    PINGS_LAUNCH_CHECK();
  }
  {
    pings::prof::Scope ps("row_chunk_sum", st);
    const size_t groups = bw.np_max;
    hipLaunchKernelGGL(row_chunk_sum_kernel, dim3((unsigned)pings::ceil_div<size_t>(groups * 16, 256)),
                       block, 0, st, P, bw.pair_off, bw.pair_owner, bw.cbeg, gs.offsets_sorted, bw.cidx,
                       bw.rows, bw.partials);
    PINGS_LAUNCH_CHECK();
  }
  const int nblocks = pings::ceil_div(P, 256);
  {
    pings::prof::Scope ps_g("gaussian_bwd", st);
    if (s->mode == PINGS_RASTER_SURFEL)
      hipLaunchKernelGGL(gaussian_bwd_kernel<MODE_SURFEL>, dim3(nblocks), block, 0, st, bp, means3D,
                         scales, rotations, gs.rect, gs.rank_of, gs.tiles_sorted, bw.pair_off, bw.partials,
                         dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dopacities, dL_dscales,
                         dL_drotations, bw.tau_partials);
    else
      hipLaunchKernelGGL(gaussian_bwd_kernel<MODE_3DGS>, dim3(nblocks), block, 0, st, bp, means3D,
                         scales, rotations, gs.rect, gs.rank_of, gs.tiles_sorted, bw.pair_off, bw.partials,
                         dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dopacities, dL_dscales,
                         dL_drotations, bw.tau_partials);
    PINGS_LAUNCH_CHECK();
    hipLaunchKernelGGL(tau_reduce_kernel, dim3(1), block, 0, st, bw.tau_partials, nblocks, dL_dtau);
    PINGS_LAUNCH_CHECK();
  }
  return PINGS_OK;
}

#ifdef PINGS_BWD_STATS
PINGS_API int pings_debug_bwd_stats(unsigned long long* out8, int reset) {
  PINGS_HIP_CHECK(hipDeviceSynchronize());
  PINGS_HIP_CHECK(hipMemcpyFromSymbol(out8, HIP_SYMBOL(pings::raster::g_bwd_stats), 64));
  if (reset) {
    unsigned long long z[8] = {0};
    PINGS_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(pings::raster::g_bwd_stats), z, 64));
  }
  return PINGS_OK;
}
#endif
