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
#include <string>

#include "knn_common.hpp"

namespace {
using namespace pings_knn;

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

#ifndef PINGS_SDF_FWD_WAVES
#define PINGS_SDF_FWD_WAVES 4   // waves per SIMD the fused forward is compiled for (128 VGPRs)
#endif
template <int IN_PAD, bool GRAD>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK, PINGS_SDF_FWD_WAVES) void sdf_forward_kernel(
    pings_knn_map m, pings_sdf_decoder dec, const float* __restrict__ features,
    const float* __restrict__ points, const float* __restrict__ orientations,
    const float* __restrict__ certainties, int after_pgo, const float* __restrict__ queries,
    long long B, float* __restrict__ sdf_out, float* __restrict__ grad_out,
    long long* __restrict__ cnt_out, float* __restrict__ cert_out, long long* __restrict__ idx_out,
    float* __restrict__ w_out, float* __restrict__ std_out, long long* __restrict__ gidx_out) {
  __shared__ long long sIdx[WAVES_PER_BLOCK][MAX_NNK];
  __shared__ long long sGIdx[WAVES_PER_BLOCK][MAX_NNK];
  __shared__ float sD2[WAVES_PER_BLOCK][MAX_NNK];
  __shared__ float sW[WAVES_PER_BLOCK][MAX_NNK];
  __shared__ __attribute__((aligned(16))) float sIn[WAVES_PER_BLOCK][MAX_NNK][IN_PAD];
  __shared__ float sVec[WAVES_PER_BLOCK][MAX_NNK][4];  // x - p_j w.r.t. the searched (global) point: d(d2)/dx
  __shared__ float sS[WAVES_PER_BLOCK][MAX_NNK];       // per-neighbour prediction / projection
  __shared__ float sPos[WAVES_PER_BLOCK][MAX_NNK * 3]; // the searched (global) positions, from the search itself
  __shared__ float sGn[WAVES_PER_BLOCK][8][4];         // per-neighbour gradient w.r.t. the direction input

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int F = dec.feat_dim, IN = F + 3, Hd = dec.hidden, nnk = m.nn_k;
  // this lane's hidden unit: weights stay in registers across queries
  float w1[IN_PAD];
  float b1 = 0.f, w2 = 0.f;
  float w1n0, w1n1, w1n2;  // direction-input weights of this unit (columns F..F+2), kept separately for the gradient
  load_w1_rows<IN_PAD>(dec.W1, IN, Hd, &sIn[0][0][0], w1, w1n0, w1n1, w1n2);
  if (lane < Hd) { b1 = dec.b1[lane]; w2 = dec.W2[lane]; }
  const float b2 = dec.b2[0];

  const LaneCtx lc = make_lane_ctx(m, lane);
  // this lane's (neighbour, 16-byte column) pairs of the feature gather, fixed across queries
  const bool f_vec = (F & 3) == 0 && ((reinterpret_cast<uintptr_t>(features) & 15u) == 0);
  const int F4 = F >> 2;
  int f_mm[2] = {-1, -1}, f_c4[2] = {0, 0};
  if (f_vec)
    for (int r = 0; r < 2; ++r) {
      const int e = lane + 64 * r;
      if (e < nnk * F4) { f_mm[r] = e / F4; f_c4[r] = e - f_mm[r] * F4; }
    }
  const long long nwaves = (long long)gridDim.x * WAVES_PER_BLOCK;
  // (Measured and dropped: a software pipeline over the wave's queries — coordinates loaded two queries ahead, the
  // next query's block entries issued during the current decode — 0.262 -> 0.316 ms at B = 131,072: the second
  // address computation and 13 spilled registers cost more than the hidden round trip saves.)
  for (long long q = (long long)blockIdx.x * WAVES_PER_BLOCK + wave; q < B; q += nwaves) {
    const float qx = queries[3 * q], qy = queries[3 * q + 1], qz = queries[3 * q + 2];
    const int count = knn_one_query(m, lc, qx, qy, qz, lane, sIdx[wave], sD2[wave], sGIdx[wave], sPos[wave]);
    __builtin_amdgcn_wave_barrier();

    // ---- this query's dependent gathers, all issued before the first of them is consumed: positions / certainty of
    // the neighbours, feature rows
    long long my_idx = -1;
    float u = 0.f, my_d2 = INVALID_D2;
    if (lane < nnk) {
      my_idx = sIdx[wave][lane];
      my_d2 = sD2[wave][lane];
      if (my_idx >= 0) u = 1.0f / (my_d2 + 1e-15f);
    }
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, cval = 0.f;
    if (my_idx >= 0) {
      p0 = points[3 * my_idx]; p1 = points[3 * my_idx + 1]; p2 = points[3 * my_idx + 2];
      if (certainties) cval = certainties[my_idx];
    }
    float4 fv[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    if (f_vec) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
        if (f_mm[r] >= 0) {
          const long long id = sIdx[wave][f_mm[r]];
          if (id >= 0) fv[r] = reinterpret_cast<const float4*>(features + id * F)[f_c4[r]];
        }
    }
    // ---- inverse-distance weights (lane i < nn_k owns neighbour i)
    const float U = wave_sum_all(u);
    const float wgt = (my_idx >= 0) ? u / U : 0.f;
    float cert = 0.f;
    if (lane < nnk) {
      sW[wave][lane] = wgt;
      if (idx_out) idx_out[q * nnk + lane] = my_idx;
      if (gidx_out) gidx_out[q * nnk + lane] = my_idx >= 0 ? sGIdx[wave][lane] : -1;
      if (w_out) w_out[q * nnk + lane] = wgt;
      float vx = 0.f, vy = 0.f, vz = 0.f;
      if (my_idx >= 0) {
        vx = qx - p0;
        vy = qy - p1;
        vz = qz - p2;
        cert = cval * wgt;
        sVec[wave][lane][0] = qx - sPos[wave][3 * lane];
        sVec[wave][lane][1] = qy - sPos[wave][3 * lane + 1];
        sVec[wave][lane][2] = qz - sPos[wave][3 * lane + 2];
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
    // ---- feature rows -> LDS (zeros for missing neighbours): 16 bytes per lane when the rows allow it
    if (f_vec) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
        if (f_mm[r] >= 0) *reinterpret_cast<float4*>(&sIn[wave][f_mm[r]][4 * f_c4[r]]) = fv[r];
      for (int e = lane + 128; e < nnk * F4; e += 64) {
        const int mm = e / F4, c4 = e - mm * F4;
        const long long id = sIdx[wave][mm];
        const float4 v = id >= 0 ? reinterpret_cast<const float4*>(features + id * F)[c4]
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(&sIn[wave][mm][4 * c4]) = v;
      }
    } else {
      for (int e = lane; e < nnk * F; e += 64) {
        const int mm = e / F, f = e - mm * F;
        const long long id = sIdx[wave][mm];
        sIn[wave][mm][f] = id >= 0 ? features[id * F + f] : 0.f;
      }
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
      if (GRAD && count > 0) {
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
      if (nnk <= 8) {
        // transposed reductions of eight sums at a time instead of one 8-step reduction per sum: without the gradient
        // all (<= 8) neighbours' output sums in one; with it {output, three direction-input gradients} of two
        // neighbours per reduction
        const int slot = reduce8_slot(lane);
        if (!GRAD) {
          float hv[8];
#pragma unroll
          for (int mm = 0; mm < 8; ++mm) {
            float pre = b1;
            if (mm < nnk) {
#pragma unroll
              for (int i = 0; i < IN_PAD; i += 4) {   // the row is broadcast from LDS sixteen bytes at a time
                const float4 t4 = *reinterpret_cast<const float4*>(&sIn[wave][mm][i]);
                pre = fmaf(w1[i], t4.x, pre);
                pre = fmaf(w1[i + 1], t4.y, pre);
                pre = fmaf(w1[i + 2], t4.z, pre);
                pre = fmaf(w1[i + 3], t4.w, pre);
              }
            }
            hv[mm] = (mm < nnk && lane < Hd) ? w2 * fmaxf(pre, 0.f) : 0.f;
          }
          const float tot = wave_reduce8(hv, lane);
          if (lane < 8) sS[wave][slot] = dec.sdf_scale * (b2 + tot);
        } else {
          // all hidden-layer chains first (eight independent dependency chains, as above), then the reductions
          float pre[8];
#pragma unroll
          for (int mm = 0; mm < 8; ++mm) {
            pre[mm] = b1;
            if (mm < nnk) {
#pragma unroll
              for (int i = 0; i < IN_PAD; i += 4) {
                const float4 t4 = *reinterpret_cast<const float4*>(&sIn[wave][mm][i]);
                pre[mm] = fmaf(w1[i], t4.x, pre[mm]);
                pre[mm] = fmaf(w1[i + 1], t4.y, pre[mm]);
                pre[mm] = fmaf(w1[i + 2], t4.z, pre[mm]);
                pre[mm] = fmaf(w1[i + 3], t4.w, pre[mm]);
              }
            }
          }
#pragma unroll
          for (int m0 = 0; m0 < 8; m0 += 2) {
            if (m0 < nnk) {
              float v[8];
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                const bool on = m0 + j < nnk && lane < Hd;
                const float gh = (on && pre[m0 + j] > 0.f) ? w2 : 0.f;
                v[4 * j] = on ? w2 * fmaxf(pre[m0 + j], 0.f) : 0.f;
                v[4 * j + 1] = w1n0 * gh; v[4 * j + 2] = w1n1 * gh; v[4 * j + 3] = w1n2 * gh;
              }
              const float tot = wave_reduce8(v, lane);
              if (lane < 8) {
                const int mm = m0 + (slot >> 2), c = slot & 3;
                if (c == 0) sS[wave][mm] = dec.sdf_scale * (b2 + tot);
                else sGn[wave][mm][c - 1] = tot;
              }
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
        for (int mm = 0; mm < nnk; ++mm) {
          const float wm = sW[wave][mm];
          S = fmaf(wm, sS[wave][mm], S);
          if (GRAD && sIdx[wave][mm] >= 0) {
            float gn0 = sGn[wave][mm][0], gn1 = sGn[wave][mm][1], gn2 = sGn[wave][mm][2];
            if (after_pgo) rot_active(orientations + 4 * sIdx[wave][mm], gn0, gn1, gn2, gn0, gn1, gn2);
            const float k = wm * dec.sdf_scale;
            gx = fmaf(k, gn0, gx); gy = fmaf(k, gn1, gy); gz = fmaf(k, gn2, gz);
          }
        }
      } else
      for (int mm = 0; mm < nnk; ++mm) {
        float pre = b1;
#pragma unroll
        for (int i = 0; i < IN_PAD; i += 4) {   // the row is broadcast from LDS sixteen bytes at a time
          const float4 t4 = *reinterpret_cast<const float4*>(&sIn[wave][mm][i]);
          pre = fmaf(w1[i], t4.x, pre);
          pre = fmaf(w1[i + 1], t4.y, pre);
          pre = fmaf(w1[i + 2], t4.z, pre);
          pre = fmaf(w1[i + 3], t4.w, pre);
        }
        const float h = fmaxf(pre, 0.f);
        const float s_m = dec.sdf_scale * (b2 + wave_sum_all(lane < Hd ? w2 * h : 0.f));
        sS[wave][mm] = s_m;
        const float wm = sW[wave][mm];
        S = fmaf(wm, s_m, S);
        if (GRAD && sIdx[wave][mm] >= 0) {
          const float gh = (lane < Hd && pre > 0.f) ? w2 : 0.f;
          float gn0 = wave_sum_all(w1n0 * gh), gn1 = wave_sum_all(w1n1 * gh),
                gn2 = wave_sum_all(w1n2 * gh);
          if (after_pgo) rot_active(orientations + 4 * sIdx[wave][mm], gn0, gn1, gn2, gn0, gn1, gn2);
          const float k = wm * dec.sdf_scale;
          gx = fmaf(k, gn0, gx); gy = fmaf(k, gn1, gy); gz = fmaf(k, gn2, gz);
        }
      }
      if (GRAD && count > 0) {
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
      if (GRAD) { grad_out[3 * q] = gx; grad_out[3 * q + 1] = gy; grad_out[3 * q + 2] = gz; }
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

namespace {
// `radius_neighborhood_search` as it stands (model/neural_gaussians.py:1061-1115): one thread per (query, cell).  The
// fused search above never forms this [B, K] pair; callers that want it (query_certainty, :1117-1133) get it here.
// Order of the reference's steps kept: the time window drops an entry first, an empty / dropped entry reads the LAST
// point (python's index -1) and gets max_valid_dist2, then a distance above the bound (a hash collision) clears the
// index but keeps its distance.
__global__ __launch_bounds__(256) void knn_cells_kernel(pings_knn_map m, const float* __restrict__ q, long long B,
                                                         long long num_points, float* __restrict__ d2,
                                                         long long* __restrict__ idx) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= B * m.K) return;
  const long long b = e / m.K;
  const int k = (int)(e - b * m.K);
  const float qx = q[3 * b], qy = q[3 * b + 1], qz = q[3 * b + 2];
  const long long gx = (long long)floorf(qx / m.resolution) + m.neighbor_dx[3 * k];
  const long long gy = (long long)floorf(qy / m.resolution) + m.neighbor_dx[3 * k + 1];
  const long long gz = (long long)floorf(qz / m.resolution) + m.neighbor_dx[3 * k + 2];
  const long long h = hash_slot(gx * P0 + gy * P1 + gz * P2, m.buffer_size, 1.0 / (double)m.buffer_size);
  long long i = table_lookup(m, h);
  const long long last = num_points - 1;                 // what index -1 reads
  if (m.time_filtering) {
    const int ts = m.point_ts_create[i >= 0 ? i : last];
    if (!(fabsf(m.travel_dist[m.cur_ts] - m.travel_dist[ts]) < m.diff_travel_dist_local)) i = -1;
  }
  const long long r = i >= 0 ? i : last;
  const float sx = m.neural_points[3 * r] - qx, sy = m.neural_points[3 * r + 1] - qy, sz = m.neural_points[3 * r + 2] - qz;
  float dd = (sx * sx + sy * sy) + sz * sz;
  if (i < 0) dd = m.max_valid_dist2;
  if (dd > m.max_valid_dist2) i = -1;
  d2[e] = dd;
  idx[e] = i;
}
}  // namespace

PINGS_API int pings_knn_cells(const pings_knn_map* m, const float* queries, int64_t B, int64_t num_points, float* d2,
                              int64_t* idx, void* stream) {
  // the raw pair has no nn_k: any K (the mapper calls it with the one-cell neighbourhood, K = 1 < nn_k)
  PINGS_ARG_CHECK(m != nullptr, "null map");
  PINGS_ARG_CHECK((m->table || m->compact) && m->buffer_size > 0 && m->buffer_size < (1LL << 31) && m->neural_points &&
                      m->neighbor_dx && m->K > 0 && m->resolution > 0.f,
                  "bad map");
  PINGS_ARG_CHECK(!m->compact || ((m->compact_mask & (m->compact_mask + 1u)) == 0u), "compact_mask must be 2^k - 1");
  PINGS_ARG_CHECK(!m->time_filtering || (m->point_ts_create && m->travel_dist), "time filtering needs ts / travel_dist");
  if (B == 0) return PINGS_OK;
  PINGS_ARG_CHECK(B > 0 && num_points > 0 && queries && d2 && idx, "bad arguments");
  hipStream_t st = pings::as_stream(stream);
  const long long total = (long long)B * m->K;
  hipLaunchKernelGGL(knn_cells_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, *m, queries,
                     (long long)B, (long long)num_points, d2, (long long*)idx);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}

PINGS_API int pings_sdf_forward(const pings_knn_map* m, const pings_sdf_decoder* dec,
                                const float* features, const float* points,
                                const float* orientations, const float* certainties,
                                int32_t after_pgo, const float* queries, int64_t B, float* sdf,
                                float* grad_x, int64_t* nn_counts, float* certainty,
                                int64_t* idx_out, float* w_out, float* sdf_std, int64_t* gidx_out,
                                void* stream) {
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
  // per-neighbour decoder, nn_k <= 8: the matrix-core kernel (sdf_fwd_mfma.hip); PINGS_SDF_FWD=vector keeps this file's
  const char* fwd_env = getenv("PINGS_SDF_FWD");   // read per call: the tests switch it in-process
  const bool force_vector = fwd_env && std::string(fwd_env) == "vector";
  if (!force_vector && sdf_forward_mfma_supported(m, dec, features))
    return sdf_forward_mfma_launch(m, dec, features, points, orientations, certainties, after_pgo, queries, B, sdf,
                                   grad_x, nn_counts, certainty, idx_out, w_out, sdf_std, gidx_out, st);
  const int in_dim = dec->feat_dim + 3;
#define PINGS_SDF_LAUNCH_G(PAD, G)                                                                          \
  hipLaunchKernelGGL((sdf_forward_kernel<PAD, G>), dim3(grid_for(B, (const void*)sdf_forward_kernel<PAD, G>)), dim3(64 * WAVES_PER_BLOCK), 0, st, *m, \
                     *dec, features, points, orientations, certainties, (int)after_pgo, queries,       \
                     (long long)B, sdf, grad_x, (long long*)nn_counts, certainty, (long long*)idx_out, w_out, sdf_std, \
                     (long long*)gidx_out)
#define PINGS_SDF_LAUNCH(PAD) do { if (grad_x) PINGS_SDF_LAUNCH_G(PAD, true); else PINGS_SDF_LAUNCH_G(PAD, false); } while (0)
  if (in_dim <= 12) PINGS_SDF_LAUNCH(12);
  else if (in_dim <= 36) PINGS_SDF_LAUNCH(36);
  else PINGS_SDF_LAUNCH(64);
#undef PINGS_SDF_LAUNCH
#undef PINGS_SDF_LAUNCH_G
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}
