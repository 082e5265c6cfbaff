// NeuralPoints.query_feature (model/neural_gaussians.py:506-725) on the device, end to end: forward, backward and
// the backward of the backward (the mapper differentiates d sdf / d x once more: utils/tools.py:409-419 with
// create_graph=True, used at utils/mapper.py:874-875 and :1445-1448).
//
//   qf_forward_kernel          one wave64 per query: hash-grid search + top-k (knn_common.hpp), inverse-distance
//                              weights (:644-662), neighbour vectors in the neural point's frame (:622-632), feature
//                              gather (:565-579), queried certainty (:691-695) and the training-mode side effects
//                              (:664-689: certainty += w, ts_update = max(ts_update, query_ts)); outputs written
//                              coalesced as [B, k, F+3] rows (or their weighted sum [B, F+3] when weighted_first, :701).
//   qf_backward_kernel         16 lanes per query, lane j = neighbour j: d/dx through the neighbour vectors and through
//                              the weights; the feature-table gradients are a row scatter-add of the UPSTREAM gradient
//                              itself (row (b, j) of d geo is the gradient of feature row idx[b, j]), done without float
//                              atomics by csrc/row_scatter.hip.
//   qf_double_backward_kernel  same layout: given the gradient w.r.t. that d/dx, the gradients w.r.t. the upstream
//                              gradients (d geo, d w), the query and — weighted_first only — the feature tables.
//
// Math (per query, neighbours j): e_j = x - P[gidx_j] (the searched, global point), u_j = 1 / (|e_j|^2 + 1e-15),
// s = sum u, w_j = u_j / s; n_j = R_j^T (x - p_j) (p_j: the row of the queried table; R_j = I before a loop closure).
// With upstream gn_j (w.r.t. n_j) and gw_j (w.r.t. w_j):  gx = sum_j R_j gn_j + sum_j gu_j (-2 u_j^2) e_j,
// gu_j = (gw_j - sum_i gw_i w_i) / s.  The second order terms are the directional derivatives of these along gg_x.
// Squared distances use ((x*x + y*y) + z*z) without FMA like the search, so weights are those of the forward bit for bit.
#include "knn_common.hpp"
#include "row_scatter.hpp"

namespace {
using namespace pings_knn;

constexpr float IDW_EPS = 1e-15f;   // neural_gaussians.py:644

__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void qf_forward_kernel(
    pings_knn_map m, pings_qf_tables t, const float* __restrict__ queries, long long B,
    float* __restrict__ geo_out, float* __restrict__ color_out, float* __restrict__ w_out,
    long long* __restrict__ idx_out, long long* __restrict__ gidx_out, long long* __restrict__ cnt_out,
    float* __restrict__ cert_out, float* __restrict__ cert_accum, const int* __restrict__ query_ts,
    int* __restrict__ ts_update, float* __restrict__ n_out) {
  __shared__ long long sIdx[WAVES_PER_BLOCK][MAX_NNK];
  __shared__ long long sGIdx[WAVES_PER_BLOCK][MAX_NNK];
  __shared__ float sD2[WAVES_PER_BLOCK][MAX_NNK];
  __shared__ float sW[WAVES_PER_BLOCK][MAX_NNK];
  __shared__ float sN[WAVES_PER_BLOCK][MAX_NNK][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nnk = m.nn_k, Fg = t.Fg, Fc = t.Fc;
  const LaneCtx lc = make_lane_ctx(m, lane);
  const long long nwaves = (long long)gridDim.x * WAVES_PER_BLOCK;
  for (long long q = (long long)blockIdx.x * WAVES_PER_BLOCK + wave; q < B; q += nwaves) {
    const float qx = queries[3 * q], qy = queries[3 * q + 1], qz = queries[3 * q + 2];
    const int count = knn_one_query(m, lc, qx, qy, qz, lane, sIdx[wave], sD2[wave], sGIdx[wave]);
    __builtin_amdgcn_wave_barrier();
    long long my_idx = -1;
    float u = 0.f;
    if (lane < nnk) {
      my_idx = sIdx[wave][lane];
      if (my_idx >= 0) u = 1.0f / (sD2[wave][lane] + IDW_EPS);
    }
    const float U = wave_sum_all(u);
    const float wgt = my_idx >= 0 ? u / U : 0.f;
    float cert = 0.f;
    if (lane < nnk) {
      sW[wave][lane] = wgt;
      idx_out[q * nnk + lane] = my_idx;
      gidx_out[q * nnk + lane] = my_idx >= 0 ? sGIdx[wave][lane] : -1;
      w_out[q * nnk + lane] = wgt;
      float nx = 0.f, ny = 0.f, nz = 0.f;
      if (my_idx >= 0) {
        const float vx = qx - t.points[3 * my_idx], vy = qy - t.points[3 * my_idx + 1],
                    vz = qz - t.points[3 * my_idx + 2];
        nx = vx; ny = vy; nz = vz;
        if (t.after_pgo) rot_passive(t.orientations + 4 * my_idx, vx, vy, vz, nx, ny, nz);
        if (t.certainties) cert = t.certainties[my_idx] * wgt;
        // training-mode side effects (:664-689): float atomics like the reference's scatter_add_ (order-dependent
        // rounding, not part of any gradient); the timestamp maximum is an integer atomic and exact
        if (cert_accum) atomicAdd(&cert_accum[my_idx], wgt);
        if (ts_update && query_ts) atomicMax(&ts_update[my_idx], query_ts[q]);
      }
      sN[wave][lane][0] = nx; sN[wave][lane][1] = ny; sN[wave][lane][2] = nz;
      if (n_out) {
        float* o = n_out + (q * nnk + lane) * 3;
        o[0] = nx; o[1] = ny; o[2] = nz;
      }
    }
    if (cert_out) {
      const float cs = wave_sum_all(cert);
      if (lane == 0) cert_out[q] = cs;
    }
    if (lane == 0) cnt_out[q] = count;
    __builtin_amdgcn_wave_barrier();
    // ---- feature rows, coalesced over the output row
#pragma unroll 1
    for (int which = 0; which < 2; ++which) {
      float* out = which ? color_out : geo_out;
      const float* tab = which ? t.color_features : t.geo_features;
      const int F = which ? Fc : Fg;
      if (!out) continue;
      const int IN = F + 3;
      if (t.weighted_first) {
        for (int i = lane; i < IN; i += 64) {
          float v = 0.f;
          for (int mm = 0; mm < nnk; ++mm) {
            const long long id = sIdx[wave][mm];
            const float e = i < F ? (id >= 0 ? tab[id * F + i] : 0.f) : sN[wave][mm][i - F];
            v += e * sW[wave][mm];               // sum_k (feature * weight), in neighbour order (:701-705)
          }
          out[q * IN + i] = v;
        }
      } else {
        for (int e = lane; e < nnk * IN; e += 64) {
          const int mm = e / IN, i = e - mm * IN;
          const long long id = sIdx[wave][mm];
          out[(q * nnk + mm) * IN + i] = i < F ? (id >= 0 ? tab[id * F + i] : 0.f) : sN[wave][mm][i - F];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------ 16 lanes per query
__device__ inline float row16_sum(float v) {
  v += __shfl_xor(v, 1, 16);
  v += __shfl_xor(v, 2, 16);
  v += __shfl_xor(v, 4, 16);
  v += __shfl_xor(v, 8, 16);
  return v;
}

// Per-pair geometry shared by the two backward kernels.
struct PairGeo {
  bool valid;
  long long id;          // row in the queried tables
  float ex, ey, ez;      // x - P[gidx] (weights)
  float u, s, w;         // inverse squared distance, row sum, weight
  float nx, ny, nz;      // neighbour vector in the point's frame
};

__device__ inline PairGeo pair_geo(const pings_qf_tables& t, const float* __restrict__ gpoints,
                                   const float* __restrict__ queries, const long long* __restrict__ idx,
                                   const long long* __restrict__ gidx, long long q, int j, int nnk, bool in_range,
                                   float& qx, float& qy, float& qz) {
  PairGeo g;
  g.valid = false; g.id = -1; g.ex = g.ey = g.ez = 0.f; g.u = 0.f; g.nx = g.ny = g.nz = 0.f;
  qx = qy = qz = 0.f;
  if (in_range) {
    qx = queries[3 * q]; qy = queries[3 * q + 1]; qz = queries[3 * q + 2];
    if (j < nnk) {
      g.id = idx[q * nnk + j];
      g.valid = g.id >= 0;
      if (g.valid) {
        const long long gi = gidx[q * nnk + j];
        g.ex = qx - gpoints[3 * gi]; g.ey = qy - gpoints[3 * gi + 1]; g.ez = qz - gpoints[3 * gi + 2];
        const float d2 = (g.ex * g.ex + g.ey * g.ey) + g.ez * g.ez;
        g.u = 1.0f / (d2 + IDW_EPS);
        const float vx = qx - t.points[3 * g.id], vy = qy - t.points[3 * g.id + 1], vz = qz - t.points[3 * g.id + 2];
        g.nx = vx; g.ny = vy; g.nz = vz;
        if (t.after_pgo) rot_passive(t.orientations + 4 * g.id, vx, vy, vz, g.nx, g.ny, g.nz);
      }
    }
  }
  g.s = row16_sum(g.u);
  g.w = g.valid ? g.u / g.s : 0.f;
  return g;
}

// dot of an upstream row [F+3] with the pair's input row [f_j, n_j]
__device__ inline float dot_in(const float* __restrict__ up, const float* __restrict__ tab, int F, long long id,
                               float nx, float ny, float nz) {
  float a = 0.f;
  for (int i = 0; i < F; ++i) a = fmaf(up[i], tab[id * F + i], a);
  a = fmaf(up[F], nx, a);
  a = fmaf(up[F + 1], ny, a);
  a = fmaf(up[F + 2], nz, a);
  return a;
}

__global__ __launch_bounds__(256) void qf_backward_kernel(
    pings_qf_tables t, const float* __restrict__ gpoints, const float* __restrict__ queries, long long B, int nnk,
    const long long* __restrict__ idx, const long long* __restrict__ gidx, const float* __restrict__ g_geo,
    const float* __restrict__ g_color, const float* __restrict__ g_n, const float* __restrict__ g_w, long long rows,
    float* __restrict__ g_x, uint32_t* __restrict__ keys, uint32_t* __restrict__ src_row,
    float* __restrict__ pair_w) {
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long q = tid >> 4;
  const int j = (int)(tid & 15);
  const bool in_range = q < B;
  float qx, qy, qz;
  const PairGeo g = pair_geo(t, gpoints, queries, idx, gidx, q, j, nnk, in_range, qx, qy, qz);
  const int Fg = t.Fg, Fc = t.Fc;
  float gnx = 0.f, gny = 0.f, gnz = 0.f, gw = 0.f;
  if (g.valid) {
    if (g_w) gw = g_w[q * nnk + j];
    if (g_n) {                       // split layout: the neighbour-vector part of the upstream travels on its own
      const float* up = g_n + (q * nnk + j) * 3;
      gnx += up[0]; gny += up[1]; gnz += up[2];
    } else if (t.weighted_first) {
      if (g_geo) {
        const float* up = g_geo + q * (Fg + 3);
        gnx += g.w * up[Fg]; gny += g.w * up[Fg + 1]; gnz += g.w * up[Fg + 2];
        gw += dot_in(up, t.geo_features, Fg, g.id, g.nx, g.ny, g.nz);
      }
      if (g_color) {
        const float* up = g_color + q * (Fc + 3);
        gnx += g.w * up[Fc]; gny += g.w * up[Fc + 1]; gnz += g.w * up[Fc + 2];
        gw += dot_in(up, t.color_features, Fc, g.id, g.nx, g.ny, g.nz);
      }
    } else {
      if (g_geo) {
        const float* up = g_geo + (q * nnk + j) * (Fg + 3) + Fg;
        gnx += up[0]; gny += up[1]; gnz += up[2];
      }
      if (g_color) {
        const float* up = g_color + (q * nnk + j) * (Fc + 3) + Fc;
        gnx += up[0]; gny += up[1]; gnz += up[2];
      }
    }
  }
  // through the neighbour vector: n = R^T (x - p)  =>  dx += R gn
  float ax = gnx, ay = gny, az = gnz;
  if (g.valid && t.after_pgo) rot_active(t.orientations + 4 * g.id, gnx, gny, gnz, ax, ay, az);
  // through the weights
  const float A = row16_sum(gw * g.w);
  if (g.valid) {
    const float gu = (gw - A) / g.s;
    const float k = -2.f * g.u * g.u * gu;
    ax = fmaf(k, g.ex, ax); ay = fmaf(k, g.ey, ay); az = fmaf(k, g.ez, az);
  } else {
    ax = ay = az = 0.f;
  }
  ax = row16_sum(ax); ay = row16_sum(ay); az = row16_sum(az);
  if (in_range && j == 0 && g_x) { g_x[3 * q] = ax; g_x[3 * q + 1] = ay; g_x[3 * q + 2] = az; }
  if (in_range && j < nnk && keys) {
    const long long pr = q * nnk + j;
    keys[pr] = g.valid ? (uint32_t)g.id : (uint32_t)rows;
    if (t.weighted_first) {
      src_row[pr] = (uint32_t)q;
      pair_w[pr] = g.w;
    }
  }
}

__global__ __launch_bounds__(256) void qf_double_backward_kernel(
    pings_qf_tables t, const float* __restrict__ gpoints, const float* __restrict__ queries, long long B, int nnk,
    const long long* __restrict__ idx, const long long* __restrict__ gidx, const float* __restrict__ g_geo,
    const float* __restrict__ g_color, const float* __restrict__ g_w, const float* __restrict__ gg_x,
    const float* __restrict__ gg_geo_feat, const float* __restrict__ gg_color_feat, long long rows,
    float* __restrict__ d_g_geo, float* __restrict__ d_g_color, float* __restrict__ d_g_n, float* __restrict__ d_g_w,
    float* __restrict__ d_x, uint32_t* __restrict__ keys, uint32_t* __restrict__ src_row,
    float* __restrict__ pair_w) {
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long q = tid >> 4;
  const int j = (int)(tid & 15);
  const bool in_range = q < B;
  float qx, qy, qz;
  const PairGeo g = pair_geo(t, gpoints, queries, idx, gidx, q, j, nnk, in_range, qx, qy, qz);
  const int Fg = t.Fg, Fc = t.Fc;
  float vx = 0.f, vy = 0.f, vz = 0.f;
  if (in_range && gg_x) { vx = gg_x[3 * q]; vy = gg_x[3 * q + 1]; vz = gg_x[3 * q + 2]; }
  // tangent of the neighbour vector along v: R^T v
  float tnx = 0.f, tny = 0.f, tnz = 0.f;
  if (g.valid) {
    tnx = vx; tny = vy; tnz = vz;
    if (t.after_pgo) rot_passive(t.orientations + 4 * g.id, vx, vy, vz, tnx, tny, tnz);
  }
  // tangent of the weights along v
  const float ev = (g.ex * vx + g.ey * vy) + g.ez * vz;
  const float ud = g.valid ? -2.f * g.u * g.u * ev : 0.f;
  const float sd = row16_sum(ud);
  const float wd = g.valid ? (ud - g.w * sd) / g.s : 0.f;
  if (in_range && j < nnk && d_g_w) d_g_w[q * nnk + j] = wd;

  // c_j: coefficient of w_j in the first backward; tt_j: <upstream, tangent input row> (weighted_first only)
  float c = 0.f, tt = 0.f;
  float upnx = 0.f, upny = 0.f, upnz = 0.f;   // weighted_first: the n-part of the upstream rows (geo + colour)
  if (g.valid) {
    if (g_w) c = g_w[q * nnk + j];
    if (t.weighted_first) {
      if (g_geo) {
        const float* up = g_geo + q * (Fg + 3);
        c += dot_in(up, t.geo_features, Fg, g.id, g.nx, g.ny, g.nz);
        upnx += up[Fg]; upny += up[Fg + 1]; upnz += up[Fg + 2];
        if (gg_geo_feat) for (int i = 0; i < Fg; ++i) tt = fmaf(up[i], gg_geo_feat[g.id * Fg + i], tt);
      }
      if (g_color) {
        const float* up = g_color + q * (Fc + 3);
        c += dot_in(up, t.color_features, Fc, g.id, g.nx, g.ny, g.nz);
        upnx += up[Fc]; upny += up[Fc + 1]; upnz += up[Fc + 2];
        if (gg_color_feat) for (int i = 0; i < Fc; ++i) tt = fmaf(up[i], gg_color_feat[g.id * Fc + i], tt);
      }
      tt = fmaf(upnx, tnx, fmaf(upny, tny, fmaf(upnz, tnz, tt)));
    }
  }

  // ---- gradient w.r.t. the upstream feature gradients
  if (t.weighted_first) {
    // d gGw[i] = sum_j (w_j tangent_in_j[i] + wd_j in_j[i])
#pragma unroll 1
    for (int which = 0; which < 2; ++which) {
      float* out = which ? d_g_color : d_g_geo;
      if (!out) continue;
      const int F = which ? Fc : Fg;
      const float* tab = which ? t.color_features : t.geo_features;
      const float* ggt = which ? gg_color_feat : gg_geo_feat;
      for (int i = 0; i < F + 3; ++i) {
        float a = 0.f;
        if (g.valid) {
          const float in_i = i < F ? tab[g.id * F + i] : (i == F ? g.nx : (i == F + 1 ? g.ny : g.nz));
          const float tin_i = i < F ? (ggt ? ggt[g.id * F + i] : 0.f) : (i == F ? tnx : (i == F + 1 ? tny : tnz));
          a = g.w * tin_i + wd * in_i;
        }
        a = row16_sum(a);
        if (in_range && j == 0) out[q * (F + 3) + i] = a;
      }
    }
  } else if (in_range && j < nnk && d_g_n) {   // split layout: only the neighbour-vector part is differentiated here
    float* o = d_g_n + (q * nnk + j) * 3;
    o[0] = tnx; o[1] = tny; o[2] = tnz;
  } else if (in_range && j < nnk) {
#pragma unroll 1
    for (int which = 0; which < 2; ++which) {
      float* out = which ? d_g_color : d_g_geo;
      if (!out) continue;
      const int F = which ? Fc : Fg;
      const float* ggt = which ? gg_color_feat : gg_geo_feat;
      float* o = out + (q * nnk + j) * (F + 3);
      for (int i = 0; i < F; ++i) o[i] = (g.valid && ggt) ? ggt[g.id * F + i] : 0.f;
      o[F] = tnx; o[F + 1] = tny; o[F + 2] = tnz;
    }
  }

  // ---- gradient w.r.t. the query: sum_j c_j d(wd_j)/dx + sum_j tt_j dw_j/dx + sum_j wd_j R_j up_n
  if (d_x) {
    // du_j/dx = -2 u^2 e ;  d(ud_j)/dx = 8 u^3 (e.v) e - 2 u^2 v
    const float k1 = g.valid ? -2.f * g.u * g.u : 0.f;
    const float dux = k1 * g.ex, duy = k1 * g.ey, duz = k1 * g.ez;
    const float k2 = g.valid ? 8.f * g.u * g.u * g.u * ev : 0.f;
    const float dudx = fmaf(k2, g.ex, k1 * vx), dudy = fmaf(k2, g.ey, k1 * vy), dudz = fmaf(k2, g.ez, k1 * vz);
    const float dsx = row16_sum(dux), dsy = row16_sum(duy), dsz = row16_sum(duz);
    const float dsdx = row16_sum(dudx), dsdy = row16_sum(dudy), dsdz = row16_sum(dudz);
    const float Cc = row16_sum(c * g.w), Dd = row16_sum(c * ud);
    const float inv_s = g.s > 0.f ? 1.0f / g.s : 0.f;
    // dw_j/dx = (du_j - w_j ds) / s
    const float dwx = (dux - g.w * dsx) * inv_s, dwy = (duy - g.w * dsy) * inv_s, dwz = (duz - g.w * dsz) * inv_s;
    const float dCx = row16_sum(c * dwx), dCy = row16_sum(c * dwy), dCz = row16_sum(c * dwz);
    const float dDx = row16_sum(c * dudx), dDy = row16_sum(c * dudy), dDz = row16_sum(c * dudz);
    const float num = Dd - sd * Cc;
    float ox = (dDx - dsdx * Cc - sd * dCx) * inv_s - num * dsx * inv_s * inv_s;
    float oy = (dDy - dsdy * Cc - sd * dCy) * inv_s - num * dsy * inv_s * inv_s;
    float oz = (dDz - dsdz * Cc - sd * dCz) * inv_s - num * dsz * inv_s * inv_s;
    if (t.weighted_first) {
      float rx = upnx, ry = upny, rz = upnz;
      if (g.valid && t.after_pgo) rot_active(t.orientations + 4 * g.id, upnx, upny, upnz, rx, ry, rz);
      const float px = g.valid ? fmaf(tt, dwx, wd * rx) : 0.f, py = g.valid ? fmaf(tt, dwy, wd * ry) : 0.f,
                  pz = g.valid ? fmaf(tt, dwz, wd * rz) : 0.f;
      ox += row16_sum(px); oy += row16_sum(py); oz += row16_sum(pz);
    }
    if (in_range && j == 0) { d_x[3 * q] = ox; d_x[3 * q + 1] = oy; d_x[3 * q + 2] = oz; }
  }
  // ---- weighted_first: the feature tables receive wd_j * upstream[:F]
  if (in_range && j < nnk && keys) {
    const long long pr = q * nnk + j;
    keys[pr] = g.valid ? (uint32_t)g.id : (uint32_t)rows;
    src_row[pr] = (uint32_t)q;
    pair_w[pr] = wd;
  }
}

size_t au(size_t v) { return (v + 255) / 256 * 256; }

struct QfScratch {
  uint32_t *keys, *src_row;
  float* pair_w;
  void* plan;
  size_t total;
};

QfScratch carve_qf(void* base, int64_t B, int nnk, int64_t rows) {
  QfScratch s;
  char* p = reinterpret_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t bytes) { char* r = p ? p + off : nullptr; off = au(off + bytes); return r; };
  const size_t n = (size_t)(B > 0 ? B : 1) * nnk;
  s.keys = (uint32_t*)take(n * 4);
  s.src_row = (uint32_t*)take(n * 4);
  s.pair_w = (float*)take(n * 4);
  s.plan = take(pings_rows::carve(nullptr, (int64_t)n, rows).total);
  s.total = off;
  return s;
}

int check_tables(const pings_qf_tables* t, int nn_k) {
  PINGS_ARG_CHECK(t != nullptr && t->points, "null tables");
  PINGS_ARG_CHECK(t->Fg >= 0 && t->Fg <= 61 && t->Fc >= 0 && t->Fc <= 61, "feature dims must be <= 61");
  PINGS_ARG_CHECK((t->Fg == 0) == (t->geo_features == nullptr), "geo feature table / dim mismatch");
  PINGS_ARG_CHECK((t->Fc == 0) == (t->color_features == nullptr), "colour feature table / dim mismatch");
  PINGS_ARG_CHECK(!t->after_pgo || t->orientations, "after_pgo needs orientations");
  PINGS_ARG_CHECK(nn_k > 0 && nn_k <= MAX_NNK, "nn_k must be in 1..16");
  return PINGS_OK;
}

}  // namespace

PINGS_API int pings_query_feature_forward(const pings_knn_map* m, const pings_qf_tables* t, const float* queries,
                                          int64_t B, float* geo_out, float* color_out, float* w_out,
                                          int64_t* idx_out, int64_t* gidx_out, int64_t* nn_counts,
                                          float* certainty, float* certainty_accum, const int32_t* query_ts,
                                          int32_t* ts_update, float* n_out, void* stream) {
  if (int e = check_map(m)) return e;
  if (int e = check_tables(t, m->nn_k)) return e;
  PINGS_ARG_CHECK(!geo_out || t->geo_features, "geo output without a geo feature table");
  PINGS_ARG_CHECK(!color_out || t->color_features, "colour output without a colour feature table");
  PINGS_ARG_CHECK(!certainty || t->certainties, "certainty output needs the certainty table");
  if (B == 0) return PINGS_OK;
  PINGS_ARG_CHECK(B > 0 && queries && w_out && idx_out && gidx_out && nn_counts, "null pointer");
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope ps("qf_forward", st);
  hipLaunchKernelGGL(qf_forward_kernel, dim3(grid_for(B)), dim3(64 * WAVES_PER_BLOCK), 0, st, *m, *t, queries,
                     (long long)B, geo_out, color_out, w_out, (long long*)idx_out, (long long*)gidx_out,
                     (long long*)nn_counts, certainty, certainty_accum, (const int*)query_ts, (int*)ts_update,
                     n_out);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}

namespace {
// certainty[row] += w for every (query, neighbour) pair of a batch (:664-689), launched BEHIND the forward kernel: every
// read of the queried certainty (:691-695) has completed by then, so the table itself can be the target and the work
// is O(B nn_k) as the reference's scatter_add_, not O(rows) (round 2: a zeroed table-sized delta + one full-table add)
__global__ __launch_bounds__(256) void qf_accumulate_kernel(const long long* __restrict__ idx, const float* __restrict__ w,
                                                            long long n_pairs, float* __restrict__ cert) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  const long long id = idx[p];
  if (id >= 0) atomicAdd(&cert[id], w[p]);
}
}  // namespace

PINGS_API int pings_query_feature_accumulate(const int64_t* idx, const float* w, int64_t n_pairs, float* certainties,
                                             void* stream) {
  PINGS_ARG_CHECK(n_pairs >= 0, "negative pair count");
  if (n_pairs == 0) return PINGS_OK;
  PINGS_ARG_CHECK(idx && w && certainties, "null pointer");
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope ps("qf_accumulate", st);
  hipLaunchKernelGGL(qf_accumulate_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, st,
                     (const long long*)idx, w, (long long)n_pairs, certainties);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}

PINGS_API size_t pings_query_feature_scratch_bytes(int64_t B, int nn_k, int64_t rows) {
  if (nn_k <= 0 || rows <= 0) return 0;
  return carve_qf(nullptr, B, nn_k, rows).total;
}

PINGS_API int pings_query_feature_backward(const pings_qf_tables* t, const float* global_points,
                                           const float* queries, int64_t B, int nn_k, const int64_t* idx,
                                           const int64_t* gidx, const float* g_geo, const float* g_color,
                                           const float* g_n, const float* g_w, int64_t rows, void* scratch,
                                           float* g_x, float* g_geo_features, float* g_color_features,
                                           void* stream) {
  if (int e = check_tables(t, nn_k)) return e;
  PINGS_ARG_CHECK(rows > 0 && rows < 0x7FFFFFF0LL, "rows out of range");
  PINGS_ARG_CHECK(B >= 0 && (int64_t)B * nn_k < 0x7FFFFFF0LL, "too many (query, neighbour) pairs");
  PINGS_ARG_CHECK(!g_geo_features || (t->geo_features && scratch), "geo feature gradient needs the table and scratch");
  PINGS_ARG_CHECK(!g_color_features || (t->color_features && scratch), "colour feature gradient needs the table and scratch");
  PINGS_ARG_CHECK(B == 0 || (global_points && queries && idx && gidx), "null pointer");
  PINGS_ARG_CHECK(!g_n || !t->weighted_first, "the split layout (g_n) exists in per-neighbour mode only");
  PINGS_ARG_CHECK(!g_n || (!g_geo_features && !g_color_features),
                  "split layout: scatter the feature rows with pings_rows_plan_apply");
  hipStream_t st = pings::as_stream(stream);
  const bool scatter = g_geo_features || g_color_features;
  QfScratch s = carve_qf(scatter ? scratch : nullptr, B, nn_k, rows);
  if (B > 0) {
    pings::prof::Scope ps("qf_backward", st);
    const long long threads = (long long)B * 16;
    hipLaunchKernelGGL(qf_backward_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, *t,
                       global_points, queries, (long long)B, nn_k, (const long long*)idx, (const long long*)gidx,
                       g_geo, g_color, g_n, g_w, (long long)rows, g_x, scatter ? s.keys : nullptr, s.src_row,
                       s.pair_w);
    PINGS_LAUNCH_CHECK();
  }
  if (!scatter) return PINGS_OK;
  pings::prof::Scope ps("qf_scatter", st);
  const int64_t n = (int64_t)B * nn_k;
  pings_rows::Plan plan = pings_rows::carve(s.plan, n, rows);
  if (int e = pings_rows::build(plan, s.keys, n, rows, st)) return e;
  const bool wf = t->weighted_first != 0;
  // the rows to add ARE the upstream gradient: row (b, j) of d geo (or, weighted_first, w_bj times row b)
  if (g_geo_features) {
    if (g_geo) {
      if (int e = pings_rows::gather_sum(plan, rows, t->Fg, g_geo, t->Fg + 3, wf ? s.src_row : nullptr,
                                         wf ? s.pair_w : nullptr, g_geo_features, st)) return e;
    } else {
      PINGS_HIP_CHECK(hipMemsetAsync(g_geo_features, 0, sizeof(float) * (size_t)rows * t->Fg, st));
    }
  }
  if (g_color_features) {
    if (g_color) {
      if (int e = pings_rows::gather_sum(plan, rows, t->Fc, g_color, t->Fc + 3, wf ? s.src_row : nullptr,
                                         wf ? s.pair_w : nullptr, g_color_features, st)) return e;
    } else {
      PINGS_HIP_CHECK(hipMemsetAsync(g_color_features, 0, sizeof(float) * (size_t)rows * t->Fc, st));
    }
  }
  return PINGS_OK;
}

PINGS_API int pings_query_feature_double_backward(
    const pings_qf_tables* t, const float* global_points, const float* queries, int64_t B, int nn_k,
    const int64_t* idx, const int64_t* gidx, const float* g_geo, const float* g_color, const float* g_w,
    const float* gg_x, const float* gg_geo_features, const float* gg_color_features, int64_t rows, void* scratch,
    float* d_g_geo, float* d_g_color, float* d_g_n, float* d_g_w, float* d_x, float* d_geo_features,
    float* d_color_features, void* stream) {
  if (int e = check_tables(t, nn_k)) return e;
  PINGS_ARG_CHECK(rows > 0 && rows < 0x7FFFFFF0LL, "rows out of range");
  PINGS_ARG_CHECK(B >= 0 && (int64_t)B * nn_k < 0x7FFFFFF0LL, "too many (query, neighbour) pairs");
  PINGS_ARG_CHECK(B == 0 || (global_points && queries && idx && gidx), "null pointer");
  PINGS_ARG_CHECK(!d_g_geo || t->geo_features, "d_g_geo without a geo table");
  PINGS_ARG_CHECK(!d_g_color || t->color_features, "d_g_color without a colour table");
  PINGS_ARG_CHECK(!d_g_n || (!t->weighted_first && !d_g_geo && !d_g_color),
                  "the split layout (d_g_n) exists in per-neighbour mode only and excludes d_g_geo / d_g_color");
  const bool wf = t->weighted_first != 0;
  const bool scatter = wf && (d_geo_features || d_color_features);
  PINGS_ARG_CHECK(wf || (!d_geo_features && !d_color_features),
                  "the feature tables receive a second-order gradient in weighted_first mode only");
  PINGS_ARG_CHECK(!scatter || scratch, "null scratch");
  hipStream_t st = pings::as_stream(stream);
  QfScratch s = carve_qf(scatter ? scratch : nullptr, B, nn_k, rows);
  if (B > 0) {
    pings::prof::Scope ps("qf_double_backward", st);
    const long long threads = (long long)B * 16;
    hipLaunchKernelGGL(qf_double_backward_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, *t,
                       global_points, queries, (long long)B, nn_k, (const long long*)idx, (const long long*)gidx,
                       g_geo, g_color, g_w, gg_x, gg_geo_features, gg_color_features, (long long)rows, d_g_geo,
                       d_g_color, d_g_n, d_g_w, d_x, scatter ? s.keys : nullptr, s.src_row, s.pair_w);
    PINGS_LAUNCH_CHECK();
  }
  if (!scatter) return PINGS_OK;
  pings::prof::Scope ps("qf_scatter", st);
  const int64_t n = (int64_t)B * nn_k;
  pings_rows::Plan plan = pings_rows::carve(s.plan, n, rows);
  if (int e = pings_rows::build(plan, s.keys, n, rows, st)) return e;
  if (d_geo_features) {
    if (g_geo) {
      if (int e = pings_rows::gather_sum(plan, rows, t->Fg, g_geo, t->Fg + 3, s.src_row, s.pair_w, d_geo_features, st))
        return e;
    } else {
      PINGS_HIP_CHECK(hipMemsetAsync(d_geo_features, 0, sizeof(float) * (size_t)rows * t->Fg, st));
    }
  }
  if (d_color_features) {
    if (g_color) {
      if (int e = pings_rows::gather_sum(plan, rows, t->Fc, g_color, t->Fc + 3, s.src_row, s.pair_w,
                                         d_color_features, st)) return e;
    } else {
      PINGS_HIP_CHECK(hipMemsetAsync(d_color_features, 0, sizeof(float) * (size_t)rows * t->Fc, st));
    }
  }
  return PINGS_OK;
}
