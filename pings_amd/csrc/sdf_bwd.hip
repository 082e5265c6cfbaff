// Backward and backward-of-the-backward of the fused neural-point SDF query
//   S(x) = sum_m w_m * scale * MLP([f_{idx_m}, n_m])          (per-neighbour mode, utils/mapper.py:2273-2289)
//   S(x) = scale * MLP(sum_m w_m [f_{idx_m}, n_m])             (weighted_first, model/neural_gaussians.py:701-705)
// with respect to the feature table and the decoder.  Training path of Mapper.sdf_mapping (utils/mapper.py:822-970)
// and of the Gaussian <-> SDF consistency loss, which differentiates d S / d x once more
// (get_gradient(create_graph=True), utils/tools.py:409-419, utils/mapper.py:1445-1448).
//
//   sdf_grad_kernel<IN_PAD, SECOND>   one wave64 per query, lane h = hidden unit h (its W1 row, b1, W2 live in
//       registers across the queries of the wave, and so do its rows of dW1 / db1 / dW2: weight gradients are never
//       reduced across lanes).  For every decoder evaluation e (one per neighbour, or one per query when
//       weighted_first) with input row X, pre-activation mask m and q = m * W2:
//           db2 += a          dW2 += a relu(pre) + o m (W1 Xd)          db1 += a q          dW1 += q (x) (a X + o Xd)
//           d X  = c W1^T q   (feature columns only: one row per evaluation, scattered into the table afterwards)
//       first order   (SECOND = false):  a = dL/dS * scale * w_m,  o = 0,  c = a     (weighted_first: a = dL/dS * scale,
//           rows carry w_m)
//       second order  (SECOND = true) :  the objective is  Phi = <v, dS/dx>  =  sum_m (wd_m s_m + w_m sd_m),  the
//           directional derivative of S along v (v = upstream gradient of dS/dx): wd_m = <dw_m/dx, v>, and sd_m the
//           tangent of the decoder output along Xd = [0, R_m^T v]  =>  a = scale * wd_m, o = scale * w_m, c = a
//           (weighted_first: X = G, Xd = sum_m (w_m [0, R_m^T v] + wd_m in_m), a = 0, o = scale, rows carry wd_m).
//       relu'' = 0 almost everywhere, as in torch's double backward.  The feature-gradient reduction d X = c W1^T q
//       runs with LANES ON COLUMNS: the mask m of a row is a 64-bit ballot, lane group g owns evaluation g and four
//       columns, and walks the hidden units reading W1[h][c..c+3] * W2[h] from LDS (no cross-lane sums at all).
//   sdf_param_reduce_kernel           fixed-order sum of the per-workgroup partials of dW1 / db1 / dW2 / db2.
//   pings_rows::build + gather_sum    deterministic scatter-add of the feature rows (csrc/row_scatter.hip).
//
// Everything is fp32 VALU work at ~1,000 wave-instructions per query; the gather + MFMA pipeline of round 1 needed
// the [B k, F+3] input rows and their gradients materialised in HBM and three more launches.
// Bitwise reproducible: no float atomics, fixed summation orders.
#include <algorithm>
#include <string>
#include "knn_common.hpp"
#include "row_scatter.hpp"

namespace {
using namespace pings_knn;

constexpr int WPB = 4;          // waves per workgroup

// value of `v` in lane `src` (wave-uniform index): v_readlane instead of a ds_bpermute round trip
__device__ inline float lane_bcast(float v, int src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}
constexpr float IDW_EPS = 1e-15f;

struct GradArgs {
  const float *W1, *b1, *W2;
  int H, F, weighted_first;
  float scale;
  const float *features, *points, *orientations, *gpoints;
  int after_pgo;
  const float* queries;
  long long B;
  int nnk;
  const long long *idx, *gidx;
  const float* w;
  const float* dL_dsdf;   // first order: [B]
  const float* vdir;      // second order: [B,3]
  float* rows;            // [evaluations][F] feature-gradient rows
  uint32_t* keys;         // [B*nnk] destination row of every (query, neighbour) pair (table_rows = invalid)
  uint32_t* src_row;      // weighted_first: pair -> its query's row
  float* pair_w;          // weighted_first: pair weight (w_m or wd_m)
  long long table_rows;
  float* partials;        // [gridDim.x][H * (IN + 2) + 1]
};

template <int IN_PAD, bool SECOND>
__global__ __launch_bounds__(64 * WPB, IN_PAD <= 36 ? 3 : 2) void sdf_grad_kernel(GradArgs a) {
  __shared__ __attribute__((aligned(16))) float sV[64 * 64];   // V[h][c] = W1[h][c] * W2[h], c < F (row stride FP)
  __shared__ __attribute__((aligned(16))) float sIn[WPB][MAX_NNK][IN_PAD];   // input rows of the wave's query (IN_PAD % 4 == 0)
  __shared__ unsigned long long sMask[WPB][MAX_NNK];
  __shared__ float sCoef[WPB][MAX_NNK];         // row coefficient c of every evaluation
  __shared__ float sRed[WPB][64];               // cross-wave reduction of the parameter gradients

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int F = a.F, IN = F + 3, H = a.H, nnk = a.nnk;
  const int FP = (F + 3) & ~3;
  const bool wf = a.weighted_first != 0;
  float w1[IN_PAD], gW1[IN_PAD];
  float b1 = 0.f, w2 = 0.f, gb1 = 0.f, gW2 = 0.f, gb2 = 0.f;
  // the three direction-input weights of this unit and their gradient accumulators are kept apart: F is a runtime
  // value, and indexing the register arrays with it would send them to scratch
  float w1n0, w1n1, w1n2;
  load_w1_rows<IN_PAD>(a.W1, IN, H, &sIn[0][0][0], w1, w1n0, w1n1, w1n2);
#pragma unroll
  for (int i = 0; i < IN_PAD; ++i) gW1[i] = 0.f;
  if (lane < H) { b1 = a.b1[lane]; w2 = a.W2[lane]; }
  float gWn0 = 0.f, gWn1 = 0.f, gWn2 = 0.f;
  for (int e = threadIdx.x; e < 64 * FP; e += 64 * WPB) {
    const int h = e / FP, c = e - h * FP;
    sV[e] = (h < H && c < F) ? a.W1[h * IN + c] * a.W2[h] : 0.f;
  }
  __syncthreads();

  const int Q4 = FP / 4;
  const int LPR = Q4 <= 1 ? 1 : (Q4 <= 2 ? 2 : (Q4 <= 4 ? 4 : (Q4 <= 8 ? 8 : 16)));  // lanes per row
  const int groups = 64 / LPR;
  const long long nwaves = (long long)gridDim.x * WPB;
  for (long long q = (long long)blockIdx.x * WPB + wave; q < a.B; q += nwaves) {
    const float qx = a.queries[3 * q], qy = a.queries[3 * q + 1], qz = a.queries[3 * q + 2];
    // ---- per-neighbour geometry on lanes m < nnk
    long long id = -1;
    float wm = 0.f, nx = 0.f, ny = 0.f, nz = 0.f, wd = 0.f, tnx = 0.f, tny = 0.f, tnz = 0.f;
    float vx = 0.f, vy = 0.f, vz = 0.f;
    if (SECOND) { vx = a.vdir[3 * q]; vy = a.vdir[3 * q + 1]; vz = a.vdir[3 * q + 2]; }
    float u = 0.f, ev = 0.f;
    if (lane < nnk) {
      id = a.idx[q * nnk + lane];
      wm = a.w[q * nnk + lane];
      if (id >= 0) {
        const float px = qx - a.points[3 * id], py = qy - a.points[3 * id + 1], pz = qz - a.points[3 * id + 2];
        nx = px; ny = py; nz = pz;
        if (a.after_pgo) rot_passive(a.orientations + 4 * id, px, py, pz, nx, ny, nz);
        if (SECOND) {
          tnx = vx; tny = vy; tnz = vz;
          if (a.after_pgo) rot_passive(a.orientations + 4 * id, vx, vy, vz, tnx, tny, tnz);
          const long long gi = a.gidx[q * nnk + lane];
          const float ex = qx - a.gpoints[3 * gi], ey = qy - a.gpoints[3 * gi + 1], ez = qz - a.gpoints[3 * gi + 2];
          u = 1.0f / (((ex * ex + ey * ey) + ez * ez) + IDW_EPS);
          ev = (ex * vx + ey * vy) + ez * vz;
        }
      }
      sIn[wave][lane][F] = nx; sIn[wave][lane][F + 1] = ny; sIn[wave][lane][F + 2] = nz;
      for (int i = IN; i < IN_PAD; ++i) sIn[wave][lane][i] = 0.f;
    }
    if (SECOND) {   // tangent of the weights along v
      const float s = wave_sum_all(u);
      const float ud = -2.f * u * u * ev;
      const float sd = wave_sum_all(ud);
      wd = (lane < nnk && id >= 0) ? (ud - wm * sd) / s : 0.f;
    }
    // feature rows -> LDS (zeros for missing neighbours)
    for (int e = lane; e < nnk * F; e += 64) {
      const int mm = e / F, f = e - mm * F;
      const long long idm = a.idx[q * nnk + mm];
      sIn[wave][mm][f] = idm >= 0 ? a.features[idm * F + f] : 0.f;
    }
    const float gS = SECOND ? 1.0f : a.dL_dsdf[q];
    __builtin_amdgcn_wave_barrier();

    if (wf) {
      // one evaluation per query on G = sum_m w_m in_m  (and, second order, the tangent input Xd)
      float xin[IN_PAD];
#pragma unroll
      for (int i = 0; i < IN_PAD; ++i) xin[i] = 0.f;
      for (int mm = 0; mm < nnk; ++mm) {
        const float wmm = lane_bcast(wm, mm);
#pragma unroll
        for (int i = 0; i < IN_PAD; ++i) xin[i] = fmaf(wmm, sIn[wave][mm][i], xin[i]);
      }
      float pre = b1;
#pragma unroll
      for (int i = 0; i < IN_PAD; ++i) pre = fmaf(w1[i], xin[i], pre);
      const bool on = lane < H && pre > 0.f;
      const float qh = on ? w2 : 0.f;
      float rowc;
      if (SECOND) {
        // Xd = sum_m (w_m [0, R^T v] + wd_m in_m)
        float xd[IN_PAD];
#pragma unroll
        for (int i = 0; i < IN_PAD; ++i) xd[i] = 0.f;
        for (int mm = 0; mm < nnk; ++mm) {
          const float wmm = lane_bcast(wm, mm), wdm = lane_bcast(wd, mm);
          const float t0 = lane_bcast(tnx, mm), t1 = lane_bcast(tny, mm), t2 = lane_bcast(tnz, mm);
#pragma unroll
          for (int i = 0; i < IN_PAD; ++i) {
            const float tin = i == F ? t0 : (i == F + 1 ? t1 : (i == F + 2 ? t2 : 0.f));
            xd[i] = fmaf(wmm, tin, fmaf(wdm, sIn[wave][mm][i], xd[i]));
          }
        }
        float xd_dot = 0.f;
#pragma unroll
        for (int i = 0; i < IN_PAD; ++i) xd_dot = fmaf(w1[i], xd[i], xd_dot);
        const float o = a.scale;
        gW2 = fmaf(o, on ? xd_dot : 0.f, gW2);
        const float qo = qh * o;
#pragma unroll
        for (int i = 0; i < IN_PAD; ++i) gW1[i] = fmaf(qo, xd[i], gW1[i]);
        rowc = a.scale;
      } else {
        const float aco = gS * a.scale;
        gb2 += aco;
        gW2 = fmaf(aco, on ? pre : 0.f, gW2);
        gb1 = fmaf(aco, qh, gb1);
        const float qa = qh * aco;
#pragma unroll
        for (int i = 0; i < IN_PAD; ++i) gW1[i] = fmaf(qa, xin[i], gW1[i]);
        rowc = aco;
      }
      const unsigned long long mk = __ballot(on);
      if (lane == 0) { sMask[wave][0] = mk; sCoef[wave][0] = rowc; }
      if (lane < nnk) {
        const long long pr = q * nnk + lane;
        a.keys[pr] = id >= 0 ? (uint32_t)id : (uint32_t)a.table_rows;
        a.src_row[pr] = (uint32_t)q;
        a.pair_w[pr] = SECOND ? wd : wm;
      }
    } else {
      for (int mm = 0; mm < nnk; ++mm) {
        float xr[IN_PAD];   // the row, broadcast from LDS sixteen bytes at a time, used for pre AND the dW1 update
#pragma unroll
        for (int i = 0; i < IN_PAD; i += 4) {
          const float4 t4 = *reinterpret_cast<const float4*>(&sIn[wave][mm][i]);
          xr[i] = t4.x; xr[i + 1] = t4.y; xr[i + 2] = t4.z; xr[i + 3] = t4.w;
        }
        float pre = b1;
#pragma unroll
        for (int i = 0; i < IN_PAD; ++i) pre = fmaf(w1[i], xr[i], pre);
        const bool on = lane < H && pre > 0.f;
        const float qh = on ? w2 : 0.f;
        const float wmm = lane_bcast(wm, mm);
        float aco, oco = 0.f;
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
        if (SECOND) {
          aco = a.scale * lane_bcast(wd, mm);
          oco = a.scale * wmm;
          t0 = lane_bcast(tnx, mm); t1 = lane_bcast(tny, mm); t2 = lane_bcast(tnz, mm);
          const float xd_dot = (w1n0 * t0 + w1n1 * t1) + w1n2 * t2;   // W1[h, n-part] . R^T v
          gW2 = fmaf(oco, on ? xd_dot : 0.f, gW2);
        } else {
          aco = gS * a.scale * wmm;
        }
        gb2 += aco;
        gW2 = fmaf(aco, on ? pre : 0.f, gW2);
        gb1 = fmaf(aco, qh, gb1);
        const float qa = qh * aco;
#pragma unroll
        for (int i = 0; i < IN_PAD; ++i) gW1[i] = fmaf(qa, xr[i], gW1[i]);
        if (SECOND) {
          const float qo = qh * oco;
          gWn0 = fmaf(qo, t0, gWn0); gWn1 = fmaf(qo, t1, gWn1); gWn2 = fmaf(qo, t2, gWn2);
        }
        const unsigned long long mk = __ballot(on);
        if (lane == 0) { sMask[wave][mm] = mk; sCoef[wave][mm] = aco; }
      }
      if (lane < nnk) a.keys[q * nnk + lane] = id >= 0 ? (uint32_t)id : (uint32_t)a.table_rows;
    }
    __builtin_amdgcn_wave_barrier();

    // ---- feature-gradient rows, lanes on columns: group g = evaluation, lane c4 = four columns
    const int nev = wf ? 1 : nnk;
    const int g = lane / LPR, c4 = 4 * (lane % LPR);
    for (int e0 = 0; e0 < nev; e0 += groups) {
      const int ev_i = e0 + g;
      const bool act = ev_i < nev && c4 < FP;
      const unsigned long long mk = act ? sMask[wave][ev_i] : 0ull;
      float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
      const float* vrow = &sV[act ? c4 : 0];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (half * 32 >= H) break;
        const uint32_t m32 = half ? (uint32_t)(mk >> 32) : (uint32_t)mk;
#pragma unroll 8
        for (int hb = 0; hb < 32; ++hb) {   // rows of V beyond H are zero: no bound check needed inside a half
          const float4 vv = *reinterpret_cast<const float4*>(vrow + (half * 32 + hb) * FP);
          const float bit = (float)((m32 >> hb) & 1u);
          r0 = fmaf(bit, vv.x, r0); r1 = fmaf(bit, vv.y, r1); r2 = fmaf(bit, vv.z, r2); r3 = fmaf(bit, vv.w, r3);
        }
      }
      if (act) {
        const float cf = sCoef[wave][ev_i];
        float* o = a.rows + ((size_t)(wf ? q : q * nnk + ev_i)) * F + c4;
        if (c4 < F) o[0] = cf * r0;
        if (c4 + 1 < F) o[1] = cf * r1;
        if (c4 + 2 < F) o[2] = cf * r2;
        if (c4 + 3 < F) o[3] = cf * r3;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }

  // ---- parameter gradients: sum the workgroup's four waves in wave order, one partial per workgroup
  const int PSZ = H * (IN + 2) + 1;
  float* out = a.partials + (size_t)blockIdx.x * PSZ;
  auto wg_sum = [&](float v) {
    __syncthreads();
    sRed[wave][lane] = v;
    __syncthreads();
    return ((sRed[0][lane] + sRed[1][lane]) + sRed[2][lane]) + sRed[3][lane];
  };
#pragma unroll
  for (int i = 0; i < IN_PAD; ++i) {
    const float extra = i == F ? gWn0 : (i == F + 1 ? gWn1 : (i == F + 2 ? gWn2 : 0.f));
    const float s = wg_sum(gW1[i] + extra);
    if (wave == 0 && lane < H && i < IN) out[lane * IN + i] = s;
  }
  {
    const float s1 = wg_sum(gb1), s2 = wg_sum(gW2), s3 = wg_sum(gb2);
    if (wave == 0 && lane < H) { out[H * IN + lane] = s1; out[H * IN + H + lane] = s2; }
    if (wave == 0 && lane == 0) out[H * IN + 2 * H] = s3;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// First-order backward of the per-neighbour decoder on the matrix cores (nn_k <= 8, F a power-of-two multiple of 4
// up to 32): the counterpart of sdf_forward_mfma_kernel.  A wave takes four queries = 32 (query, neighbour) columns:
//   pre  = [W1 | b1] X                       32x32x2 MFMA, X staged in LDS in operand order (as in the forward)
//   Gp   = a_n W2[h] 1[pre > 0]              (dL/dpre; a_n = dL/dS scale w_n)   from the accumulator registers
//   dX^T = W1^T Gp                           MFMA with the Gp ACCUMULATOR as the B operand (its column is on the lane
//                                            and its rows in the 16 registers: the operand layout in row order
//                                            rowmap32) -> the feature-gradient rows, 16-byte stores
//   dW1 += Gp X^T                            MFMA over the columns: Gp takes one trip through LDS (transposed read),
//                                            X is read from its staging area with the column as k
//   dW1[:, F..F+2], db1 (= the bias column)  when they do not fit the 32-wide block (F = 32): per-lane FMAs on the
//                                            transposed Gp against the four tail inputs (LDS broadcast)
//   dW2 += sum_n a_n relu(pre)               the same transposed trip, summed in the lane
// The weight-gradient accumulators stay in registers across the wave's queries; one partial per workgroup, summed in
// fixed order.  ~200 vector instructions per query instead of ~1,100.
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int GQ = 4, GNB = 8, GCOLS = GQ * GNB;
constexpr int GROW = 36;   // row of a transposed 32x32 block in LDS: [k = n & 1][t = n >> 1] + 4 (conflict-free 16-B reads)

template <int IN_PAD>
__global__ __launch_bounds__(64 * WPB, IN_PAD > 32 ? 2 : 3) void sdf_grad_mfma_kernel(GradArgs a) {
  constexpr int HALF = IN_PAD / 2, HALF_PAD = (HALF + 3) & ~3, ROW = 2 * HALF_PAD + 4;
  constexpr bool TAIL = IN_PAD > 32;                       // inputs 32.. (direction + bias column) outside the main block
  constexpr int XW = GCOLS * ROW > 32 * GROW ? GCOLS * ROW : 32 * GROW;
  __shared__ __attribute__((aligned(16))) float sX[WPB][XW];
  __shared__ __attribute__((aligned(16))) float sA[2][HALF_PAD / 4][64][4];   // forward A operand
  __shared__ __attribute__((aligned(16))) float sWT[2][4][64][4];             // dX A operand: W1[h = rowmap(s)][i = lane]
  __shared__ __attribute__((aligned(16))) float sTail[WPB][GCOLS][4];         // inputs 32..35 of every column
  __shared__ float sW2[64];
  __shared__ int sId[WPB][GCOLS];
  __shared__ float sRed[WPB][64];
  static_assert(WPB * XW >= 64 * IN_PAD, "the weight staging area must hold W1");

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int F = a.F, IN = F + 3, H = a.H, nnk = a.nnk;
  const int r32 = lane & 31, kh = lane >> 5;
  {
    float* stage = &sX[0][0];
    const int stride = IN | 1;
    for (int h = wave; h < H; h += WPB)
      for (int c = lane; c < IN; c += 64) stage[h * stride + c] = a.W1[h * IN + c];
    for (int h = threadIdx.x; h < 64; h += 64 * WPB) sW2[h] = h < H ? a.W2[h] : 0.f;
    __syncthreads();
    if (wave < 2) {
      const int row = wave * 32 + r32;
      for (int t = 0; t < HALF_PAD; ++t) {
        const int c = 2 * t + kh;
        float v = 0.f;
        if (row < H && t < HALF) v = c < IN ? stage[row * stride + c] : (c == IN ? a.b1[row] : 0.f);
        sA[wave][t >> 2][lane][t & 3] = v;
      }
      for (int s = 0; s < 16; ++s) {
        const int h = wave * 32 + rowmap32(s, kh);
        sWT[wave][s >> 2][lane][s & 3] = (h < H && r32 < F) ? stage[h * stride + r32] : 0.f;
      }
    }
    __syncthreads();
  }
  const bool two_blocks = H > 32;
  const int F4 = F >> 2;
  const int f4_shift = 31 - __clz(F4);
  const int n_gather = (GCOLS * F4 + 63) >> 6;
  float* X = sX[wave];
  const int j_of = r32 >> 3, mm_of = r32 & 7;

  f32x16 accW1[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int s = 0; s < 16; ++s) accW1[mb][s] = 0.f;
  float tailW[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  float accW2[2] = {0.f, 0.f};
  float accB2 = 0.f;

  const long long ngroups = (a.B + GQ - 1) / GQ;
  const long long nwaves = (long long)gridDim.x * WPB;
  for (long long g = (long long)blockIdx.x * WPB + wave; g < ngroups; g += nwaves) {
    // ---- column owner: neighbour, weight, upstream gradient, geometry (lanes 32..63 mirror 0..31)
    const long long q_mine = g * GQ + j_of;
    const bool slot_on = mm_of < nnk && q_mine < a.B;
    long long id = -1;
    float an = 0.f, qx = 0.f, qy = 0.f, qz = 0.f;
    if (slot_on) {
      id = a.idx[q_mine * nnk + mm_of];
      an = a.dL_dsdf[q_mine] * a.scale * a.w[q_mine * nnk + mm_of];
      qx = a.queries[3 * q_mine]; qy = a.queries[3 * q_mine + 1]; qz = a.queries[3 * q_mine + 2];
    }
    if (lane < GCOLS) sId[wave][r32] = (int)id;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (id >= 0) {
      const float px = qx - a.points[3 * id], py = qy - a.points[3 * id + 1], pz = qz - a.points[3 * id + 2];
      nx = px; ny = py; nz = pz;
      if (a.after_pgo) rot_passive(a.orientations + 4 * id, px, py, pz, nx, ny, nz);
    } else {
      an = 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = 0; i < n_gather; ++i) {
      const int e = lane + 64 * i;
      if (e < GCOLS * F4) {
        const int col = e >> f4_shift, c4 = e & (F4 - 1);
        const int idc = sId[wave][col];
        const float4 v = idc >= 0 ? reinterpret_cast<const float4*>(a.features + (size_t)idc * F)[c4]
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float2*>(&X[col * ROW + 2 * c4]) = make_float2(v.x, v.z);
        *reinterpret_cast<float2*>(&X[col * ROW + HALF_PAD + 2 * c4]) = make_float2(v.y, v.w);
      }
    }
    if (lane < GCOLS) {
      float* row = &X[r32 * ROW];
      const float tail[4] = {nx, ny, nz, 1.f};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int col = F + c;
        row[(col & 1) * HALF_PAD + (col >> 1)] = tail[c];
      }
      for (int col = IN + 1; col < IN_PAD; ++col) row[(col & 1) * HALF_PAD + (col >> 1)] = 0.f;
      if (TAIL) *reinterpret_cast<float4*>(sTail[wave][r32]) = make_float4(nx, ny, nz, 1.f);
      if (slot_on) a.keys[q_mine * nnk + mm_of] = id >= 0 ? (uint32_t)id : (uint32_t)a.table_rows;
      accB2 += an;
    }
    __builtin_amdgcn_wave_barrier();

    // ---- operands out of the staging area: forward B (column r32, k half kh) and dW1 B (column 2t + kh, input r32)
    float bx[HALF_PAD];
#pragma unroll
    for (int t = 0; t < HALF_PAD; t += 4) {
      const float4 v = *reinterpret_cast<const float4*>(&X[r32 * ROW + kh * HALF_PAD + t]);
      bx[t] = v.x; bx[t + 1] = v.y; bx[t + 2] = v.z; bx[t + 3] = v.w;
    }
    float bxT[16];
#pragma unroll
    for (int t = 0; t < 16; ++t)
      bxT[t] = r32 < IN_PAD ? X[(2 * t + kh) * ROW + (r32 & 1) * HALF_PAD + (r32 >> 1)] : 0.f;
    __builtin_amdgcn_wave_barrier();   // the staging area is free from here on (transposes below)

    f32x16 dX = {0};
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      if (mb == 1 && !two_blocks) break;
      f32x16 acc = {0};
#pragma unroll
      for (int t = 0; t < HALF_PAD; t += 4) {
        const float4 a4 = *reinterpret_cast<const float4*>(sA[mb][t >> 2][lane]);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, bx[t], acc, 0, 0, 0);
        if (t + 1 < HALF) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, bx[t + 1], acc, 0, 0, 0);
        if (t + 2 < HALF) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, bx[t + 2], acc, 0, 0, 0);
        if (t + 3 < HALF) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, bx[t + 3], acc, 0, 0, 0);
      }
      // Gp = a_n W2 1[pre > 0], R = a_n relu(pre)   (column r32 on the lane, hidden rowmap32(s, kh) in register s)
      float gp[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) gp[s] = acc[s] > 0.f ? an * sW2[mb * 32 + rowmap32(s, kh)] : 0.f;
      // dX^T += W1^T Gp  (accumulator layout as the B operand)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const float4 w4 = *reinterpret_cast<const float4*>(sWT[mb][s4][lane]);
        dX = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, gp[4 * s4], dX, 0, 0, 0);
        dX = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, gp[4 * s4 + 1], dX, 0, 0, 0);
        dX = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, gp[4 * s4 + 2], dX, 0, 0, 0);
        dX = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.w, gp[4 * s4 + 3], dX, 0, 0, 0);
      }
      // Gp transposed through LDS: lane (h = r32, k half kh) reads Gp[h][n = 2t + kh], t = 0..15
#pragma unroll
      for (int s = 0; s < 16; ++s) X[rowmap32(s, kh) * GROW + (r32 & 1) * 16 + (r32 >> 1)] = gp[s];
      __builtin_amdgcn_wave_barrier();
      float gt[16];
#pragma unroll
      for (int t = 0; t < 16; t += 4) {
        const float4 v = *reinterpret_cast<const float4*>(&X[r32 * GROW + kh * 16 + t]);
        gt[t] = v.x; gt[t + 1] = v.y; gt[t + 2] = v.z; gt[t + 3] = v.w;
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int t = 0; t < 16; ++t) accW1[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(gt[t], bxT[t], accW1[mb], 0, 0, 0);
      if (TAIL) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const float4 x4 = *reinterpret_cast<const float4*>(sTail[wave][2 * t + kh]);
          tailW[mb][0] = fmaf(gt[t], x4.x, tailW[mb][0]);
          tailW[mb][1] = fmaf(gt[t], x4.y, tailW[mb][1]);
          tailW[mb][2] = fmaf(gt[t], x4.z, tailW[mb][2]);
          tailW[mb][3] = fmaf(gt[t], x4.w, tailW[mb][3]);
        }
      }
      // R the same way, summed over its columns in the lane
#pragma unroll
      for (int s = 0; s < 16; ++s) X[rowmap32(s, kh) * GROW + (r32 & 1) * 16 + (r32 >> 1)] = an * fmaxf(acc[s], 0.f);
      __builtin_amdgcn_wave_barrier();
      float rs = 0.f;
#pragma unroll
      for (int t = 0; t < 16; t += 4) {
        const float4 v = *reinterpret_cast<const float4*>(&X[r32 * GROW + kh * 16 + t]);
        rs += (v.x + v.y) + (v.z + v.w);
      }
      accW2[mb] += rs;
      __builtin_amdgcn_wave_barrier();
    }
    // ---- feature-gradient rows: dX^T[i = rowmap32(s, kh)][column r32], four consecutive inputs per 16-byte store
    if (slot_on) {
      float* o = a.rows + (size_t)(q_mine * nnk + mm_of) * F;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int i0 = 8 * g4 + 4 * kh;
        if (i0 < F)
          *reinterpret_cast<float4*>(o + i0) = make_float4(dX[4 * g4], dX[4 * g4 + 1], dX[4 * g4 + 2], dX[4 * g4 + 3]);
      }
    }
  }

  // ---- parameter gradients: the four waves of the workgroup in wave order, one partial per workgroup
  const int PSZ = H * (IN + 2) + 1;
  float* out = a.partials + (size_t)blockIdx.x * PSZ;
  auto wg_sum = [&](float v) {
    __syncthreads();
    sRed[wave][lane] = v;
    __syncthreads();
    return ((sRed[0][lane] + sRed[1][lane]) + sRed[2][lane]) + sRed[3][lane];
  };
  auto half_sum = [&](float v) {   // the two lane halves hold the even / odd columns' shares of the same hidden unit
    float x = v, y = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
  };
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float v = wg_sum(accW1[mb][s]);      // dW1aug[h = mb*32 + rowmap32(s, kh)][i = r32]
      const int h = mb * 32 + rowmap32(s, kh);
      if (wave == 0 && h < H) {
        if (r32 < IN) out[h * IN + r32] = v;
        else if (!TAIL && r32 == IN) out[H * IN + h] = v;   // the bias column is db1
      }
    }
    if (TAIL) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float v = wg_sum(half_sum(tailW[mb][c]));
        const int h = mb * 32 + r32;
        if (wave == 0 && kh == 0 && h < H) {
          if (c < 3) out[h * IN + 32 + c] = v; else out[H * IN + h] = v;
        }
      }
    }
    const float v2 = wg_sum(half_sum(accW2[mb]));
    if (wave == 0 && kh == 0 && mb * 32 + r32 < H) out[H * IN + H + mb * 32 + r32] = v2;
  }
  {
    float v = lane < GCOLS ? accB2 : 0.f;
    v = wave_sum_all(v);
    const float s3 = wg_sum(v);
    if (wave == 0 && lane == 0) out[H * IN + 2 * H] = s3;
  }
}

bool grad_mfma_supported(bool second, const pings_sdf_decoder* dec, int nn_k, const float* features, const float* rows) {
  const int F = dec->feat_dim, F4 = F >> 2;
  const char* e = getenv("PINGS_SDF_BWD");
  if (e && std::string(e) == "vector") return false;
  return !second && !dec->weighted_first && nn_k <= GNB && (F & 3) == 0 && F4 > 0 && (F4 & (F4 - 1)) == 0 && F <= 32 &&
         dec->hidden <= 64 && ((reinterpret_cast<uintptr_t>(features) | reinterpret_cast<uintptr_t>(rows)) & 15u) == 0;
}

// out[e] = sum over the per-workgroup partials, one wave per output element: lane l adds partials l, l + 64, ... in
// order (independent loads, all in flight), the wave then adds its 64 lane sums in a fixed tree: the same bits every
// run.  dW1 [H, IN] | db1 [H] | dW2 [H] | db2 [1]
__global__ __launch_bounds__(256) void sdf_param_reduce_kernel(const float* __restrict__ partials, int nblocks, int PSZ,
                                                                float* __restrict__ dW1, float* __restrict__ db1,
                                                                float* __restrict__ dW2, float* __restrict__ db2, int H,
                                                                int IN) {
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= PSZ) return;
  float s = 0.f;
  for (int b = lane; b < nblocks; b += 64) s += partials[(size_t)b * PSZ + e];
  s = wave_sum_all(s);
  if (lane != 0) return;
  if (e < H * IN) dW1[e] = s;
  else if (e < H * IN + H) db1[e - H * IN] = s;
  else if (e < H * IN + 2 * H) dW2[e - H * IN - H] = s;
  else db2[0] = s;
}

size_t au(size_t v) { return (v + 255) / 256 * 256; }

int grad_blocks(int64_t B) {
  const long long want = (B + WPB - 1) / WPB;
  return (int)(want < 1024 ? (want > 0 ? want : 1) : 1024);
}

struct Scratch {
  float *rows, *pair_w, *partials;
  uint32_t *keys, *src_row;
  void* plan;
  size_t total;
};

Scratch carve(void* base, int64_t B, int nnk, int F, int H, int64_t table_rows) {
  Scratch s;
  const size_t n = (size_t)(B > 0 ? B : 1) * nnk;
  char* p = reinterpret_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t bytes) { char* r = p ? p + off : nullptr; off = au(off + bytes); return r; };
  s.rows = (float*)take(n * F * sizeof(float));
  s.pair_w = (float*)take(n * 4);
  s.keys = (uint32_t*)take(n * 4);
  s.src_row = (uint32_t*)take(n * 4);
  s.partials = (float*)take((size_t)grad_blocks(B) * ((size_t)H * (F + 5) + 1) * sizeof(float));
  s.plan = take(pings_rows::carve(nullptr, (int64_t)n, table_rows).total);
  s.total = off;
  return s;
}

int run(bool second, const pings_sdf_decoder* dec, const float* features, int64_t feature_rows, const float* points,
        const float* orientations, const float* global_points, int32_t after_pgo, const float* queries, int64_t B,
        int nn_k, const int64_t* idx, const int64_t* gidx, const float* w, const float* upstream, void* scratch,
        float* d_features, float* dW1, float* db1, float* dW2, float* db2, void* stream) {
  PINGS_ARG_CHECK(dec && dec->W1 && dec->b1 && dec->W2 && dec->b2, "null decoder");
  PINGS_ARG_CHECK(dec->hidden > 0 && dec->hidden <= 64, "hidden must be in 1..64");
  PINGS_ARG_CHECK(dec->feat_dim > 0 && dec->feat_dim <= 61, "feature dim must be <= 61");
  PINGS_ARG_CHECK(nn_k > 0 && nn_k <= MAX_NNK, "nn_k must be in 1..16");
  PINGS_ARG_CHECK(feature_rows > 0 && feature_rows < 0x7FFFFFF0LL, "feature_rows out of range");
  PINGS_ARG_CHECK(scratch && d_features && dW1 && db1 && dW2 && db2, "null output");
  PINGS_ARG_CHECK(!after_pgo || orientations, "after_pgo needs orientations");
  PINGS_ARG_CHECK(B == 0 || (features && points && queries && idx && w && upstream), "null pointer");
  PINGS_ARG_CHECK(!second || B == 0 || (gidx && global_points), "the second-order pass needs gidx and the global points");
  PINGS_ARG_CHECK((int64_t)B * nn_k < 0x7FFFFFF0LL, "too many (query, neighbour) pairs");
  hipStream_t st = pings::as_stream(stream);
  const int F = dec->feat_dim, H = dec->hidden, IN = F + 3;
  Scratch s = carve(scratch, B, nn_k, F, H, feature_rows);
  // one resident round (knn_common.hpp grid_for); grad_blocks(B) is what the scratch was sized for
  const bool mfma = grad_mfma_supported(second, dec, nn_k, features, s.rows);
  const int need = F + 4;
  const void* kfn = nullptr;
  if (mfma) {
    kfn = need <= 12 ? (const void*)sdf_grad_mfma_kernel<12>
                     : (need <= 20 ? (const void*)sdf_grad_mfma_kernel<20> : (const void*)sdf_grad_mfma_kernel<36>);
  } else if (IN <= 12) kfn = second ? (const void*)sdf_grad_kernel<12, true> : (const void*)sdf_grad_kernel<12, false>;
  else if (IN <= 20) kfn = second ? (const void*)sdf_grad_kernel<20, true> : (const void*)sdf_grad_kernel<20, false>;
  else if (IN <= 36) kfn = second ? (const void*)sdf_grad_kernel<36, true> : (const void*)sdf_grad_kernel<36, false>;
  else kfn = second ? (const void*)sdf_grad_kernel<64, true> : (const void*)sdf_grad_kernel<64, false>;
  const long long work = mfma ? (B + GQ - 1) / GQ : B;   // wave-steps: four queries each on the matrix-core kernel
  int nblocks = std::max(1, std::min(grad_blocks(B), (int)grid_for(work > 0 ? work : 1, kfn)));
  if (const char* e = getenv("PINGS_SDF_GRAD_BLOCKS")) nblocks = std::max(1, std::min(nblocks, atoi(e)));   // A/B runs
  const int PSZ = H * (IN + 2) + 1;
  GradArgs a;
  a.W1 = dec->W1; a.b1 = dec->b1; a.W2 = dec->W2;
  a.H = H; a.F = F; a.weighted_first = dec->weighted_first; a.scale = dec->sdf_scale;
  a.features = features; a.points = points; a.orientations = orientations; a.gpoints = global_points;
  a.after_pgo = after_pgo; a.queries = queries; a.B = B; a.nnk = nn_k;
  a.idx = (const long long*)idx; a.gidx = (const long long*)gidx; a.w = w;
  a.dL_dsdf = second ? nullptr : upstream; a.vdir = second ? upstream : nullptr;
  a.rows = s.rows; a.keys = s.keys; a.src_row = s.src_row; a.pair_w = s.pair_w;
  a.table_rows = feature_rows; a.partials = s.partials;
  {
    pings::prof::Scope ps(second ? "sdf_bwd2_grad" : "sdf_bwd_grad", st);
    if (B == 0) PINGS_HIP_CHECK(hipMemsetAsync(s.partials, 0, sizeof(float) * (size_t)nblocks * PSZ, st));
#define PINGS_SDF_GRAD(PAD)                                                                                   \
  do {                                                                                                       \
    if (second) hipLaunchKernelGGL((sdf_grad_kernel<PAD, true>), dim3(nblocks), dim3(64 * WPB), 0, st, a);    \
    else hipLaunchKernelGGL((sdf_grad_kernel<PAD, false>), dim3(nblocks), dim3(64 * WPB), 0, st, a);         \
  } while (0)
    if (B > 0 && mfma) {
      if (need <= 12) hipLaunchKernelGGL(sdf_grad_mfma_kernel<12>, dim3(nblocks), dim3(64 * WPB), 0, st, a);
      else if (need <= 20) hipLaunchKernelGGL(sdf_grad_mfma_kernel<20>, dim3(nblocks), dim3(64 * WPB), 0, st, a);
      else hipLaunchKernelGGL(sdf_grad_mfma_kernel<36>, dim3(nblocks), dim3(64 * WPB), 0, st, a);
      PINGS_LAUNCH_CHECK();
    } else if (B > 0) {
      if (IN <= 12) PINGS_SDF_GRAD(12);
      else if (IN <= 20) PINGS_SDF_GRAD(20);
      else if (IN <= 36) PINGS_SDF_GRAD(36);
      else PINGS_SDF_GRAD(64);
      PINGS_LAUNCH_CHECK();
    }
#undef PINGS_SDF_GRAD
    hipLaunchKernelGGL(sdf_param_reduce_kernel, dim3((PSZ + 3) / 4), dim3(256), 0, st, s.partials, nblocks, PSZ,
                       dW1, db1, dW2, db2, H, IN);
    PINGS_LAUNCH_CHECK();
  }
  // scatter-add of the feature rows into the table; every row of d_features is written
  pings::prof::Scope ps("sdf_bwd_scatter", st);
  const int64_t n = (int64_t)B * nn_k;
  pings_rows::Plan plan = pings_rows::carve(s.plan, n, feature_rows);
  if (int e = pings_rows::build(plan, s.keys, n, feature_rows, st)) return e;
  const bool wf = dec->weighted_first != 0;
  return pings_rows::gather_sum(plan, feature_rows, F, s.rows, F, wf ? s.src_row : nullptr, wf ? s.pair_w : nullptr,
                                d_features, st);
}

}  // namespace

PINGS_API size_t pings_sdf_backward_scratch_bytes(int64_t B, int nn_k, int feat_dim, int hidden,
                                                  int64_t feature_rows) {
  if (nn_k <= 0 || feat_dim <= 0 || hidden <= 0 || feature_rows <= 0) return 0;
  return carve(nullptr, B, nn_k, feat_dim, hidden, feature_rows).total;
}

PINGS_API int pings_sdf_backward(const pings_sdf_decoder* dec, const float* features,
                                 int64_t feature_rows, const float* points,
                                 const float* orientations, int32_t after_pgo, const float* queries,
                                 int64_t B, int nn_k, const int64_t* idx, const float* w,
                                 const float* dL_dsdf, void* scratch, float* dL_dfeatures,
                                 float* dL_dW1, float* dL_db1, float* dL_dW2, float* dL_db2,
                                 void* stream) {
  return run(false, dec, features, feature_rows, points, orientations, nullptr, after_pgo, queries, B, nn_k, idx,
             nullptr, w, dL_dsdf, scratch, dL_dfeatures, dL_dW1, dL_db1, dL_dW2, dL_db2, stream);
}

PINGS_API int pings_sdf_double_backward(const pings_sdf_decoder* dec, const float* features,
                                        int64_t feature_rows, const float* points, const float* orientations,
                                        const float* global_points, int32_t after_pgo, const float* queries,
                                        int64_t B, int nn_k, const int64_t* idx, const int64_t* gidx,
                                        const float* w, const float* v, void* scratch, float* d_features,
                                        float* d_W1, float* d_b1, float* d_W2, float* d_b2, void* stream) {
  return run(true, dec, features, feature_rows, points, orientations, global_points, after_pgo, queries, B, nn_k, idx,
             gidx, w, v, scratch, d_features, d_W1, d_b1, d_W2, d_b2, stream);
}
