// Fused SDF query, decoder on the matrix cores: the variant of sdf_forward_kernel (knn_sdf.hip) that pings_sdf_forward
// launches for the per-neighbour decoder with nn_k <= 8 (every shipped config; weighted_first and larger nn_k stay on
// the vector kernel).
//
// Why.  With the cell-block index the search is 0.10 of the vector kernel's 0.26 ms (1M points, B = 131,072); the rest
// is the decoder: lane = hidden unit, every lane reading every input row from LDS (216 floats per lane and query: the
// LDS return path alone is ~120 us of the launch) and 288 dependent-chain FMAs per query on the vector ALU while the
// matrix pipe idles.  Here a wave takes FOUR queries at a time, i.e. 4 x 8 = 32 (query, neighbour) columns, and the
// hidden layer is one 64 x IN_PAD by IN_PAD x 32 product on v_mfma_f32_32x32x2_f32 (bitwise an fp32 fma chain: the
// 1e-4 parity bound holds without mixed precision):
//   A = W1 augmented with b1 as one more input column (the matching input is the constant 1), laid out once per
//       workgroup in LDS in operand order (lane l: rows l%32 (+32) at k = 2t + l/32) and read 16 bytes at a time — in
//       registers it would cost IN_PAD of the 128 a wave has at four waves per SIMD;
//   B = the 32 input rows [feature row of the neighbour | R^T(x - p) | 1], staged once through LDS in the operand's
//       k order (each lane reads only its own row half: 18 floats per 4 queries instead of 216 per query);
//   D = the pre-activations, column (query, neighbour) on the lane and hidden units in the 16 registers: the output
//       layer and the three direction-input gradient sums are 4-5 vector ops per register, then one half-swap add.
// Everything after that — IDW weights, the weighted prediction, its spread, the analytic gradient through the direction
// input and through the weights — is lane-parallel over the 32 columns with 8-lane group sums (3 DPP adds).
#include "knn_common.hpp"

namespace {
using namespace pings_knn;

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int QPW = 4;          // queries per wave step
constexpr int NBR = 8;          // neighbour slots per query (nn_k <= 8)
constexpr int COLS = QPW * NBR; // 32 MFMA columns
#ifndef PINGS_SDF_SEARCH_GROUP
#define PINGS_SDF_SEARCH_GROUP 2
#endif
constexpr int SQ = PINGS_SDF_SEARCH_GROUP;   // queries whose candidate loads are issued together (1, 2 or 4): measured
                                            // at B = 131,072 0.201 / 0.189 / 0.205 ms (4 spills 38 registers); the same pairing
                                            // in qf_forward_kernel, which writes 140-B rows per neighbour, lost (0.233 -> 0.288)

template <int IN_PAD>
struct XLayout {
  static constexpr int HALF = IN_PAD / 2;                 // k-steps: element 2t + k lives at [k][t]
  static constexpr int HALF_PAD = (HALF + 3) & ~3;        // 16-byte reads
  static constexpr int ROW = 2 * HALF_PAD + 4;            // +4: rows of an 8-lane group land on distinct banks
};

template <int IN_PAD, bool GRAD>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK, IN_PAD > 36 ? 2 : 4) void sdf_forward_mfma_kernel(
    pings_knn_map m, pings_sdf_decoder dec, const float* __restrict__ features,
    const float* __restrict__ points, const float* __restrict__ orientations,
    const float* __restrict__ certainties, int after_pgo, const float* __restrict__ queries,
    long long B, float* __restrict__ sdf_out, float* __restrict__ grad_out,
    long long* __restrict__ cnt_out, float* __restrict__ cert_out, long long* __restrict__ idx_out,
    float* __restrict__ w_out, float* __restrict__ std_out, long long* __restrict__ gidx_out) {
  using XL = XLayout<IN_PAD>;
  constexpr int HALF = XL::HALF, ROW = XL::ROW, HALF_PAD = XL::HALF_PAD;
  __shared__ long long sIdx[WAVES_PER_BLOCK][COLS];
  __shared__ long long sGIdx[WAVES_PER_BLOCK][COLS];
  __shared__ float sD2[WAVES_PER_BLOCK][COLS];
  __shared__ float sPos[WAVES_PER_BLOCK][COLS * 3];
  __shared__ int sCnt[WAVES_PER_BLOCK][QPW];
  __shared__ __attribute__((aligned(16))) float sX[WAVES_PER_BLOCK][COLS * ROW];
  __shared__ __attribute__((aligned(16))) float sC[64][4];   // {w2, w2 w1n0, w2 w1n1, w2 w1n2} of every hidden unit
  __shared__ __attribute__((aligned(16))) float sA[2][HALF_PAD / 4][64][4];   // A operand: [block][t / 4][lane][t % 4]
  static_assert(WAVES_PER_BLOCK * COLS * ROW >= 64 * IN_PAD, "the weight staging area must hold W1");

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int F = dec.feat_dim, IN = F + 3, Hd = dec.hidden, nnk = m.nn_k;
  const int r32 = lane & 31, kh = lane >> 5;

  // ---- A operand: W1 | b1, row r32 (+32) at k = 2t + kh, staged with coalesced global reads, then re-laid
  {
    float* stage = &sX[0][0];
    const int stride = IN | 1;   // odd: conflict-free row reads; stride <= IN_PAD because IN < IN_PAD
    for (int h = wave; h < Hd; h += WAVES_PER_BLOCK)
      for (int c = lane; c < IN; c += 64) stage[h * stride + c] = dec.W1[h * IN + c];
    for (int h = threadIdx.x; h < 64; h += 64 * WAVES_PER_BLOCK) {
      const bool on = h < Hd;
      const float w2 = on ? dec.W2[h] : 0.f;
      sC[h][0] = w2;
      sC[h][1] = on ? w2 * dec.W1[h * IN + F] : 0.f;
      sC[h][2] = on ? w2 * dec.W1[h * IN + F + 1] : 0.f;
      sC[h][3] = on ? w2 * dec.W1[h * IN + F + 2] : 0.f;
    }
    __syncthreads();
    if (wave < 2) {   // wave mb lays out block mb
      const int row = wave * 32 + r32;
      for (int t = 0; t < HALF_PAD; ++t) {
        const int c = 2 * t + kh;
        float v = 0.f;
        if (row < Hd && t < HALF) v = c < IN ? stage[row * stride + c] : (c == IN ? dec.b1[row] : 0.f);
        sA[wave][t >> 2][lane][t & 3] = v;
      }
    }
    __syncthreads();
  }
  const float b2 = dec.b2[0], scale = dec.sdf_scale;
  const LaneCtx lc = make_lane_ctx(m, lane);
  const bool two_blocks = Hd > 32;

  // feature gather: column = e >> f4_shift, 16-byte piece = e & (F4 - 1), e = lane + 64 i
  const int F4 = F >> 2;
  const int f4_shift = 31 - __clz(F4);
  const int n_gather = (COLS * F4 + 63) >> 6;

  float* X = sX[wave];
  const int j_of = r32 >> 3, mm_of = r32 & 7;   // this lane's (query of the step, neighbour slot) as a column owner
  const long long ngroups = (B + QPW - 1) / QPW;
  const long long nwaves = (long long)gridDim.x * WAVES_PER_BLOCK;
  for (long long g = (long long)blockIdx.x * WAVES_PER_BLOCK + wave; g < ngroups; g += nwaves) {
    // ---- the four searches, SQ at a time (knn_common.hpp: candidates_blocks_multi)
#pragma unroll 1
    for (int j = 0; j < QPW; j += SQ) {
      const long long q0 = g * QPW + j;
      if (lc.use_blocks && q0 + SQ - 1 < B) {
        float qq[SQ][3];
#pragma unroll
        for (int u = 0; u < SQ; ++u) {
          qq[u][0] = queries[3 * (q0 + u)]; qq[u][1] = queries[3 * (q0 + u) + 1]; qq[u][2] = queries[3 * (q0 + u) + 2];
        }
        Cand32 c[SQ];
        candidates_blocks_multi<SQ>(m, lc, qq, c);
        if (SQ == 2) {
          int cnt0, cnt1;
          Cand32 (&c2)[2] = reinterpret_cast<Cand32 (&)[2]>(c);
          select_topk32_pair(m, c2, lane, sIdx[wave] + NBR * j, sD2[wave] + NBR * j, sGIdx[wave] + NBR * j,
                             sPos[wave] + 3 * NBR * j, sIdx[wave] + NBR * (j + 1), sD2[wave] + NBR * (j + 1),
                             sGIdx[wave] + NBR * (j + 1), sPos[wave] + 3 * NBR * (j + 1), cnt0, cnt1);
          if (lane == 0) { sCnt[wave][j] = cnt0; sCnt[wave][j + 1] = cnt1; }
          continue;
        }
#pragma unroll
        for (int u = 0; u < SQ; ++u) {
          const int count = select_topk32(m, c[u], lane, sIdx[wave] + NBR * (j + u), sD2[wave] + NBR * (j + u),
                                          sGIdx[wave] + NBR * (j + u), sPos[wave] + 3 * NBR * (j + u));
          if (lane == 0) sCnt[wave][j + u] = count;
        }
        continue;
      }
#pragma unroll 1
      for (int u = 0; u < SQ; ++u) {
        const long long q = q0 + u;
        if (q < B) {
          const float qx = queries[3 * q], qy = queries[3 * q + 1], qz = queries[3 * q + 2];
          const int count = knn_one_query(m, lc, qx, qy, qz, lane, sIdx[wave] + NBR * (j + u), sD2[wave] + NBR * (j + u),
                                          sGIdx[wave] + NBR * (j + u), sPos[wave] + 3 * NBR * (j + u));
          if (lane == 0) sCnt[wave][j + u] = count;
        } else if (lane < NBR) {
          sIdx[wave][NBR * (j + u) + lane] = -1;
          sD2[wave][NBR * (j + u) + lane] = INVALID_D2;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();

    // ---- column owner (lanes 0..31; lanes 32..63 mirror them so that the group sums below are uniform code)
    const long long q_mine = g * QPW + j_of;
    const bool slot_on = mm_of < nnk && q_mine < B;
    long long my_idx = -1;
    float d2 = INVALID_D2, u = 0.f;
    if (slot_on) {
      my_idx = sIdx[wave][r32];
      d2 = sD2[wave][r32];
      if (my_idx >= 0) u = 1.0f / (d2 + 1e-15f);
    }
    float qx = 0.f, qy = 0.f, qz = 0.f, p0 = 0.f, p1 = 0.f, p2 = 0.f, cval = 0.f;
    if (q_mine < B) { qx = queries[3 * q_mine]; qy = queries[3 * q_mine + 1]; qz = queries[3 * q_mine + 2]; }
    if (my_idx >= 0) {
      p0 = points[3 * my_idx]; p1 = points[3 * my_idx + 1]; p2 = points[3 * my_idx + 2];
      if (certainties) cval = certainties[my_idx];
    }
    // feature rows, 16 bytes per lane and round, stored in the operand's k order
    for (int i = 0; i < n_gather; ++i) {
      const int e = lane + 64 * i;
      if (e < COLS * F4) {
        const int col = e >> f4_shift, c4 = e & (F4 - 1);
        const long long id = ((col & 7) < nnk) ? sIdx[wave][col] : -1;
        const float4 v = id >= 0 ? reinterpret_cast<const float4*>(features + id * F)[c4]
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
        // elements 4 c4 .. 4 c4 + 3 = (t, k) = (2 c4, 0), (2 c4, 1), (2 c4 + 1, 0), (2 c4 + 1, 1)
        *reinterpret_cast<float2*>(&X[col * ROW + 2 * c4]) = make_float2(v.x, v.z);
        *reinterpret_cast<float2*>(&X[col * ROW + HALF_PAD + 2 * c4]) = make_float2(v.y, v.w);
      }
    }
    const float U = group8_sum(u);
    const float wgt = my_idx >= 0 ? u / U : 0.f;
    float vx = 0.f, vy = 0.f, vz = 0.f, ex = 0.f, ey = 0.f, ez = 0.f;
    if (my_idx >= 0) {
      vx = qx - p0; vy = qy - p1; vz = qz - p2;
      ex = qx - sPos[wave][3 * r32]; ey = qy - sPos[wave][3 * r32 + 1]; ez = qz - sPos[wave][3 * r32 + 2];
    }
    float nx = vx, ny = vy, nz = vz;
    if (after_pgo && my_idx >= 0) rot_passive(orientations + 4 * my_idx, vx, vy, vz, nx, ny, nz);
    if (lane < COLS) {
      // direction input at columns F, F+1, F+2, the constant 1 (bias column) at F+3 = IN, zeros behind it
      float* row = &X[r32 * ROW];
      const float tail[4] = {nx, ny, nz, 1.f};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int col = F + c;
        row[(col & 1) * HALF_PAD + (col >> 1)] = tail[c];
      }
      for (int col = IN + 1; col < IN_PAD; ++col) row[(col & 1) * HALF_PAD + (col >> 1)] = 0.f;
      if (slot_on) {
        if (idx_out) idx_out[q_mine * nnk + mm_of] = my_idx;
        if (gidx_out) gidx_out[q_mine * nnk + mm_of] = my_idx >= 0 ? sGIdx[wave][r32] : -1;
        if (w_out) w_out[q_mine * nnk + mm_of] = wgt;
      }
    }
    __builtin_amdgcn_wave_barrier();

    // ---- B operand of this lane: its column r32, k half kh
    float bx[HALF_PAD];
#pragma unroll
    for (int t = 0; t < HALF_PAD; t += 4) {
      const float4 v = *reinterpret_cast<const float4*>(&X[r32 * ROW + kh * HALF_PAD + t]);
      bx[t] = v.x; bx[t + 1] = v.y; bx[t + 2] = v.z; bx[t + 3] = v.w;
    }
    // ---- hidden layer on the matrix cores, output layer on the fly
    float hv = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f;
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
#pragma unroll
      for (int rq = 0; rq < 16; ++rq) {
        const float4 c4 = *reinterpret_cast<const float4*>(sC[mb * 32 + rowmap32(rq, kh)]);
        const float pre = acc[rq];
        hv = fmaf(c4.x, fmaxf(pre, 0.f), hv);
        if (GRAD) {
          const bool pos = pre > 0.f;
          g0 += pos ? c4.y : 0.f; g1 += pos ? c4.z : 0.f; g2 += pos ? c4.w : 0.f;
        }
      }
    }
    // the two lane halves hold the two halves of the hidden units of the same column
    {
      float a = hv, b = hv;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
      hv = a + b;
      if (GRAD) {
        a = g0; b = g0; asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); g0 = a + b;
        a = g1; b = g1; asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); g1 = a + b;
        a = g2; b = g2; asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); g2 = a + b;
      }
    }
    // ---- per-column prediction, IDW of the predictions (mapper.py:2279), spread, analytic gradient
    const float s_m = scale * (b2 + hv);
    const float S = group8_sum(wgt * s_m);
    const float cs = cert_out ? group8_sum(cval * wgt) : 0.f;
    float var = 0.f;
    if (std_out) {
      const float d = s_m - S;
      var = group8_sum(wgt * (d * d));
    }
    float gx = 0.f, gy = 0.f, gz = 0.f;
    if (GRAD) {
      float a0 = g0, a1 = g1, a2 = g2;
      if (after_pgo && my_idx >= 0) rot_active(orientations + 4 * my_idx, g0, g1, g2, a0, a1, a2);
      float tx = 0.f, ty = 0.f, tz = 0.f;
      if (my_idx >= 0) {
        const float kd = wgt * scale;                    // through the direction input
        const float kw = (s_m - S) * (-2.f * u * u) / U; // through the weights
        tx = fmaf(kw, ex, kd * a0); ty = fmaf(kw, ey, kd * a1); tz = fmaf(kw, ez, kd * a2);
      }
      gx = group8_sum(tx); gy = group8_sum(ty); gz = group8_sum(tz);
    }
    if (lane < COLS && mm_of == 0 && q_mine < B) {
      sdf_out[q_mine] = S;
      if (GRAD) { grad_out[3 * q_mine] = gx; grad_out[3 * q_mine + 1] = gy; grad_out[3 * q_mine + 2] = gz; }
      if (cnt_out) cnt_out[q_mine] = sCnt[wave][j_of];
      if (cert_out) cert_out[q_mine] = cs;
      if (std_out) std_out[q_mine] = sqrtf(var);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

namespace pings_knn {

bool sdf_forward_mfma_supported(const pings_knn_map* m, const pings_sdf_decoder* dec, const float* features) {
  const int F = dec->feat_dim;
  const int F4 = F >> 2;
  return !dec->weighted_first && m->nn_k <= NBR && (F & 3) == 0 && F4 > 0 && (F4 & (F4 - 1)) == 0 &&
         F + 4 <= 64 && dec->hidden <= 64 && ((reinterpret_cast<uintptr_t>(features) & 15u) == 0);
}

int sdf_forward_mfma_launch(const pings_knn_map* m, const pings_sdf_decoder* dec, const float* features,
                            const float* points, const float* orientations, const float* certainties,
                            int32_t after_pgo, const float* queries, int64_t B, float* sdf, float* grad_x,
                            int64_t* nn_counts, float* certainty, int64_t* idx_out, float* w_out, float* sdf_std,
                            int64_t* gidx_out, hipStream_t st) {
  const int need = dec->feat_dim + 4;   // inputs + the bias column
  const long long groups = (B + QPW - 1) / QPW;
#define PINGS_SDF_MFMA_G(PAD, G)                                                                                     \
  hipLaunchKernelGGL((sdf_forward_mfma_kernel<PAD, G>),                                                              \
                     dim3(grid_for(groups, (const void*)sdf_forward_mfma_kernel<PAD, G>)),                           \
                     dim3(64 * WAVES_PER_BLOCK), 0, st, *m, *dec, features, points, orientations, certainties,       \
                     (int)after_pgo, queries, (long long)B, sdf, grad_x, (long long*)nn_counts, certainty,           \
                     (long long*)idx_out, w_out, sdf_std, (long long*)gidx_out)
#define PINGS_SDF_MFMA(PAD) do { if (grad_x) PINGS_SDF_MFMA_G(PAD, true); else PINGS_SDF_MFMA_G(PAD, false); } while (0)
  if (need <= 12) PINGS_SDF_MFMA(12);
  else if (need <= 20) PINGS_SDF_MFMA(20);
  else if (need <= 36) PINGS_SDF_MFMA(36);
  else PINGS_SDF_MFMA(64);
#undef PINGS_SDF_MFMA
#undef PINGS_SDF_MFMA_G
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}

}  // namespace pings_knn
