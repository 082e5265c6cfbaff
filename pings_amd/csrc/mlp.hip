// Decoder MLP  y = relu(x W1^T + b1) W2^T + b2  on the gfx950 matrix cores, forward and backward.
//
// fp32 in / fp32 accumulate (v_mfma_f32_32x32x2_f32): bitwise an fp32 fma chain, so the 1e-4 parity
// bound of the reference's fp32 decoders (model/decoder.py:62-82) holds without mixed precision.
//
// Work decomposition: a workgroup has NW = HID/32 waves and walks 32-row tiles of x; wave w owns
// hidden units 32w .. 32w+31 for ALL products, so every matrix it needs from the previous product is
// already in its accumulators:
//   GEMM1 (transposed)  H^T[hid, row]  = W1[hid, :] . x[row, :]^T        A = W1 (LDS), B = x tile (LDS)
//   GEMM2               Y^T[o, row]   += W2[o, hid] . H^T[hid, row]      B = the GEMM1 accumulator itself
//                       (accumulator-as-operand: the 32x32 C tile has its column on the lane and its rows
//                        in the 16 registers, which IS the B-operand layout when k runs over the tile's
//                        rows in the order (s&3) + 8(s>>2) + 4(lane>>5); the A operand is read from LDS
//                        in that same k order)
//   backward            gH^T = (W2^T gY^T) * relu'(H^T) ;  gX^T += W1^T gH^T (B = the gH^T accumulator);
//                       gW2^T[hid,o] += H^T gY and gW1[hid,i] += gH^T x sum over ROWS, i.e. over the
//                       accumulators' lane index, so H^T and gH^T take one trip through LDS (transposed
//                       read) — the only transposes in the kernel.
// Partial Y^T / gX^T tiles of the NW waves are summed through LDS in fixed order; weight gradients are
// kept in registers across all row tiles of the workgroup, written as per-workgroup partials and summed
// by a second kernel in fixed order (bitwise reproducible, no atomics).
#include <cstdlib>

#include "common.hpp"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int TR = 32;        // rows per tile
constexpr int MAX_INP = 64;   // padded input width limit
constexpr int OUTP = 32;      // padded output width

__device__ inline int rowmap(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

__device__ inline f32x16 mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

struct Dims {
  long long N;
  int IN, INP, HID, OUT;  // INP = IN rounded up to even
  int ldw1, ldw2, ldx, ldg;  // LDS leading dimensions (odd -> conflict-free column walks)
};

__host__ __device__ inline Dims make_dims(long long N, int IN, int HID, int OUT) {
  Dims d;
  d.N = N; d.IN = IN; d.HID = HID; d.OUT = OUT;
  d.INP = (IN + 1) & ~1;
  d.ldw1 = d.INP + 1;
  d.ldw2 = HID + 1;
  d.ldx = d.INP + 1;
  d.ldg = OUTP + 1;
  return d;
}

// ---------------------------------------------------------------- forward
template <int PF_X>
__global__ __launch_bounds__(256) void mlp_fwd_kernel(Dims d, const float* __restrict__ x, const float* __restrict__ W1,
                               const float* __restrict__ b1, const float* __restrict__ W2,
                               const float* __restrict__ b2, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int NW = d.HID / 32, nthreads = NW * 64;
  float* sW1 = lds;                                  // [HID][ldw1]
  float* sW2 = sW1 + d.HID * d.ldw1;                 // [OUTP][ldw2]
  float* sX = sW2 + OUTP * d.ldw2;                   // [TR][ldx]
  float* sY = sX + TR * d.ldx;                       // [NW][OUTP][TR+1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;

  for (int e = tid; e < d.HID * d.INP; e += nthreads) {
    const int j = e / d.INP, i = e - j * d.INP;
    sW1[j * d.ldw1 + i] = i < d.IN ? W1[(size_t)j * d.IN + i] : 0.f;
  }
  for (int e = tid; e < OUTP * d.HID; e += nthreads) {
    const int o = e / d.HID, j = e - o * d.HID;
    sW2[o * d.ldw2 + j] = o < d.OUT ? W2[(size_t)o * d.HID + j] : 0.f;
  }
  float bias1[16];
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) bias1[reg] = b1[wave * 32 + rowmap(reg, h)];

  const long long ntiles = (d.N + TR - 1) / TR;
  // The next tile of x is fetched into registers while the current one is multiplied (one wave per SIMD:
  // without this the matrix pipe idles for the whole HBM round trip of every tile).
  float px[PF_X];
  auto fetch = [&](long long t) {
#pragma unroll
    for (int u = 0; u < PF_X; ++u) {
      const int e = tid + u * nthreads;
      const int rr = e / d.INP, i = e - rr * d.INP;
      const long long gr = t * TR + rr;
      px[u] = (e < TR * d.INP && gr < d.N && i < d.IN) ? x[(size_t)gr * d.IN + i] : 0.f;
    }
  };
  if ((long long)blockIdx.x < ntiles) fetch(blockIdx.x);
  for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long long row0 = t * TR;
    __syncthreads();  // previous tile's sX / sY consumed (also orders the weight staging on the first trip)
#pragma unroll
    for (int u = 0; u < PF_X; ++u) {
      const int e = tid + u * nthreads;
      if (e < TR * d.INP) {
        const int rr = e / d.INP, i = e - rr * d.INP;
        sX[rr * d.ldx + i] = px[u];
      }
    }
    __syncthreads();
    if (t + gridDim.x < ntiles) fetch(t + gridDim.x);
    f32x16 acc = {0};
    {
      // the trip count is a kernel argument, which keeps hipcc from unrolling an MFMA loop: groups of four with a
      // static inner loop let it issue the eight LDS operand reads of a group together, ahead of the four MFMAs
      const float* pa = sW1 + (wave * 32 + r) * d.ldw1 + h;
      const float* pb = sX + r * d.ldx + h;
      const int half = d.INP / 2;
      int s = 0;
      for (; s + 4 <= half; s += 4) {
        float a4[4], b4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a4[u] = pa[2 * (s + u)]; b4[u] = pb[2 * (s + u)]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = mfma(a4[u], b4[u], acc);
      }
      for (; s < half; ++s) acc = mfma(pa[2 * s], pb[2 * s], acc);
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) acc[reg] = fmaxf(acc[reg] + bias1[reg], 0.f);
    f32x16 acc2 = {0};
#pragma unroll
    for (int s = 0; s < 16; ++s)
      acc2 = mfma(sW2[r * d.ldw2 + wave * 32 + rowmap(s, h)], acc[s], acc2);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) sY[(wave * OUTP + rowmap(reg, h)) * (TR + 1) + r] = acc2[reg];
    __syncthreads();
    for (int e = tid; e < TR * d.OUT; e += nthreads) {
      const int rr = e / d.OUT, o = e - rr * d.OUT;
      const long long gr = row0 + rr;
      if (gr < d.N) {
        float v = b2[o];
        for (int w = 0; w < NW; ++w) v += sY[(w * OUTP + o) * (TR + 1) + rr];
        y[(size_t)gr * d.OUT + o] = v;
      }
    }
  }
}

// ---------------------------------------------------------------- forward, one independent wave per 32-row tile
// For the GS decoders (HID = 128, IN <= 32, OUT <= 32: every shipped config).  A wave takes a 32-row tile from x to y
// with no workgroup barrier, no LDS and no weight traffic after its prologue: both weight matrices live in its
// registers as MFMA A-operand fragments (W1: 4 hidden blocks x 17 k-steps, W2: 4 x 16), the x tile is loaded straight
// into the B-operand layout (the k order of the first product is free: lane half h takes inputs 16 h .. 16 h + 15,
// sixteen contiguous floats of its row), the bias b1 rides along as one more k-step against a constant 1, and each
// 32-unit block of the hidden layer goes accumulator -> ReLU -> B operand of the second product (accumulator-as-
// operand chaining), which accumulates Y^T over the four blocks in ONE accumulator initialised with b2: no
// cross-wave sum.  132 MFMAs per tile and wave; the next tile's x is in flight meanwhile.  Results are bitwise those
// of the workgroup kernel's fma order up to the order of the hidden-block sum, i.e. within 1e-6.
__device__ __forceinline__ void mlp_fwd_wave_body(long long N, int IN, int OUT, const float* __restrict__ x,
                                                  const float* __restrict__ W1, const float* __restrict__ b1,
                                                  const float* __restrict__ W2, const float* __restrict__ b2,
                                                  float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int r = lane & 31, h = lane >> 5;
  // ---- weight fragments
  float w1f[4][17], w2f[4][16], b2f[16];
  const bool vecw = (IN % 4 == 0) && ((reinterpret_cast<uintptr_t>(W1) & 15) == 0) &&
                    ((reinterpret_cast<uintptr_t>(W2) & 15) == 0);
#pragma unroll
  for (int hb = 0; hb < 4; ++hb) {
    const int hid = hb * 32 + r;
    if (vecw) {  // 16-byte loads: a lane's 16 inputs of a hidden unit, and its four runs of four hidden units of W2
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int k = 16 * h + 4 * q4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < IN) v = *reinterpret_cast<const float4*>(W1 + (size_t)hid * IN + k);
        w1f[hb][4 * q4] = v.x; w1f[hb][4 * q4 + 1] = v.y; w1f[hb][4 * q4 + 2] = v.z; w1f[hb][4 * q4 + 3] = v.w;
        float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < OUT) u = *reinterpret_cast<const float4*>(W2 + (size_t)r * 128 + hb * 32 + 8 * q4 + 4 * h);
        w2f[hb][4 * q4] = u.x; w2f[hb][4 * q4 + 1] = u.y; w2f[hb][4 * q4 + 2] = u.z; w2f[hb][4 * q4 + 3] = u.w;
      }
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int k = 16 * h + s;
        w1f[hb][s] = k < IN ? W1[(size_t)hid * IN + k] : 0.f;
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) w2f[hb][t] = r < OUT ? W2[(size_t)r * 128 + hb * 32 + rowmap(t, h)] : 0.f;
    }
    w1f[hb][16] = h == 0 ? b1[hid] : 0.f;            // k-step 16: (constant 1, zero) against (b1, 0)
  }
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int o = rowmap(t, h);
    b2f[t] = o < OUT ? b2[o] : 0.f;
  }
  const bool vec = (IN % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  const long long ntiles = (N + 31) / 32;
  const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
  const long long wave0 = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);

  float xf[16], xn[16];
  auto fetch = [&](long long t, float (&dst)[16]) {
    const long long row = t * 32 + r;
    const bool ok = row < N;
    if (vec) {
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int k = 16 * h + 4 * q4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok && k < IN) v = *reinterpret_cast<const float4*>(x + (size_t)row * IN + k);
        dst[4 * q4 + 0] = v.x; dst[4 * q4 + 1] = v.y; dst[4 * q4 + 2] = v.z; dst[4 * q4 + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) {
        const int k = 16 * h + s2;
        dst[s2] = (ok && k < IN) ? x[(size_t)row * IN + k] : 0.f;
      }
    }
  };
  if (wave0 < ntiles) fetch(wave0, xn);
  for (long long t = wave0; t < ntiles; t += nwaves) {
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) xf[s2] = xn[s2];
    if (t + nwaves < ntiles) fetch(t + nwaves, xn);
    const float one = h == 0 ? 1.f : 0.f;
    f32x16 yacc;
#pragma unroll
    for (int q = 0; q < 16; ++q) yacc[q] = b2f[q];
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) {
      f32x16 acc = {0};
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) acc = mfma(w1f[hb][s2], xf[s2], acc);
      acc = mfma(w1f[hb][16], one, acc);
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = fmaxf(acc[q], 0.f);
#pragma unroll
      for (int q = 0; q < 16; ++q) yacc = mfma(w2f[hb][q], acc[q], yacc);
    }
    // Y^T[o = rowmap(q, h)][row = r]: four runs of four consecutive outputs per lane
    const long long row = t * 32 + r;
    if (row < N) {
      float* dst = y + (size_t)row * OUT;
      if (OUT % 4 == 0 && ((reinterpret_cast<uintptr_t>(y) & 15) == 0)) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int o = 8 * gq + 4 * h;
          if (o < OUT)
            *reinterpret_cast<float4*>(dst + o) = make_float4(yacc[4 * gq], yacc[4 * gq + 1], yacc[4 * gq + 2], yacc[4 * gq + 3]);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int o = rowmap(q, h);
          if (o < OUT) dst[o] = yacc[q];
        }
      }
    }
  }
}

__global__ __launch_bounds__(256, 2) void mlp_fwd_wave_kernel(long long N, int IN, int OUT, const float* __restrict__ x,
                                                              const float* __restrict__ W1, const float* __restrict__ b1,
                                                              const float* __restrict__ W2, const float* __restrict__ b2,
                                                              float* __restrict__ y) {
  mlp_fwd_wave_body(N, IN, OUT, x, W1, b1, W2, b2, y);
}

// Several decoders over the same rows in ONE launch (blockIdx.y = decoder): the five spawn decoders of a view
// (gaussian_renderer/__init__.py:605-716) share their row count and, four of them, their input; one grid of
// 5 x 512 workgroups keeps every CU busy through the weight prologues and costs one launch instead of five.
constexpr int MAX_JOBS = 8;
struct MlpJobs {
  const float *x[MAX_JOBS], *W1[MAX_JOBS], *b1[MAX_JOBS], *W2[MAX_JOBS], *b2[MAX_JOBS];
  float* y[MAX_JOBS];
  const float* gy[MAX_JOBS];
  float* gx[MAX_JOBS];
  float* partials[MAX_JOBS];
  size_t per_block[MAX_JOBS];
  float *gW1[MAX_JOBS], *gb1[MAX_JOBS], *gW2[MAX_JOBS], *gb2[MAX_JOBS];
  int IN[MAX_JOBS], OUT[MAX_JOBS];
  int wg0[MAX_JOBS + 1];   // backward: job g owns workgroups [wg0[g], wg0[g + 1]) of a 1-D grid (cost-proportional shares)
};

__global__ __launch_bounds__(256, 2) void mlp_fwd_wave_grouped_kernel(long long N, MlpJobs j,
                                                                      const int* __restrict__ n_dev) {
  const int g = blockIdx.y;
  if (n_dev) N = min(N, (long long)*n_dev);   // N sized the grid and the buffers; the rows to decode are counted on the device
  mlp_fwd_wave_body(N, j.IN[g], j.OUT[g], j.x[g], j.W1[g], j.b1[g], j.W2[g], j.b2[g], j.y[g]);
}

// ---------------------------------------------------------------- forward, SDF decoder shape (HID 64, OUT 1)
// `Decoder.sdf` (model/decoder.py:100-104) on [N, F + 3] rows: the general kernels above pad OUT = 1 to a 32-wide
// second product (96 % of its MFMAs multiply zeros) and stage the tile through LDS with three barriers.  Here a wave
// owns a 32-row tile outright, no LDS, no barrier:
//   H^T[64 x 32] = [W1 | b1] [x | 1]^T   two 32x32 accumulators (hidden blocks), KS k-steps of v_mfma_f32_32x32x2_f32;
//                                        the k order is free, so lane half h takes input columns KS*h .. KS*h + KS - 1:
//                                        KS consecutive floats of the lane's own row (loaded straight into the B
//                                        operand, the next tile's row in flight meanwhile), W1 columns in the same
//                                        order as register-resident A fragments; column IN is the bias (x = 1);
//   y[row] = b2 + sum_h W2[h] relu(H^T[h][row])   on the accumulator layout (column = row of the tile on the lane,
//                                        16 hidden units per register set and lane half): 32 fma + one half swap.
// 2 * KS MFMAs per 32 rows (36 for F = 32) against 18 + 16 + 16 * 2 of the padded product.
template <int KS>
__global__ __launch_bounds__(256, 2) void mlp_fwd_h64o1_kernel(long long N, int IN, const float* __restrict__ x,
                                                               const float* __restrict__ W1, const float* __restrict__ b1,
                                                               const float* __restrict__ W2, const float* __restrict__ b2,
                                                               float* __restrict__ y) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  float w1f[2][KS], w2f[2][16];
#pragma unroll
  for (int hb = 0; hb < 2; ++hb) {
    const int hid = hb * 32 + r;                     // A operand: row m = r, k = half
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) {
      const int c = KS * h + s2;
      w1f[hb][s2] = c < IN ? W1[(size_t)hid * IN + c] : (c == IN ? b1[hid] : 0.f);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) w2f[hb][q] = W2[hb * 32 + rowmap(q, h)];
  }
  const float bias2 = b2[0];
  const long long ntiles = (N + 31) / 32;
  const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
  const long long wave0 = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  float xf[KS], xn[KS];
  auto fetch = [&](long long t, float (&dst)[KS]) {
    const long long row = t * 32 + r;
    const bool ok = row < N;
    const float* src = x + (size_t)(ok ? row : 0) * IN;
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) {
      const int c = KS * h + s2;
      dst[s2] = c < IN ? (ok ? src[c] : 0.f) : (c == IN ? 1.f : 0.f);
    }
  };
  if (wave0 < ntiles) fetch(wave0, xn);
  for (long long t = wave0; t < ntiles; t += nwaves) {
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) xf[s2] = xn[s2];
    if (t + nwaves < ntiles) fetch(t + nwaves, xn);
    f32x16 a0 = {0}, a1 = {0};
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) {               // two independent accumulator chains, interleaved
      a0 = mfma(w1f[0][s2], xf[s2], a0);
      a1 = mfma(w1f[1][s2], xf[s2], a1);
    }
    float p = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) p = fmaf(w2f[0][q], fmaxf(a0[q], 0.f), p);
#pragma unroll
    for (int q = 0; q < 16; ++q) p = fmaf(w2f[1][q], fmaxf(a1[q], 0.f), p);
    p += __shfl_xor(p, 32, 64);                     // the other 32 hidden units of this row
    const long long row = t * 32 + r;
    if (h == 0 && row < N) y[row] = p + bias2;
  }
}

// ---------------------------------------------------------------- backward
// scratch layout per workgroup: [HID*IN] gW1, [OUT*HID] gW2, [HID] gb1, [OUT] gb2
__host__ __device__ inline size_t partial_floats(int IN, int HID, int OUT) {
  return (size_t)HID * IN + (size_t)OUT * HID + HID + OUT;
}

template <int PF_X, int PF_G>
__global__ __launch_bounds__(256) void mlp_bwd_kernel(Dims d, const float* __restrict__ x, const float* __restrict__ gy,
                               const float* __restrict__ W1, const float* __restrict__ b1,
                               const float* __restrict__ W2, float* __restrict__ gx,
                               float* __restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int NW = d.HID / 32, nthreads = NW * 64;
  const int NIB = (d.INP + 31) / 32;                 // 32-wide blocks of the input dimension (1 or 2)
  float* sW1 = lds;                                  // [HID][ldw1]
  float* sW2 = sW1 + d.HID * d.ldw1;                 // [OUTP][ldw2]
  float* sX = sW2 + OUTP * d.ldw2;                   // [TR][ldx]
  float* sGY = sX + TR * d.ldx;                      // [TR][ldg]
  float* sHT = sGY + TR * d.ldg;                     // [NW][32 hid][TR+1]
  float* sGH = sHT + NW * 32 * (TR + 1);             // [NW][32 hid][TR+1]
  float* sGX = sGH + NW * 32 * (TR + 1);             // [NW][NIB*32][TR+1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;

  for (int e = tid; e < d.HID * d.INP; e += nthreads) {
    const int j = e / d.INP, i = e - j * d.INP;
    sW1[j * d.ldw1 + i] = i < d.IN ? W1[(size_t)j * d.IN + i] : 0.f;
  }
  for (int e = tid; e < OUTP * d.HID; e += nthreads) {
    const int o = e / d.HID, j = e - o * d.HID;
    sW2[o * d.ldw2 + j] = o < d.OUT ? W2[(size_t)o * d.HID + j] : 0.f;
  }
  float bias1[16];
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) bias1[reg] = b1[wave * 32 + rowmap(reg, h)];

  // persistent accumulators of this wave's 32 hidden units
  f32x16 aW2T = {0};          // gW2^T tile: [hid (rows, reg map)] x [o (lane)]
  f32x16 aW1a = {0}, aW1b = {0};  // gW1 tiles: [hid] x [i 0..31], [hid] x [i 32..63]
  f32x16 aB1 = {0};           // per-lane (row) partial of gb1 for hid = rowmap(reg, h)
  float aB2 = 0.f;            // thread o < OUT of wave 0: column sum of gY

  float* myHT = sHT + wave * 32 * (TR + 1);
  float* myGH = sGH + wave * 32 * (TR + 1);
  float* myGX = sGX + wave * NIB * 32 * (TR + 1);

  const long long ntiles = (d.N + TR - 1) / TR;
  // next tile of x / gY prefetched into registers during the current tile's products
  float px[PF_X], pg[PF_G];
  auto fetch = [&](long long t) {
#pragma unroll
    for (int u = 0; u < PF_X; ++u) {
      const int e = tid + u * nthreads;
      const int rr = e / d.INP, i = e - rr * d.INP;
      const long long gr = t * TR + rr;
      px[u] = (e < TR * d.INP && gr < d.N && i < d.IN) ? x[(size_t)gr * d.IN + i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < PF_G; ++u) {
      const int e = tid + u * nthreads;
      const int rr = e / OUTP, o = e - rr * OUTP;
      const long long gr = t * TR + rr;
      pg[u] = (e < TR * OUTP && gr < d.N && o < d.OUT) ? gy[(size_t)gr * d.OUT + o] : 0.f;
    }
  };
  if ((long long)blockIdx.x < ntiles) fetch(blockIdx.x);
  for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long long row0 = t * TR;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PF_X; ++u) {
      const int e = tid + u * nthreads;
      if (e < TR * d.INP) {
        const int rr = e / d.INP, i = e - rr * d.INP;
        sX[rr * d.ldx + i] = px[u];
      }
    }
#pragma unroll
    for (int u = 0; u < PF_G; ++u) {
      const int e = tid + u * nthreads;
      if (e < TR * OUTP) {
        const int rr = e / OUTP, o = e - rr * OUTP;
        sGY[rr * d.ldg + o] = pg[u];
      }
    }
    __syncthreads();
    if (t + gridDim.x < ntiles) fetch(t + gridDim.x);
    if (wave == 0 && lane < d.OUT) {
      float s = 0.f;
      for (int rr = 0; rr < TR; ++rr) s += sGY[rr * d.ldg + lane];
      aB2 += s;
    }
    // H^T (hidden units of this wave) for the tile's rows
    f32x16 hT = {0};
    {
      const float* pa = sW1 + (wave * 32 + r) * d.ldw1 + h;
      const float* pb = sX + r * d.ldx + h;
      const int half = d.INP / 2;
      int s = 0;
      for (; s + 4 <= half; s += 4) {
        float a4[4], b4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a4[u] = pa[2 * (s + u)]; b4[u] = pb[2 * (s + u)]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) hT = mfma(a4[u], b4[u], hT);
      }
      for (; s < half; ++s) hT = mfma(pa[2 * s], pb[2 * s], hT);
    }
    // gH^T = W2^T gY^T  (A = W2^T: row = hid on the lane, k = o)
    f32x16 gT = {0};
#pragma unroll
    for (int s = 0; s < OUTP / 2; ++s)
      gT = mfma(sW2[(2 * s + h) * d.ldw2 + wave * 32 + r], sGY[r * d.ldg + 2 * s + h], gT);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const float pre = hT[reg] + bias1[reg];
      hT[reg] = fmaxf(pre, 0.f);
      gT[reg] = pre > 0.f ? gT[reg] : 0.f;
      aB1[reg] += gT[reg];
    }
    // gX^T partial of this wave: [i, row] = sum_{hid in wave} W1[hid][i] gH^T[hid][row]
    //   A = W1^T: row = i on the lane, k = hid in accumulator order; B = the gH^T accumulator
    if (gx) {
      for (int ib = 0; ib < NIB; ++ib) {
        f32x16 ax = {0};
        const int i = ib * 32 + r;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const float a = i < d.INP ? sW1[(wave * 32 + rowmap(s, h)) * d.ldw1 + i] : 0.f;
          ax = mfma(a, gT[s], ax);
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) myGX[(ib * 32 + rowmap(reg, h)) * (TR + 1) + r] = ax[reg];
      }
    }
    // transpose H^T / gH^T through LDS: element [hid (reg map)][row (lane)]
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      myHT[rowmap(reg, h) * (TR + 1) + r] = hT[reg];
      myGH[rowmap(reg, h) * (TR + 1) + r] = gT[reg];
    }
    __builtin_amdgcn_wave_barrier();
    // gW2^T[hid][o] += sum_row H^T[hid][row] gY[row][o]   (A: row = hid on the lane, k = data row)
    // gW1[hid][i]   += sum_row gH^T[hid][row] x[row][i]
#pragma unroll
    for (int s = 0; s < TR / 2; ++s) {
      const int k = 2 * s + h;
      const float aH = myHT[r * (TR + 1) + k];
      const float aG = myGH[r * (TR + 1) + k];
      aW2T = mfma(aH, sGY[k * d.ldg + r], aW2T);
      aW1a = mfma(aG, r < d.INP ? sX[k * d.ldx + r] : 0.f, aW1a);
      if (NIB > 1) aW1b = mfma(aG, (32 + r) < d.INP ? sX[k * d.ldx + 32 + r] : 0.f, aW1b);
    }
    __syncthreads();
    if (gx) {
      for (int e = tid; e < TR * d.IN; e += nthreads) {
        const int rr = e / d.IN, i = e - rr * d.IN;
        const long long gr = row0 + rr;
        if (gr < d.N) {
          float v = 0.f;
          for (int w = 0; w < NW; ++w) v += sGX[(w * NIB * 32 + i) * (TR + 1) + rr];
          gx[(size_t)gr * d.IN + i] = v;
        }
      }
    }
  }

  // ---- write this workgroup's partial weight gradients
  float* P = partials + (size_t)blockIdx.x * partial_floats(d.IN, d.HID, d.OUT);
  float* pW1 = P;
  float* pW2 = pW1 + (size_t)d.HID * d.IN;
  float* pB1 = pW2 + (size_t)d.OUT * d.HID;
  float* pB2 = pB1 + d.HID;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int hid = wave * 32 + rowmap(reg, h);
    if (r < d.OUT) pW2[(size_t)r * d.HID + hid] = aW2T[reg];           // lane = o
    if (r < d.IN) pW1[(size_t)hid * d.IN + r] = aW1a[reg];             // lane = i
    if (NIB > 1 && 32 + r < d.IN) pW1[(size_t)hid * d.IN + 32 + r] = aW1b[reg];
  }
  // gb1: sum the per-row partials over the 32 lanes that share h
  __syncthreads();
  float* sRed = sHT;  // reuse: [NW][64 lanes][16]
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) sRed[(wave * 64 + lane) * 17 + reg] = aB1[reg];
  __syncthreads();
  for (int e = tid; e < d.HID; e += nthreads) {
    const int w = e >> 5, m = e & 31;          // hidden unit m of wave w: find (reg, hh) with rowmap = m
    const int hh = (m >> 2) & 1, reg = (m & 3) + 4 * (m >> 3);
    float s = 0.f;
    for (int l = 0; l < 32; ++l) s += sRed[(w * 64 + hh * 32 + l) * 17 + reg];
    pB1[e] = s;
  }
  if (wave == 0 && lane < d.OUT) pB2[lane] = aB2;
}

// ---------------------------------------------------------------- backward, one independent wave per 32-row tile
// Same idea as mlp_fwd_wave_kernel, for HID = 128, IN <= 32: a wave takes a 32-row tile through all five products for
// all four hidden blocks; nothing but the read-only weight images in LDS is shared, so the only barrier is the one
// after staging them.  Per hidden block hb (k orders are free, lane half h takes k = 16 h .. 16 h + 15):
//   A  pre^T  = W1_hb x^T (+ b1 as a 17th k-step)          A = register fragments of W1, B = x rows
//   B  gH^T   = W2_hb^T gY^T, masked by pre > 0             A = W2 image in LDS,          B = gY rows
//   C  gX^T  += W1_hb^T gH^T                                A = W1 image in LDS,          B = the gH^T accumulator
//   D  gW2^T_hb += H^T gY,  gW1_hb += gH^T x  (sums over rows = lanes of the accumulators): H^T and gH^T make one
//      trip through the wave's private LDS to become A operands, B = x / gY read column-wise from global memory;
//      gb1 falls out of the transposed gH^T fragments, gb2 of the gY columns.
// 324 MFMAs per tile.  Weight-gradient accumulators stay in registers across the wave's tiles (8 x 16 + 5 VGPRs),
// are added across the four waves in LDS (wave order) and written as one partial per workgroup, summed by
// mlp_reduce_kernel in fixed order: bitwise reproducible.
constexpr int BW_LD = 33;  // leading dimension of the private transpose tiles and of the W1 image
constexpr int BW_LDS_FLOATS = 128 * BW_LD + 32 * 129 + 4 * 2 * 32 * BW_LD + 4 * 4 * 32 * BW_LD;   // mlp_bwd_wave_dispatch
constexpr int BW_REGION = 128 * 32 + 32 * 129 + 160;   // one wave's weight-gradient region in the epilogue
static_assert(4 * BW_REGION <= BW_LDS_FLOATS, "the four epilogue regions must fit the kernel's LDS");
#ifdef PINGS_MLP_STATS
__device__ unsigned long long g_mlp_stats[8];
__device__ unsigned long long g_mlp_clock[2];   // shader-clock cycles and 100 MHz real-time ticks of one workgroup's life
#endif
// Optional scheduling barriers of mlp_bwd_wave_body (bit k of PINGS_MLP_SB = barrier k; A/B builds, tools/mlp_sb_ab.sh):
// A, C, D pin the operand reads of the NEXT product above the MFMAs of the current one, B the mask below the products
// it depends on.  Measured at 125k points, five decoders (ms): all four 0.306, A 0.310, A + C 0.307, A + C + D 0.309,
// A + B + C 0.300, NONE 0.281 — the source order (reads of the next product written ahead of the current product's
// MFMAs) is enough for the compiler's scheduler, and hard barriers only keep it from overlapping the tails: default none.
#ifndef PINGS_MLP_SB
#define PINGS_MLP_SB 0
#endif
#define MLP_SB_(k_) do { if (PINGS_MLP_SB & (1 << (k_))) __builtin_amdgcn_sched_barrier(0); } while (0)
// The wave barriers around the wave-private LDS round trips (tile views; before / after the transpose writes) are NOT
// optional: builds without the two around the transposes were no faster (0.285-0.288 ms) and failed tests/test_mlp.py —
// the compiler does move the transposed reads across the writes without the fence.
// (wave barrier = scheduling fence; the empty asm with a memory clobber states the memory ordering explicitly)
#define MLP_WB_T do { __builtin_amdgcn_wave_barrier(); __asm__ volatile("" ::: "memory"); } while (0)
#define MLP_WB_W MLP_WB_T
#define MLP_WB_R MLP_WB_T
#define MLP_SB_A MLP_SB_(0)
#define MLP_SB_B MLP_SB_(1)
#define MLP_SB_C MLP_SB_(2)
#define MLP_SB_D MLP_SB_(3)
// Scheduling GROUPS for the stretch between two transposes (PINGS_MLP_SGB = variant; 0 = none).  Left alone, the
// compiler sinks each LDS operand read to just above its MFMA pair (ds_read2 -> s_waitcnt -> 2 MFMAs): a wave that is
// alone on its SIMD then exposes (LDS latency - one MFMA) per pair.  A group barrier chain states the order by
// instruction class only — reads first, then the matrix instructions — and leaves the rest to the scheduler.
#ifndef PINGS_MLP_SGB
#define PINGS_MLP_SGB 0
#endif
#define MLP_SGB_DSR 0x100
#define MLP_SGB_DSW 0x200
#define MLP_SGB_MFMA 0x008
#define MLP_SGB_VALU 0x002
#define MLP_SGB_VMEM_R 0x020

// Round 4: the same five products, re-issued so that the matrix pipe does not wait for operands.
//  * A PMC pass of round 3 put the pipe at 57 % busy; the ISA showed why: every LDS operand was fetched just in time
//    (ds_read -> s_waitcnt -> two MFMAs), so each pair of MFMAs (128 cycles) exposed an LDS round trip, and the
//    bounds-checked row / column loads of a tile compiled to ~100 basic blocks (one branch per load) that nothing
//    could be scheduled across.  Now the A operands of a product are read from LDS into registers ONE PRODUCT AHEAD
//    (while the previous product's MFMAs execute; `sched_barrier`s keep the compiler from sinking the reads back to
//    their uses), and every global load is unconditional on a clamped address with a select afterwards.
//  * OH = k-steps of product B (gH^T = W2^T gY^T) per lane half: OUT / 2 when the decoder's output count is one of the
//    shipped classes (8, 24, 32: alpha; xyz / scale / colour; rotation — gaussian_renderer/__init__.py:609-708), so
//    that product does not multiply the zero padding of a 32-wide output tile; 16 with zero padding otherwise.
template <int OH, bool VECX, bool VECG>
__device__ __forceinline__ void mlp_bwd_wave_body(long long N, int IN, int OUT, const float* __restrict__ x,
                                                  const float* __restrict__ gy, const float* __restrict__ W1,
                                                  const float* __restrict__ b1, const float* __restrict__ W2,
                                                  float* __restrict__ gx, float* __restrict__ sW1, float* __restrict__ sW2, float* __restrict__ sT,
                                                  float* __restrict__ sXG, const int blk, const int nblk,
                                                  f32x16 (&aW2T)[4], f32x16 (&aW1)[4], float (&aB1)[4], float& aB2) {
  // LDS (declared once in mlp_bwd_wave_dispatch): sW1 = W1[hid][i] as [128][BW_LD], zero beyond IN; sW2 = W2[o][hid] as
  // [32][129], zero beyond OUT; sT = per wave H^T and gH^T as [4][2][32 * BW_LD] ([hid_local][row]); sXG = per wave the
  // double-buffered x and gY tiles [4][x0, x1, g0, g1][32 * BW_LD]
  const int tid = threadIdx.x, lane = tid & 63;
  // the wave index as a scalar: the tile index, the tile's base addresses and the wave's LDS windows stay in SGPRs
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

#pragma unroll
  for (int hb = 0; hb < 4; ++hb) {
    aB1[hb] = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) { aW2T[hb][q] = 0.f; aW1[hb][q] = 0.f; }
  }
  aB2 = 0.f;

  float b1f[4];  // bias k-step of product A: (b1, 0) against (1, 0); loaded with the weight images below
  float* myH = sT + (wave * 2 + 0) * 32 * BW_LD;
  float* myG = sT + (wave * 2 + 1) * 32 * BW_LD;
  // the wave's private, double-buffered images of its tile of x and gY, [row][column] with leading dimension BW_LD
  // (conflict-free by rows and by columns): the row view (B operands of products A / B) and the column view (B operands
  // of the weight-gradient products) are both read from here, so a tile's rows are fetched from HBM ONCE, coalesced
  // (round 3 fetched the column view with 32 more scattered, individually guarded loads per tile)
  float* myX = sXG + (wave * 4 + 0) * 32 * BW_LD;   // + buf * 32 * BW_LD
  float* myGY = sXG + (wave * 4 + 2) * 32 * BW_LD;
  // lane half h of product B takes outputs o0 .. o0 + OH - 1
  const int o0 = h * OH;
  // every LDS access below is ONE per-lane base plus a compile-time offset (the instruction's immediate field): written
  // as (r, h)-dependent index expressions the compiler hoisted ~100 loop-invariant addresses out of the tile loop
  // and spilled them
  const float* const baseA = sW1 + r * BW_LD + 16 * h;        // + hb * 32 * BW_LD + s          (A operands of product A)
  const float* const baseB = sW2 + o0 * 129 + r;              // + s * 129 + hb * 32            (A operands of product B)
  const float* const baseC = sW1 + 4 * h * BW_LD + r;         // + (hb * 32 + rm(q)) * BW_LD    (A operands of product C)
  float* const wrH = myH + 4 * h * BW_LD + r;                 // + rm(q) * BW_LD                (transpose: write)
  float* const wrG = myG + 4 * h * BW_LD + r;
  const float* const rdH = myH + r * BW_LD + 16 * h;          // + s                            (transpose: read)
  const float* const rdG = myG + r * BW_LD + 16 * h;
  const long long ntiles = (N + 31) / 32;
  const long long nwaves = (long long)nblk * 4;       // this decoder's share of the grid (mlp_bwd_wave_grouped_kernel)
  const long long wave0 = (long long)blk * 4 + wave;

  // Row r of tile t: columns 16 h .. 16 h + 15 of x and of gY (zero beyond IN / OUT and beyond row N).  Every load is
  // issued whatever the row / column — the address is clamped into the array and the value masked afterwards with
  // integer ops — so a tile's fetch is ONE basic block with all loads in flight together.
  // fetch_rows only LOADS (raw values stay in flight in xd / gd for the whole tile); stage_rows masks them (integer and:
  // no select the compiler could turn back into a branch) and writes the LDS image.
  auto fetch_rows = [&](long long t, float (&xd)[16], float (&gd)[16]) {
    // address = wave-uniform tile base (SGPR pair) + a 32-bit per-lane element offset.  (As 64-bit per-lane pointers
    // the compiler kept ~25 loop-invariant address pairs, spilled them and re-read them from scratch at the top of
    // every tile: 28 scratch loads in front of the fetch.)  The lane half goes through an opaque move so that the
    // clamped column offsets are a handful of integer ops per tile instead of hoisted registers.
    long long tb = t * 32;
    if (tb > N - 1) tb = N - 1;                      // beyond the last tile: row N - 1 again, never used
    const long long below = N - 1 - tb;              // rows of the array after the tile's first one
    const int rl = below < 31 ? (r < (int)below ? r : (int)below) : r;
    int hh = h;
    __asm__ volatile("" : "+v"(hh));
    const float* xb = x + (size_t)tb * IN;
    const float* gb = gy + (size_t)tb * OUT;
    const uint32_t xro = (uint32_t)(rl * IN), gro = (uint32_t)(rl * OUT);   // unsigned: the saddr + 32-bit voffset form
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int k = 16 * hh + 4 * q4;
      if (VECX) {    // IN a multiple of four, 16-byte aligned rows
        const float4 v = *reinterpret_cast<const float4*>(xb + (xro + (uint32_t)(k < IN ? k : 0)));
        xd[4 * q4] = v.x; xd[4 * q4 + 1] = v.y; xd[4 * q4 + 2] = v.z; xd[4 * q4 + 3] = v.w;
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) xd[4 * q4 + u] = xb[xro + (uint32_t)(k + u < IN ? k + u : 0)];
      }
      if (VECG) {
        const float4 u4 = *reinterpret_cast<const float4*>(gb + (gro + (uint32_t)(k < OUT ? k : 0)));
        gd[4 * q4] = u4.x; gd[4 * q4 + 1] = u4.y; gd[4 * q4 + 2] = u4.z; gd[4 * q4 + 3] = u4.w;
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) gd[4 * q4 + u] = gb[gro + (uint32_t)(k + u < OUT ? k + u : 0)];
      }
    }
  };
  auto stage_rows = [&](int buf, long long t, const float (&xd)[16], const float (&gd)[16]) {
    const uint32_t live = (t * 32 + r) < N ? 0xFFFFFFFFu : 0u;
    float* dx = myX + buf * 32 * BW_LD + r * BW_LD + 16 * h;
    float* dg = myGY + buf * 32 * BW_LD + r * BW_LD + 16 * h;
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      const int c = 16 * h + s2;
      dx[s2] = __uint_as_float(__float_as_uint(xd[s2]) & (c < IN ? live : 0u));
      dg[s2] = __uint_as_float(__float_as_uint(gd[s2]) & (c < OUT ? live : 0u));
    }
  };
  // A operands of products A (+ bias step) and B of hidden block hb, from the LDS weight images
  float opA[17], opB[OH];
  auto load_AB = [&](int hb) {
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) opA[s2] = baseA[hb * 32 * BW_LD + s2];
    opA[16] = b1f[hb];
#pragma unroll
    for (int s2 = 0; s2 < OH; ++s2) opB[s2] = baseB[s2 * 129 + hb * 32];
  };

#ifdef PINGS_MLP_STATS   // diagnostic build only (tools/build_stats_lib.sh): shader-clock ticks per phase of the tile loop
  unsigned long long st_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define MLP_TICK(k_) do { const unsigned long long now_ = __builtin_readcyclecounter(); st_t[k_] += now_ - st_last; st_last = now_; } while (0)
  unsigned long long st_last = __builtin_readcyclecounter();
#else
#define MLP_TICK(k_) do { } while (0)
#endif
  // row views (B operands of products A / B) of the tile about to be processed: read from the staged image at the END
  // of the previous tile, under its last weight-gradient products, so that a tile starts with its MFMAs
  float xf[16], gyf[OH];
  auto read_rows = [&](int b) {
    const float* bxr = myX + b * 32 * BW_LD + r * BW_LD + 16 * h;
    const float* bgr = myGY + b * 32 * BW_LD + r * BW_LD + o0;
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) xf[s2] = bxr[s2];
#pragma unroll
    for (int s2 = 0; s2 < OH; ++s2) gyf[s2] = bgr[s2];
  };
  float xn[16], gn[16];
  int buf = 0;
  fetch_rows(wave0 < ntiles ? wave0 : 0, xn, gn);
  // the weight images are staged under the first tile's fetch
  for (int e = tid; e < 128 * 32; e += 256) {
    const int j = e >> 5, i = e & 31;
    sW1[j * BW_LD + i] = i < IN ? W1[(size_t)j * IN + i] : 0.f;
  }
  for (int e = tid; e < 32 * 128; e += 256) {
    const int o = e >> 7, j = e & 127;
    sW2[o * 129 + j] = o < OUT ? W2[(size_t)o * 128 + j] : 0.f;
  }
#pragma unroll
  for (int hb = 0; hb < 4; ++hb) b1f[hb] = h == 0 ? b1[hb * 32 + r] : 0.f;
  __syncthreads();
  stage_rows(0, wave0 < ntiles ? wave0 : 0, xn, gn);
  MLP_WB_T;
  read_rows(0);
  load_AB(0);
  const float one = h == 0 ? 1.f : 0.f;
  MLP_TICK(0);   // prologue
  for (long long t = wave0; t < ntiles; t += nwaves) {
    // the next tile's rows: in flight for the whole of this tile, staged into the other LDS buffer at its end
    // (beyond the last tile the clamped addresses re-read row N - 1 and the values are never used)
    fetch_rows(t + nwaves, xn, gn);
    float xcol[16], gycol[16];     // column views (B operands of the weight-gradient products): first used in product D
    {
      const float* bxc = myX + buf * 32 * BW_LD + 16 * h * BW_LD + r;
      const float* bgc = myGY + buf * 32 * BW_LD + 16 * h * BW_LD + r;
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) {
        xcol[s2] = bxc[s2 * BW_LD];
        gycol[s2] = bgc[s2 * BW_LD];
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) aB2 += gycol[s2];
    f32x16 gxacc = {0};
    MLP_TICK(1);   // tile start: operand views from LDS
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) {
      // ---- products A and B on the operands read one product ago; meanwhile the A operands of product C
      float opC[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) opC[q] = baseC[(hb * 32 + (q & 3) + 8 * (q >> 2)) * BW_LD];
      MLP_SB_A;
      f32x16 pre = {0}, gH = {0};
#pragma unroll
      for (int s2 = 0; s2 < 17; ++s2) {
        pre = mfma(opA[s2], s2 < 16 ? xf[s2] : one, pre);
        if (s2 < OH) gH = mfma(opB[s2], gyf[s2], gH);
      }
      MLP_SB_B;
      MLP_TICK(2);   // products A / B issued
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        gH[q] = pre[q] > 0.f ? gH[q] : 0.f;
        pre[q] = fmaxf(pre[q], 0.f);
      }
      MLP_TICK(3);   // mask (waits for the products)
      // ---- product C; meanwhile H^T and gH^T take their trip through the wave's private LDS ([hid_local][row])
      MLP_WB_W;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        wrH[((q & 3) + 8 * (q >> 2)) * BW_LD] = pre[q];
        wrG[((q & 3) + 8 * (q >> 2)) * BW_LD] = gH[q];
      }
      // last hidden block: the next tile's rows (in flight since the top of this tile) go into the other image
      if (hb == 3) stage_rows(buf ^ 1, t + nwaves, xn, gn);
      MLP_WB_R;
      float aH[16], aG[16];
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) {
        aH[s2] = rdH[s2];
        aG[s2] = rdG[s2];
      }
      MLP_SB_C;
#if PINGS_MLP_SGB == 1
      // the transposed operands, then product C, then every other LDS read of the stretch, then the rest of the MFMAs
      __builtin_amdgcn_sched_group_barrier(MLP_SGB_DSR, 16, 0);
      __builtin_amdgcn_sched_group_barrier(MLP_SGB_MFMA, 16, 0);
      __builtin_amdgcn_sched_group_barrier(MLP_SGB_DSR, 64, 0);
      __builtin_amdgcn_sched_group_barrier(MLP_SGB_MFMA, 96, 0);
#elif PINGS_MLP_SGB == 2
      // as 1, with the second batch of reads spread under product C (four MFMAs, eight reads, ...)
      __builtin_amdgcn_sched_group_barrier(MLP_SGB_DSR, 16, 0);
#pragma unroll
      for (int g_ = 0; g_ < 4; ++g_) {
        __builtin_amdgcn_sched_group_barrier(MLP_SGB_MFMA, 4, 0);
        __builtin_amdgcn_sched_group_barrier(MLP_SGB_DSR, 12, 0);
      }
      __builtin_amdgcn_sched_group_barrier(MLP_SGB_MFMA, 96, 0);
#elif PINGS_MLP_SGB == 3
      // every MFMA of the stretch preceded by one LDS read while there are any
#pragma unroll
      for (int g_ = 0; g_ < 64; ++g_) {
        __builtin_amdgcn_sched_group_barrier(MLP_SGB_DSR, 1, 0);
        __builtin_amdgcn_sched_group_barrier(MLP_SGB_MFMA, 1, 0);
      }
#elif PINGS_MLP_SGB == 4
      // reads two at a time, one MFMA between the pairs
      __builtin_amdgcn_sched_group_barrier(MLP_SGB_DSR, 8, 0);
#pragma unroll
      for (int g_ = 0; g_ < 40; ++g_) {
        __builtin_amdgcn_sched_group_barrier(MLP_SGB_MFMA, 1, 0);
        __builtin_amdgcn_sched_group_barrier(MLP_SGB_DSR, 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(MLP_SGB_MFMA, 96, 0);
#endif
#pragma unroll
      for (int q = 0; q < 16; ++q) gxacc = mfma(opC[q], gH[q], gxacc);   // (computed even when gx is null: no branch)
      MLP_TICK(4);   // transposes + product C issued
      // ---- the weight-gradient products; meanwhile the operands of the next hidden block's A and B
      load_AB((hb + 1) & 3);
      if (hb == 3) read_rows(buf ^ 1);
      MLP_SB_D;
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) {
        aB1[hb] += aG[s2];
        aW2T[hb] = mfma(aH[s2], gycol[s2], aW2T[hb]);
        aW1[hb] = mfma(aG[s2], xcol[s2], aW1[hb]);
      }
      MLP_TICK(5);   // products D issued
    }
    buf ^= 1;
    MLP_TICK(6);
    if (gx) {
      const long long row = t * 32 + r;
      if (row < N) {
        float* dst = gx + (size_t)row * IN;
        if (IN % 4 == 0 && ((reinterpret_cast<uintptr_t>(gx) & 15) == 0)) {
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int i = 8 * gq + 4 * h;
            if (i < IN)
              *reinterpret_cast<float4*>(dst + i) =
                  make_float4(gxacc[4 * gq], gxacc[4 * gq + 1], gxacc[4 * gq + 2], gxacc[4 * gq + 3]);
          }
        } else {
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int i = rowmap(q, h);
            if (i < IN) dst[i] = gxacc[q];
          }
        }
      }
    }
  }

  MLP_TICK(7);   // gX store of the last tile
#ifdef PINGS_MLP_STATS
  if (lane == 0)
    for (int k_ = 0; k_ < 8; ++k_) atomicAdd(&g_mlp_stats[k_], st_t[k_]);
#endif
}

// The workgroup's partial weight gradients (after mlp_bwd_wave_body; its own function so that the body's LDS pointers,
// which promise not to alias each other, are out of scope when the whole array is re-carved).  Every wave writes its
// accumulators into its OWN region of the (now dead) LDS at once, then all 256 threads add the four regions in wave
// order and write the partial in the global layout.  (Round 3 let the waves take turns adding into one image: four
// serial rounds and five barriers, ~5 us of the launch's ~35 us of fixed cost.)  Region: gW1 as [hid][32] (lane = i),
// gW2 as [o][129] (lane = o), gb1, gb2.
__device__ __forceinline__ void mlp_bwd_wave_epilogue(int IN, int OUT, float* __restrict__ partials, size_t per_block,
                                                      float* __restrict__ sAll, const int blk, const f32x16 (&aW2T)[4],
                                                      const f32x16 (&aW1)[4], const float (&aB1)[4], const float aB2) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  __syncthreads();
  {
    float* R = sAll + wave * BW_REGION;
    float* rW1 = R, *rW2 = R + 128 * 32, *rB = R + 128 * 32 + 32 * 129;
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int hid = hb * 32 + rowmap(q, h);
        rW2[r * 129 + hid] = aW2T[hb][q];
        rW1[hid * 32 + r] = aW1[hb][q];
      }
      const float v = aB1[hb] + __shfl_xor(aB1[hb], 32, 64);  // the two row halves of hidden unit hb*32 + r
      if (h == 0) rB[hb * 32 + r] = v;
    }
    const float v2 = aB2 + __shfl_xor(aB2, 32, 64);
    if (h == 0) rB[128 + r] = v2;
  }
  __syncthreads();
  float* P = partials + (size_t)blk * per_block;
  const int nW1 = 128 * IN, nW2 = OUT * 128;
  auto sum4 = [&](int off) {
    return ((sAll[off] + sAll[BW_REGION + off]) + sAll[2 * BW_REGION + off]) + sAll[3 * BW_REGION + off];
  };
  {
    const int i = tid & 31;
    if (i < IN)
      for (int j = tid >> 5; j < 128; j += 8) P[j * IN + i] = sum4(j * 32 + i);
  }
  for (int e = tid; e < nW2; e += 256) P[nW1 + e] = sum4(128 * 32 + (e >> 7) * 129 + (e & 127));
  if (tid < 128) P[nW1 + nW2 + tid] = sum4(128 * 32 + 32 * 129 + tid);
  if (tid < OUT) P[nW1 + nW2 + 128 + tid] = sum4(128 * 32 + 32 * 129 + 128 + tid);
}

// one instantiation per output class (the B product's k-steps are compile-time); the choice is uniform per workgroup
__device__ __forceinline__ void mlp_bwd_wave_dispatch(long long N, int IN, int OUT, const float* __restrict__ x,
                                                      const float* __restrict__ gy, const float* __restrict__ W1,
                                                      const float* __restrict__ b1, const float* __restrict__ W2,
                                                      float* __restrict__ gx, float* __restrict__ partials,
                                                      size_t per_block, const int blk, const int nblk) {
  // one array (the epilogue re-carves it into four per-wave regions): W1 image, W2 image, transposes, x / gY tiles.
  // 135 KB in all: one workgroup per CU, as the registers dictate anyway
  __shared__ float sAll[BW_LDS_FLOATS];
  float* const sW1 = sAll;
  float* const sW2 = sW1 + 128 * BW_LD;
  float* const sT = sW2 + 32 * 129;
  float* const sXG = sT + 4 * 2 * 32 * BW_LD;
  // 16-byte row loads where the row length and the base allow (the colour decoder's 19 inputs: scalar loads of x)
  const bool vecx = (IN % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  const bool vecg = (OUT % 4 == 0) && ((reinterpret_cast<uintptr_t>(gy) & 15) == 0);
#define PINGS_BWD_BODY(OH_, VX_, VG_) \
  mlp_bwd_wave_body<OH_, VX_, VG_>(N, IN, OUT, x, gy, W1, b1, W2, gx, sW1, sW2, sT, sXG, blk, nblk, aW2T, aW1, aB1, aB2)
#define PINGS_BWD_CLASS(VX_)                          \
  do {                                                \
    if (!vecg) PINGS_BWD_BODY(16, VX_, false);        \
    else if (OUT == 24) PINGS_BWD_BODY(12, VX_, true); \
    else if (OUT == 8) PINGS_BWD_BODY(4, VX_, true);  \
    else PINGS_BWD_BODY(16, VX_, true);               \
  } while (0)
  f32x16 aW2T[4], aW1[4];   // the wave's weight-gradient accumulators: 128 registers for the whole launch
  float aB1[4], aB2;
  if (vecx) PINGS_BWD_CLASS(true);
  else PINGS_BWD_CLASS(false);
#undef PINGS_BWD_CLASS
#undef PINGS_BWD_BODY
  mlp_bwd_wave_epilogue(IN, OUT, partials, per_block, sAll, blk, aW2T, aW1, aB1, aB2);
}

// ---------------------------------------------------------------- backward, TWO waves per SIMD (round 4)
// mlp_bwd_wave_body keeps one wave per SIMD (its 128 accumulator registers + every operand read one product ahead need
// ~430 of the 512 registers), and a wave that is alone on its SIMD pays every latency it cannot schedule around: the
// launch runs its MFMAs at 72 % of the pipe in steady state where the bare MFMA stream of one wave reaches 88 %
// (tools/dec_scale.py with timing-only ablation builds; profiles/mfma_calib.hip: one wave per SIMD 140-148 TFLOP/s,
// two waves 154).  This body fits a wave into 256 registers so that EIGHT waves share a CU: the operands are read
// from LDS where they are used (the other wave of the SIMD covers the round trip), H^T and gH^T take turns in ONE
// private transpose tile (the two weight-gradient products run one after the other), the tile of x / gY is single-
// buffered (the next tile's rows are fetched under the last product and staged after it).  Same
// products, same k orders, same accumulation order per accumulator as mlp_bwd_wave_body: bit-identical partials per
// wave; a workgroup's partial adds eight waves instead of four.
template <int OH, bool VECX, bool VECG>
__device__ __forceinline__ void mlp_bwd_wave2_body(long long N, int IN, int OUT, const float* __restrict__ x,
                                                   const float* __restrict__ gy, const float* __restrict__ W1,
                                                   const float* __restrict__ b1, const float* __restrict__ W2,
                                                   float* __restrict__ gx, float* __restrict__ sW1,
                                                   float* __restrict__ sW2, float* __restrict__ sP, const int blk,
                                                   const int nblk, f32x16 (&aW2T)[4], f32x16 (&aW1)[4],
                                                   float (&aB1)[4], float& aB2) {
  // LDS: sW1 = W1[hid][i] as [128][BW_LD]; sW2 = W2^T[hid][o] as [128][BW_LD] (the k-steps of product B are then
  // neighbours in memory, like product A's); sP = per wave [T, X, G][32 * BW_LD]:
  // T = the transpose tile ([hid_local][row]), X / G = this tile's rows of x and gY ([row][column])
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int hb = 0; hb < 4; ++hb) {
    aB1[hb] = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) { aW2T[hb][q] = 0.f; aW1[hb][q] = 0.f; }
  }
  aB2 = 0.f;
  float* const myT = sP + (wave * 3 + 0) * 32 * BW_LD;
  float* const myX = sP + (wave * 3 + 1) * 32 * BW_LD;
  float* const myG = sP + (wave * 3 + 2) * 32 * BW_LD;
  const long long ntiles = (N + 31) / 32;
  const long long nwaves = (long long)nblk * 8;
  const long long wave0 = (long long)blk * 8 + wave;

  auto fetch_rows = [&](long long t, float (&xd)[16], float (&gd)[16]) {   // as in mlp_bwd_wave_body
    long long tb = t * 32;
    if (tb > N - 1) tb = N - 1;
    // (opaque scalar: without it the compiler turns every load's address into its own 64-bit per-lane induction
    // variable across the tile loop — 20 register pairs that a 256-register wave does not have)
    __asm__ volatile("" : "+s"(tb));
    const long long below = N - 1 - tb;
    const int rl = below < 31 ? (r < (int)below ? r : (int)below) : r;
    int hh = h;
    __asm__ volatile("" : "+v"(hh));
    const float* xb = x + (size_t)tb * IN;
    const float* gb = gy + (size_t)tb * OUT;
    const uint32_t xro = (uint32_t)(rl * IN), gro = (uint32_t)(rl * OUT);
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int k = 16 * hh + 4 * q4;
      if (VECX) {
        const float4 v = *reinterpret_cast<const float4*>(xb + (xro + (uint32_t)(k < IN ? k : 0)));
        xd[4 * q4] = v.x; xd[4 * q4 + 1] = v.y; xd[4 * q4 + 2] = v.z; xd[4 * q4 + 3] = v.w;
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) xd[4 * q4 + u] = xb[xro + (uint32_t)(k + u < IN ? k + u : 0)];
      }
      if (VECG) {
        const float4 u4 = *reinterpret_cast<const float4*>(gb + (gro + (uint32_t)(k < OUT ? k : 0)));
        gd[4 * q4] = u4.x; gd[4 * q4 + 1] = u4.y; gd[4 * q4 + 2] = u4.z; gd[4 * q4 + 3] = u4.w;
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) gd[4 * q4 + u] = gb[gro + (uint32_t)(k + u < OUT ? k + u : 0)];
      }
    }
  };
  auto stage_rows = [&](long long t, const float (&xd)[16], const float (&gd)[16]) {
    const uint32_t live = (t * 32 + r) < N ? 0xFFFFFFFFu : 0u;
    int rr = r, hh = h;
    __asm__ volatile("" : "+v"(rr), "+v"(hh));
    float* dx = myX + rr * BW_LD + 16 * hh;
    float* dg = myG + rr * BW_LD + 16 * hh;
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      const int c = 16 * hh + s2;
      dx[s2] = __uint_as_float(__float_as_uint(xd[s2]) & (c < IN ? live : 0u));
      dg[s2] = __uint_as_float(__float_as_uint(gd[s2]) & (c < OUT ? live : 0u));
    }
  };

  float xn[16], gn[16];
  fetch_rows(wave0 < ntiles ? wave0 : 0, xn, gn);
  // the weight images are staged under the first tile's fetch
  for (int e = tid; e < 128 * 32; e += 512) {
    const int j = e >> 5, i = e & 31;
    sW1[j * BW_LD + i] = i < IN ? W1[(size_t)j * IN + i] : 0.f;
  }
  for (int e = tid; e < 32 * 128; e += 512) {
    const int o = e >> 7, j = e & 127;
    sW2[j * BW_LD + o] = o < OUT ? W2[(size_t)o * 128 + j] : 0.f;
  }
  float b1f[4];  // bias k-step of product A: (b1, 0) against (1, 0)
#pragma unroll
  for (int hb = 0; hb < 4; ++hb) b1f[hb] = h == 0 ? b1[hb * 32 + r] : 0.f;
  stage_rows(wave0 < ntiles ? wave0 : 0, xn, gn);
  __syncthreads();
  const float one = h == 0 ? 1.f : 0.f;

  // Every LDS address is formed where it is used from the lane's (r, h), which go through an opaque move first: left to
  // itself the compiler hoists ~60 loop-invariant address registers out of the tile loop, or keeps a hidden block's
  // nine bases alive through all of its phases, and spills them (189 dwords of scratch in the first build, reloaded
  // one by one in front of the reads).  A handful of integer ops per phase instead.
#define MLP_RH int rr = r, hh = h; __asm__ volatile("" : "+v"(rr), "+v"(hh))
  for (long long t = wave0; t < ntiles; t += nwaves) {
    f32x16 gxacc = {0};
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) {
      // ---- A: pre^T = W1_hb x^T + b1;  B: gH^T = W2_hb^T gY^T
      f32x16 pre = {0}, gH = {0};
      {
        MLP_RH;
        const float* const baseA = sW1 + (hb * 32 + rr) * BW_LD + 16 * hh;   // + s
        const float* const baseB = sW2 + (hb * 32 + rr) * BW_LD + hh * OH;   // + s
        const float* const xrow = myX + rr * BW_LD + 16 * hh;                // + s
        const float* const grow = myG + rr * BW_LD + hh * OH;                // + s
#pragma unroll
        for (int s2 = 0; s2 < 17; ++s2) {
          pre = mfma(s2 < 16 ? baseA[s2] : b1f[hb], s2 < 16 ? xrow[s2] : one, pre);
          if (s2 < OH) gH = mfma(baseB[s2], grow[s2], gH);
        }
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const bool on = pre[q] > 0.f;
        gH[q] = on ? gH[q] : 0.f;
        pre[q] = on ? pre[q] : 0.f;
      }
      // ---- H^T through the transpose tile
      MLP_WB_W;
      {
        MLP_RH;
        float* const wrT = myT + 4 * hh * BW_LD + rr;                        // + rm(q) * BW_LD
#pragma unroll
        for (int q = 0; q < 16; ++q) wrT[((q & 3) + 8 * (q >> 2)) * BW_LD] = pre[q];
      }
      MLP_WB_R;
      // ---- C: gX^T += W1_hb^T gH^T (operands in registers / the weight image: covers the transpose's round trip)
      {
        MLP_RH;
        const float* const baseC = sW1 + (hb * 32 + 4 * hh) * BW_LD + rr;    // + rm(q) * BW_LD
#pragma unroll
        for (int q = 0; q < 16; ++q) gxacc = mfma(baseC[((q & 3) + 8 * (q >> 2)) * BW_LD], gH[q], gxacc);
      }
      // ---- D1: gW2^T_hb += H^T gY
      {
        MLP_RH;
        const float* const rdT = myT + rr * BW_LD + 16 * hh;                 // + s
        const float* const gcolp = myG + 16 * hh * BW_LD + rr;               // + s * BW_LD
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) aW2T[hb] = mfma(rdT[s2], gcolp[s2 * BW_LD], aW2T[hb]);
      }
      // ---- gH^T takes the tile over
      MLP_WB_W;
      {
        MLP_RH;
        float* const wrT = myT + 4 * hh * BW_LD + rr;
#pragma unroll
        for (int q = 0; q < 16; ++q) wrT[((q & 3) + 8 * (q >> 2)) * BW_LD] = gH[q];
      }
      MLP_WB_R;
      // the next tile's rows: fetched into the 32 registers that pre / gH have just vacated (a wave has 256), in flight
      // under the last product and the gX store; what is left of the latency is the SIMD's other wave's to cover
      if (hb == 3) fetch_rows(t + nwaves, xn, gn);
      // ---- D2: gW1_hb += gH^T x, gb1 from the transposed fragments (the add is pinned where the fragment arrives:
      // scheduled freely, the sixteen fragments of every hidden block were kept for a packed add at the tile's end)
      {
        MLP_RH;
        const float* const rdT = myT + rr * BW_LD + 16 * hh;
        const float* const xcolp = myX + 16 * hh * BW_LD + rr;               // + s * BW_LD
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
          const float aG = rdT[s2];
          __asm__ volatile("v_add_f32 %0, %0, %1" : "+v"(aB1[hb]) : "v"(aG));
          aW1[hb] = mfma(aG, xcolp[s2 * BW_LD], aW1[hb]);
        }
      }
    }
    {
      MLP_RH;
      const float* const gcolp = myG + 16 * hh * BW_LD + rr;
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) aB2 += gcolp[s2 * BW_LD];
    }
    if (gx) {
      long long t32 = t * 32;
      __asm__ volatile("" : "+s"(t32));          // scalar tile base + 32-bit lane offset, as in fetch_rows
      const long long row = t32 + r;
      if (row < N) {
        float* dst = gx + (size_t)t32 * IN + (uint32_t)(r * IN);
        if (IN % 4 == 0 && ((reinterpret_cast<uintptr_t>(gx) & 15) == 0)) {
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int i = 8 * gq + 4 * h;
            if (i < IN)
              *reinterpret_cast<float4*>(dst + i) =
                  make_float4(gxacc[4 * gq], gxacc[4 * gq + 1], gxacc[4 * gq + 2], gxacc[4 * gq + 3]);
          }
        } else {
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int i = rowmap(q, h);
            if (i < IN) dst[i] = gxacc[q];
          }
        }
      }
    }
    // ---- the next tile's rows replace this one's (every read of X / G above is complete: same wave, program order)
    MLP_WB_W;
    stage_rows(t + nwaves, xn, gn);
    MLP_WB_R;
  }
}
#undef MLP_RH

// eight waves: waves 0-3 store their accumulators into the four regions, waves 4-7 add theirs on top, then all
// threads add the four regions in order: ((w0 + w4) + (w1 + w5)) + (w2 + w6)) + (w3 + w7)
__device__ __forceinline__ void mlp_bwd_wave2_epilogue(int IN, int OUT, float* __restrict__ partials, size_t per_block,
                                                       float* __restrict__ sAll, const int blk, const f32x16 (&aW2T)[4],
                                                       const f32x16 (&aW1)[4], const float (&aB1)[4], const float aB2) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  for (int round = 0; round < 2; ++round) {
    __syncthreads();
    if ((wave >> 2) == round) {
      float* R = sAll + (wave & 3) * BW_REGION;
      float* rW1 = R, *rW2 = R + 128 * 32, *rB = R + 128 * 32 + 32 * 129;
#pragma unroll
      for (int hb = 0; hb < 4; ++hb) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int hid = hb * 32 + rowmap(q, h);
          if (round == 0) { rW2[r * 129 + hid] = aW2T[hb][q]; rW1[hid * 32 + r] = aW1[hb][q]; }
          else { rW2[r * 129 + hid] += aW2T[hb][q]; rW1[hid * 32 + r] += aW1[hb][q]; }
        }
        const float v = aB1[hb] + __shfl_xor(aB1[hb], 32, 64);
        if (h == 0) { if (round == 0) rB[hb * 32 + r] = v; else rB[hb * 32 + r] += v; }
      }
      const float v2 = aB2 + __shfl_xor(aB2, 32, 64);
      if (h == 0) { if (round == 0) rB[128 + r] = v2; else rB[128 + r] += v2; }
    }
  }
  __syncthreads();
  float* P = partials + (size_t)blk * per_block;
  const int nW1 = 128 * IN, nW2 = OUT * 128;
  auto sum4 = [&](int off) {
    return ((sAll[off] + sAll[BW_REGION + off]) + sAll[2 * BW_REGION + off]) + sAll[3 * BW_REGION + off];
  };
  {
    const int i = tid & 31;
    if (i < IN)
      for (int j = tid >> 5; j < 128; j += 16) P[j * IN + i] = sum4(j * 32 + i);
  }
  for (int e = tid; e < nW2; e += 512) P[nW1 + e] = sum4(128 * 32 + (e >> 7) * 129 + (e & 127));
  if (tid < 128) P[nW1 + nW2 + tid] = sum4(128 * 32 + 32 * 129 + tid);
  if (tid < OUT) P[nW1 + nW2 + 128 + tid] = sum4(128 * 32 + 32 * 129 + 128 + tid);
}

__device__ __forceinline__ void mlp_bwd_wave2_dispatch(long long N, int IN, int OUT, const float* __restrict__ x,
                                                       const float* __restrict__ gy, const float* __restrict__ W1,
                                                       const float* __restrict__ b1, const float* __restrict__ W2,
                                                       float* __restrict__ gx, float* __restrict__ partials,
                                                       size_t per_block, const int blk, const int nblk) {
  constexpr int LDS2 = 2 * 128 * BW_LD + 8 * 3 * 32 * BW_LD;   // W1 image, W2^T image, 8 waves x [T, X, G] tiles: 135 KB
  static_assert(4 * BW_REGION <= LDS2, "the four epilogue regions must fit the kernel's LDS");
  __shared__ float sAll[LDS2];
  float* const sW1 = sAll;
  float* const sW2 = sW1 + 128 * BW_LD;
  float* const sP = sW2 + 128 * BW_LD;
  const bool vecx = (IN % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  const bool vecg = (OUT % 4 == 0) && ((reinterpret_cast<uintptr_t>(gy) & 15) == 0);
  f32x16 aW2T[4], aW1[4];
  float aB1[4], aB2;
#define PINGS_BWD2_BODY(OH_, VX_, VG_) \
  mlp_bwd_wave2_body<OH_, VX_, VG_>(N, IN, OUT, x, gy, W1, b1, W2, gx, sW1, sW2, sP, blk, nblk, aW2T, aW1, aB1, aB2)
#define PINGS_BWD2_CLASS(VX_)                           \
  do {                                                 \
    if (!vecg) PINGS_BWD2_BODY(16, VX_, false);         \
    else if (OUT == 24) PINGS_BWD2_BODY(12, VX_, true); \
    else if (OUT == 8) PINGS_BWD2_BODY(4, VX_, true);   \
    else PINGS_BWD2_BODY(16, VX_, true);                \
  } while (0)
  if (vecx) PINGS_BWD2_CLASS(true);
  else PINGS_BWD2_CLASS(false);
#undef PINGS_BWD2_CLASS
#undef PINGS_BWD2_BODY
  mlp_bwd_wave2_epilogue(IN, OUT, partials, per_block, sAll, blk, aW2T, aW1, aB1, aB2);
}

__global__ __launch_bounds__(512) void mlp_bwd_wave2_grouped_kernel(long long N, MlpJobs j, int njobs) {
#ifdef PINGS_MLP_STATS
  const unsigned long long c0_ = __builtin_readcyclecounter(), w0_ = wall_clock64();
#endif
  int g = 0;
  while (g + 1 < njobs && (int)blockIdx.x >= j.wg0[g + 1]) ++g;
  mlp_bwd_wave2_dispatch(N, j.IN[g], j.OUT[g], j.x[g], j.gy[g], j.W1[g], j.b1[g], j.W2[g], j.gx[g], j.partials[g],
                         j.per_block[g], (int)blockIdx.x - j.wg0[g], j.wg0[g + 1] - j.wg0[g]);
#ifdef PINGS_MLP_STATS
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    g_mlp_clock[0] = __builtin_readcyclecounter() - c0_;
    g_mlp_clock[1] = wall_clock64() - w0_;
  }
#endif
}

__global__ __launch_bounds__(256, 1) void mlp_bwd_wave_kernel(long long N, int IN, int OUT, const float* __restrict__ x,
                                                              const float* __restrict__ gy, const float* __restrict__ W1,
                                                              const float* __restrict__ b1, const float* __restrict__ W2,
                                                              float* __restrict__ gx, float* __restrict__ partials,
                                                              size_t per_block) {
  mlp_bwd_wave_dispatch(N, IN, OUT, x, gy, W1, b1, W2, gx, partials, per_block, (int)blockIdx.x, (int)gridDim.x);
}

// 1-D grid; decoder g owns workgroups [wg0[g], wg0[g + 1]): shares proportional to the decoders' MFMAs per tile (the
// 32-wide rotation decoder issues 324 per tile, the 8-wide alpha decoder 276), so that they finish together — with
// equal shares the launch lasted as long as its most expensive decoder (6 % more)
__global__ __launch_bounds__(256, 1) void mlp_bwd_wave_grouped_kernel(long long N, MlpJobs j, int njobs) {
#ifdef PINGS_MLP_STATS
  const unsigned long long c0_ = __builtin_readcyclecounter(), w0_ = wall_clock64();
#endif
  int g = 0;
  while (g + 1 < njobs && (int)blockIdx.x >= j.wg0[g + 1]) ++g;
  mlp_bwd_wave_dispatch(N, j.IN[g], j.OUT[g], j.x[g], j.gy[g], j.W1[g], j.b1[g], j.W2[g], j.gx[g], j.partials[g],
                        j.per_block[g], (int)blockIdx.x - j.wg0[g], j.wg0[g + 1] - j.wg0[g]);
#ifdef PINGS_MLP_STATS
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    g_mlp_clock[0] = __builtin_readcyclecounter() - c0_;
    g_mlp_clock[1] = wall_clock64() - w0_;
  }
#endif
}

// ---------------------------------------------------------------- backward of the SDF decoder shape: HID = 64, OUT = 1
// (`Decoder.sdf`, decoder.py:102-104: [B k, F + 3] -> 64 -> 1; F + 3 = 35 or 11).  Same wave-per-tile scheme as
// mlp_bwd_wave_kernel, specialised for the single output: gH^T = relu'(pre) * W2[hid] * gy[row] is elementwise in the
// accumulator layout, gW2 / gb1 / gb2 accumulate per lane over all of the wave's tiles (one cross-lane reduction at
// the very end), and only gH^T makes the LDS trip for gW1 += gH^T x.  Inputs are consumed two per MFMA step
// (k = 2 s + h), so IN = 35 costs 18 steps of product A instead of a padded 32.  NS = k-steps, IB = 32-wide input
// blocks.  Per tile and hidden block: NS + 1 (A) + 16 IB (C) + 16 IB (D) MFMAs.
template <int NS, int IB, int NT>
__global__ __launch_bounds__(256, 1) void mlp_bwd_wave_h64o1_kernel(long long N, int IN, const float* __restrict__ x,
                                                                    const float* __restrict__ gy,
                                                                    const float* __restrict__ W1,
                                                                    const float* __restrict__ b1,
                                                                    const float* __restrict__ W2, float* __restrict__ gx,
                                                                    float* __restrict__ partials, size_t per_block) {
  constexpr int WI = 32 * IB + NT;          // input columns held in the W1 image
  constexpr int LD1 = WI | 1;               // odd leading dimension: conflict-free in both orientations
  static_assert(NT == 0 || IB == 1, "tail inputs follow a single 32-wide block");
  __shared__ float sW1[64 * LD1];          // W1[hid][i], zero beyond IN; reused for the workgroup's gW1
  __shared__ float sG[4][32 * BW_LD];      // per wave: gH^T as [hid_local][row]
  __shared__ float sB[64 + 64 + 1];        // workgroup sums of gW2, gb1, gb2
  __shared__ float sW2[64];                // W2[0][hid]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  for (int e = tid; e < 64 * WI; e += 256) {
    const int j = e / WI, i = e - j * WI;
    sW1[j * LD1 + i] = i < IN ? W1[(size_t)j * IN + i] : 0.f;
  }
  if (tid < 64) sW2[tid] = W2[tid];
  float b1f[2];
#pragma unroll
  for (int hb = 0; hb < 2; ++hb) b1f[hb] = h == 0 ? b1[hb * 32 + r] : 0.f;
  __syncthreads();

  f32x16 aW1[2][IB];
  float aW2[2][16], aB1[2][16], aB2 = 0.f;
  float aT[NT > 0 ? NT : 1][2][16];  // gW1 of the NT tail inputs: per-lane sums over rows, like aW2 / aB1
#pragma unroll
  for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      aW2[hb][q] = 0.f;
      aB1[hb][q] = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j) aT[j][hb][q] = 0.f;
#pragma unroll
      for (int ib = 0; ib < IB; ++ib) aW1[hb][ib][q] = 0.f;
    }
  }
  float* myG = &sG[wave][0];
  const long long ntiles = (N + 31) / 32;
  const long long nwaves = (long long)gridDim.x * 4;
  const float one = h == 0 ? 1.f : 0.f;
  for (long long t = (long long)blockIdx.x * 4 + wave; t < ntiles; t += nwaves) {
    // (a register prefetch of the next tile's operands, as in mlp_bwd_wave_kernel, measured 12 % SLOWER here)
    asm volatile("" ::: "memory");  // keep the loop-invariant LDS operands in LDS (hoisting them costs ~100 VGPRs)
    const long long row = t * 32 + r;
    const bool ok = row < N;
    float xf[NS];
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) xf[s2] = (ok && 2 * s2 + h < IN) ? x[(size_t)row * IN + 2 * s2 + h] : 0.f;
    const float gyr = ok ? gy[row] : 0.f;
    float xcol[IB][16];  // x[row = 16 h + s][column = 32 ib + r]
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      const long long rc = t * 32 + 16 * h + s2;
#pragma unroll
      for (int ib = 0; ib < IB; ++ib)
        xcol[ib][s2] = (rc < N && 32 * ib + r < IN) ? x[(size_t)rc * IN + 32 * ib + r] : 0.f;
    }
    aB2 += h == 0 ? gyr : 0.f;
    // tail inputs 32 .. 32 + NT - 1 of this lane's row (xf holds the inputs of parity h: fetch the others from the
    // lane of the other half, same row) and their gX, both on the vector ALU: a second 32-wide MFMA block for three
    // position inputs would double products C and D
    float xt[NT > 0 ? NT : 1], gxt[NT > 0 ? NT : 1];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float own = xf[16 + j / 2], other = __shfl_xor(own, 32, 64);
      xt[j] = (j & 1) == h ? own : other;
      gxt[j] = 0.f;
    }
    f32x16 gxacc[IB];
#pragma unroll
    for (int ib = 0; ib < IB; ++ib) gxacc[ib] = f32x16{0};
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      f32x16 pre = {0};
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) pre = mfma(sW1[(hb * 32 + r) * LD1 + 2 * s2 + h], xf[s2], pre);
      pre = mfma(b1f[hb], one, pre);
      float gH[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        gH[q] = pre[q] > 0.f ? sW2[hb * 32 + rowmap(q, h)] * gyr : 0.f;
        aW2[hb][q] = fmaf(fmaxf(pre[q], 0.f), gyr, aW2[hb][q]);
        aB1[hb][q] += gH[q];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          aT[j][hb][q] = fmaf(gH[q], xt[j], aT[j][hb][q]);
          gxt[j] = fmaf(sW1[(hb * 32 + rowmap(q, h)) * LD1 + 32 + j], gH[q], gxt[j]);
        }
      }
      if (gx) {
#pragma unroll
        for (int ib = 0; ib < IB; ++ib)
#pragma unroll
          for (int q = 0; q < 16; ++q)
            gxacc[ib] = mfma(sW1[(hb * 32 + rowmap(q, h)) * LD1 + 32 * ib + r], gH[q], gxacc[ib]);
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 16; ++q) myG[rowmap(q, h) * BW_LD + r] = gH[q];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) {
        const float aG = myG[r * BW_LD + 16 * h + s2];
#pragma unroll
        for (int ib = 0; ib < IB; ++ib) aW1[hb][ib] = mfma(aG, xcol[ib][s2], aW1[hb][ib]);
      }
    }
    if (gx && ok) {
      float* dst = gx + (size_t)row * IN;
#pragma unroll
      for (int ib = 0; ib < IB; ++ib)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int i = 32 * ib + rowmap(q, h);
          if (i < IN) dst[i] = gxacc[ib][q];
        }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float v = gxt[j] + __shfl_xor(gxt[j], 32, 64);  // the two halves hold disjoint hidden units
      if (gx && ok && h == 0 && 32 + j < IN) gx[(size_t)row * IN + 32 + j] = v;
    }
  }

  // per-lane sums over rows -> sums over the 32 lanes that share h (hidden unit hb*32 + rowmap(q, h))
#pragma unroll
  for (int hb = 0; hb < 2; ++hb)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) {
        aW2[hb][q] += __shfl_xor(aW2[hb][q], off, 64);
        aB1[hb][q] += __shfl_xor(aB1[hb][q], off, 64);
#pragma unroll
        for (int j = 0; j < NT; ++j) aT[j][hb][q] += __shfl_xor(aT[j][hb][q], off, 64);
      }
    }
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) aB2 += __shfl_xor(aB2, off, 64);

  // the four waves add theirs in wave order: gW1 into the (now dead) W1 image, the vectors into sB
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int hb = 0; hb < 2; ++hb)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int hid = hb * 32 + rowmap(q, h);
#pragma unroll
          for (int ib = 0; ib < IB; ++ib) {
            float* d1 = &sW1[hid * LD1 + 32 * ib + r];
            if (w == 0) *d1 = aW1[hb][ib][q]; else *d1 += aW1[hb][ib][q];
          }
          if (r == 0) {
            if (w == 0) { sB[hid] = aW2[hb][q]; sB[64 + hid] = aB1[hb][q]; }
            else { sB[hid] += aW2[hb][q]; sB[64 + hid] += aB1[hb][q]; }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              if (w == 0) sW1[hid * LD1 + 32 + j] = aT[j][hb][q]; else sW1[hid * LD1 + 32 + j] += aT[j][hb][q];
            }
          }
        }
      if (lane == 0) { if (w == 0) sB[128] = aB2; else sB[128] += aB2; }
    }
  }
  __syncthreads();
  float* P = partials + (size_t)blockIdx.x * per_block;  // [64 IN | 64 | 64 | 1]
  const int nW1 = 64 * IN;
  for (int e = tid; e < nW1; e += 256) P[e] = sW1[(e / IN) * LD1 + (e % IN)];
  if (tid < 64) { P[nW1 + tid] = sB[tid]; P[nW1 + 64 + tid] = sB[64 + tid]; }
  if (tid == 0) P[nW1 + 128] = sB[128];
}

// ---------------------------------------------------------------- backward of the backward, SDF decoder shape
// The mapper's Eikonal / consistency terms differentiate dS/dx once more (utils/tools.py:409-419 `get_gradient` with
// create_graph=True, used at utils/mapper.py:1445-1448): the first-order backward
//     gx[n, :] = gy[n] * sum_j m[n, j] W2[j] W1[j, :]          (m = [W1 x + b1 > 0])
// is itself a graph node, and a loss on gx sends a cotangent a = dL/dgx [N, IN] back through it:
//     ggy[n]    = sum_j m[n, j] W2[j] u[n, j]                  u = a W1^T
//     gW2[j]    = sum_n gy[n] m[n, j] u[n, j]
//     gW1[j, :] = sum_n gy[n] m[n, j] W2[j] a[n, :]
// (nothing reaches x or b1: the mask is piecewise constant, as in torch's own relu).  Same wave-per-tile scheme and
// accumulator layouts as mlp_bwd_wave_h64o1_kernel: product A twice (x for the mask, a for u, sharing the W1
// fragments), the elementwise part in the accumulator layout, gH2^T = (m W2 gy)^T through the wave's private LDS
// tile for gW1 += gH2^T a.  Partials in the layout of the first-order kernel ([64 IN | gW2 64 | 64 zeros | 0]) so
// that mlp_reduce_kernel sums them.
template <int NS, int IB>
__global__ __launch_bounds__(256, 1) void mlp_dbl_wave_h64o1_kernel(long long N, int IN, const float* __restrict__ x,
                                                                    const float* __restrict__ a,
                                                                    const float* __restrict__ gy,
                                                                    const float* __restrict__ W1,
                                                                    const float* __restrict__ b1,
                                                                    const float* __restrict__ W2,
                                                                    float* __restrict__ ggy,
                                                                    float* __restrict__ partials, size_t per_block) {
  constexpr int WI = 32 * IB;
  constexpr int LD1 = WI | 1;
  __shared__ float sW1[64 * LD1];          // W1[hid][i], zero beyond IN; reused for the workgroup's gW1
  __shared__ float sG[4][32 * BW_LD];      // per wave: gH2^T as [hid_local][row]
  __shared__ float sB[64];                 // workgroup sum of gW2
  __shared__ float sW2[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  for (int e = tid; e < 64 * WI; e += 256) {
    const int j = e / WI, i = e - j * WI;
    sW1[j * LD1 + i] = i < IN ? W1[(size_t)j * IN + i] : 0.f;
  }
  if (tid < 64) sW2[tid] = W2[tid];
  float b1f[2];
#pragma unroll
  for (int hb = 0; hb < 2; ++hb) b1f[hb] = h == 0 ? b1[hb * 32 + r] : 0.f;
  __syncthreads();

  f32x16 aW1[2][IB];
  float aW2[2][16];
#pragma unroll
  for (int hb = 0; hb < 2; ++hb)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      aW2[hb][q] = 0.f;
#pragma unroll
      for (int ib = 0; ib < IB; ++ib) aW1[hb][ib][q] = 0.f;
    }
  float* myG = &sG[wave][0];
  const long long ntiles = (N + 31) / 32;
  const long long nwaves = (long long)gridDim.x * 4;
  const float one = h == 0 ? 1.f : 0.f;
  for (long long t = (long long)blockIdx.x * 4 + wave; t < ntiles; t += nwaves) {
    asm volatile("" ::: "memory");  // keep the loop-invariant LDS operands in LDS (as in the first-order kernel)
    const long long row = t * 32 + r;
    const bool ok = row < N;
    float xf[NS], af[NS];
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) {
      const bool in = ok && 2 * s2 + h < IN;
      xf[s2] = in ? x[(size_t)row * IN + 2 * s2 + h] : 0.f;
      af[s2] = in ? a[(size_t)row * IN + 2 * s2 + h] : 0.f;
    }
    const float gyr = ok ? gy[row] : 0.f;
    float acol[IB][16];  // a[row = 16 h + s][column = 32 ib + r]
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      const long long rc = t * 32 + 16 * h + s2;
#pragma unroll
      for (int ib = 0; ib < IB; ++ib)
        acol[ib][s2] = (rc < N && 32 * ib + r < IN) ? a[(size_t)rc * IN + 32 * ib + r] : 0.f;
    }
    float ggy_part = 0.f;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      f32x16 pre = {0}, u = {0};
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) {
        const float w = sW1[(hb * 32 + r) * LD1 + 2 * s2 + h];
        pre = mfma(w, xf[s2], pre);
        u = mfma(w, af[s2], u);
      }
      pre = mfma(b1f[hb], one, pre);
      float gH[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const bool on = pre[q] > 0.f;
        const float w2 = sW2[hb * 32 + rowmap(q, h)];
        gH[q] = on ? w2 * gyr : 0.f;
        aW2[hb][q] = fmaf(on ? u[q] : 0.f, gyr, aW2[hb][q]);
        ggy_part = fmaf(on ? w2 : 0.f, u[q], ggy_part);
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 16; ++q) myG[rowmap(q, h) * BW_LD + r] = gH[q];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) {
        const float aG = myG[r * BW_LD + 16 * h + s2];
#pragma unroll
        for (int ib = 0; ib < IB; ++ib) aW1[hb][ib] = mfma(aG, acol[ib][s2], aW1[hb][ib]);
      }
    }
    ggy_part += __shfl_xor(ggy_part, 32, 64);   // the two lane halves hold disjoint hidden units of the same row
    if (ok && h == 0) ggy[row] = ggy_part;
  }

  // per-lane sums over rows -> sums over the 32 lanes that share h (hidden unit hb*32 + rowmap(q, h))
#pragma unroll
  for (int hb = 0; hb < 2; ++hb)
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) aW2[hb][q] += __shfl_xor(aW2[hb][q], off, 64);
  // the four waves add theirs in wave order: gW1 into the (now dead) W1 image, gW2 into sB
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int hb = 0; hb < 2; ++hb)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int hid = hb * 32 + rowmap(q, h);
#pragma unroll
          for (int ib = 0; ib < IB; ++ib) {
            float* d1 = &sW1[hid * LD1 + 32 * ib + r];
            if (w == 0) *d1 = aW1[hb][ib][q]; else *d1 += aW1[hb][ib][q];
          }
          if (r == 0) { if (w == 0) sB[hid] = aW2[hb][q]; else sB[hid] += aW2[hb][q]; }
        }
    }
  }
  __syncthreads();
  float* P = partials + (size_t)blockIdx.x * per_block;  // [64 IN | gW2 64 | 64 zeros | 0]
  const int nW1 = 64 * IN;
  for (int e = tid; e < nW1; e += 256) P[e] = sW1[(e / IN) * LD1 + (e % IN)];
  if (tid < 64) { P[nW1 + tid] = sB[tid]; P[nW1 + 64 + tid] = 0.f; }
  if (tid == 0) P[nW1 + 128] = 0.f;
}

__global__ void mlp_reduce_kernel(const float* __restrict__ partials, int nblocks, size_t per_block, int IN,
                                  int HID, int OUT, float* __restrict__ gW1, float* __restrict__ gb1,
                                  float* __restrict__ gW2, float* __restrict__ gb2) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= per_block) return;
  // eight independent accumulation chains (fixed assignment -> still deterministic) keep loads in flight
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int b = 0;
  for (; b + 8 <= nblocks; b += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += partials[(size_t)(b + u) * per_block + e];
  }
  for (; b < nblocks; ++b) a[0] += partials[(size_t)b * per_block + e];
  const float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  const size_t nW1 = (size_t)HID * IN, nW2 = (size_t)OUT * HID;
  if (e < nW1) gW1[e] = s;
  else if (e < nW1 + nW2) gW2[e - nW1] = s;
  else if (e < nW1 + nW2 + HID) gb1[e - nW1 - nW2] = s;
  else gb2[e - nW1 - nW2 - HID] = s;
}

__global__ void mlp_reduce_grouped_kernel(MlpJobs j, int HID) {
  const int g = blockIdx.y;
  const int nblocks = j.wg0[g + 1] - j.wg0[g];
  const size_t per_block = j.per_block[g];
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= per_block) return;
  const float* partials = j.partials[g];
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int b = 0;
  for (; b + 8 <= nblocks; b += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += partials[(size_t)(b + u) * per_block + e];
  }
  for (; b < nblocks; ++b) a[0] += partials[(size_t)b * per_block + e];
  const float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  const size_t nW1 = (size_t)HID * j.IN[g], nW2 = (size_t)j.OUT[g] * HID;
  if (e < nW1) j.gW1[g][e] = s;
  else if (e < nW1 + nW2) j.gW2[g][e - nW1] = s;
  else if (e < nW1 + nW2 + HID) j.gb1[g][e - nW1 - nW2] = s;
  else j.gb2[g][e - nW1 - nW2 - HID] = s;
}

int check_dims(int64_t N, int IN, int HID, int OUT) {
  PINGS_ARG_CHECK(N >= 0, "negative N");
  PINGS_ARG_CHECK(IN > 0 && IN <= MAX_INP, "IN must be in 1..64");
  PINGS_ARG_CHECK(HID > 0 && HID <= 128 && HID % 32 == 0, "HID must be 32, 64, 96 or 128");
  PINGS_ARG_CHECK(OUT > 0 && OUT <= OUTP, "OUT must be in 1..32");
  return PINGS_OK;
}

constexpr int MAX_BWD_BLOCKS = 1024;  // per-workgroup partials (generic kernel) or per-wave partials (wave kernel)

size_t fwd_lds_bytes(const Dims& d) {
  const int NW = d.HID / 32;
  return sizeof(float) * ((size_t)d.HID * d.ldw1 + (size_t)OUTP * d.ldw2 + (size_t)TR * d.ldx +
                          (size_t)NW * OUTP * (TR + 1));
}

size_t bwd_lds_bytes(const Dims& d) {
  const int NW = d.HID / 32, NIB = (d.INP + 31) / 32;
  size_t tail = (size_t)2 * NW * 32 * (TR + 1) + (size_t)NW * NIB * 32 * (TR + 1);
  const size_t red = (size_t)NW * 64 * 17;  // gb1 reduction reuses the sHT/sGH region
  if (tail < red) tail = red;
  return sizeof(float) * ((size_t)d.HID * d.ldw1 + (size_t)OUTP * d.ldw2 + (size_t)TR * d.ldx +
                          (size_t)TR * d.ldg + tail);
}

}  // namespace

PINGS_API size_t pings_mlp_backward_scratch_bytes(int IN, int HID, int OUT) {
  if (IN <= 0 || HID <= 0 || OUT <= 0) return 0;
  return sizeof(float) * partial_floats(IN, HID, OUT) * MAX_BWD_BLOCKS;
}

PINGS_API int pings_mlp_forward(const float* x, int64_t N, int IN, int HID, int OUT, const float* W1,
                                const float* b1, const float* W2, const float* b2, float* y,
                                void* stream) {
  if (int e = check_dims(N, IN, HID, OUT)) return e;
  if (N == 0) return PINGS_OK;
  PINGS_ARG_CHECK(x && W1 && b1 && W2 && b2 && y, "null pointer");
  hipStream_t st = pings::as_stream(stream);
  const Dims d = make_dims(N, IN, HID, OUT);
  const long long ntiles = (N + TR - 1) / TR;
  if (HID == 128 && IN <= 32 && getenv("PINGS_MLP_FWD_WG") == nullptr) {
    // wave-per-tile kernel: 2 workgroups of 4 waves per CU (register-resident weights: 2 waves per SIMD)
    pings::prof::Scope ps("mlp_fwd", st);
    const long long want = (ntiles + 3) / 4;
    const unsigned grid_w = (unsigned)(want < 512 ? want : 512);
    hipLaunchKernelGGL(mlp_fwd_wave_kernel, dim3(grid_w), dim3(256), 0, st, (long long)N, IN, OUT, x, W1, b1, W2, b2, y);
    PINGS_LAUNCH_CHECK();
    return PINGS_OK;
  }
  if (HID == 64 && OUT == 1 && IN <= 35 && getenv("PINGS_MLP_FWD_WG") == nullptr) {
    // the SDF decoder (Decoder.sdf): column-per-lane kernel, two workgroups of four waves per CU
    pings::prof::Scope ps("mlp_fwd", st);
    const long long want = (ntiles + 3) / 4;
    const unsigned grid_w = (unsigned)(want < 512 ? want : 512);
#define PINGS_MLP_H64O1(KS_)                                                                                       \
  hipLaunchKernelGGL(mlp_fwd_h64o1_kernel<KS_>, dim3(grid_w), dim3(256), 0, st, (long long)N, IN, x, W1, b1, W2, b2, y)
    if (IN + 1 <= 12) PINGS_MLP_H64O1(6);
    else if (IN + 1 <= 20) PINGS_MLP_H64O1(10);
    else PINGS_MLP_H64O1(18);
#undef PINGS_MLP_H64O1
    PINGS_LAUNCH_CHECK();
    return PINGS_OK;
  }
  const unsigned grid = (unsigned)(ntiles < 512 ? ntiles : 512);  // two resident workgroups per CU: weights are staged once each
  const size_t lds = fwd_lds_bytes(d);
  // prefetch registers per thread: ceil(32 * INP / threads), in three size classes
  const int need = (TR * d.INP + 64 * (HID / 32) - 1) / (64 * (HID / 32));
  pings::prof::Scope ps("mlp_fwd", st);
#define PINGS_MLP_FWD(PX)                                                                                   \
  do {                                                                                                      \
    PINGS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fwd_kernel<PX>),                  \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));            \
    hipLaunchKernelGGL(mlp_fwd_kernel<PX>, dim3(grid), dim3(64 * (HID / 32)), lds, st, d, x, W1, b1, W2, b2, y); \
  } while (0)
  if (need <= 5) PINGS_MLP_FWD(5);
  else if (need <= 10) PINGS_MLP_FWD(10);
  else if (need <= 18) PINGS_MLP_FWD(18);
  else PINGS_MLP_FWD(32);
#undef PINGS_MLP_FWD
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}

PINGS_API int pings_mlp_backward(const float* x, const float* dL_dy, int64_t N, int IN, int HID,
                                 int OUT, const float* W1, const float* b1, const float* W2,
                                 void* scratch, float* dL_dx, float* dL_dW1, float* dL_db1,
                                 float* dL_dW2, float* dL_db2, void* stream) {
  if (int e = check_dims(N, IN, HID, OUT)) return e;
  PINGS_ARG_CHECK(W1 && b1 && W2 && scratch && dL_dW1 && dL_db1 && dL_dW2 && dL_db2, "null pointer");
  hipStream_t st = pings::as_stream(stream);
  const size_t per_block = partial_floats(IN, HID, OUT);
  if (N == 0) {
    PINGS_HIP_CHECK(hipMemsetAsync(dL_dW1, 0, sizeof(float) * HID * IN, st));
    PINGS_HIP_CHECK(hipMemsetAsync(dL_dW2, 0, sizeof(float) * OUT * HID, st));
    PINGS_HIP_CHECK(hipMemsetAsync(dL_db1, 0, sizeof(float) * HID, st));
    PINGS_HIP_CHECK(hipMemsetAsync(dL_db2, 0, sizeof(float) * OUT, st));
    return PINGS_OK;
  }
  PINGS_ARG_CHECK(x && dL_dy, "null pointer");
  const Dims d = make_dims(N, IN, HID, OUT);
  const long long ntiles = (N + TR - 1) / TR;
  if (HID == 128 && IN <= 32 && getenv("PINGS_MLP_BWD_WG") == nullptr) {
    pings::prof::Scope ps("mlp_bwd", st);
    const long long want = (ntiles + 3) / 4;
    const int grid_w = (int)(want < 256 ? want : 256);     // one workgroup (four independent waves) per CU
    hipLaunchKernelGGL(mlp_bwd_wave_kernel, dim3(grid_w), dim3(256), 0, st, (long long)N, IN, OUT, x, dL_dy, W1, b1, W2,
                       dL_dx, reinterpret_cast<float*>(scratch), per_block);
    PINGS_LAUNCH_CHECK();
    hipLaunchKernelGGL(mlp_reduce_kernel, dim3((unsigned)pings::ceil_div<size_t>(per_block, 256)), dim3(256), 0,
                       st, reinterpret_cast<const float*>(scratch), grid_w, per_block, IN, HID, OUT, dL_dW1,
                       dL_db1, dL_dW2, dL_db2);
    PINGS_LAUNCH_CHECK();
    return PINGS_OK;
  }
  if (HID == 64 && OUT == 1 && getenv("PINGS_MLP_BWD_WG") == nullptr) {
    pings::prof::Scope ps("mlp_bwd", st);
    const long long want = (ntiles + 3) / 4;
    const int grid_w = (int)(want < 256 ? want : 256);
#define PINGS_H64O1(NS_, IB_, NT_)                                                                                 \
  hipLaunchKernelGGL((mlp_bwd_wave_h64o1_kernel<NS_, IB_, NT_>), dim3(grid_w), dim3(256), 0, st, (long long)N, IN, x, \
                     dL_dy, W1, b1, W2, dL_dx, reinterpret_cast<float*>(scratch), per_block)
    if (IN <= 12) PINGS_H64O1(6, 1, 0);
    else if (IN <= 32) PINGS_H64O1(16, 1, 0);
    else if (IN <= 36) PINGS_H64O1(18, 2, 0);   // <18, 1, 3> (tail inputs on the vector ALU) measured no faster
    else PINGS_H64O1(32, 2, 0);
#undef PINGS_H64O1
    PINGS_LAUNCH_CHECK();
    hipLaunchKernelGGL(mlp_reduce_kernel, dim3((unsigned)pings::ceil_div<size_t>(per_block, 256)), dim3(256), 0,
                       st, reinterpret_cast<const float*>(scratch), grid_w, per_block, IN, HID, OUT, dL_dW1,
                       dL_db1, dL_dW2, dL_db2);
    PINGS_LAUNCH_CHECK();
    return PINGS_OK;
  }
  const int grid = (int)(ntiles < 512 ? ntiles : 512);
  const size_t lds = bwd_lds_bytes(d);
  const int nthr = 64 * (HID / 32);
  const int need = (TR * d.INP + nthr - 1) / nthr;   // x prefetch registers per thread
  pings::prof::Scope ps("mlp_bwd", st);
#define PINGS_MLP_BWD(PX, PG)                                                                              \
  do {                                                                                                     \
    PINGS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bwd_kernel<PX, PG>),             \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));           \
    hipLaunchKernelGGL((mlp_bwd_kernel<PX, PG>), dim3(grid), dim3(nthr), lds, st, d, x, dL_dy, W1, b1, W2,  \
                       dL_dx, reinterpret_cast<float*>(scratch));                                         \
  } while (0)
  // gY prefetch registers: 32*32 / threads = 16, 8, 6, 4 for 1..4 waves
  if (HID == 128) { if (need <= 5) PINGS_MLP_BWD(5, 4); else if (need <= 9) PINGS_MLP_BWD(9, 4); else PINGS_MLP_BWD(16, 4); }
  else if (HID == 96) { if (need <= 6) PINGS_MLP_BWD(6, 6); else PINGS_MLP_BWD(11, 6); }
  else if (HID == 64) { if (need <= 9) PINGS_MLP_BWD(9, 8); else PINGS_MLP_BWD(16, 8); }
  else { if (need <= 18) PINGS_MLP_BWD(18, 16); else PINGS_MLP_BWD(32, 16); }
#undef PINGS_MLP_BWD
  PINGS_LAUNCH_CHECK();
  hipLaunchKernelGGL(mlp_reduce_kernel, dim3((unsigned)pings::ceil_div<size_t>(per_block, 256)), dim3(256), 0,
                     st, reinterpret_cast<const float*>(scratch), grid, per_block, IN, HID, OUT, dL_dW1,
                     dL_db1, dL_dW2, dL_db2);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}

PINGS_API int pings_mlp_double_backward_supported(int IN, int HID, int OUT) {
  return (HID == 64 && OUT == 1 && IN > 0 && IN <= 64) ? 1 : 0;
}

PINGS_API int pings_mlp_double_backward(const float* x, const float* ddx, const float* dL_dy, int64_t N, int IN,
                                        int HID, int OUT, const float* W1, const float* b1, const float* W2,
                                        void* scratch, float* d_dy, float* d_W1, float* d_W2, void* stream) {
  if (int e = check_dims(N, IN, HID, OUT)) return e;
  PINGS_ARG_CHECK(pings_mlp_double_backward_supported(IN, HID, OUT), "double backward: hidden 64, one output only");
  PINGS_ARG_CHECK(W1 && b1 && W2 && d_W1 && d_W2 && scratch, "null pointer");
  hipStream_t st = pings::as_stream(stream);
  if (N == 0) {
    PINGS_HIP_CHECK(hipMemsetAsync(d_W1, 0, sizeof(float) * HID * IN, st));
    PINGS_HIP_CHECK(hipMemsetAsync(d_W2, 0, sizeof(float) * OUT * HID, st));
    return PINGS_OK;
  }
  PINGS_ARG_CHECK(x && ddx && dL_dy && d_dy, "null pointer");
  const size_t per_block = partial_floats(IN, HID, OUT);
  const long long ntiles = (N + TR - 1) / TR, want = (ntiles + 3) / 4;
  const int grid_w = (int)(want < 256 ? want : 256);
  // scratch: MAX_BWD_BLOCKS partials (pings_mlp_backward_scratch_bytes), then 65 floats that take the reduce kernel's
  // (all-zero) gb1 / gb2 columns
  float* part = reinterpret_cast<float*>(scratch);
  float* dummy = part + (size_t)MAX_BWD_BLOCKS * per_block - 72;   // the launch uses at most 256 of the 1024 partials
  pings::prof::Scope ps("mlp_dbl", st);
#define PINGS_DBL(NS_, IB_)                                                                                     \
  hipLaunchKernelGGL((mlp_dbl_wave_h64o1_kernel<NS_, IB_>), dim3(grid_w), dim3(256), 0, st, (long long)N, IN, x,  \
                     ddx, dL_dy, W1, b1, W2, d_dy, part, per_block)
  if (IN <= 12) PINGS_DBL(6, 1);
  else if (IN <= 32) PINGS_DBL(16, 1);
  else if (IN <= 36) PINGS_DBL(18, 2);
  else PINGS_DBL(32, 2);
#undef PINGS_DBL
  PINGS_LAUNCH_CHECK();
  hipLaunchKernelGGL(mlp_reduce_kernel, dim3((unsigned)pings::ceil_div<size_t>(per_block, 256)), dim3(256), 0, st,
                     (const float*)part, grid_w, per_block, IN, HID, OUT, d_W1, dummy, d_W2, dummy + 64);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}

#ifdef PINGS_MLP_STATS
PINGS_API int pings_debug_mlp_stats(unsigned long long* out8, int reset) {
  PINGS_HIP_CHECK(hipDeviceSynchronize());
  PINGS_HIP_CHECK(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_mlp_stats), 64));
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    PINGS_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_mlp_stats), z, 64));
  }
  return PINGS_OK;
}
// out2 = {shader-clock cycles, 100 MHz real-time ticks} of workgroup 0 of the last grouped backward launch: the clock
// the kernel really ran at = 100 MHz x out2[0] / out2[1]
PINGS_API int pings_debug_mlp_clock(unsigned long long* out2) {
  PINGS_HIP_CHECK(hipDeviceSynchronize());
  PINGS_HIP_CHECK(hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_mlp_clock), 16));
  return PINGS_OK;
}
#endif

// ---------------------------------------------------------------- grouped launches (several decoders, same rows)
namespace {
int fill_jobs(const pings_mlp_job* jobs, int njobs, MlpJobs& J) {
  PINGS_ARG_CHECK(jobs && njobs > 0 && njobs <= MAX_JOBS, "1..8 jobs");
  for (int g = 0; g < njobs; ++g) {
    const pings_mlp_job& q = jobs[g];
    PINGS_ARG_CHECK(q.IN > 0 && q.IN <= 32 && q.OUT > 0 && q.OUT <= OUTP, "grouped MLP: IN <= 32, OUT <= 32");
    PINGS_ARG_CHECK(q.x && q.W1 && q.b1 && q.W2 && q.b2, "null pointer in job");
    J.x[g] = q.x; J.W1[g] = q.W1; J.b1[g] = q.b1; J.W2[g] = q.W2; J.b2[g] = q.b2; J.y[g] = q.y;
    J.gy[g] = q.dL_dy; J.gx[g] = q.dL_dx; J.gW1[g] = q.dL_dW1; J.gb1[g] = q.dL_db1; J.gW2[g] = q.dL_dW2;
    J.gb2[g] = q.dL_db2; J.IN[g] = q.IN; J.OUT[g] = q.OUT;
    J.per_block[g] = partial_floats(q.IN, 128, q.OUT);
    J.partials[g] = nullptr;
  }
  return PINGS_OK;
}
}  // namespace

PINGS_API int pings_mlp_forward_grouped(const pings_mlp_job* jobs, int njobs, int64_t N, void* stream) {
  return pings_mlp_forward_grouped_dyn(jobs, njobs, N, nullptr, stream);
}

PINGS_API int pings_mlp_forward_grouped_dyn(const pings_mlp_job* jobs, int njobs, int64_t N, const int32_t* n_rows_dev,
                                            void* stream) {
  MlpJobs J;
  if (int e = fill_jobs(jobs, njobs, J)) return e;
  PINGS_ARG_CHECK(N >= 0, "negative N");
  if (N == 0) return PINGS_OK;
  for (int g = 0; g < njobs; ++g) PINGS_ARG_CHECK(J.y[g] != nullptr, "null output");
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope ps("mlp_fwd", st);
  // two resident workgroups per CU over ALL jobs together: every workgroup stages its weights once and then walks
  // njobs times more tiles than in a per-decoder launch of 512 workgroups
  const long long ntiles = (N + TR - 1) / TR, want = (ntiles + 3) / 4;
  const long long cap = 512 / njobs;   // floor: one workgroup too many would run alone in a second round
  const unsigned grid_w = (unsigned)(want < cap ? want : cap);
  hipLaunchKernelGGL(mlp_fwd_wave_grouped_kernel, dim3(grid_w, njobs), dim3(256), 0, st, (long long)N, J,
                     (const int*)n_rows_dev);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}

PINGS_API size_t pings_mlp_backward_grouped_scratch_bytes(const pings_mlp_job* jobs, int njobs) {
  size_t total = 0;
  if (!jobs) return 0;
  for (int g = 0; g < njobs; ++g) total += sizeof(float) * partial_floats(jobs[g].IN, 128, jobs[g].OUT) * 256;
  return total;
}

PINGS_API int pings_mlp_backward_grouped(const pings_mlp_job* jobs, int njobs, int64_t N, void* scratch,
                                         void* stream) {
  MlpJobs J;
  if (int e = fill_jobs(jobs, njobs, J)) return e;
  PINGS_ARG_CHECK(N > 0 && scratch, "grouped backward needs rows and scratch");
  float* p = reinterpret_cast<float*>(scratch);
  size_t max_pb = 0;
  for (int g = 0; g < njobs; ++g) {
    PINGS_ARG_CHECK(J.gy[g] && J.gW1[g] && J.gb1[g] && J.gW2[g] && J.gb2[g], "null gradient pointer in job");
    J.partials[g] = p;
    p += J.per_block[g] * 256;
    if (J.per_block[g] > max_pb) max_pb = J.per_block[g];
  }
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope ps("mlp_bwd", st);
  // one resident workgroup per CU over all jobs together (see the forward), split between the jobs in proportion to
  // their MFMAs per tile: 4 x (17 + OH + 16 + 32), OH = k-steps of product B (mlp_bwd_wave_dispatch)
  // PINGS_MLP_BWD_WAVES = 2: the two-waves-per-SIMD kernel (mlp_bwd_wave2_body).  Measured equal to the default
  // one-wave kernel at every size (125k points: 0.255 vs 0.247 ms; slope 1.70 vs 1.71 us per 1000 points) — both sit at
  // ~83 % of the matrix pipe at the clock the launch really runs at (2.1-2.2 GHz by s_memtime / s_memrealtime, not the
  // data sheet's 2.4) — so the default stays the kernel whose partials are bit-identical to the single launches.
  const char* wenv = getenv("PINGS_MLP_BWD_WAVES");
  const int wps = wenv && atoi(wenv) == 2 ? 2 : 1;
  const int wpw = 4 * wps;   // waves per workgroup
  const long long ntiles = (N + TR - 1) / TR, want = (ntiles + wpw - 1) / wpw;
  int cost[MAX_JOBS];
  for (int g = 0; g < njobs; ++g) {
    const bool vecg = (J.OUT[g] % 4 == 0) && ((reinterpret_cast<uintptr_t>(J.gy[g]) & 15) == 0);
    const int oh = !vecg ? 16 : (J.OUT[g] == 24 ? 12 : (J.OUT[g] == 8 ? 4 : 16));
    cost[g] = 17 + oh + 16 + 32;
  }
  // 256 workgroups (one per CU) over the jobs so that the LAST wave finishes as early as possible: a job with w
  // workgroups needs ceil(ntiles / 4w) rounds of cost[g] MFMAs.  Greedy on the maximum (one more workgroup to the job
  // that finishes last) is optimal for a minimum over non-increasing step functions.  Shares merely proportional to
  // the costs left 5.9 % on the table at 125k points: 3907 tiles over 54 x 4 waves are 18.09 -> 19 rounds.
  int share[MAX_JOBS];
  for (int g = 0; g < njobs; ++g) share[g] = 1;
  for (int left = 256 - njobs; left > 0; --left) {
    int worst = -1;
    long long worst_t = -1;
    for (int g = 0; g < njobs; ++g) {
      const long long t_g = ((ntiles + (long long)wpw * share[g] - 1) / ((long long)wpw * share[g])) * cost[g];
      if (t_g > worst_t) { worst_t = t_g; worst = g; }
    }
    if (share[worst] >= want) break;      // one tile per wave already: more workgroups would idle
    ++share[worst];
  }
  J.wg0[0] = 0;
  for (int g = 0; g < njobs; ++g) J.wg0[g + 1] = J.wg0[g] + share[g];
  const int grid_w = J.wg0[njobs];
  if (wps == 2)
    hipLaunchKernelGGL(mlp_bwd_wave2_grouped_kernel, dim3(grid_w), dim3(512), 0, st, (long long)N, J, njobs);
  else
    hipLaunchKernelGGL(mlp_bwd_wave_grouped_kernel, dim3(grid_w), dim3(256), 0, st, (long long)N, J, njobs);
  PINGS_LAUNCH_CHECK();
  hipLaunchKernelGGL(mlp_reduce_grouped_kernel, dim3((unsigned)pings::ceil_div<size_t>(max_pb, 256), njobs), dim3(256),
                     0, st, J, 128);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}
