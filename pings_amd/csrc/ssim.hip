// Fused SSIM forward / backward for gfx950.
//
// Semantics follow the reference's in-tree torch SSIM
// (gaussian_splatting/utils/loss_utils.py:189-219; window loss_utils.py:53-55,
// 182-186): depth-wise 11x11 Gaussian (sigma 1.5, normalised) convolution with
// zero padding 5, C1 = 0.01^2, C2 = 0.03^2, mean over every element.  The 2-D
// window is an outer product, so both passes here are separable 11-tap
// convolutions staged through LDS.
//
// Layout: images are [planes, H, W] fp32.  One 256-thread workgroup produces a
// 32x32 output tile of one plane: the 42x42 halo tile of both images is staged
// in LDS (coalesced row loads), the horizontal pass leaves five running
// statistics per (row, col) in LDS, the vertical pass finishes them per pixel.
// HBM traffic per pixel: fwd 8 B read (+12 B written when train), bwd 20 B read
// + 4 B written; the kernel is HBM/L2-stream bound, the 1.72x halo re-read is
// served by L2.
#include <cstdlib>

#include "common.hpp"

namespace {

constexpr int TS = 32;            // output tile edge
constexpr int HALO = 5;           // window radius
constexpr int IN = TS + 2 * HALO; // 42
constexpr int INP = IN + 1;       // padded LDS row
constexpr int NT = 256;
constexpr int PERSISTENT_WGS = 256 * 3;   // three workgroups per CU fit by LDS (42 KB each)

// Which tiles a persistent workgroup walks.  Workgroups b and b + 8 share an XCD and with it an L2 (observed dispatch,
// MI355X_MICROARCH.md: a speed matter only); tiles are numbered row-major.  Giving every XCD label a contiguous band
// of tiles keeps neighbouring tiles — whose 42 x 42 halo tiles overlap by 5 pixels on every side, 1.72x the bytes of
// the image — in ONE L2; dealt round-robin (tile t to workgroup t % grid) every neighbour sat behind a different L2
// and the overlap came from HBM again (PMC, r03: forward 96.5 MB fetched for 49.8 MB of images).
struct TileWalk { long long first, end, step; };
__device__ inline TileWalk tile_walk(long long ntiles) {
  const long long g = gridDim.x, b = blockIdx.x;
  if ((g & 7) != 0 || ntiles < 8 * 8) return {b, ntiles, g};
  const long long band = (ntiles + 7) / 8, lo = (b & 7) * band;
  const long long hi = lo + band < ntiles ? lo + band : ntiles;
  return {lo + (b >> 3), hi, g >> 3};
}

// exp(-(k-5)^2 / (2*1.5^2)) / sum, k = 0..10, rounded to fp32.
__device__ __constant__ float kWin[11] = {
    0.00102838012f, 0.00759875821f, 0.0360007733f, 0.109360687f, 0.213005528f, 0.266011715f,
    0.213005528f, 0.109360687f, 0.0360007733f, 0.00759875821f, 0.00102838012f};

constexpr float kC1 = 0.01f * 0.01f;
constexpr float kC2 = 0.03f * 0.03f;

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

template <bool TRAIN>
__global__ __launch_bounds__(NT, 3) void ssim_fwd_kernel(
    const float* __restrict__ img1, const float* __restrict__ img2, int planes, int H, int W,
    float* __restrict__ dm_dmu1, float* __restrict__ dm_dsigma1_sq,
    float* __restrict__ dm_dsigma12, float* __restrict__ partials) {
  __shared__ float sA[IN][INP];
  __shared__ float sB[IN][INP];
  __shared__ float sH[5][IN][TS + 1];
  __shared__ float sRed[NT / 64];

  const int tid = threadIdx.x;
  constexpr int PF = (IN * IN + NT - 1) / NT;   // halo-tile elements per thread
  const int GX = (W + TS - 1) / TS, GY = (H + TS - 1) / TS;
  const long long ntiles = (long long)GX * GY * planes;
  // Persistent workgroups over the tiles with a register prefetch: the next tile's halo tile is in flight while the
  // current one is convolved (first version: one tile per workgroup, three workgroups per CU by LDS, every tile's
  // global loads exposed: 0.092 ms forward at 1080p x 3 against a 0.04 ms traffic floor).
  float ra[PF], rb[PF];
  auto fetch = [&](long long t) {
    const int bx = (int)(t % GX), by = (int)((t / GX) % GY);
    const size_t off = (size_t)(t / ((long long)GX * GY)) * H * W;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int i = tid + u * NT;
      const int r = i / IN, c = i - r * IN;
      const int gy = by * TS + r - HALO, gx = bx * TS + c - HALO;
      const bool in = i < IN * IN && gy >= 0 && gy < H && gx >= 0 && gx < W;
      ra[u] = in ? img1[off + (size_t)gy * W + gx] : 0.f;
      rb[u] = in ? img2[off + (size_t)gy * W + gx] : 0.f;
    }
  };
  const TileWalk tw = tile_walk(ntiles);
  if (tw.first < tw.end) fetch(tw.first);
  for (long long t = tw.first; t < tw.end; t += tw.step) {
    const int x0 = (int)(t % GX) * TS, y0 = (int)((t / GX) % GY) * TS;
    const size_t plane_off = (size_t)(t / ((long long)GX * GY)) * H * W;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int i = tid + u * NT;
      if (i < IN * IN) {
        const int r = i / IN, c = i - r * IN;
        sA[r][c] = ra[u];
        sB[r][c] = rb[u];
      }
    }
    __syncthreads();
    if (t + tw.step < tw.end) fetch(t + tw.step);

    // horizontal pass: 42 rows x 32 cols, five statistics each.  A thread owns FOUR adjacent output columns of a row:
    // the 14 inputs they share are read from LDS once (7 reads per output and image instead of 22; the pass was
    // LDS-read bound: 84 ds_read_b32 per pixel over both passes).  Every output still accumulates its 11 taps in
    // the order k = 0..10 with the same operations, so the statistics are bit-identical to the one-output form.
    for (int i = tid; i < IN * (TS / 4); i += NT) {
      const int r = i / (TS / 4), c0 = 4 * (i - r * (TS / 4));
      float m1[4] = {0.f, 0.f, 0.f, 0.f}, m2[4] = {0.f, 0.f, 0.f, 0.f}, s11[4] = {0.f, 0.f, 0.f, 0.f},
            s22[4] = {0.f, 0.f, 0.f, 0.f}, s12[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 14; ++j) {
        const float a = sA[r][c0 + j], b = sB[r][c0 + j];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const int k = j - o;
          if (k >= 0 && k < 11) {
            const float w = kWin[k];
            const float wa = w * a, wb = w * b;
            m1[o] += wa;
            m2[o] += wb;
            s11[o] = fmaf(wa, a, s11[o]);
            s22[o] = fmaf(wb, b, s22[o]);
            s12[o] = fmaf(wa, b, s12[o]);
          }
        }
      }
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        sH[0][r][c0 + o] = m1[o];
        sH[1][r][c0 + o] = m2[o];
        sH[2][r][c0 + o] = s11[o];
        sH[3][r][c0 + o] = s22[o];
        sH[4][r][c0 + o] = s12[o];
      }
    }
    __syncthreads();

    // vertical pass: a thread owns FOUR adjacent rows of one column (32 columns x 8 row groups = the 256 threads): 14
    // rows of the five statistics are read once for four outputs (17.5 LDS reads per output instead of 55), taps
    // again accumulated in the order k = 0..10
    float local = 0.f;
    float vmu1[4] = {0.f, 0.f, 0.f, 0.f}, vmu2[4] = {0.f, 0.f, 0.f, 0.f}, ve11[4] = {0.f, 0.f, 0.f, 0.f},
          ve22[4] = {0.f, 0.f, 0.f, 0.f}, ve12[4] = {0.f, 0.f, 0.f, 0.f};
    const int vc = tid & (TS - 1), vr0 = 4 * (tid / TS);
#pragma unroll
    for (int j = 0; j < 14; ++j) {
      const float h0 = sH[0][vr0 + j][vc], h1 = sH[1][vr0 + j][vc], h2 = sH[2][vr0 + j][vc],
                  h3 = sH[3][vr0 + j][vc], h4 = sH[4][vr0 + j][vc];
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int k = j - o;
        if (k >= 0 && k < 11) {
          const float w = kWin[k];
          vmu1[o] = fmaf(w, h0, vmu1[o]);
          vmu2[o] = fmaf(w, h1, vmu2[o]);
          ve11[o] = fmaf(w, h2, ve11[o]);
          ve22[o] = fmaf(w, h3, ve22[o]);
          ve12[o] = fmaf(w, h4, ve12[o]);
        }
      }
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const int r = vr0 + o, c = vc;
      const int gy = y0 + r, gx = x0 + c;
      if (gy >= H || gx >= W) continue;
      const float mu1 = vmu1[o], mu2 = vmu2[o], e11 = ve11[o], e22 = ve22[o], e12 = ve12[o];
      const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
      const float sig1 = e11 - mu1_sq, sig2 = e22 - mu2_sq, sig12 = e12 - mu12;
      const float A1 = 2.f * mu12 + kC1;
      const float A2 = 2.f * sig12 + kC2;
      const float B1 = mu1_sq + mu2_sq + kC1;
      const float B2 = sig1 + sig2 + kC2;
      const float inv_B1 = __builtin_amdgcn_rcpf(B1), inv_B2 = __builtin_amdgcn_rcpf(B2);   // as ssim_fwd_sw_kernel
      const float m = (A1 * A2) * (inv_B1 * inv_B2);
      local += m;
      if (TRAIN) {
        // partials of the map w.r.t. (mu1, E[x^2], E[xy]) at this pixel; the
        // chain through sigma1^2 = E[x^2]-mu1^2 and sigma12 = E[xy]-mu1*mu2 is
        // folded into d/dmu1.
        const float d_s1 = -m * inv_B2;               // dm/dsigma1_sq
        const float d_s12 = 2.f * A1 * inv_B1 * inv_B2;  // dm/dsigma12
        const float d_mu1 = 2.f * mu2 * A2 * inv_B1 * inv_B2 - 2.f * mu1 * m * inv_B1 -
                            2.f * mu1 * d_s1 - mu2 * d_s12;
        const size_t o = plane_off + (size_t)gy * W + gx;
        dm_dmu1[o] = d_mu1;
        dm_dsigma1_sq[o] = d_s1;
        dm_dsigma12[o] = d_s12;
      }
    }
    local = wave_sum(local);
    if ((tid & 63) == 0) sRed[tid >> 6] = local;
    __syncthreads();
    if (tid == 0) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NT / 64; ++i) s += sRed[i];
      partials[t] = s;
    }
    __syncthreads();   // sA / sB / sH / sRed are rewritten by the next tile
    }
  }

// ---------------------------------------------------------------- forward, sliding window (round 4)
// The tile kernel above spends most of its 72 us in its phase structure (stage -> barrier -> horizontal -> barrier ->
// vertical, three workgroups per CU by the 42 KB of LDS each), not in memory: PMC traffic is 1.02x the algorithmic
// bytes.  Here there is no LDS image and no barrier.  A WAVE owns a strip of SW_OUT = 54 output columns (lanes 5..58;
// all 64 lanes hold input columns x0 - 5 .. x0 + 58) and walks a band of rows top to bottom:
//   horizontal  the 11 taps of a row come from the neighbouring lanes through full-wave DPP shifts of the two input
//               values (ten shifts per image), every output accumulates its taps in the order k = 0..10 with the tile
//               kernel's operations;
//   vertical    an input row's five statistics are added at once into the eleven output rows it belongs to (eleven
//               accumulator sets in registers, slots static through an 11-fold unroll); output row r receives its taps
//               in the order k = 0..10 as well, so every statistic is bit-identical to the tile kernel's;
//   loads       one 256-byte row segment per image and input row, issued four rows ahead into a register ring.
// Per input row and lane: 2 loads, 20 DPP moves, ~130 vector operations.
constexpr int SW_OUT = 64 - 2 * HALO;   // output columns per wave
typedef float v2f __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ inline float dpp_wave(float v) {   // 0x138 wave_shr:1 (lane l takes lane l - 1), 0x130 wave_shl:1; edge lanes take 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

template <bool TRAIN>
__global__ __launch_bounds__(256) void ssim_fwd_sw_kernel(
    const float* __restrict__ img1, const float* __restrict__ img2, int planes, int H, int W, int RB, int strips,
    int bands, int alt, float* __restrict__ dm_dmu1, float* __restrict__ dm_dsigma1_sq, float* __restrict__ dm_dsigma12,
    float* __restrict__ partials) {
  const int lane = threadIdx.x & 63;
  // units = (plane, band, strip), strips fastest; the four waves of a workgroup take four neighbouring strips of one
  // band, and every XCD label (blockIdx & 7) a contiguous range of workgroups, so that the 10-column / 10-row overlaps
  // of neighbouring units meet in one L2 (see tile_walk above)
  const long long units = (long long)planes * bands * strips;
  const long long nblk = gridDim.x;
  long long blk = blockIdx.x;
  if ((nblk & 7) == 0) blk = (blk & 7) * (nblk >> 3) + (blk >> 3);
  const long long unit = blk * 4 + (threadIdx.x >> 6);
  if (unit >= units) return;
  const int strip = (int)(unit % strips), band = (int)((unit / strips) % bands);
  const size_t plane_off = (size_t)(unit / ((long long)strips * bands)) * H * W;
  const int x = strip * SW_OUT + lane - HALO;           // this lane's input (and, for lanes 5..58, output) column
  const bool x_in = x >= 0 && x < W;
  const bool x_out = lane >= HALO && lane < 64 - HALO && x < W;
  const int yb = band * RB, ye = min(H, yb + RB);       // output rows [yb, ye)
  // Odd bands walk bottom-up (alt): a band's last rows are then its lower neighbour's FIRST rows' neighbours in time as
  // well as in space, so the ten halo rows two bands share are read by both within a few row steps and the second read
  // hits the XCD's L2 (walking every band top-down, band i reads them ~RB steps after band i + 1 did: 40 us and 8 MB of
  // other rows later — PMC: 1.23x the algorithmic bytes).  The window is symmetric, so a reversed band applies the same
  // eleven weights; its outputs add their taps in the opposite row order (fp32 association: ~1e-7 relative).
  const bool up = alt && (band & 1);
  const int y_first = up ? ye + HALO - 1 : yb - HALO, y_step = up ? -1 : 1;      // input row of step t: y_first + t y_step
  const int r_first = up ? ye + 2 * HALO - 1 : yb - 2 * HALO;                    // output row finished at step t
  // wave-uniform plane bases and 32-bit element offsets (one plane holds fewer than 2^31 pixels): the loads and stores
  // take the scalar-base form and the per-row address is one 32-bit multiply-add, not a 64-bit chain per access
  const float* const p1 = img1 + plane_off;
  const float* const p2 = img2 + plane_off;
  const int xc = x_in ? x : 0;

  constexpr int PFD = 4;     // input rows in flight ahead of the one being filtered
  v2f rab[11];               // (img1, img2) of input row t in slot t % 11; only PFD of them are live at a time
  // pending output rows: slot (t + 5 - k) % 11 for the row that takes tap k of input row t; (mu1, mu2), (e11, e22), e12
  v2f accm[11], accs[11];
  float accx[11];
#pragma unroll
  for (int i = 0; i < 11; ++i) {
    rab[i] = v2f{0.f, 0.f};
    if (i < PFD) {
      const int y = y_first + i * y_step;
      const bool in = x_in && y >= 0 && y < H;
      const int o = (y >= 0 && y < H ? y : 0) * W + xc;
      const float va = p1[o], vb = p2[o];
      rab[i] = v2f{in ? va : 0.f, in ? vb : 0.f};
    }
    accm[i] = v2f{0.f, 0.f}; accs[i] = v2f{0.f, 0.f}; accx[i] = 0.f;
  }
  float local = 0.f;
  const int t_end = (ye - yb) + 2 * HALO;                // input rows yb - 5 .. ye + 4
  for (int t0 = 0; t0 < t_end; t0 += 11) {
#pragma unroll
    for (int i = 0; i < 11; ++i) {
      const int t = t0 + i;
      if (t < t_end) {                                   // wave-uniform
        const v2f c0 = rab[i];
        {   // the row PFD steps ahead
          const int y = y_first + (t + PFD) * y_step;
          const bool in = x_in && y >= 0 && y < H && t + PFD < t_end;
          const int o = (y >= 0 && y < H ? y : 0) * W + xc;
          const float va = p1[o], vb = p2[o];
          rab[(i + PFD) % 11] = v2f{in ? va : 0.f, in ? vb : 0.f};
        }
        // ---- horizontal: input columns x - 5 .. x + 5 of this row, tap k at column x - 5 + k
        // (the left neighbours first — tap 0 is the FARTHEST one — then the right ones as the taps reach them, so that at
        // most six pairs are live).  The two images travel as ONE register pair through the shift chains and the taps:
        // formed per tap from two separate registers, the pair cost two moves each (35 of 234 vector instructions per row
        // step in the first version of this kernel).
        v2f l[6];
        l[0] = c0;
#pragma unroll
        for (int j = 1; j <= 5; ++j) l[j] = v2f{dpp_wave<0x138>(l[j - 1].x), dpp_wave<0x138>(l[j - 1].y)};   // lane l - j
        // packed fp32 (v_pk_mul / v_pk_add / v_pk_fma_f32: two IEEE operations per instruction, the same roundings as the
        // scalar forms): (m1, m2) and (s11, s22) travel as pairs
        v2f m12 = {0.f, 0.f}, s1122 = {0.f, 0.f};
        float s12 = 0.f;
        auto tap = [&](float w, v2f ab) {
          const v2f wab = ab * w;
          m12 += wab;
          s1122 = __builtin_elementwise_fma(wab, ab, s1122);
          s12 = fmaf(wab.x, ab.y, s12);
        };
        tap(kWin[0], l[5]); tap(kWin[1], l[4]); tap(kWin[2], l[3]);
        tap(kWin[3], l[2]); tap(kWin[4], l[1]); tap(kWin[5], l[0]);
        v2f rr = c0;
#pragma unroll
        for (int k = 6; k < 11; ++k) {
          rr = v2f{dpp_wave<0x130>(rr.x), dpp_wave<0x130>(rr.y)};          // lane l + (k - 5)
          tap(kWin[k], rr);
        }
        // ---- vertical: this input row is tap k of output row (yb - 5 + t) + 5 - k
#pragma unroll
        for (int k = 0; k < 11; ++k) {
          const int sl = (i + 5 - k + 11) % 11;
          const float w = kWin[k];
          const v2f ww = {w, w};
          accm[sl] = __builtin_elementwise_fma(ww, m12, accm[sl]);
          accs[sl] = __builtin_elementwise_fma(ww, s1122, accs[sl]);
          accx[sl] = fmaf(w, s12, accx[sl]);
        }
        // ---- the output row that just took its last tap (k = 10): r = yb - 10 + t
        const int sl = (i + 5 - 10 + 11) % 11;
        const int r = r_first + t * y_step;
        if (r >= yb && r < ye && x_out) {
          const float mu1 = accm[sl].x, mu2 = accm[sl].y, e11 = accs[sl].x, e22 = accs[sl].y, e12 = accx[sl];
          const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
          const float sig1 = e11 - mu1_sq, sig2 = e22 - mu2_sq, sig12 = e12 - mu12;
          const float A1 = 2.f * mu12 + kC1;
          const float A2 = 2.f * sig12 + kC2;
          const float B1 = mu1_sq + mu2_sq + kC1;
          const float B2 = sig1 + sig2 + kC2;
          // 1-ulp reciprocals (v_rcp_f32) instead of two IEEE divisions (ten instructions each under
          // -fhip-fp32-correctly-rounded-divide-sqrt): B1, B2 >= C1, C2 > 0; 1.2e-7 relative, the tolerance is 1e-4
          const float inv_B1 = __builtin_amdgcn_rcpf(B1), inv_B2 = __builtin_amdgcn_rcpf(B2);
          const float m = (A1 * A2) * (inv_B1 * inv_B2);
          local += m;
          if (TRAIN) {
            const float d_s1 = -m * inv_B2;
            const float d_s12 = 2.f * A1 * inv_B1 * inv_B2;
            const float d_mu1 = 2.f * mu2 * A2 * inv_B1 * inv_B2 - 2.f * mu1 * m * inv_B1 - 2.f * mu1 * d_s1 - mu2 * d_s12;
            const int o = r * W + x;
            (dm_dmu1 + plane_off)[o] = d_mu1;
            (dm_dsigma1_sq + plane_off)[o] = d_s1;
            (dm_dsigma12 + plane_off)[o] = d_s12;
          }
        }
        accm[sl] = v2f{0.f, 0.f}; accs[sl] = v2f{0.f, 0.f}; accx[sl] = 0.f;   // the slot now belongs to output row r + 11
      }
    }
  }
  local = wave_sum(local);
  if (lane == 0) partials[unit] = local;
}

// ---------------------------------------------------------------- forward, sliding window, vertical pass first
// ssim_fwd_sw_kernel filters every INPUT row horizontally (two images, three products per tap) and adds the five row
// statistics into eleven pending output rows.  The two passes commute: here an input row's raw moments (a, b, a^2, b^2,
// ab: formed once per input pixel) are added into the pending rows first, and a finished OUTPUT row takes the horizontal
// pass — 39 rows of a 49-step band instead of 49, on five values instead of on two images x three products per tap.
// Same strips, bands, prefetch ring and alternating band direction; the sums associate differently (fp32: ~1e-7
// relative against the tile kernel, nothing is bit-identical to it any more).
template <bool TRAIN>
__global__ __launch_bounds__(256) void ssim_fwd_vf_kernel(
    const float* __restrict__ img1, const float* __restrict__ img2, int planes, int H, int W, int RB, int strips,
    int bands, int alt, float* __restrict__ dm_dmu1, float* __restrict__ dm_dsigma1_sq, float* __restrict__ dm_dsigma12,
    float* __restrict__ partials) {
  const int lane = threadIdx.x & 63;
  // units = (plane, band, strip), strips fastest; the four waves of a workgroup take four neighbouring strips of one
  // band, and every XCD label (blockIdx & 7) a contiguous range of workgroups, so that the 10-column / 10-row overlaps
  // of neighbouring units meet in one L2 (see tile_walk above)
  const long long units = (long long)planes * bands * strips;
  const long long nblk = gridDim.x;
  long long blk = blockIdx.x;
  if ((nblk & 7) == 0) blk = (blk & 7) * (nblk >> 3) + (blk >> 3);
  const long long unit = blk * 4 + (threadIdx.x >> 6);
  if (unit >= units) return;
  const int strip = (int)(unit % strips), band = (int)((unit / strips) % bands);
  const size_t plane_off = (size_t)(unit / ((long long)strips * bands)) * H * W;
  const int x = strip * SW_OUT + lane - HALO;           // this lane's input (and, for lanes 5..58, output) column
  const bool x_in = x >= 0 && x < W;
  // the horizontal pass below leaves the result for column x - 5 on this lane: lanes 10..63 hold the strip's 54 outputs
  const int xo = x - HALO;
  const bool x_out = lane >= 2 * HALO && xo < W;
  const int yb = band * RB, ye = min(H, yb + RB);       // output rows [yb, ye)
  // Odd bands walk bottom-up (alt): a band's last rows are then its lower neighbour's FIRST rows' neighbours in time as
  // well as in space, so the ten halo rows two bands share are read by both within a few row steps and the second read
  // hits the XCD's L2 (walking every band top-down, band i reads them ~RB steps after band i + 1 did: 40 us and 8 MB of
  // other rows later — PMC: 1.23x the algorithmic bytes).  The window is symmetric, so a reversed band applies the same
  // eleven weights; its outputs add their taps in the opposite row order (fp32 association: ~1e-7 relative).
  const bool up = alt && (band & 1);
  const int y_first = up ? ye + HALO - 1 : yb - HALO, y_step = up ? -1 : 1;      // input row of step t: y_first + t y_step
  const int r_first = up ? ye + 2 * HALO - 1 : yb - 2 * HALO;                    // output row finished at step t
  // wave-uniform plane bases and 32-bit element offsets (one plane holds fewer than 2^31 pixels): the loads and stores
  // take the scalar-base form and the per-row address is one 32-bit multiply-add, not a 64-bit chain per access
  const float* const p1 = img1 + plane_off;
  const float* const p2 = img2 + plane_off;
  const int xc = x_in ? x : 0;

  constexpr int PFD = 4;     // input rows in flight ahead of the one being filtered
  v2f rab[11];               // (img1, img2) of input row t in slot t % 11; only PFD of them are live at a time
  // pending output rows: slot (t + 5 - k) % 11 for the row that takes tap k of input row t; (mu1, mu2), (e11, e22), e12
  v2f accm[11], accs[11];
  float accx[11];
#pragma unroll
  for (int i = 0; i < 11; ++i) {
    rab[i] = v2f{0.f, 0.f};
    if (i < PFD) {
      const int y = y_first + i * y_step;
      const bool in = x_in && y >= 0 && y < H;
      const int o = (y >= 0 && y < H ? y : 0) * W + xc;
      const float va = p1[o], vb = p2[o];
      rab[i] = v2f{in ? va : 0.f, in ? vb : 0.f};
    }
    accm[i] = v2f{0.f, 0.f}; accs[i] = v2f{0.f, 0.f}; accx[i] = 0.f;
  }
  float local = 0.f;
  const int t_end = (ye - yb) + 2 * HALO;                // input rows yb - 5 .. ye + 4
  for (int t0 = 0; t0 < t_end; t0 += 11) {
#pragma unroll
    for (int i = 0; i < 11; ++i) {
      const int t = t0 + i;
      if (t < t_end) {                                   // wave-uniform
        const v2f c0 = rab[i];
        {   // the row PFD steps ahead
          const int y = y_first + (t + PFD) * y_step;
          const bool in = x_in && y >= 0 && y < H && t + PFD < t_end;
          const int o = (y >= 0 && y < H ? y : 0) * W + xc;
          const float va = p1[o], vb = p2[o];
          rab[(i + PFD) % 11] = v2f{in ? va : 0.f, in ? vb : 0.f};
        }
        // ---- vertical FIRST, on the raw moments of this input row (formed once per input pixel): the row is tap k of
        // output row (yb - 5 + t) + 5 - k
        const v2f sq = c0 * c0;
        const float xy = c0.x * c0.y;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
          const int sl = (i + 5 - k + 11) % 11;
          const float w = kWin[k];
          const v2f ww = {w, w};
          accm[sl] = __builtin_elementwise_fma(ww, c0, accm[sl]);
          accs[sl] = __builtin_elementwise_fma(ww, sq, accs[sl]);
          accx[sl] = fmaf(w, xy, accx[sl]);
        }
        // ---- the output row that just took its last tap (k = 10): r = yb - 10 + t.  Only the band's own rows take the
        // horizontal pass (39 of the 49 steps of a 39-row band): the five column sums travel ACROSS the lanes in
        // transposed form — acc <- shift(acc) + w_k v, one add with a DPP operand per tap, the six distinct products
        // w_k v formed once (the window is symmetric) — 16 instructions per statistic instead of 10 moves + 11 fmas on
        // each of two images and three products per tap.
        const int sl = (i + 5 - 10 + 11) % 11;
        const int r = r_first + t * y_step;
        if (r >= yb && r < ye) {                          // wave-uniform
          // the five chains advance in lockstep: a DPP operand may not be read within two issue slots of its write, and
          // four other chains' adds fill them (one chain after the other cost one s_nop per add)
          const float hv[5] = {accm[sl].x, accm[sl].y, accs[sl].x, accs[sl].y, accx[sl]};
          float tk[5][6], hacc[5];
#pragma unroll
          for (int j = 0; j < 5; ++j) {
#pragma unroll
            for (int k = 0; k < 6; ++k) tk[j][k] = kWin[k] * hv[j];
            hacc[j] = tk[j][0];
          }
#pragma unroll
          for (int k = 1; k < 11; ++k) {
#pragma unroll
            for (int j = 0; j < 5; ++j) hacc[j] = dpp_wave<0x138>(hacc[j]) + tk[j][k < 6 ? k : 10 - k];
          }
          // lane l now holds sum_k w_k v[l - 10 + k]: the filtered values at column x - 5
          const float mu1 = hacc[0], mu2 = hacc[1], e11 = hacc[2], e22 = hacc[3], e12 = hacc[4];
          if (x_out) {
            const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
            const float sig1 = e11 - mu1_sq, sig2 = e22 - mu2_sq, sig12 = e12 - mu12;
            const float A1 = 2.f * mu12 + kC1;
            const float A2 = 2.f * sig12 + kC2;
            const float B1 = mu1_sq + mu2_sq + kC1;
            const float B2 = sig1 + sig2 + kC2;
            const float inv_B1 = __builtin_amdgcn_rcpf(B1), inv_B2 = __builtin_amdgcn_rcpf(B2);   // as ssim_fwd_sw_kernel
            const float m = (A1 * A2) * (inv_B1 * inv_B2);
            local += m;
            if (TRAIN) {
              const float d_s1 = -m * inv_B2;
              const float d_s12 = 2.f * A1 * inv_B1 * inv_B2;
              const float d_mu1 = 2.f * mu2 * A2 * inv_B1 * inv_B2 - 2.f * mu1 * m * inv_B1 - 2.f * mu1 * d_s1 - mu2 * d_s12;
              const int o = r * W + xo;
              (dm_dmu1 + plane_off)[o] = d_mu1;
              (dm_dsigma1_sq + plane_off)[o] = d_s1;
              (dm_dsigma12 + plane_off)[o] = d_s12;
            }
          }
        }
        accm[sl] = v2f{0.f, 0.f}; accs[sl] = v2f{0.f, 0.f}; accx[sl] = 0.f;   // the slot now belongs to output row r + 11
      }
    }
  }
  local = wave_sum(local);
  if (lane == 0) partials[unit] = local;
}

  __global__ __launch_bounds__(NT) void ssim_reduce_kernel(const float* __restrict__ partials,
                                                            size_t n, double inv_count,
                                                            float* __restrict__ out) {
    __shared__ double sRed[NT];
    double s = 0.0;
    for (size_t i = threadIdx.x; i < n; i += NT) s += (double)partials[i];
    sRed[threadIdx.x] = s;
    __syncthreads();
    for (int off = NT / 2; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off) sRed[threadIdx.x] += sRed[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(sRed[0] * inv_count);
  }

  __global__ __launch_bounds__(NT) void ssim_bwd_kernel(
      const float* __restrict__ img1, const float* __restrict__ img2, int planes, int H, int W,
      const float* __restrict__ dL_dmean, float inv_count,
      const float* __restrict__ dm_dmu1, const float* __restrict__ dm_dsigma1_sq,
      const float* __restrict__ dm_dsigma12, float* __restrict__ dL_dimg1) {
    __shared__ float sM[3][IN][INP];
    __shared__ float sH[3][IN][TS + 1];

    const int tid = threadIdx.x;
    constexpr int PF = (IN * IN + NT - 1) / NT;
    const int GX = (W + TS - 1) / TS, GY = (H + TS - 1) / TS;
    const long long ntiles = (long long)GX * GY * planes;
    float r0[PF], r1[PF], r2[PF];   // the next tile's three derivative maps, prefetched (see ssim_fwd_kernel)
    auto fetch = [&](long long t) {
      const int bx = (int)(t % GX), by = (int)((t / GX) % GY);
      const size_t off = (size_t)(t / ((long long)GX * GY)) * H * W;
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int i = tid + u * NT;
        const int r = i / IN, c = i - r * IN;
        const int gy = by * TS + r - HALO, gx = bx * TS + c - HALO;
        const bool in = i < IN * IN && gy >= 0 && gy < H && gx >= 0 && gx < W;
        const size_t o = off + (size_t)gy * W + gx;
        r0[u] = in ? dm_dmu1[o] : 0.f;
        r1[u] = in ? dm_dsigma1_sq[o] : 0.f;
        r2[u] = in ? dm_dsigma12[o] : 0.f;
      }
    };
    const TileWalk tw = tile_walk(ntiles);
    if (tw.first < tw.end) fetch(tw.first);
  for (long long t = tw.first; t < tw.end; t += tw.step) {
    const int x0 = (int)(t % GX) * TS, y0 = (int)((t / GX) % GY) * TS;
    const size_t plane_off = (size_t)(t / ((long long)GX * GY)) * H * W;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int i = tid + u * NT;
      if (i < IN * IN) {
        const int r = i / IN, c = i - r * IN;
        sM[0][r][c] = r0[u];
        sM[1][r][c] = r1[u];
        sM[2][r][c] = r2[u];
      }
    }
    __syncthreads();
    if (t + tw.step < tw.end) fetch(t + tw.step);
    // four adjacent outputs per thread in both passes (see ssim_fwd_kernel): same taps in the same order, 3.1x fewer
    // LDS reads
    for (int i = tid; i < IN * (TS / 4); i += NT) {
      const int r = i / (TS / 4), c0 = 4 * (i - r * (TS / 4));
      float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f}, d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 14; ++j) {
        const float v0 = sM[0][r][c0 + j], v1 = sM[1][r][c0 + j], v2 = sM[2][r][c0 + j];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const int k = j - o;
          if (k >= 0 && k < 11) {
            const float w = kWin[k];
            a[o] = fmaf(w, v0, a[o]);
            b[o] = fmaf(w, v1, b[o]);
            d[o] = fmaf(w, v2, d[o]);
          }
        }
      }
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        sH[0][r][c0 + o] = a[o];
        sH[1][r][c0 + o] = b[o];
        sH[2][r][c0 + o] = d[o];
      }
    }
    __syncthreads();
    const float g = dL_dmean[0] * inv_count;
    {
      const int vc = tid & (TS - 1), vr0 = 4 * (tid / TS);
      float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f}, d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 14; ++j) {
        const float v0 = sH[0][vr0 + j][vc], v1 = sH[1][vr0 + j][vc], v2 = sH[2][vr0 + j][vc];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const int k = j - o;
          if (k >= 0 && k < 11) {
            const float w = kWin[k];
            a[o] = fmaf(w, v0, a[o]);
            b[o] = fmaf(w, v1, b[o]);
            d[o] = fmaf(w, v2, d[o]);
          }
        }
      }
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int gy = y0 + vr0 + o, gx = x0 + vc;
        if (gy >= H || gx >= W) continue;
        const size_t oo = plane_off + (size_t)gy * W + gx;
        const float x = img1[oo], y = img2[oo];
        dL_dimg1[oo] = g * (a[o] + 2.f * x * b[o] + y * d[o]);
      }
    }
    __syncthreads();   // sM / sH are rewritten by the next tile
    }
  }

// ---------------------------------------------------------------- backward, sliding window (round 4)
// The same walk as ssim_fwd_sw_kernel for the adjoint: the three derivative maps are filtered with the window in both
// directions (taps in the order k = 0..10, fused multiply-adds as in the tile kernel: bit-identical), the output row
// r = y - 5 is finished with the two images at (r, x):  dL/dimg1 = g (A + 2 img1 B + img2 D).
__global__ __launch_bounds__(256) void ssim_bwd_sw_kernel(
    const float* __restrict__ img1, const float* __restrict__ img2, int planes, int H, int W, int RB, int strips,
    int bands, int alt, const float* __restrict__ dL_dmean, float inv_count, const float* __restrict__ dm_dmu1,
    const float* __restrict__ dm_dsigma1_sq, const float* __restrict__ dm_dsigma12, float* __restrict__ dL_dimg1) {
  const int lane = threadIdx.x & 63;
  const long long units = (long long)planes * bands * strips;
  const long long nblk = gridDim.x;
  long long blk = blockIdx.x;
  if ((nblk & 7) == 0) blk = (blk & 7) * (nblk >> 3) + (blk >> 3);
  const long long unit = blk * 4 + (threadIdx.x >> 6);
  if (unit >= units) return;
  const int strip = (int)(unit % strips), band = (int)((unit / strips) % bands);
  const size_t plane_off = (size_t)(unit / ((long long)strips * bands)) * H * W;
  const int x = strip * SW_OUT + lane - HALO;
  const bool x_in = x >= 0 && x < W;
  const bool x_out = lane >= HALO && lane < 64 - HALO && x < W;
  const int yb = band * RB, ye = min(H, yb + RB);
  const bool up = alt && (band & 1);                       // odd bands bottom-up, see ssim_fwd_sw_kernel
  const int y_first = up ? ye + HALO - 1 : yb - HALO, y_step = up ? -1 : 1;
  const int r_first = up ? ye + 2 * HALO - 1 : yb - 2 * HALO;
  const size_t col = plane_off + (x_in ? x : 0);
  const float g = dL_dmean[0] * inv_count;

  constexpr int PFD = 4;
  float r0[11], r1[11], r2[11];   // input rows of the three maps in flight (slot t % 11), PFD live
  float i1[11], i2[11];           // the images at the output row that completes at step t (slot t % 11), PFD live
  v2f accab[11];                  // pending output rows: (A, B) and D, slot (t + 5 - k) % 11
  float accd[11];
#pragma unroll
  for (int i = 0; i < 11; ++i) {
    r0[i] = r1[i] = r2[i] = 0.f;
    i1[i] = i2[i] = 0.f;
    if (i < PFD) {
      const int y = y_first + i * y_step;
      const bool in = x_in && y >= 0 && y < H;
      const size_t o = col + (size_t)(y >= 0 && y < H ? y : 0) * W;
      const float v0 = dm_dmu1[o], v1 = dm_dsigma1_sq[o], v2 = dm_dsigma12[o];
      r0[i] = in ? v0 : 0.f; r1[i] = in ? v1 : 0.f; r2[i] = in ? v2 : 0.f;
      const int ro = r_first + i * y_step;                    // output row finished at step i
      const size_t oo = col + (size_t)(ro >= 0 && ro < H ? ro : 0) * W;
      i1[i] = img1[oo]; i2[i] = img2[oo];
    }
    accab[i] = v2f{0.f, 0.f}; accd[i] = 0.f;
  }
  const int t_end = (ye - yb) + 2 * HALO;
  for (int t0 = 0; t0 < t_end; t0 += 11) {
#pragma unroll
    for (int i = 0; i < 11; ++i) {
      const int t = t0 + i;
      if (t < t_end) {
        const float c0 = r0[i], c1 = r1[i], c2 = r2[i];
        const float px = i1[i], py = i2[i];
        {   // PFD steps ahead: the maps' input row and the images' output row
          const int y = y_first + (t + PFD) * y_step;
          const bool in = x_in && y >= 0 && y < H && t + PFD < t_end;
          const size_t o = col + (size_t)(y >= 0 && y < H ? y : 0) * W;
          const float v0 = dm_dmu1[o], v1 = dm_dsigma1_sq[o], v2 = dm_dsigma12[o];
          r0[(i + PFD) % 11] = in ? v0 : 0.f; r1[(i + PFD) % 11] = in ? v1 : 0.f; r2[(i + PFD) % 11] = in ? v2 : 0.f;
          const int ro = r_first + (t + PFD) * y_step;
          const size_t oo = col + (size_t)(ro >= 0 && ro < H ? ro : 0) * W;
          i1[(i + PFD) % 11] = img1[oo]; i2[(i + PFD) % 11] = img2[oo];
        }
        // ---- horizontal
        float l0[6], l1[6], l2[6];
        l0[0] = c0; l1[0] = c1; l2[0] = c2;
#pragma unroll
        for (int j = 1; j <= 5; ++j) {
          l0[j] = dpp_wave<0x138>(l0[j - 1]);
          l1[j] = dpp_wave<0x138>(l1[j - 1]);
          l2[j] = dpp_wave<0x138>(l2[j - 1]);
        }
        v2f hab = {0.f, 0.f};
        float hd = 0.f;
        auto tap = [&](float w, float v0, float v1, float v2) {
          const v2f ww = {w, w}, vv = {v0, v1};
          hab = __builtin_elementwise_fma(ww, vv, hab);
          hd = fmaf(w, v2, hd);
        };
        tap(kWin[0], l0[5], l1[5], l2[5]); tap(kWin[1], l0[4], l1[4], l2[4]); tap(kWin[2], l0[3], l1[3], l2[3]);
        tap(kWin[3], l0[2], l1[2], l2[2]); tap(kWin[4], l0[1], l1[1], l2[1]); tap(kWin[5], l0[0], l1[0], l2[0]);
        float q0 = c0, q1 = c1, q2 = c2;
#pragma unroll
        for (int k = 6; k < 11; ++k) {
          q0 = dpp_wave<0x130>(q0); q1 = dpp_wave<0x130>(q1); q2 = dpp_wave<0x130>(q2);
          tap(kWin[k], q0, q1, q2);
        }
        // ---- vertical
#pragma unroll
        for (int k = 0; k < 11; ++k) {
          const int sl = (i + 5 - k + 11) % 11;
          const float w = kWin[k];
          const v2f ww = {w, w};
          accab[sl] = __builtin_elementwise_fma(ww, hab, accab[sl]);
          accd[sl] = fmaf(w, hd, accd[sl]);
        }
        const int sl = (i + 5 - 10 + 11) % 11;
        const int r = r_first + t * y_step;
        if (r >= yb && r < ye && x_out)
          dL_dimg1[plane_off + (size_t)r * W + x] = g * (accab[sl].x + 2.f * px * accab[sl].y + py * accd[sl]);
        accab[sl] = v2f{0.f, 0.f}; accd[sl] = 0.f;
      }
    }
  }
}

// ---------------------------------------------------------------- backward, sliding window, vertical pass first
// As ssim_fwd_vf_kernel: the three derivative maps are summed down the columns first (33 fused multiply-adds per input
// row), a finished output row takes the horizontal pass in transposed form (3 x 16 instructions, only for the band's
// own rows) and is completed with the two images read at the column the result lands on.
__global__ __launch_bounds__(256) void ssim_bwd_vf_kernel(
    const float* __restrict__ img1, const float* __restrict__ img2, int planes, int H, int W, int RB, int strips,
    int bands, int alt, const float* __restrict__ dL_dmean, float inv_count, const float* __restrict__ dm_dmu1,
    const float* __restrict__ dm_dsigma1_sq, const float* __restrict__ dm_dsigma12, float* __restrict__ dL_dimg1) {
  const int lane = threadIdx.x & 63;
  const long long units = (long long)planes * bands * strips;
  const long long nblk = gridDim.x;
  long long blk = blockIdx.x;
  if ((nblk & 7) == 0) blk = (blk & 7) * (nblk >> 3) + (blk >> 3);
  const long long unit = blk * 4 + (threadIdx.x >> 6);
  if (unit >= units) return;
  const int strip = (int)(unit % strips), band = (int)((unit / strips) % bands);
  const size_t plane_off = (size_t)(unit / ((long long)strips * bands)) * H * W;
  const int x = strip * SW_OUT + lane - HALO;
  const bool x_in = x >= 0 && x < W;
  const int xo = x - HALO;                                  // the horizontal pass leaves column x - 5 on this lane
  const bool x_out = lane >= 2 * HALO && xo < W;
  const int yb = band * RB, ye = min(H, yb + RB);
  const bool up = alt && (band & 1);                       // odd bands bottom-up, see ssim_fwd_sw_kernel
  const int y_first = up ? ye + HALO - 1 : yb - HALO, y_step = up ? -1 : 1;
  const int r_first = up ? ye + 2 * HALO - 1 : yb - 2 * HALO;
  const size_t col = plane_off + (x_in ? x : 0);
  const size_t colo = plane_off + (x_out ? xo : 0);          // the images are read where the output is written
  const float g = dL_dmean[0] * inv_count;

  constexpr int PFD = 4;
  float r0[11], r1[11], r2[11];   // input rows of the three maps in flight (slot t % 11), PFD live
  float i1[11], i2[11];           // the images at the output row that completes at step t (slot t % 11), PFD live
  v2f accab[11];                  // pending output rows: (A, B) and D, slot (t + 5 - k) % 11
  float accd[11];
#pragma unroll
  for (int i = 0; i < 11; ++i) {
    r0[i] = r1[i] = r2[i] = 0.f;
    i1[i] = i2[i] = 0.f;
    if (i < PFD) {
      const int y = y_first + i * y_step;
      const bool in = x_in && y >= 0 && y < H;
      const size_t o = col + (size_t)(y >= 0 && y < H ? y : 0) * W;
      const float v0 = dm_dmu1[o], v1 = dm_dsigma1_sq[o], v2 = dm_dsigma12[o];
      r0[i] = in ? v0 : 0.f; r1[i] = in ? v1 : 0.f; r2[i] = in ? v2 : 0.f;
      const int ro = r_first + i * y_step;                    // output row finished at step i
      const size_t oo = colo + (size_t)(ro >= 0 && ro < H ? ro : 0) * W;
      i1[i] = img1[oo]; i2[i] = img2[oo];
    }
    accab[i] = v2f{0.f, 0.f}; accd[i] = 0.f;
  }
  const int t_end = (ye - yb) + 2 * HALO;
  for (int t0 = 0; t0 < t_end; t0 += 11) {
#pragma unroll
    for (int i = 0; i < 11; ++i) {
      const int t = t0 + i;
      if (t < t_end) {
        const float c0 = r0[i], c1 = r1[i], c2 = r2[i];
        const float px = i1[i], py = i2[i];
        {   // PFD steps ahead: the maps' input row and the images' output row
          const int y = y_first + (t + PFD) * y_step;
          const bool in = x_in && y >= 0 && y < H && t + PFD < t_end;
          const size_t o = col + (size_t)(y >= 0 && y < H ? y : 0) * W;
          const float v0 = dm_dmu1[o], v1 = dm_dsigma1_sq[o], v2 = dm_dsigma12[o];
          r0[(i + PFD) % 11] = in ? v0 : 0.f; r1[(i + PFD) % 11] = in ? v1 : 0.f; r2[(i + PFD) % 11] = in ? v2 : 0.f;
          const int ro = r_first + (t + PFD) * y_step;
          const size_t oo = colo + (size_t)(ro >= 0 && ro < H ? ro : 0) * W;
          i1[(i + PFD) % 11] = img1[oo]; i2[(i + PFD) % 11] = img2[oo];
        }
        // ---- vertical first (see ssim_fwd_vf_kernel): the input row of the three maps is tap k of output row ... + 5 - k
        {
          const v2f c01 = {c0, c1};
#pragma unroll
          for (int k = 0; k < 11; ++k) {
            const int sl = (i + 5 - k + 11) % 11;
            const float w = kWin[k];
            const v2f ww = {w, w};
            accab[sl] = __builtin_elementwise_fma(ww, c01, accab[sl]);
            accd[sl] = fmaf(w, c2, accd[sl]);
          }
        }
        const int sl = (i + 5 - 10 + 11) % 11;
        const int r = r_first + t * y_step;
        if (r >= yb && r < ye) {                          // wave-uniform: the band's own rows take the horizontal pass
          const float hv[3] = {accab[sl].x, accab[sl].y, accd[sl]};
          float tk[3][6], hacc[3];
#pragma unroll
          for (int j = 0; j < 3; ++j) {
#pragma unroll
            for (int k = 0; k < 6; ++k) tk[j][k] = kWin[k] * hv[j];
            hacc[j] = tk[j][0];
          }
#pragma unroll
          for (int k = 1; k < 11; ++k) {
#pragma unroll
            for (int j = 0; j < 3; ++j) hacc[j] = dpp_wave<0x138>(hacc[j]) + tk[j][k < 6 ? k : 10 - k];
          }
          if (x_out) dL_dimg1[plane_off + (size_t)r * W + xo] = g * (hacc[0] + 2.f * px * hacc[1] + py * hacc[2]);
        }
        accab[sl] = v2f{0.f, 0.f}; accd[sl] = 0.f;
      }
    }
  }
}

  }  // namespace

  namespace {
  // rows per band of the sliding-window forward: enough waves to fill the chip six deep, at least 16 rows per band
  // (a band re-reads 10 halo rows).  PINGS_SSIM_RB overrides (A/B runs); PINGS_SSIM_FWD=tile keeps the tile kernel.
  int sw_rows_per_band(int planes, int H, int W) {
    if (const char* e = getenv("PINGS_SSIM_RB")) { const int v = atoi(e); if (v > 0) return v; }
    // about three waves per SIMD in all (the kernels hold four): fewer, taller bands re-read fewer halo rows, more bands
    // balance the chip better.  1080p x 3, forward / backward ms by band height: 20 rows 0.075 / 0.049, 24 0.069 / 0.046,
    // 28 0.071 / 0.048, 32 0.066 / 0.044, 40 0.063 / 0.042, 64 0.065 / 0.043 (the round-3 tile kernels: 0.074 / 0.050)
    const long long strips = pings::ceil_div(W, SW_OUT);
    const long long want_bands = (256LL * 4 * 3) / (strips * planes);
    long long rb = pings::ceil_div<long long>(H, want_bands > 0 ? want_bands : 1);
    if (rb < 16) rb = 16;
    if (rb > H) rb = H;
    return (int)rb;
  }
  // odd bands bottom-up (halo rows meet in L2); PINGS_SSIM_ALT=0: every band top-down (bit-identical to the tile kernels)
  int sw_alternate() {
    if (const char* e = getenv("PINGS_SSIM_ALT")) return atoi(e) != 0;
    return 1;
  }
  size_t sw_units(int planes, int H, int W) {
    const int rb = sw_rows_per_band(planes, H, W);
    return (size_t)planes * pings::ceil_div(H, rb) * pings::ceil_div(W, SW_OUT);
  }
  }  // namespace

  PINGS_API size_t pings_ssim_partials_count(int planes, int H, int W) {
    if (planes <= 0 || H <= 0 || W <= 0) return 0;
    const size_t tiles = (size_t)planes * pings::ceil_div(H, TS) * pings::ceil_div(W, TS);
    const size_t units = sw_units(planes, H, W);
    return tiles > units ? tiles : units;      // either forward kernel fits
  }

  PINGS_API int pings_ssim_forward(const float* img1, const float* img2, int planes, int H, int W,
                                   int train, float* out_mean, float* dm_dmu1,
                                   float* dm_dsigma1_sq, float* dm_dsigma12, float* partials,
                                   void* stream) {
    PINGS_ARG_CHECK(img1 && img2 && out_mean && partials, "null pointer");
    PINGS_ARG_CHECK(planes > 0 && H > 0 && W > 0, "empty image");
    PINGS_ARG_CHECK(planes <= 65535, "too many planes");
    PINGS_ARG_CHECK(!train || (dm_dmu1 && dm_dsigma1_sq && dm_dsigma12),
                    "train=1 needs the three derivative maps");
    hipStream_t st = pings::as_stream(stream);
    const char* fwd_env = getenv("PINGS_SSIM_FWD");
    if (!(fwd_env && fwd_env[0] == 't')) {      // default: the sliding-window kernel
      const int rb = sw_rows_per_band(planes, H, W);
      const int strips = pings::ceil_div(W, SW_OUT), bands = pings::ceil_div(H, rb);
      const size_t units = (size_t)planes * bands * strips;
      unsigned nblk = (unsigned)pings::ceil_div<size_t>(units, 4);
      nblk = (nblk + 7u) & ~7u;                  // a multiple of eight: the XCD-contiguous numbering above
      pings::prof::Scope ps("ssim_fwd", st);
      const bool hfirst = fwd_env && fwd_env[0] == 's';   // PINGS_SSIM_FWD=sw: the horizontal-first kernel (A/B runs)
#define PINGS_SSIM_FWD_LAUNCH(K_, T_, A_, B_, C_)                                                                   \
  hipLaunchKernelGGL(K_<T_>, dim3(nblk), dim3(256), 0, st, img1, img2, planes, H, W, rb, strips, bands, sw_alternate(), \
                     A_, B_, C_, partials)
      if (train) {
        if (hfirst) PINGS_SSIM_FWD_LAUNCH(ssim_fwd_sw_kernel, true, dm_dmu1, dm_dsigma1_sq, dm_dsigma12);
        else PINGS_SSIM_FWD_LAUNCH(ssim_fwd_vf_kernel, true, dm_dmu1, dm_dsigma1_sq, dm_dsigma12);
      } else {
        if (hfirst) PINGS_SSIM_FWD_LAUNCH(ssim_fwd_sw_kernel, false, (float*)nullptr, (float*)nullptr, (float*)nullptr);
        else PINGS_SSIM_FWD_LAUNCH(ssim_fwd_vf_kernel, false, (float*)nullptr, (float*)nullptr, (float*)nullptr);
      }
#undef PINGS_SSIM_FWD_LAUNCH
      PINGS_LAUNCH_CHECK();
      const double inv_count = 1.0 / ((double)planes * H * W);
      hipLaunchKernelGGL(ssim_reduce_kernel, dim3(1), dim3(NT), 0, st, partials, units, inv_count, out_mean);
      PINGS_LAUNCH_CHECK();
      return PINGS_OK;
    }
    const size_t n = (size_t)planes * pings::ceil_div(H, TS) * pings::ceil_div(W, TS);
    const dim3 grid((unsigned)(n < (size_t)PERSISTENT_WGS ? n : (size_t)PERSISTENT_WGS));
    pings::prof::Scope ps("ssim_fwd", st);
    if (train) {
      hipLaunchKernelGGL(ssim_fwd_kernel<true>, grid, dim3(NT), 0, st, img1, img2, planes, H, W, dm_dmu1,
                         dm_dsigma1_sq, dm_dsigma12, partials);
    } else {
      hipLaunchKernelGGL(ssim_fwd_kernel<false>, grid, dim3(NT), 0, st, img1, img2, planes, H, W,
                         (float*)nullptr, (float*)nullptr, (float*)nullptr, partials);
    }
    PINGS_LAUNCH_CHECK();
    const double inv_count = 1.0 / ((double)planes * H * W);
    hipLaunchKernelGGL(ssim_reduce_kernel, dim3(1), dim3(NT), 0, st, partials, n, inv_count,
                       out_mean);
    PINGS_LAUNCH_CHECK();
    return PINGS_OK;
  }

  PINGS_API int pings_ssim_backward(const float* img1, const float* img2, int planes, int H, int W,
                                    const float* dL_dmean, const float* dm_dmu1,
                                    const float* dm_dsigma1_sq, const float* dm_dsigma12,
                                    float* dL_dimg1, void* stream) {
    PINGS_ARG_CHECK(img1 && img2 && dL_dmean && dm_dmu1 && dm_dsigma1_sq && dm_dsigma12 && dL_dimg1,
                    "null pointer");
    PINGS_ARG_CHECK(planes > 0 && H > 0 && W > 0, "empty image");
    PINGS_ARG_CHECK(planes <= 65535, "too many planes");
    hipStream_t st = pings::as_stream(stream);
    const float inv_count = (float)(1.0 / ((double)planes * H * W));
    const char* bwd_env = getenv("PINGS_SSIM_BWD");
    if (!(bwd_env && bwd_env[0] == 't')) {      // default: the sliding-window kernel
      const int rb = sw_rows_per_band(planes, H, W);
      const int strips = pings::ceil_div(W, SW_OUT), bands = pings::ceil_div(H, rb);
      const size_t units = (size_t)planes * bands * strips;
      unsigned nblk = (unsigned)pings::ceil_div<size_t>(units, 4);
      nblk = (nblk + 7u) & ~7u;
      pings::prof::Scope ps("ssim_bwd", st);
      if (bwd_env && bwd_env[0] == 's')              // PINGS_SSIM_BWD=sw: the horizontal-first kernel (A/B runs)
        hipLaunchKernelGGL(ssim_bwd_sw_kernel, dim3(nblk), dim3(256), 0, st, img1, img2, planes, H, W, rb, strips, bands,
                           sw_alternate(), dL_dmean, inv_count, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1);
      else
        hipLaunchKernelGGL(ssim_bwd_vf_kernel, dim3(nblk), dim3(256), 0, st, img1, img2, planes, H, W, rb, strips, bands,
                           sw_alternate(), dL_dmean, inv_count, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1);
      PINGS_LAUNCH_CHECK();
      return PINGS_OK;
    }
    const size_t n = (size_t)planes * pings::ceil_div(H, TS) * pings::ceil_div(W, TS);
    const dim3 grid((unsigned)(n < (size_t)PERSISTENT_WGS ? n : (size_t)PERSISTENT_WGS));
    pings::prof::Scope ps("ssim_bwd", st);
    hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(NT), 0, st, img1, img2, planes, H, W, dL_dmean,
                       inv_count, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1);
    PINGS_LAUNCH_CHECK();
    return PINGS_OK;
  }
