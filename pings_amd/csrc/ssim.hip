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
      const float inv_B1 = 1.f / B1, inv_B2 = 1.f / B2;
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

  }  // namespace

  PINGS_API size_t pings_ssim_partials_count(int planes, int H, int W) {
    if (planes <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)planes * pings::ceil_div(H, TS) * pings::ceil_div(W, TS);
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
    const size_t n = pings_ssim_partials_count(planes, H, W);
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
    const size_t n = pings_ssim_partials_count(planes, H, W);
    const dim3 grid((unsigned)(n < (size_t)PERSISTENT_WGS ? n : (size_t)PERSISTENT_WGS));
    const float inv_count = (float)(1.0 / ((double)planes * H * W));
    pings::prof::Scope ps("ssim_bwd", st);
    hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(NT), 0, st, img1, img2, planes, H, W, dL_dmean,
                       inv_count, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1);
    PINGS_LAUNCH_CHECK();
    return PINGS_OK;
  }
