// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns this library uses
// (MI355X_MICROARCH.md §HBM: "other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern before trusting an absolute").  Each kernel moves a KNOWN number of bytes in one pattern; the buffers are
// larger than the 256 MiB Infinity Cache unless the name says `_mall` (a table that stays resident in it).
//
//   hipcc --offload-arch=gfx950 -O3 profiles/pmc_calib.hip -o profiles/_build/pmc_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out/f -o f --output-format csv -- profiles/_build/pmc_calib
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out/w -o w --output-format csv -- profiles/_build/pmc_calib
//   python profiles/pmc_calib_summary.py out/f/f_counter_collection.csv out/w/w_counter_collection.csv
//
// The program prints one line per kernel: name, bytes the lanes asked for, bytes at 64-B sector and at 128-B line
// granularity; the summary script divides the counters by those.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned mix(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// 16 B per lane, consecutive lanes consecutive addresses
__global__ void read_stream16(const float4* __restrict__ a, size_t n, float* sink) {
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = a[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 1234.5f) *sink = acc;
}
// 4 B per lane, consecutive
__global__ void read_stream4(const float* __restrict__ a, size_t n, float* sink) {
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += a[i];
  if (acc == 1234.5f) *sink = acc;
}
// 16x16 tile of a planar [H][W] fp32 image per workgroup, 4 B per lane: a wave reads four 64-B row pieces
__global__ void read_tile_rows4(const float* __restrict__ img, int W, int H, float* sink) {
  const int tx = blockIdx.x * 16 + (threadIdx.x & 15), ty = blockIdx.y * 16 + (threadIdx.x >> 4);
  float acc = 0.f;
  if (tx < W && ty < H) acc = img[(size_t)ty * W + tx];
  if (acc == 1234.5f) *sink = acc;
}
// same tile, 4 pixels per lane as one 16-B load (a wave reads sixteen 64-B row pieces)
__global__ void read_tile_rows16(const float* __restrict__ img, int W, int H, float* sink) {
  const int tx = blockIdx.x * 16 + (threadIdx.x & 3) * 4, ty = blockIdx.y * 16 + (threadIdx.x >> 2);
  float acc = 0.f;
  if (tx < W && ty < H) {
    const float4 v = *reinterpret_cast<const float4*>(img + (size_t)ty * W + tx);
    acc = v.x + v.y + v.z + v.w;
  }
  if (acc == 1234.5f) *sink = acc;
}
// one random 8-B entry per lane (the hash-table probe of the neighbour search)
__global__ void read_gather8(const uint2* __restrict__ tab, unsigned mask, size_t n, float* sink) {
  unsigned acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint2 e = tab[mix((unsigned)i) & mask];
    acc += e.x ^ e.y;
  }
  if (acc == 0x12345u) *sink = (float)acc;
}
// one random 48-B record per lane as three 16-B loads (the Gaussian record of the blend kernels)
__global__ void read_gather48(const float4* __restrict__ rec, unsigned mask, size_t n, float* sink) {
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = (size_t)(mix((unsigned)i) & mask) * 3;
    const float4 a = rec[r], b = rec[r + 1], c = rec[r + 2];
    acc += a.x + b.y + c.z;
  }
  if (acc == 1234.5f) *sink = acc;
}
// one random 12-B position per lane as three 4-B loads (neural_points[3 i ..])
__global__ void read_gather12(const float* __restrict__ p, unsigned mask, size_t n, float* sink) {
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = (size_t)(mix((unsigned)i) & mask) * 3;
    acc += p[r] + p[r + 1] + p[r + 2];
  }
  if (acc == 1234.5f) *sink = acc;
}
__global__ void write_stream16(float4* __restrict__ a, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    a[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void write_stream4(float* __restrict__ a, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (float)i;
}
__global__ void write_tile_rows4(float* __restrict__ img, int W, int H) {
  const int tx = blockIdx.x * 16 + (threadIdx.x & 15), ty = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (tx < W && ty < H) img[(size_t)ty * W + tx] = (float)tx;
}
// random 48-B rows written as three 16-B stores (per-Gaussian gradient rows)
__global__ void write_scatter48(float4* __restrict__ rec, unsigned mask, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = (size_t)(mix((unsigned)i) & mask) * 3;
    rec[r] = make_float4(1, 2, 3, 4); rec[r + 1] = make_float4(5, 6, 7, 8); rec[r + 2] = make_float4(9, 10, 11, 12);
  }
}

static void line(const char* name, double asked, double sect64, double line128) {
  printf("%-22s asked=%.0f sector64=%.0f line128=%.0f\n", name, asked, sect64, line128);
}

int main() {
  const size_t BIG = (size_t)1 << 30;       // 1 GiB: four times the Infinity Cache
  const size_t MALL = (size_t)32 << 20;     // 32 MiB table: resident in the Infinity Cache, not in one XCD's L2
  float *big, *small, *sink;
  CK(hipMalloc(&big, BIG)); CK(hipMalloc(&small, MALL)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(big, 0, BIG)); CK(hipMemset(small, 0, MALL));
  const int G = 256 * 16, T = 256;
  const int W = 8192, H = 8192;             // 256 MiB plane inside `big`
  const size_t NG = (size_t)1 << 24;        // gathers per launch
  for (int rep = 0; rep < 3; ++rep) {
    read_stream16<<<G, T>>>((const float4*)big, BIG / 16, sink);
    read_stream4<<<G, T>>>(big, BIG / 4, sink);
    read_tile_rows4<<<dim3(W / 16, H / 16), 256>>>(big, W, H, sink);
    read_tile_rows16<<<dim3(W / 16, H / 16), 64>>>(big, W, H, sink);
    read_gather8<<<G, T>>>((const uint2*)big, (unsigned)(BIG / 8 - 1), NG, sink);
    read_gather8<<<G, T>>>((const uint2*)small, (unsigned)(MALL / 8 - 1), NG, sink);
    read_gather48<<<G, T>>>((const float4*)big, (unsigned)((1u << 24) - 1), NG, sink);   // 16M records x 48 B = 768 MiB
    read_gather12<<<G, T>>>(big, (unsigned)((1u << 26) - 1), NG, sink);                  // 64M positions x 12 B = 768 MiB
    write_stream16<<<G, T>>>((float4*)big, BIG / 16);
    write_stream4<<<G, T>>>(big, BIG / 4);
    write_tile_rows4<<<dim3(W / 16, H / 16), 256>>>(big, W, H);
    write_scatter48<<<G, T>>>((float4*)big, (unsigned)((1u << 24) - 1), NG);
  }
  CK(hipDeviceSynchronize());
  const double plane = (double)W * H * 4;
  line("read_stream16", BIG, BIG, BIG);
  line("read_stream4", BIG, BIG, BIG);
  line("read_tile_rows4", plane, plane, plane);   // every 128-B line is read by two workgroups: line128 counts it once
  line("read_tile_rows16", plane, plane, plane);
  line("read_gather8_big", NG * 8.0, NG * 64.0, NG * 128.0);
  line("read_gather8_mall", NG * 8.0, NG * 64.0, NG * 128.0);
  // a 48-B record at a 48-B stride touches one 64-B sector with probability 1/4 + ... : 16-B aligned starts, 4 per
  // sector; starts 0 and 16 stay inside one sector (1), 32 and 48 straddle (2) -> 1.5 sectors; lines: start offsets
  // 0..112 step 16 inside a 128-B line, straddles when start > 80 (96, 112) -> 1.25 lines
  line("read_gather48", NG * 48.0, NG * 1.5 * 64.0, NG * 1.25 * 128.0);
  // 12-B position at a 12-B stride: 4-B aligned start; straddles a 64-B sector when start%64 > 52 (56, 60): 2/16
  line("read_gather12", NG * 12.0, NG * 1.125 * 64.0, NG * (1.0 + 2.0 / 32.0) * 128.0);
  line("write_stream16", BIG, BIG, BIG);
  line("write_stream4", BIG, BIG, BIG);
  line("write_tile_rows4", plane, plane, plane);
  line("write_scatter48", NG * 48.0, NG * 1.5 * 64.0, NG * 1.25 * 128.0);
  return 0;
}
