// Deterministic scatter-add of rows (see row_scatter.hpp): counting sort by destination row + one gather pass.
#include <hipcub/hipcub.hpp>

#include <cstdlib>

#include "row_scatter.hpp"

namespace pings_rows {
namespace {

size_t au(size_t v) { return (v + 255) / 256 * 256; }
constexpr uint32_t NO_ROW = 0xFFFFFFFFu;

__global__ __launch_bounds__(256) void hist_kernel(const uint32_t* __restrict__ keys, long long n, uint32_t rows,
                                                   uint32_t* __restrict__ count) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t k = keys[p];
  if (k < rows) atomicAdd(&count[k], 1u);
}

// Placement: a pair takes the next free slot of its row's bucket (integer atomics: which slot is arrival order, the
// SET of pairs in a bucket is not), rank_kernel then orders every bucket by pair id.
__global__ __launch_bounds__(256) void place_kernel(const uint32_t* __restrict__ keys, long long n, uint32_t rows,
                                                    const uint32_t* __restrict__ offset, uint32_t* __restrict__ count,
                                                    uint32_t* __restrict__ bucket) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t k = keys[p];
  if (k >= rows) return;
  const uint32_t slot = atomicSub(&count[k], 1u) - 1u;
  bucket[offset[k] + slot] = (uint32_t)p;
}

// Every placed pair finds its rank among its bucket mates (number of smaller pair ids): sorted[s + rank] = pair.
// O(len) loads per pair, all pairs in parallel; buckets are short (pairs per touched row).
__global__ __launch_bounds__(256) void rank_kernel(const uint32_t* __restrict__ keys, long long n,
                                                   const uint32_t* __restrict__ offset, uint32_t rows,
                                                   const uint32_t* __restrict__ bucket, uint32_t* __restrict__ sorted,
                                                   uint32_t* __restrict__ first) {
  const long long pos = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= n) return;
  if (pos >= (long long)offset[rows]) {   // behind the last bucket (pairs with an invalid key were not placed)
    first[pos] = NO_ROW;
    return;
  }
  const uint32_t pr = bucket[pos];
  const uint32_t k = keys[pr];
  const uint32_t s = offset[k], e = offset[k + 1];
  uint32_t rank = 0;
  for (uint32_t q = s; q < e; ++q) rank += bucket[q] < pr ? 1u : 0u;
  sorted[s + rank] = pr;
  first[s + rank] = rank == 0u ? k : NO_ROW;   // every position of the bucket is written by exactly one pair
}

// G lanes own one destination row (4 columns each); the row's pairs are summed in ascending pair id.
template <int G>
__global__ __launch_bounds__(256) void gather_sum_kernel(const uint32_t* __restrict__ offset,
                                                         const uint32_t* __restrict__ sorted, long long rows, int F,
                                                         const float* __restrict__ src, long long ld,
                                                         const uint32_t* __restrict__ src_row,
                                                         const float* __restrict__ w, float* __restrict__ out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long r = t / G;
  const int c0 = 4 * (int)(t % G);
  if (r >= rows || c0 >= F) return;
  const uint32_t s = offset[r], e = offset[r + 1];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const bool c1 = c0 + 1 < F, c2 = c0 + 2 < F, c3 = c0 + 3 < F;
  for (uint32_t q = s; q < e; ++q) {
    const uint32_t pr = sorted[q];
    const size_t row = (size_t)(src_row ? src_row[pr] : pr) * (size_t)ld + (size_t)c0;
    const float ww = w ? w[pr] : 1.0f;
    a0 = fmaf(ww, src[row], a0);
    if (c1) a1 = fmaf(ww, src[row + 1], a1);
    if (c2) a2 = fmaf(ww, src[row + 2], a2);
    if (c3) a3 = fmaf(ww, src[row + 3], a3);
  }
  float* o = out + (size_t)r * F + c0;
  if ((F & 3) == 0) {
    *reinterpret_cast<float4*>(o) = make_float4(a0, a1, a2, a3);
  } else {
    o[0] = a0;
    if (c1) o[1] = a1;
    if (c2) o[2] = a2;
    if (c3) o[3] = a3;
  }
}

// The same sum, driven by the buckets: G lanes per sorted position, only the first position of a bucket works (the
// table was zeroed before).  Same order of additions as gather_sum_kernel: bitwise the same rows.
template <int G>
__global__ __launch_bounds__(256) void gather_bucket_kernel(const uint32_t* __restrict__ offset,
                                                            const uint32_t* __restrict__ sorted,
                                                            const uint32_t* __restrict__ first, long long n, int F,
                                                            const float* __restrict__ src, long long ld,
                                                            const uint32_t* __restrict__ src_row,
                                                            const float* __restrict__ w, float* __restrict__ out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long pos = t / G;
  const int c0 = 4 * (int)(t % G);
  if (pos >= n || c0 >= F) return;
  const uint32_t r = first[pos];
  if (r == NO_ROW) return;
  const uint32_t e = offset[r + 1];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const bool c1 = c0 + 1 < F, c2 = c0 + 2 < F, c3 = c0 + 3 < F;
  for (uint32_t q = (uint32_t)pos; q < e; ++q) {
    const uint32_t pr = sorted[q];
    const size_t row = (size_t)(src_row ? src_row[pr] : pr) * (size_t)ld + (size_t)c0;
    const float ww = w ? w[pr] : 1.0f;
    a0 = fmaf(ww, src[row], a0);
    if (c1) a1 = fmaf(ww, src[row + 1], a1);
    if (c2) a2 = fmaf(ww, src[row + 2], a2);
    if (c3) a3 = fmaf(ww, src[row + 3], a3);
  }
  float* o = out + (size_t)r * F + c0;
  if ((F & 3) == 0) {
    *reinterpret_cast<float4*>(o) = make_float4(a0, a1, a2, a3);
  } else {
    o[0] = a0;
    if (c1) o[1] = a1;
    if (c2) o[2] = a2;
    if (c3) o[3] = a3;
  }
}

}  // namespace

Plan carve(void* base, int64_t n_pairs, int64_t rows) {
  Plan p;
  char* b = reinterpret_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t bytes) { char* r = b ? b + off : nullptr; off = au(off + bytes); return r; };
  const size_t n = (size_t)(n_pairs > 0 ? n_pairs : 1), R = (size_t)(rows > 0 ? rows : 1);
  p.count = (uint32_t*)take((R + 1) * 4);
  p.offset = (uint32_t*)take((R + 2) * 4);
  p.bucket = (uint32_t*)take(n * 4);
  p.sorted = (uint32_t*)take(n * 4);
  p.first = (uint32_t*)take(n * 4);
  p.n = n_pairs > 0 ? n_pairs : 0;
  size_t tb = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)(R + 1));
  p.temp_bytes = au(tb) + 256;
  p.temp = take(p.temp_bytes);
  p.total = off;
  return p;
}

int build(const Plan& p, const uint32_t* keys, int64_t n, int64_t rows, hipStream_t st) {
  PINGS_ARG_CHECK(rows > 0 && rows < 0x7FFFFFF0LL && n >= 0 && n < 0x7FFFFFF0LL, "row / pair count out of range");
  PINGS_HIP_CHECK(hipMemsetAsync(p.count, 0, sizeof(uint32_t) * (size_t)(rows + 1), st));
  const unsigned grid = (unsigned)((n + 255) / 256);
  if (n > 0) {
    hipLaunchKernelGGL(hist_kernel, dim3(grid), dim3(256), 0, st, keys, (long long)n, (uint32_t)rows, p.count);
    PINGS_LAUNCH_CHECK();
  }
  size_t tb = p.temp_bytes;
  PINGS_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(p.temp, tb, p.count, p.offset, (int)(rows + 1), st));
  if (n > 0) {
    hipLaunchKernelGGL(place_kernel, dim3(grid), dim3(256), 0, st, keys, (long long)n, (uint32_t)rows, p.offset,
                       p.count, p.bucket);
    PINGS_LAUNCH_CHECK();
    hipLaunchKernelGGL(rank_kernel, dim3(grid), dim3(256), 0, st, keys, (long long)n, p.offset, (uint32_t)rows,
                       p.bucket, p.sorted, p.first);
    PINGS_LAUNCH_CHECK();
  }
  return PINGS_OK;
}

int gather_sum(const Plan& p, int64_t rows, int F, const float* src, int64_t ld, const uint32_t* src_row,
               const float* w, float* out, hipStream_t st) {
  PINGS_ARG_CHECK(F > 0 && F <= 64 && rows > 0 && out, "gather_sum: bad shape");
  const uint32_t* sorted = p.sorted;
  const int groups = (F + 3) / 4;
  int G = 1;
  while (G < groups) G <<= 1;
  static const int force = [] {   // PINGS_ROWS_GATHER=table|buckets (A/B runs)
    const char* e = getenv("PINGS_ROWS_GATHER");
    return !e ? 0 : (e[0] == 't' ? 1 : 2);
  }();
  if ((force == 2 || (force == 0 && 4 * p.n < rows)) && p.n >= 0) {   // measured: 98k pairs / 1M rows 37 -> 30 us, 786k pairs 109 -> 112 us
    PINGS_HIP_CHECK(hipMemsetAsync(out, 0, sizeof(float) * (size_t)rows * F, st));
    if (p.n == 0) return PINGS_OK;
    const long long bt = (long long)p.n * G;
    const unsigned bgrid = (unsigned)((bt + 255) / 256);
#define PINGS_GB(GG)                                                                                                 \
  hipLaunchKernelGGL(gather_bucket_kernel<GG>, dim3(bgrid), dim3(256), 0, st, p.offset, sorted, p.first, (long long)p.n, \
                     F, src, (long long)ld, src_row, w, out)
    switch (G) {
      case 1: PINGS_GB(1); break;
      case 2: PINGS_GB(2); break;
      case 4: PINGS_GB(4); break;
      case 8: PINGS_GB(8); break;
      default: PINGS_GB(16); break;
    }
#undef PINGS_GB
    PINGS_LAUNCH_CHECK();
    return PINGS_OK;
  }
  const long long threads = (long long)rows * G;
  const unsigned grid = (unsigned)((threads + 255) / 256);
#define PINGS_GS(GG)                                                                                              \
  hipLaunchKernelGGL(gather_sum_kernel<GG>, dim3(grid), dim3(256), 0, st, p.offset, sorted, (long long)rows, F, src, \
                     (long long)ld, src_row, w, out)
  switch (G) {
    case 1: PINGS_GS(1); break;
    case 2: PINGS_GS(2); break;
    case 4: PINGS_GS(4); break;
    case 8: PINGS_GS(8); break;
    default: PINGS_GS(16); break;
  }
#undef PINGS_GS
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}
}  // namespace pings_rows

namespace {
__global__ __launch_bounds__(256) void keys_from_i64_kernel(const long long* __restrict__ dst, long long n,
                                                            uint32_t rows, uint32_t* __restrict__ keys) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const long long d = dst[p];
  keys[p] = (d >= 0 && d < (long long)rows) ? (uint32_t)d : rows;
}
__global__ __launch_bounds__(256) void u32_from_i64_kernel(const long long* __restrict__ a, long long n,
                                                           uint32_t* __restrict__ o) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) o[p] = (uint32_t)a[p];
}
size_t au2(size_t v) { return (v + 255) / 256 * 256; }
}  // namespace

PINGS_API size_t pings_rows_scatter_add_scratch_bytes(int64_t n_pairs, int64_t rows) {
  const size_t n = (size_t)(n_pairs > 0 ? n_pairs : 1);
  return pings_rows::carve(nullptr, n_pairs, rows).total + 2 * au2(n * 4);
}

PINGS_API int pings_rows_scatter_add(const int64_t* dst_row, int64_t n_pairs, const float* src, int64_t ld,
                                     int32_t F, const float* w, const int64_t* src_row, int64_t rows,
                                     void* scratch, float* out, void* stream) {
  PINGS_ARG_CHECK(rows > 0 && n_pairs >= 0 && F > 0 && F <= 64 && ld >= F, "bad shape");
  PINGS_ARG_CHECK(scratch && out && (n_pairs == 0 || (dst_row && src)), "null pointer");
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope ps("rows_scatter_add", st);
  const size_t n = (size_t)(n_pairs > 0 ? n_pairs : 1);
  char* base = reinterpret_cast<char*>(scratch);
  uint32_t* keys = reinterpret_cast<uint32_t*>(base);
  uint32_t* srow = reinterpret_cast<uint32_t*>(base + au2(n * 4));
  pings_rows::Plan plan = pings_rows::carve(base + 2 * au2(n * 4), n_pairs, rows);
  const unsigned grid = (unsigned)((n_pairs + 255) / 256);
  if (n_pairs > 0) {
    hipLaunchKernelGGL(keys_from_i64_kernel, dim3(grid), dim3(256), 0, st, (const long long*)dst_row,
                       (long long)n_pairs, (uint32_t)rows, keys);
    PINGS_LAUNCH_CHECK();
    if (src_row) {
      hipLaunchKernelGGL(u32_from_i64_kernel, dim3(grid), dim3(256), 0, st, (const long long*)src_row,
                         (long long)n_pairs, srow);
      PINGS_LAUNCH_CHECK();
    }
  }
  if (int e = pings_rows::build(plan, keys, n_pairs, rows, st)) return e;
  return pings_rows::gather_sum(plan, rows, F, src, ld, src_row ? srow : nullptr, w, out, st);
}

// The same in two steps, for several tables that share one destination index (geo and colour features of one query
// batch): build the plan once, apply it per table.  `plan` = pings_rows_plan_bytes(n_pairs, rows) bytes, caller-owned.
PINGS_API size_t pings_rows_plan_bytes(int64_t n_pairs, int64_t rows) {
  const size_t n = (size_t)(n_pairs > 0 ? n_pairs : 1);
  return pings_rows::carve(nullptr, n_pairs, rows).total + au2(n * 4);
}

PINGS_API int pings_rows_plan_build(const int64_t* dst_row, int64_t n_pairs, int64_t rows, void* plan, void* stream) {
  PINGS_ARG_CHECK(rows > 0 && n_pairs >= 0 && plan && (n_pairs == 0 || dst_row), "bad arguments");
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope ps("rows_plan_build", st);
  const size_t n = (size_t)(n_pairs > 0 ? n_pairs : 1);
  char* base = reinterpret_cast<char*>(plan);
  uint32_t* keys = reinterpret_cast<uint32_t*>(base);
  pings_rows::Plan pl = pings_rows::carve(base + au2(n * 4), n_pairs, rows);
  if (n_pairs > 0) {
    hipLaunchKernelGGL(keys_from_i64_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, st,
                       (const long long*)dst_row, (long long)n_pairs, (uint32_t)rows, keys);
    PINGS_LAUNCH_CHECK();
  }
  return pings_rows::build(pl, keys, n_pairs, rows, st);
}

PINGS_API int pings_rows_plan_apply(const void* plan, int64_t n_pairs, int64_t rows, const float* src, int64_t ld,
                                    int32_t F, const float* w, float* out, void* stream) {
  PINGS_ARG_CHECK(rows > 0 && n_pairs >= 0 && plan && out && F > 0 && F <= 64 && ld >= F, "bad arguments");
  PINGS_ARG_CHECK(n_pairs == 0 || src, "null source");
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope ps("rows_plan_apply", st);
  const size_t n = (size_t)(n_pairs > 0 ? n_pairs : 1);
  char* base = reinterpret_cast<char*>(const_cast<void*>(plan));
  pings_rows::Plan pl = pings_rows::carve(base + au2(n * 4), n_pairs, rows);
  return pings_rows::gather_sum(pl, rows, F, src, ld, nullptr, w, out, st);
}
