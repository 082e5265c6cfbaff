// Cell-block index of the neural-point map (include/pings_hip.h: pings_knn_blocks_build) for gfx950.
//
// Why.  The search kernels (knn_common.hpp) were measured at 122 half-line requests per query — 81 table probes keyed
// by a hash of the cell, i.e. 81 unrelated sectors, plus position / flag / index gathers per live candidate — and at
// 76 % of the rate at which the chip serves such requests at all (profiles/r02/pmc_calibration.json: 54.5 G/s).  The
// way down is fewer requests, so the same information is laid out by LOCATION: 4x4x4 cells share one 32-byte block
// entry (coordinate, occupancy mask, record base) and every registered point has one 32-byte record with everything
// the reference's tests read (model/neural_gaussians.py:1088-1105, :544-554).
//
// Exactness.  The reference answers "which point lives in cell c" with table[hash(c)]; the index answers it with "the
// registered point whose own cell is c".  Point i is REGISTERED iff table[hash(cell(i))] == i.
//   table path returns i for cell c  <=>  table[hash(c)] == i.
//   (a) If every non-empty slot holds a registered point, then hash(cell(i)) == hash(c).
//   (b) An accepted candidate is within sqrt(max_valid_dist2) of the query, so cell(i) and c differ by at most
//       D = ceil(sqrt(max_valid_dist2) / resolution) + 1 + max|neighbor_dx| per axis; if no non-zero offset within
//       [-D, D]^3 hashes to 0 (mod buffer_size), cell(i) == c.
// Under (a) and (b) both paths hand the same point to the same tests in the same lane, hence identical outputs (rejected
// far-away collisions are -1 in both).  (a) is counted on the device (registered points == non-empty slots: registered
// points occupy distinct slots), (b) is an integer check on the host; status[0] = 1 only if both hold and every block
// coordinate fits its 21 bits, and the search kernels read that word before taking this path.
#include "knn_common.hpp"

namespace {
using namespace pings_knn;

struct BuildArgs {
  pings_knn_map m;
  long long N, T;
  BlockEntry* blocks;
  unsigned block_mask;
  BlockRec* recs;
  int* status;  // [0] ok, [1] registered, [2] non-empty slots, [3] out-of-range, [4] records
};

// cell, slot and registration of point i
__device__ inline bool point_cell(const BuildArgs& a, long long i, double inv_S, int& cx, int& cy, int& cz, bool& in_range) {
  const float px = a.m.neural_points[3 * i], py = a.m.neural_points[3 * i + 1], pz = a.m.neural_points[3 * i + 2];
  const float fx = floorf(px / a.m.resolution), fy = floorf(py / a.m.resolution), fz = floorf(pz / a.m.resolution);
  const long long gx = (long long)fx, gy = (long long)fy, gz = (long long)fz;
  const long long h = hash_slot(gx * P0 + gy * P1 + gz * P2, a.m.buffer_size, inv_S);
  const bool reg = a.m.table[h] == i;
  const float lim = (float)(4 * BLOCK_COORD_LIMIT - 256);
  in_range = fabsf(fx) < lim && fabsf(fy) < lim && fabsf(fz) < lim;
  cx = in_range ? (int)fx : 0; cy = in_range ? (int)fy : 0; cz = in_range ? (int)fz : 0;
  return reg;
}

__device__ inline void block_add(int* counter, int v) {  // one atomic per workgroup
  __shared__ int s;
  if (threadIdx.x == 0) s = 0;
  __syncthreads();
  const unsigned long long b = __ballot(v != 0);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(&s, __popcll(b));
  __syncthreads();
  if (threadIdx.x == 0 && s) atomicAdd(counter, s);
  __syncthreads();
}

__global__ __launch_bounds__(256) void blocks_insert_kernel(BuildArgs a) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const double inv_S = 1.0 / (double)a.m.buffer_size;
  int cx = 0, cy = 0, cz = 0;
  bool in_range = true, reg = false;
  if (i < a.N) reg = point_cell(a, i, inv_S, cx, cy, cz, in_range);
  if (reg && in_range) {
    const unsigned long long key = block_key(cx >> 2, cy >> 2, cz >> 2);
    unsigned s = block_slot(key, a.block_mask);
    for (;;) {
      const unsigned long long prev = atomicCAS(&a.blocks[s].key, 0ull, key);
      if (prev == 0ull || prev == key) break;
      s = (s + 1u) & a.block_mask;
    }
    atomicOr(&a.blocks[s].mask, 1ull << cell_bit(cx, cy, cz));
  }
  block_add(&a.status[1], reg ? 1 : 0);
  block_add(&a.status[3], (reg && !in_range) ? 1 : 0);
}

// record base of every block: workgroup scan of the occupancy counts, one atomic per workgroup
__global__ __launch_bounds__(256) void blocks_base_kernel(BuildArgs a) {
  __shared__ unsigned s_wave[4];
  __shared__ unsigned s_base;
  const unsigned e = blockIdx.x * 256u + threadIdx.x;
  const unsigned n = a.block_mask + 1u;
  const unsigned cnt = e < n ? (unsigned)__popcll(a.blocks[e].mask) : 0u;
  // inclusive scan inside the wave
  unsigned v = cnt;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  if (lane == 63) s_wave[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    s_base = total ? (unsigned)atomicAdd(&a.status[4], (int)total) : 0u;
  }
  __syncthreads();
  unsigned off = s_base;
  for (int w = 0; w < wave; ++w) off += s_wave[w];
  if (e < n && cnt) a.blocks[e].base = off + v - cnt;
}

__global__ __launch_bounds__(256) void blocks_fill_kernel(BuildArgs a) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.N) return;
  const double inv_S = 1.0 / (double)a.m.buffer_size;
  int cx, cy, cz;
  bool in_range;
  if (!point_cell(a, i, inv_S, cx, cy, cz, in_range) || !in_range) return;
  const unsigned long long key = block_key(cx >> 2, cy >> 2, cz >> 2);
  unsigned s = block_slot(key, a.block_mask);
  while (a.blocks[s].key != key) s = (s + 1u) & a.block_mask;
  const unsigned bit = cell_bit(cx, cy, cz);
  const unsigned r = a.blocks[s].base + (unsigned)__popcll(a.blocks[s].mask & ((1ull << bit) - 1ull));
  BlockRec rec;
  rec.x = a.m.neural_points[3 * i];
  rec.y = a.m.neural_points[3 * i + 1];
  rec.z = a.m.neural_points[3 * i + 2];
  rec.gidx = (int)i;
  rec.loc = a.m.global2local ? (int)a.m.global2local[i] : (int)i;
  rec.td = 0.f;
  if (a.m.point_ts_create && a.m.travel_dist && a.T > 0) {
    long long t = a.m.point_ts_create[i];
    t = t < 0 ? 0 : (t >= a.T ? a.T - 1 : t);
    rec.td = a.m.travel_dist[t];
  }
  rec.flags = ((a.m.free_mask && a.m.free_mask[i]) ? 1u : 0u) | ((!a.m.valid_mask || a.m.valid_mask[i]) ? 2u : 0u);
  rec.pad = 0u;
  a.recs[r] = rec;
}

__global__ __launch_bounds__(256) void table_count_kernel(const long long* __restrict__ table, long long S, int* counter) {
  const long long n2 = S / 2;  // 16 bytes per lane
  const long long nthreads = (long long)gridDim.x * blockDim.x;
  int c = 0;
  const longlong2* t2 = reinterpret_cast<const longlong2*>(table);
  for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n2; j += nthreads) {
    const longlong2 v = t2[j];
    c += (v.x >= 0) + (v.y >= 0);
  }
  if ((S & 1) && blockIdx.x == 0 && threadIdx.x == 0) c += table[S - 1] >= 0;
  // workgroup sum, one atomic
  __shared__ int s[4];
  for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = s[0] + s[1] + s[2] + s[3];
    if (t) atomicAdd(counter, t);
  }
}

__global__ void blocks_finish_kernel(int* status, int host_ok) {
  status[0] = (host_ok && status[1] == status[2] && status[3] == 0 && status[4] == status[1]) ? 1 : 0;
}

// (b) of the header: no non-zero cell offset within [-D, D]^3 hashes to slot 0
bool offsets_collision_free(long long S, int D) {
  for (int x = -D; x <= D; ++x)
    for (int y = -D; y <= D; ++y)
      for (int z = -D; z <= D; ++z) {
        if (x == 0 && y == 0 && z == 0) continue;
        if ((x * P0 + y * P1 + z * P2) % S == 0) return false;
      }
  return true;
}

}  // namespace

PINGS_API size_t pings_knn_blocks_entries(int64_t num_points) {
  size_t cap = 1024;
  const size_t want = (size_t)(num_points > 0 ? num_points : 1) * 2;  // every point its own block: load factor 0.5
  while (cap < want) cap <<= 1;
  return cap;
}

PINGS_API int pings_knn_blocks_build(const pings_knn_map* m, int64_t num_points, int64_t num_timestamps,
                                     int32_t max_abs_dx, void* blocks, size_t entries, void* records,
                                     int32_t* status, void* stream) {
  PINGS_ARG_CHECK(m && m->table && m->neural_points && m->buffer_size > 0 && m->buffer_size < (1LL << 31), "bad map");
  PINGS_ARG_CHECK(m->resolution > 0.f && m->max_valid_dist2 > 0.f && max_abs_dx >= 0 && max_abs_dx < 64, "bad geometry");
  PINGS_ARG_CHECK(blocks && records && status, "null output");
  PINGS_ARG_CHECK(num_points >= 0 && num_points < (1LL << 31), "num_points out of range");
  PINGS_ARG_CHECK(entries >= 1024 && (entries & (entries - 1)) == 0 && entries <= (1ull << 31) &&
                      entries >= pings_knn_blocks_entries(num_points), "entries must be a power of two >= 2 num_points");
  PINGS_ARG_CHECK((((uintptr_t)blocks | (uintptr_t)records) & 31u) == 0 && (((uintptr_t)m->table) & 15u) == 0,
                  "blocks / records must be 32-byte aligned, the table 16-byte aligned");
  static_assert(sizeof(BlockEntry) == 32 && sizeof(BlockRec) == 32, "32-byte entries");
  hipStream_t st = pings::as_stream(stream);
  pings::prof::Scope ps("knn_blocks_build", st);
  const int D = (int)ceilf(sqrtf(m->max_valid_dist2) / m->resolution) + 1 + max_abs_dx;
  const int host_ok = (D <= 64 && offsets_collision_free(m->buffer_size, D)) ? 1 : 0;
  BuildArgs a;
  a.m = *m;
  a.N = num_points;
  a.T = num_timestamps;
  a.blocks = reinterpret_cast<BlockEntry*>(blocks);
  a.block_mask = (unsigned)(entries - 1);
  a.recs = reinterpret_cast<BlockRec*>(records);
  a.status = status;
  PINGS_HIP_CHECK(hipMemsetAsync(blocks, 0, entries * sizeof(BlockEntry), st));
  PINGS_HIP_CHECK(hipMemsetAsync(status, 0, 8 * sizeof(int32_t), st));
  // record 0 is what lanes without a candidate read: keep it defined
  PINGS_HIP_CHECK(hipMemsetAsync(records, 0, sizeof(BlockRec), st));
  if (num_points > 0) {
    const unsigned gp = (unsigned)((num_points + 255) / 256);
    hipLaunchKernelGGL(blocks_insert_kernel, dim3(gp), dim3(256), 0, st, a);
    hipLaunchKernelGGL(blocks_base_kernel, dim3((unsigned)(entries / 256)), dim3(256), 0, st, a);
    hipLaunchKernelGGL(blocks_fill_kernel, dim3(gp), dim3(256), 0, st, a);
  }
  hipLaunchKernelGGL(table_count_kernel, dim3(256 * 16), dim3(256), 0, st, (const long long*)m->table,
                     (long long)m->buffer_size, status + 2);
  hipLaunchKernelGGL(blocks_finish_kernel, dim3(1), dim3(1), 0, st, status, host_ok);
  PINGS_LAUNCH_CHECK();
  return PINGS_OK;
}
