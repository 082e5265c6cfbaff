// General C-ABI entry points: version + thread-local error string.
#include <chrono>
#include <cstdarg>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "common.hpp"

namespace pings {
static thread_local char g_err[1024] = {0};

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace pings

// ---------------------------------------------------------------- stage profiler
namespace pings {
namespace prof {
namespace {
struct Entry {
  const char* name;
  hipEvent_t a, b;
};
bool g_on = false;
std::string g_only;  // non-empty: record only the stage of this name
std::vector<Entry> g_entries;
std::vector<hipEvent_t> g_pool;
hipEvent_t g_open_a = nullptr;
const char* g_open_name = nullptr;

hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
}  // namespace

bool enabled(const char* name) { return g_on && (g_only.empty() || g_only == name); }

void begin(const char* name, hipStream_t st) {
  g_open_a = get_event();
  g_open_name = name;
  if (g_open_a) (void)hipEventRecord(g_open_a, st);
}

void end(hipStream_t st) {
  hipEvent_t b = get_event();
  if (b) (void)hipEventRecord(b, st);
  if (g_open_a && b) g_entries.push_back({g_open_name, g_open_a, b});
  g_open_a = nullptr;
}
}  // namespace prof
}  // namespace pings

// ---------------------------------------------------------------- polled read-back (common.hpp: host_read_words)
namespace pings {
namespace {
struct ReadRecord {
  uint32_t w[8];
  uint32_t seq;
  uint32_t pad[7];
};
struct ReadPtrs {
  const uint32_t* p[8];
};
__global__ __launch_bounds__(64) void read_words_kernel(ReadPtrs ptrs, int n, ReadRecord* out, uint32_t seq) {
  const int lane = threadIdx.x;
  if (lane < 8) out->w[lane] = (lane < n && ptrs.p[lane]) ? *ptrs.p[lane] : 0u;
  __threadfence_system();
  __syncthreads();
  if (lane == 0) __hip_atomic_store(&out->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace

int host_read_words(const uint32_t* const* dev_words, int n, uint32_t* out, hipStream_t st) {
  PINGS_ARG_CHECK(n >= 1 && n <= 8 && dev_words && out, "1..8 words");
  static thread_local ReadRecord* rec = nullptr;
  static thread_local ReadRecord* rec_dev = nullptr;
  static thread_local uint32_t counter = 0;
  if (!rec) {
    PINGS_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&rec), sizeof(ReadRecord), hipHostMallocMapped | hipHostMallocPortable));
    PINGS_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&rec_dev), rec, 0));
    rec->seq = 0;
  }
  ReadPtrs ptrs;
  for (int i = 0; i < 8; ++i) ptrs.p[i] = i < n ? dev_words[i] : nullptr;
  const uint32_t seq = ++counter ? counter : ++counter;
  hipLaunchKernelGGL(read_words_kernel, dim3(1), dim3(64), 0, st, ptrs, n, rec_dev, seq);
  PINGS_LAUNCH_CHECK();
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  while (__atomic_load_n(&rec->seq, __ATOMIC_ACQUIRE) != seq) {
    if ((++spins & 0x3FFu) == 0) {
      const hipError_t q = hipStreamQuery(st);
      if (q == hipErrorNotReady) (void)hipGetLastError();
      else if (q != hipSuccess) PINGS_HIP_CHECK(q);
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
        PINGS_HIP_CHECK(hipStreamSynchronize(st));
        PINGS_ARG_CHECK(__atomic_load_n(&rec->seq, __ATOMIC_ACQUIRE) == seq, "read-back never arrived");
      }
    }
    __builtin_ia32_pause();
  }
  for (int i = 0; i < n; ++i) out[i] = rec->w[i];
  return PINGS_OK;
}
}  // namespace pings

PINGS_API int pings_prof_enable(int on) {
  pings::prof::g_on = on != 0;
  return PINGS_OK;
}

PINGS_API int pings_prof_only(const char* stage) {
  pings::prof::g_only = stage ? stage : "";
  return PINGS_OK;
}

// Synchronises the device, sums the recorded stages by name and writes
// "name count total_ms\n" lines into buf (host).  Clears the record.
PINGS_API int pings_prof_report(char* buf, size_t cap) {
  using namespace pings::prof;
  if (!buf || cap == 0) return PINGS_ERR_ARG;
  PINGS_HIP_CHECK(hipDeviceSynchronize());
  std::map<std::string, std::pair<int, double>> agg;
  std::vector<std::string> order;
  for (auto& e : g_entries) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
      auto it = agg.find(e.name);
      if (it == agg.end()) {
        agg[e.name] = {1, (double)ms};
        order.push_back(e.name);
      } else {
        it->second.first += 1;
        it->second.second += ms;
      }
    }
    g_pool.push_back(e.a);
    g_pool.push_back(e.b);
  }
  g_entries.clear();
  size_t off = 0;
  buf[0] = 0;
  for (auto& n : order) {
    int w = snprintf(buf + off, cap - off, "%s %d %.6f\n", n.c_str(), agg[n].first, agg[n].second);
    if (w < 0 || (size_t)w >= cap - off) return PINGS_ERR_CAPACITY;
    off += (size_t)w;
  }
  return PINGS_OK;
}

PINGS_API int pings_abi_version(void) { return PINGS_ABI_VERSION; }

PINGS_API const char* pings_last_error(void) { return pings::g_err; }
