// General C-ABI entry points: version + thread-local error string.
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

PINGS_API int pings_abi_version(void) { return 4; }

PINGS_API const char* pings_last_error(void) { return pings::g_err; }
