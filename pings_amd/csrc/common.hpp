// Shared helpers for the libpings_hip translation units (device + host).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#include "pings_hip.h"

namespace pings {

void set_error(const char* fmt, ...);  // abi.hip

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

template <typename T>
__host__ __device__ constexpr T ceil_div(T a, T b) { return (a + b - 1) / b; }

constexpr int kWave = 64;  // gfx950 wavefront

// Optional per-stage HIP-event timing (off by default; bench.py turns it on to get the
// per-kernel durations its roofline figures are computed from).
namespace prof {
bool enabled(const char* name);
void begin(const char* name, hipStream_t st);
void end(hipStream_t st);
struct Scope {
  hipStream_t st;
  bool on;
  Scope(const char* name, hipStream_t s) : st(s), on(enabled(name)) { if (on) begin(name, st); }
  ~Scope() { if (on) end(st); }
};
}  // namespace prof

}  // namespace pings

// Every C-ABI entry point returns 0 on success; HIP failures are turned into
// PINGS_ERR_HIP with a message retrievable through pings_last_error().
#define PINGS_HIP_CHECK(expr)                                                        \
  do {                                                                               \
    hipError_t _e = (expr);                                                          \
    if (_e != hipSuccess) {                                                          \
      pings::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr,                  \
                       hipGetErrorString(_e));                                       \
      return PINGS_ERR_HIP;                                                          \
    }                                                                                \
  } while (0)

#define PINGS_LAUNCH_CHECK() PINGS_HIP_CHECK(hipGetLastError())

#define PINGS_ARG_CHECK(cond, msg)                                                   \
  do {                                                                               \
    if (!(cond)) {                                                                   \
      pings::set_error("%s:%d invalid argument: %s (%s)", __FILE__, __LINE__, msg,   \
                       #cond);                                                       \
      return PINGS_ERR_ARG;                                                          \
    }                                                                                \
  } while (0)
