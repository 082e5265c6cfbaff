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

// Stream-ordered read of up to 8 device words (32 bit each) into host memory WITHOUT a blocking wait inside the HIP
// runtime: one wave copies the words into pinned, device-mapped host memory behind everything already queued on `st` and
// the host polls a sequence number (abi.hip).  A blocking hipStreamSynchronize wakes through an interrupt, which on one
// box of the pool arrived only with a 60 Hz tick (16 ms per wait); polling does not depend on it and costs a few
// microseconds on every box.  Returns a PINGS status.
int host_read_words(const uint32_t* const* dev_words, int n, uint32_t* out, hipStream_t st);

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
