// General C-ABI entry points: version + thread-local error string.
#include <cstdarg>
#include <cstring>

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

PINGS_API int pings_abi_version(void) { return 1; }

PINGS_API const char* pings_last_error(void) { return pings::g_err; }
