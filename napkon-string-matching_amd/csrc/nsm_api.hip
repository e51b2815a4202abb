// Library-wide pieces of the C ABI (include/nsm_hip.h): version, error text.
#include <stdarg.h>
#include <stdio.h>

#include "nsm_common.hpp"

namespace nsm {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}

int hip_status(hipError_t err, const char* what) {
  if (err == hipSuccess) return 0;
  set_error("%s: %s", what, hipGetErrorString(err));
  return static_cast<int>(err);
}

}  // namespace nsm

extern "C" int nsm_abi_version(void) { return NSM_ABI_VERSION; }
extern "C" const char* nsm_last_error(void) { return nsm::g_error; }
