// api.hip — library-level entry points of the C-ABI (version, error text, device probe).
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void lmx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int lmx_version(void) { return LMX_VERSION; }
extern "C" const char* lmx_last_error(void) { return g_err; }
extern "C" int lmx_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    lmx_set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
    (void)hipGetLastError();
    return LMX_EHIP;
  }
  return n;
}
