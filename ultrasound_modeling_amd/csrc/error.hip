// Error reporting for the C ABI: a thread-local message, never an exception or an exit.
#include <stdarg.h>
#include "common.h"

static thread_local char g_err[512] = "";

void usseg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int usseg_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    usseg_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return USSEG_ERR_LAUNCH;
  }
  return USSEG_OK;
}

extern "C" const char* usseg_last_error(void) { return g_err; }
extern "C" int usseg_version(void) { return 1; }
