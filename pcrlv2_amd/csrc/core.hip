// Library core: version string, thread-local error message, launch check.
#include "common.h"

static thread_local char g_err[512] = "";

int pcrl_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int pcrl_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pcrl_fail(PCRL_ELAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
  return PCRL_OK;
}

extern "C" const char* pcrl_version(void) { return "pcrl_hip 0.1 (gfx950)"; }
extern "C" const char* pcrl_last_error(void) { return g_err; }
