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

// Zero-fill of a device buffer on the caller's stream (the runtime's fill: no kernel of ours, none of ATen's) -- e.g. dx of a 1x1 stride-2
// convolution's data gradient, three of whose four parity classes receive nothing (pcrl_conv2d_dgrad_s2).
extern "C" int pcrl_zero(void* p, size_t bytes, pcrl_stream_t stream) {
  PCRL_REQUIRE(p && bytes > 0, "zero: bad arguments");
  const hipError_t e = hipMemsetAsync(p, 0, bytes, as_stream(stream));
  if (e != hipSuccess) return pcrl_fail(PCRL_ELAUNCH, "zero: hipMemsetAsync failed: %s", hipGetErrorString(e));
  return PCRL_OK;
}
