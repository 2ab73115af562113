// Error reporting for the C ABI: functions return an int status; the message of the last failure on
// the calling thread is available through lvc_last_error() (reference behaviour being replaced:
// AT_ASSERTM/AT_ERROR -> C++ exception -> Python RuntimeError, csrc/ROIAlign/ROIAlign_cuda.cu:318-324;
// the Python host re-raises RuntimeError with this text).
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

extern "C" void lvc_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

extern "C" const char* lvc_last_error(void) { return g_err; }

extern "C" int lvc_abi_version(void) { return 1; }
