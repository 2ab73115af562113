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

// Range slot of the NEXT fp16-split conv/GEMM launch on this thread: the kernels raise bit 1 of workspace word
// LVC_MAX_WORKERS + slot when an operand leaves their range, so the host can tell WHICH layer it was (slot 0 = the shared word).
#define LVC_RANGE_SLOTS 1024
static thread_local int g_range_slot = 0;
extern "C" void lvc_set_range_slot(int slot) { g_range_slot = (slot > 0 && slot < LVC_RANGE_SLOTS) ? slot : 0; }
extern "C" int lvc_range_slot(void) { return g_range_slot; }
extern "C" int lvc_range_slots(void) { return LVC_RANGE_SLOTS; }
