// Shared helpers for the lvc_amd C-ABI library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define LVC_OK 0
#define LVC_ERR_INVALID 1   // bad argument (shape / alignment / null)
#define LVC_ERR_HIP 2       // a HIP runtime call failed (see lvc_last_error)
#define LVC_ERR_DOMAIN 3    // data-dependent precondition violated (e.g. negative RoI size)

extern "C" void lvc_set_error(const char* fmt, ...);
extern "C" int lvc_range_slot(void);   // see common.cpp
extern "C" int lvc_range_slots(void);

#define LVC_CHECK_ARG(cond, msg)                       \
  do {                                                 \
    if (!(cond)) {                                     \
      lvc_set_error("%s: %s", __func__, msg);          \
      return LVC_ERR_INVALID;                          \
    }                                                  \
  } while (0)

#define LVC_CHECK_LAUNCH()                                                         \
  do {                                                                             \
    hipError_t e_ = hipGetLastError();                                             \
    if (e_ != hipSuccess) {                                                        \
      lvc_set_error("%s: kernel launch failed: %s", __func__, hipGetErrorString(e_)); \
      return LVC_ERR_HIP;                                                          \
    }                                                                              \
  } while (0)

static inline int lvc_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t lvc_cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// XCD-aware bijective remap of a 1-D workgroup id: consecutive *logical* ids land on the same
// XCD (hardware places block b on XCD b % 8), so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int lvc_xcd_remap(int bid, int nwg) {
  const int NX = 8;
  int xcd = bid % NX, within = bid / NX;
  int q = nwg / NX, r = nwg % NX;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + within;
}
