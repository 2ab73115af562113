// weights.hip -- device-side packing of convolution weights: the reference's OIHW parameter (state_dict layout,
// detectron2/layers/wrappers.py:41-99) -> the [Kpad][Kg] fp32 operand of the conv kernels, and its split planes.
// In training every optimizer step changes the parameters, so the pack runs once per layer per step: one launch here
// instead of a dozen elementwise torch ops (and no host sync for the fp16 range check of the two-way split).
//
//   lvc_pack_conv_weights   mode 0: wp[ko][(c/32, r, s, c%32)] = w[ko][c][r][s]                       (forward operand)
//                           mode 1: wp[c ][(k/32, r, s, k%32)] = w[k][c][R-1-r][S-1-s] * scale[k]      (data gradient:
//                                   rows = the forward conv's INPUT channels, contraction over its output channels,
//                                   taps flipped; scale = the FrozenBatchNorm2d scale that follows the conv or NULL)
//   lvc_split_weights       planes of wp: 3 x bf16 (hi, mid, lo; w == hi + mid + lo) or 2 x fp16 (w1 = fp16(w),
//                           w2 = fp16((w - w1) * 2048)); an fp16 overflow raises bit 1 (value 2) of *err_word.
// Built with -ffp-contract=off (the residuals must be the exact differences).
#include "common.h"
#include <hip/hip_fp16.h>

__device__ __forceinline__ unsigned short bf16_rne(float x) {
  unsigned int u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_to_f(unsigned short h) { return __uint_as_float((unsigned int)h << 16); }

// the split planes of one packed weight: 3 x bf16 (hi, mid, lo) or 2 x fp16 (w1, (w - w1) * 2048)
__device__ __forceinline__ void emit_planes(float x, long long i, long long n, int planes, unsigned short* __restrict__ out,
                                            int* __restrict__ err_word) {
  if (planes == 3) {
    const unsigned short hi = bf16_rne(x);
    const float r1 = x - bf16_to_f(hi);
    const unsigned short mid = bf16_rne(r1);
    const float r2 = r1 - bf16_to_f(mid);
    out[i] = hi;
    out[n + i] = mid;
    out[2 * n + i] = bf16_rne(r2);
  } else {
    if (fabsf(x) > 65504.f && err_word) atomicOr(err_word, 2);
    const __half w1 = __float2half_rn(x);
    const __half w2 = __float2half_rn((x - __half2float(w1)) * 2048.f);
    out[i] = __half_as_ushort(w1);
    out[n + i] = __half_as_ushort(w2);
  }
}

__global__ __launch_bounds__(256) void pack_conv_weights_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                                                float* __restrict__ wp, int K, int C, int R, int S,
                                                                int rows, int Cin_pad, int Kg, int mode, long long total,
                                                                unsigned short* __restrict__ planes_out, int planes,
                                                                int* __restrict__ err_word) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int row = (int)(i / Kg);
  int k = (int)(i % Kg);
  const int ci = k & 31; k >>= 5;
  const int s = k % S; k /= S;
  const int r = k % R; k /= R;
  const int cin = k * 32 + ci;
  float v = 0.f;
  if (mode == 0) {
    if (row < K && cin < C) v = w[(((size_t)row * C + cin) * R + r) * S + s];
  } else {
    if (row < C && cin < K) {
      v = w[(((size_t)cin * C + row) * R + (R - 1 - r)) * S + (S - 1 - s)];
      if (scale) v *= scale[cin];
    }
  }
  wp[i] = v;
  if (planes_out) emit_planes(v, i, total, planes, planes_out, err_word);
  (void)rows; (void)Cin_pad;
}

extern "C" int lvc_pack_conv_weights(const float* w, const float* scale, float* wp, int K, int C, int R, int S, int rows_pad,
                                     int cin_pad, int mode, void* stream) {
  LVC_CHECK_ARG(w && wp && K > 0 && C > 0 && R > 0 && S > 0 && (mode == 0 || mode == 1), "bad arguments");
  LVC_CHECK_ARG(cin_pad % 32 == 0 && cin_pad >= (mode == 0 ? C : K) && rows_pad >= (mode == 0 ? K : C), "bad padding");
  const int Kg = R * S * cin_pad;
  const long long total = (long long)rows_pad * Kg;
  hipLaunchKernelGGL(pack_conv_weights_kernel, dim3((unsigned)lvc_cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                     scale, wp, K, C, R, S, rows_pad, cin_pad, Kg, mode, total, (unsigned short*)nullptr, 0, (int*)nullptr);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// lvc_pack_conv_weights + lvc_split_weights in one launch (a training step re-packs every trainable layer twice, for the
// forward and for the data gradient: ~110 launches of each kind per step): planes_out [planes][rows_pad][Kg].
extern "C" int lvc_pack_split_conv_weights(const float* w, const float* scale, float* wp, void* planes_out, int planes,
                                           int* err_word, int K, int C, int R, int S, int rows_pad, int cin_pad, int mode,
                                           void* stream) {
  LVC_CHECK_ARG(w && wp && planes_out && K > 0 && C > 0 && R > 0 && S > 0 && (mode == 0 || mode == 1), "bad arguments");
  LVC_CHECK_ARG(planes == 2 || planes == 3, "planes must be 2 (fp16) or 3 (bf16)");
  LVC_CHECK_ARG(cin_pad % 32 == 0 && cin_pad >= (mode == 0 ? C : K) && rows_pad >= (mode == 0 ? K : C), "bad padding");
  const int Kg = R * S * cin_pad;
  const long long total = (long long)rows_pad * Kg;
  hipLaunchKernelGGL(pack_conv_weights_kernel, dim3((unsigned)lvc_cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                     scale, wp, K, C, R, S, rows_pad, cin_pad, Kg, mode, total, (unsigned short*)planes_out, planes, err_word);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ wp, long long n, int planes,
                                                            unsigned short* __restrict__ out, int* __restrict__ err_word) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  emit_planes(wp[i], i, n, planes, out, err_word);
}

extern "C" int lvc_split_weights(const float* wp, long long n, int planes, void* out, int* err_word, void* stream) {
  LVC_CHECK_ARG(wp && out && n > 0 && (planes == 2 || planes == 3), "bad arguments");
  hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)lvc_cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, wp, n,
                     planes, (unsigned short*)out, err_word);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}
