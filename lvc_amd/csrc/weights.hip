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


// ---- every stale layer of a training step in a few launches (round 5) --------------------------------------------------------------
// One optimizer step invalidates the packed operands of every trainable layer: the forward one (mode 0; fp16 planes plain or
// row-scaled, or bf16 planes) and the data-gradient one (mode 1).  Per layer that was a pack launch, a split launch and a
// torch multiply for the row factors: ~330 launches of 5-7 us in the box-corrector step (R101, 110 trainable convolutions).
// Here one workgroup owns one packed row of one job; jobs travel as kernel arguments.
//   fmt 0: wp only; 2: + fp16 planes (w1, (w - w1) * 2048); 3: + bf16 planes (hi, mid, lo); 4: + ROW-SCALED fp16 planes (the
//   one-accumulator kernels' operand, see split_rowscaled_kernel in conv3x3_halo_s1.hip: e = 13 - floor(log2 max|row|), w1 =
//   fp16(w 2^e), w2 = fp16(w 2^e - w1)) and fac[row] = 2^-e / 16 * (fac_scale ? fac_scale[row] : 1) for rows < K.
#define PK_GROUP_MAX 24
struct PackJob {
  const float* w;
  const float* scale;        // mode 1: the FrozenBN scale folded into the data-gradient operand (or NULL)
  const float* fac_scale;    // fmt 4: the layer's epilogue scale multiplied into the row factors (or NULL)
  float* wp;
  unsigned short* planes;
  float* fac;
  int K, C, R, S, rows_pad, cin_pad, mode, fmt;
  int row_begin, pad_;
};
struct PackGroup { PackJob job[PK_GROUP_MAX]; int njobs, total_rows; int* err_word; };

__device__ __forceinline__ float pk_gather(const PackJob& f, int row, int k) {
  const int ci = k & 31; k >>= 5;
  const int s = k % f.S; k /= f.S;
  const int r = k % f.R; k /= f.R;
  const int cin = k * 32 + ci;
  float v = 0.f;
  if (f.mode == 0) {
    if (row < f.K && cin < f.C) v = f.w[(((size_t)row * f.C + cin) * f.R + r) * f.S + s];
  } else if (row < f.C && cin < f.K) {
    v = f.w[(((size_t)cin * f.C + row) * f.R + (f.R - 1 - r)) * f.S + (f.S - 1 - s)];
    if (f.scale) v *= f.scale[cin];
  }
  return v;
}

__global__ __launch_bounds__(256) void pack_group_kernel(const PackGroup g) {
  __shared__ float red[256];
  int j = 0;
  for (int q = 1; q < g.njobs; ++q)
    if ((int)blockIdx.x >= g.job[q].row_begin) j = q;
  const PackJob& f = g.job[j];
  const int row = blockIdx.x - f.row_begin;
  const int Kg = f.R * f.S * f.cin_pad;
  const long long plane = (long long)f.rows_pad * Kg;
  float s2 = 1.f;
  if (f.fmt == 4) {
    float mx = 0.f;
    for (int i = threadIdx.x; i < Kg; i += 256) {
      const float v = fabsf(pk_gather(f, row, i));
      mx = (v > mx || v != v) ? v : mx;
    }
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) {
        const float o = red[threadIdx.x + s];
        if (o > red[threadIdx.x] || o != o) red[threadIdx.x] = o;
      }
      __syncthreads();
    }
    mx = red[0];
    int e = 0;
    if (mx > 0.f && mx < INFINITY) {
      int ex;
      frexpf(mx, &ex);
      e = 13 - (ex - 1);
      e = e > 100 ? 100 : e < -100 ? -100 : e;
    }
    s2 = ldexpf(1.f, e);
    if (threadIdx.x == 0) {
      float fa = ldexpf(1.f, -e) * (1.f / 16.f);
      const int nrow = f.mode == 0 ? f.K : f.C;
      if (f.fac_scale && row < nrow) fa *= f.fac_scale[row];
      f.fac[row] = fa;
    }
  }
  for (int i = threadIdx.x; i < Kg; i += 256) {
    const float v = pk_gather(f, row, i);
    const long long o = (long long)row * Kg + i;
    f.wp[o] = v;
    if (f.fmt == 2 || f.fmt == 3) {
      emit_planes(v, o, plane, f.fmt, f.planes, g.err_word);
    } else if (f.fmt == 4) {
      const float vs = v * s2;
      const _Float16 h = (_Float16)vs;
      const _Float16 l = (_Float16)(vs - (float)h);
      f.planes[o] = __builtin_bit_cast(unsigned short, h);
      f.planes[plane + o] = __builtin_bit_cast(unsigned short, l);
    }
  }
}

// ptrs: 6 pointers per job (w, scale, fac_scale, wp, planes, fac); shapes: 8 ints per job (K, C, R, S, rows_pad, cin_pad, mode, fmt)
extern "C" int lvc_pack_group(int njobs, const void* const* ptrs, const int* shapes, int* err_word, void* stream) {
  LVC_CHECK_ARG(njobs > 0 && ptrs && shapes, "null pointer / no jobs");
  for (int j0 = 0; j0 < njobs; j0 += PK_GROUP_MAX) {
    const int nj = njobs - j0 < PK_GROUP_MAX ? njobs - j0 : PK_GROUP_MAX;
    PackGroup g;
    int rows = 0;
    for (int j = 0; j < nj; ++j) {
      const void* const* pp = ptrs + (size_t)(j0 + j) * 6;
      const int* sh = shapes + (size_t)(j0 + j) * 8;
      PackJob& f = g.job[j];
      f.w = (const float*)pp[0]; f.scale = (const float*)pp[1]; f.fac_scale = (const float*)pp[2];
      f.wp = (float*)pp[3]; f.planes = (unsigned short*)pp[4]; f.fac = (float*)pp[5];
      f.K = sh[0]; f.C = sh[1]; f.R = sh[2]; f.S = sh[3]; f.rows_pad = sh[4]; f.cin_pad = sh[5]; f.mode = sh[6]; f.fmt = sh[7];
      LVC_CHECK_ARG(f.w && f.wp && f.K > 0 && f.C > 0 && f.R > 0 && f.S > 0 && (f.mode == 0 || f.mode == 1), "bad job");
      LVC_CHECK_ARG(f.fmt == 0 || ((f.fmt == 2 || f.fmt == 3 || f.fmt == 4) && f.planes), "bad plane format / null planes");
      LVC_CHECK_ARG(f.fmt != 4 || f.fac, "fmt 4 needs the row-factor output");
      LVC_CHECK_ARG(f.cin_pad % 32 == 0 && f.cin_pad >= (f.mode == 0 ? f.C : f.K) && f.rows_pad >= (f.mode == 0 ? f.K : f.C), "bad padding");
      f.row_begin = rows; f.pad_ = 0;
      rows += f.rows_pad;
    }
    g.njobs = nj; g.total_rows = rows; g.err_word = err_word;
    hipLaunchKernelGGL(pack_group_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, g);
    LVC_CHECK_LAUNCH();
  }
  return LVC_OK;
}
