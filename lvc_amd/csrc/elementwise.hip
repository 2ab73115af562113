// elementwise.hip -- the non-GEMM layers of the trunk, NHWC, float4-vectorised (HBM-bound).
//   lvc_preprocess_nhwc4 : GeneralizedRCNN.preprocess_image (lvc/modeling/meta_arch/rcnn.py:324-333:
//                          (x - mean) / std per channel) + ImageList.from_tensors zero padding
//                          (detectron2/structures/image_list.py:95-119) + CHW -> NHWC4 relayout.
//   lvc_maxpool2d_nhwc   : F.max_pool2d of BasicStem (resnet.py:591, k3 s2 p1) and LastLevelMaxPool
//                          (fpn.py:176, k1 s2 p0); padding behaves as -inf like ATen.
#include "common.h"

template <typename T>
__global__ void preprocess_kernel(const T* __restrict__ img, int h, int w, float m0, float m1, float m2, float s0,
                                  float s1, float s2, float* __restrict__ out, int Hp, int Wp) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= Wp) return;
  float4 v = {0.f, 0.f, 0.f, 0.f};
  if (y < h && x < w) {
    const size_t plane = (size_t)h * w, o = (size_t)y * w + x;
    v.x = ((float)img[o] - m0) / s0;
    v.y = ((float)img[plane + o] - m1) / s1;
    v.z = ((float)img[2 * plane + o] - m2) / s2;
  }
  *reinterpret_cast<float4*>(out + ((size_t)y * Wp + x) * 4) = v;
}

// image: CHW (3 planes) float32 (dtype 0) or uint8 (dtype 1) on device; out: this image's [Hp,Wp,4] slot.
extern "C" int lvc_preprocess_nhwc4(const void* image, int dtype, int h, int w, const float* mean3,
                                    const float* std3, float* out, int Hp, int Wp, void* stream) {
  LVC_CHECK_ARG(image && out && mean3 && std3, "null pointer");
  LVC_CHECK_ARG(h > 0 && w > 0 && Hp >= h && Wp >= w, "bad sizes");
  LVC_CHECK_ARG(dtype == 0 || dtype == 1, "dtype must be 0 (f32) or 1 (u8)");
  dim3 block(256), grid(lvc_cdiv(Wp, 256), Hp);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0)
    hipLaunchKernelGGL(preprocess_kernel<float>, grid, block, 0, st, (const float*)image, h, w, mean3[0], mean3[1],
                       mean3[2], std3[0], std3[1], std3[2], out, Hp, Wp);
  else
    hipLaunchKernelGGL(preprocess_kernel<unsigned char>, grid, block, 0, st, (const unsigned char*)image, h, w,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], out, Hp, Wp);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// The same for a whole batch in one launch (blockIdx.z = image): up to 16 images per launch, their pointers and sizes by value
// (no pointer table in device memory).  Results identical to lvc_preprocess_nhwc4 image by image.
struct PreprocessBatch {
  const void* img[16];
  int h[16], w[16];
};

template <typename T>
__global__ void preprocess_batch_kernel(PreprocessBatch b, float m0, float m1, float m2, float s0, float s1, float s2,
                                        float* __restrict__ out, int Hp, int Wp) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, n = blockIdx.z;
  if (x >= Wp) return;
  const T* img = static_cast<const T*>(b.img[n]);
  const int h = b.h[n], w = b.w[n];
  float4 v = {0.f, 0.f, 0.f, 0.f};
  if (y < h && x < w) {
    const size_t plane = (size_t)h * w, o = (size_t)y * w + x;
    v.x = ((float)img[o] - m0) / s0;
    v.y = ((float)img[plane + o] - m1) / s1;
    v.z = ((float)img[2 * plane + o] - m2) / s2;
  }
  *reinterpret_cast<float4*>(out + (((size_t)n * Hp + y) * Wp + x) * 4) = v;
}

// images: B device pointers to CHW images (all float32: dtype 0, or all uint8: dtype 1) of sizes hs[i] x ws[i]; out [B,Hp,Wp,4].
extern "C" int lvc_preprocess_batch_nhwc4(const void* const* images, int dtype, const int* hs, const int* ws, int B,
                                          const float* mean3, const float* std3, float* out, int Hp, int Wp, void* stream) {
  LVC_CHECK_ARG(images && hs && ws && out && mean3 && std3 && B > 0, "null pointer / empty batch");
  LVC_CHECK_ARG(dtype == 0 || dtype == 1, "dtype must be 0 (f32) or 1 (u8)");
  hipStream_t st = (hipStream_t)stream;
  for (int b0 = 0; b0 < B; b0 += 16) {
    PreprocessBatch pb;
    const int nb = B - b0 < 16 ? B - b0 : 16;
    for (int i = 0; i < 16; ++i) {
      const int j = i < nb ? b0 + i : b0;
      LVC_CHECK_ARG(images[j] && hs[j] > 0 && ws[j] > 0 && Hp >= hs[j] && Wp >= ws[j], "bad image");
      pb.img[i] = images[j]; pb.h[i] = hs[j]; pb.w[i] = ws[j];
    }
    dim3 block(256), grid(lvc_cdiv(Wp, 256), Hp, nb);
    float* o = out + (size_t)b0 * Hp * Wp * 4;
    if (dtype == 0)
      hipLaunchKernelGGL(preprocess_batch_kernel<float>, grid, block, 0, st, pb, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], o, Hp, Wp);
    else
      hipLaunchKernelGGL(preprocess_batch_kernel<unsigned char>, grid, block, 0, st, pb, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], o, Hp, Wp);
  }
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

__global__ void maxpool_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C4,
                                    int Ho, int Wo, int k, int stride, int pad) {
  const long long total = (long long)N * Ho * Wo * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long long p = i / C4;
    const int wo = (int)(p % Wo); p /= Wo;
    const int ho = (int)(p % Ho);
    const int n = (int)(p / Ho);
    float4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int r = 0; r < k; ++r) {
      const int hi = ho * stride - pad + r;
      if (hi < 0 || hi >= H) continue;
      for (int s = 0; s < k; ++s) {
        const int wi = wo * stride - pad + s;
        if (wi < 0 || wi >= W) continue;
        const float4 v = *reinterpret_cast<const float4*>(x + (((size_t)n * H + hi) * W + wi) * C4 * 4 + c * 4);
        m.x = v.x > m.x ? v.x : m.x; m.y = v.y > m.y ? v.y : m.y;
        m.z = v.z > m.z ? v.z : m.z; m.w = v.w > m.w ? v.w : m.w;
      }
    }
    *reinterpret_cast<float4*>(y + (size_t)i * 4) = m;
  }
}

extern "C" int lvc_maxpool2d_nhwc(const float* x, float* y, int N, int H, int W, int C, int k, int stride, int pad,
                                  void* stream) {
  LVC_CHECK_ARG(x && y, "null pointer");
  LVC_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "C must be a positive multiple of 4");
  LVC_CHECK_ARG(k >= 1 && stride >= 1 && pad >= 0 && pad <= k / 2, "bad window");
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  LVC_CHECK_ARG(Ho > 0 && Wo > 0, "empty output");
  const long long total = (long long)N * Ho * Wo * (C / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(maxpool_nhwc_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C / 4, Ho,
                     Wo, k, stride, pad);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// Row-wise L2 normalisation: y[m,:] = (x[m,:] - mu) / den,  den = |x-mu| + eps (mode 0, the form of
// CosineSimOutputLayers, lvc/modeling/roi_heads/fast_rcnn.py:822-833) or max(|x-mu|, eps) (mode 1, the
// form inside F.cosine_similarity used by tools/run_nearest_neighbours.py:150-153).  One wave per row.
// VEC (D % 4 == 0 and 16-byte aligned rows): lane l owns elements 256 i + 4 l .. + 3 (one dwordx4 per 256-element slice) and adds
// their squares in that order; otherwise lane l owns elements l, l + 64, ...  The 64 partial sums meet in a butterfly.
// rownorm_h_kernel below sums in exactly the same orders.
template <bool VEC>
__device__ __forceinline__ float rownorm_sumsq(const float* __restrict__ xr, const float* __restrict__ mu, int D, int lane) {
  float ss = 0.f;
  if constexpr (VEC) {
    for (int d = lane * 4; d < D; d += 256) {
      float4 v = *reinterpret_cast<const float4*>(xr + d);
      if (mu) { const float4 m = *reinterpret_cast<const float4*>(mu + d); v.x -= m.x; v.y -= m.y; v.z -= m.z; v.w -= m.w; }
      ss += v.x * v.x; ss += v.y * v.y; ss += v.z * v.z; ss += v.w * v.w;
    }
  } else {
    for (int d = lane; d < D; d += 64) {
      float v = xr[d] - (mu ? mu[d] : 0.f);
      ss += v * v;
    }
  }
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  return ss;
}

template <bool VEC>
__global__ __launch_bounds__(256) void rownorm_kernel(const float* __restrict__ x, const float* __restrict__ mu,
                                                      float* __restrict__ y, int M, int D, int ldx, int ldy,
                                                      float eps, int mode) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* xr = x + (size_t)row * ldx;
  const float nrm = sqrtf(rownorm_sumsq<VEC>(xr, mu, D, lane));
  const float den = mode == 0 ? nrm + eps : (nrm > eps ? nrm : eps);
  float* yr = y + (size_t)row * ldy;
  if constexpr (VEC) {
    for (int d = lane * 4; d < D; d += 256) {
      float4 v = *reinterpret_cast<const float4*>(xr + d);
      if (mu) { const float4 m = *reinterpret_cast<const float4*>(mu + d); v.x -= m.x; v.y -= m.y; v.z -= m.z; v.w -= m.w; }
      v.x /= den; v.y /= den; v.z /= den; v.w /= den;
      *reinterpret_cast<float4*>(yr + d) = v;
    }
  } else {
    for (int d = lane; d < D; d += 64) yr[d] = (xr[d] - (mu ? mu[d] : 0.f)) / den;
  }
}

// lvc_rownorm for the two-stage kNN sweep: the normalised rows rounded to fp16 (round to nearest even; the operand of the
// pre-filter GEMM, gemm_h.hip), the denominators den [M] (so that a consumer can redo (x - mu) / den bit for bit), the
// 2-norm of each row's fp16 rounding residual resid [M] (what bounds the pre-filter's error for that row) and -- optionally --
// the fp32 rows themselves (bit-identical to rownorm_kernel -- same summation order, same division -- for the widths this file
// vectorises here, D <= 2048 with 16-byte aligned rows, and for unaligned rows; wider aligned rows are summed in the scalar order
// here and in the vectorised order there: equal to fp32 rounding, not bit for bit).
template <int NV, bool VEC>   // VEC: NV * 256 >= D, the centred row stays in registers between the two passes; else it is read twice
__global__ __launch_bounds__(256) void rownorm_h_kernel(const float* __restrict__ x, const float* __restrict__ mu,
                                                        float* __restrict__ y, _Float16* __restrict__ yh, float* __restrict__ dens,
                                                        float* __restrict__ resid, int M, int D, int ldx, float eps, int mode) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* xr = x + (size_t)row * ldx;
  float ss = 0.f;
  float4 c[NV > 0 ? NV : 1];
  if constexpr (VEC) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int d = i * 256 + lane * 4;
      float4 v = {0.f, 0.f, 0.f, 0.f};
      if (d < D) {
        v = *reinterpret_cast<const float4*>(xr + d);
        if (mu) { const float4 m = *reinterpret_cast<const float4*>(mu + d); v.x -= m.x; v.y -= m.y; v.z -= m.z; v.w -= m.w; }
      }
      c[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (i * 256 + lane * 4 < D) {     // the same order as rownorm_sumsq<true>
        ss += c[i].x * c[i].x; ss += c[i].y * c[i].y; ss += c[i].z * c[i].z; ss += c[i].w * c[i].w;
      }
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  } else {
    ss = rownorm_sumsq<false>(xr, mu, D, lane);
  }
  const float nrm = sqrtf(ss);
  const float den = mode == 0 ? nrm + eps : (nrm > eps ? nrm : eps);
  if (dens && lane == 0) dens[row] = den;
  float* yr = y ? y + (size_t)row * D : nullptr;
  _Float16* hr = yh + (size_t)row * D;
  float rs = 0.f;     // sum of squares of the fp16 rounding residuals v - fp16(v) (each residual is exact in fp32)
  if constexpr (VEC) {
    typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int d = i * 256 + lane * 4;
      if (d < D) {
        float4 v = c[i];
        v.x /= den; v.y /= den; v.z /= den; v.w /= den;
        const h4_t h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        if (yr) *reinterpret_cast<float4*>(yr + d) = v;
        *reinterpret_cast<h4_t*>(hr + d) = h;
        const float r0 = v.x - (float)h[0], r1 = v.y - (float)h[1], r2 = v.z - (float)h[2], r3 = v.w - (float)h[3];
        rs += r0 * r0; rs += r1 * r1; rs += r2 * r2; rs += r3 * r3;
      }
    }
  } else {
    for (int d = lane; d < D; d += 64) {
      const float v = (xr[d] - (mu ? mu[d] : 0.f)) / den;
      const _Float16 h = (_Float16)v;
      if (yr) yr[d] = v;
      hr[d] = h;
      const float r = v - (float)h;
      rs += r * r;
    }
  }
  if (resid) {
    for (int o = 32; o > 0; o >>= 1) rs += __shfl_xor(rs, o);
    if (lane == 0) resid[row] = sqrtf(rs);
  }
}

// rows that one dwordx4 per lane can walk: D a multiple of 4 and every pointer / row pitch 16-byte aligned
static bool rownorm_vec_ok(const void* x, const void* mu, const void* y, int D, int ldx, int ldy) {
  return D % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ((((uintptr_t)x) | ((uintptr_t)mu) | ((uintptr_t)y)) & 15) == 0;
}

extern "C" int lvc_rownorm_h(const float* x, const float* mu, float* y, unsigned short* yh, float* den, float* resid, int M, int D,
                             int ldx, float eps, int mode, void* stream) {
  LVC_CHECK_ARG(M >= 0 && D > 0, "bad shape");
  if (M == 0) return LVC_OK;
  LVC_CHECK_ARG(x && yh, "null pointer");
  const dim3 grid(lvc_cdiv(M, 4)), block(256);
  const int ldxx = ldx > 0 ? ldx : D;
#define RH_LAUNCH(N, V) hipLaunchKernelGGL((rownorm_h_kernel<N, V>), grid, block, 0, (hipStream_t)stream, x, mu, y, (_Float16*)yh, den, resid, M, D, ldxx, eps, mode)
  // the fp16 rows are written 8 bytes per lane: D % 4 == 0 keeps every row of yh 8-byte aligned
  const bool vec = rownorm_vec_ok(x, mu, y, D, ldxx, D) && (((uintptr_t)yh) & 7) == 0 && D <= 2048;
  if (!vec) RH_LAUNCH(0, false);
  else if (D <= 512) RH_LAUNCH(2, true);
  else if (D <= 1024) RH_LAUNCH(4, true);
  else RH_LAUNCH(8, true);
#undef RH_LAUNCH
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

extern "C" int lvc_rownorm(const float* x, const float* mu, float* y, int M, int D, int ldx, int ldy, float eps,
                           int mode, void* stream) {
  LVC_CHECK_ARG(M >= 0 && D > 0, "bad shape");
  if (M == 0) return LVC_OK;
  LVC_CHECK_ARG(x && y, "null pointer");
  const int ldxx = ldx > 0 ? ldx : D, ldyy = ldy > 0 ? ldy : D;
  if (rownorm_vec_ok(x, mu, y, D, ldxx, ldyy))
    hipLaunchKernelGGL(rownorm_kernel<true>, dim3(lvc_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, mu, y, M, D, ldxx, ldyy, eps, mode);
  else
    hipLaunchKernelGGL(rownorm_kernel<false>, dim3(lvc_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, mu, y, M, D, ldxx, ldyy, eps, mode);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// Backward of y = x / (|x| + eps) row-wise (mode 0 of lvc_rownorm without mu; CosineSimOutputLayers' input
// normalisation, lvc/modeling/roi_heads/fast_rcnn.py:823-825: torch.norm's backward is x/|x|):
//   dx = dy / (n + eps) - x * (dy . x) / ((n + eps)^2 * n),  n = |x|;  rows with n == 0 get dy / eps.
// dx_accum != 0: added to what dx already holds (the bbox_pred branch's gradient of the same x).
__global__ __launch_bounds__(256) void rownorm_backward_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               float* __restrict__ dx, int M, int D, float eps, int accum) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* xr = x + (size_t)row * D;
  const float* gr = dy + (size_t)row * D;
  float ss = 0.f, dot = 0.f;
  for (int d = lane; d < D; d += 64) {
    ss += xr[d] * xr[d];
    dot += gr[d] * xr[d];
  }
  for (int o = 32; o > 0; o >>= 1) { ss += __shfl_xor(ss, o); dot += __shfl_xor(dot, o); }
  const float n = sqrtf(ss), den = n + eps;
  const float a = 1.f / den, b = n > 0.f ? dot / (den * den * n) : 0.f;
  float* o_ = dx + (size_t)row * D;
  for (int d = lane; d < D; d += 64) {
    const float v = gr[d] * a - xr[d] * b;
    o_[d] = accum ? o_[d] + v : v;
  }
}

extern "C" int lvc_rownorm_backward(const float* x, const float* dy, float* dx, int M, int D, float eps, int accumulate,
                                    void* stream) {
  LVC_CHECK_ARG(M >= 0 && D > 0, "bad shape");
  if (M == 0) return LVC_OK;
  LVC_CHECK_ARG(x && dy && dx, "null pointer");
  hipLaunchKernelGGL(rownorm_backward_kernel, dim3(lvc_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, M, D, eps,
                     accumulate);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// Context / pad crops for the label-verification descriptors: reference
// lvc/data/utils.py:485-519 get_padding + get_crops_qe: the box window (already widened / clamped on the host, which
// is integer bookkeeping) is zero-padded to a square and resized to out x out with F.interpolate(mode='nearest'),
// i.e. src = min(floor(dst * in/out), in-1) with the scale held in fp32.
// win[k] = (x1, y1, x2, y2, l_pad, t_pad, side_w, side_h) int32, inclusive window.
__global__ void crop_resize_nearest_kernel(const float* __restrict__ img, int C, int H, int W, const int* __restrict__ win,
                                           int K, int out, float* __restrict__ y) {
  const long long total = (long long)K * C * out * out;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % out);
    long long t = i / out;
    const int oy = (int)(t % out); t /= out;
    const int c = (int)(t % C);
    const int k = (int)(t / C);
    const int* w = win + k * 8;
    const int side_w = w[6], side_h = w[7];
    const float sx = (float)side_w / (float)out, sy = (float)side_h / (float)out;
    int px = (int)floorf((float)ox * sx), py = (int)floorf((float)oy * sy);
    px = px < side_w - 1 ? px : side_w - 1;
    py = py < side_h - 1 ? py : side_h - 1;
    const int xx = px - w[4] + w[0], yy = py - w[5] + w[1];
    float v = 0.f;
    if (xx >= w[0] && xx <= w[2] && yy >= w[1] && yy <= w[3]) v = img[((long long)c * H + yy) * W + xx];
    y[i] = v;
  }
}

extern "C" int lvc_crop_resize_nearest(const float* image_chw, int C, int H, int W, const int* d_windows, int K, int out_size,
                                       float* out, void* stream) {
  LVC_CHECK_ARG(K >= 0 && C > 0 && H > 0 && W > 0 && out_size > 0, "bad shape");
  if (K == 0) return LVC_OK;
  LVC_CHECK_ARG(image_chw && d_windows && out, "null pointer");
  const long long total = (long long)K * C * out_size * out_size;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(crop_resize_nearest_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, image_chw, C, H, W,
                     d_windows, K, out_size, out);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}
