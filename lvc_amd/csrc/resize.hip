// resize.hip -- test-time input pipeline on the device: ResizeShortestEdge's
// `Image.fromarray(img).resize((new_w, new_h), BILINEAR)` (reference detectron2/data/transforms/transform.py:101-109;
// sizes from augmentation_impl.py:214-234) fused with GeneralizedRCNN.preprocess_image (lvc/modeling/meta_arch/rcnn.py:
// 324-333: (x - mean) / std, zero-pad to the batch's padded size), so a uint8 HWC image goes to the trunk's NHWC4 slot
// without a float CHW copy in between.
//
// The resample is Pillow's (third-party, src/libImaging/Resample.c ImagingResampleHorizontal_8bpc / Vertical_8bpc),
// bit for bit: 22-bit fixed-point coefficients (computed on the host in double exactly as precompute_coeffs /
// normalize_coeffs_8bpc do, lvc_amd/data/transforms.py), a horizontal pass into a uint8 intermediate, then a vertical
// pass; each output = clip8((2^21 + sum k * pixel) >> 22) in int32.  Integer work: bit-exact by construction.
// HBM-bound byte streams; lanes walk x so the source windows of neighbouring lanes overlap in cache.
#include "common.h"

#define RS_PREC 22

__device__ __forceinline__ unsigned char clip8(int v) {
  v >>= RS_PREC;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// src [H][W][3] u8 -> dst [H][new_w][3] u8
__global__ __launch_bounds__(256) void resize_h_kernel(const unsigned char* __restrict__ src, int H, int W, int new_w,
                                                       const int* __restrict__ bounds, const int* __restrict__ kk,
                                                       int ksize, unsigned char* __restrict__ dst) {
  const int xo = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (xo >= new_w) return;
  const int xmin = bounds[2 * xo], cnt = bounds[2 * xo + 1];
  const int* k = kk + (size_t)xo * ksize;
  const unsigned char* row = src + ((size_t)y * W + xmin) * 3;
  int s0 = 1 << (RS_PREC - 1), s1 = s0, s2 = s0;
  for (int x = 0; x < cnt; ++x) {
    const int c = k[x];
    s0 += row[3 * x] * c;
    s1 += row[3 * x + 1] * c;
    s2 += row[3 * x + 2] * c;
  }
  unsigned char* o = dst + ((size_t)y * new_w + xo) * 3;
  o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

// src [H][new_w][3] u8 -> optional out_u8 [new_h][new_w][3] and/or out_f [Hp][Wp][4] = (v - mean) / std, zero padded
__global__ __launch_bounds__(256) void resize_v_kernel(const unsigned char* __restrict__ src, int H, int new_w, int new_h,
                                                       const int* __restrict__ bounds, const int* __restrict__ kk,
                                                       int ksize, unsigned char* __restrict__ out_u8,
                                                       float* __restrict__ out_f, int Hp, int Wp, float m0, float m1,
                                                       float m2, float d0, float d1, float d2) {
  const int xo = blockIdx.x * 256 + threadIdx.x, yo = blockIdx.y;
  if (xo >= Wp) return;
  float4 v = {0.f, 0.f, 0.f, 0.f};
  if (yo < new_h && xo < new_w) {
    unsigned char r0, r1, r2;
    if (kk) {
      const int ymin = bounds[2 * yo], cnt = bounds[2 * yo + 1];
      const int* k = kk + (size_t)yo * ksize;
      int s0 = 1 << (RS_PREC - 1), s1 = s0, s2 = s0;
      for (int y = 0; y < cnt; ++y) {
        const unsigned char* p = src + ((size_t)(ymin + y) * new_w + xo) * 3;
        const int c = k[y];
        s0 += p[0] * c; s1 += p[1] * c; s2 += p[2] * c;
      }
      r0 = clip8(s0); r1 = clip8(s1); r2 = clip8(s2);
    } else {   // height unchanged: Pillow skips the vertical pass
      const unsigned char* p = src + ((size_t)yo * new_w + xo) * 3;
      r0 = p[0]; r1 = p[1]; r2 = p[2];
    }
    if (out_u8) {
      unsigned char* o = out_u8 + ((size_t)yo * new_w + xo) * 3;
      o[0] = r0; o[1] = r1; o[2] = r2;
    }
    v.x = ((float)r0 - m0) / d0;
    v.y = ((float)r1 - m1) / d1;
    v.z = ((float)r2 - m2) / d2;
  }
  if (out_f && yo < Hp) *reinterpret_cast<float4*>(out_f + ((size_t)yo * Wp + xo) * 4) = v;
}

// image [H][W][3] uint8 (device).  xb/xk: horizontal bounds [new_w][2] / coefficients [new_w][kxs] (NULL when
// new_w == W), yb/yk likewise for the rows (NULL when new_h == H).  tmp: new_w*H*3 bytes of scratch (unused when
// xk == NULL).  out_u8 [new_h][new_w][3] and out_nhwc4 [Hp][Wp][4] are each optional; mean3/std3 host pointers.
extern "C" int lvc_resize_bilinear_u8(const unsigned char* image, int H, int W, int new_h, int new_w, const int* xb,
                                      const int* xk, int kxs, const int* yb, const int* yk, int kys, unsigned char* tmp,
                                      unsigned char* out_u8, float* out_nhwc4, int Hp, int Wp, const float* mean3,
                                      const float* std3, void* stream) {
  LVC_CHECK_ARG(image && H > 0 && W > 0 && new_h > 0 && new_w > 0, "bad image");
  LVC_CHECK_ARG((xk != nullptr) == (new_w != W) && (yk != nullptr) == (new_h != H), "coefficients must match the size change");
  LVC_CHECK_ARG((!xk || (xb && tmp && kxs > 0)) && (!yk || (yb && kys > 0)), "missing bounds / scratch");
  LVC_CHECK_ARG(out_u8 || out_nhwc4, "no output requested");
  LVC_CHECK_ARG(!out_nhwc4 || (Hp >= new_h && Wp >= new_w && mean3 && std3), "bad padded size / normaliser");
  hipStream_t st = (hipStream_t)stream;
  const unsigned char* mid = image;
  if (xk) {
    hipLaunchKernelGGL(resize_h_kernel, dim3(lvc_cdiv(new_w, 256), H), dim3(256), 0, st, image, H, W, new_w, xb, xk, kxs, tmp);
    mid = tmp;
  }
  const int gw = out_nhwc4 ? Wp : new_w, gh = out_nhwc4 ? Hp : new_h;
  const float one[3] = {1.f, 1.f, 1.f}, zero[3] = {0.f, 0.f, 0.f};
  const float* m = mean3 ? mean3 : zero;
  const float* d = std3 ? std3 : one;
  hipLaunchKernelGGL(resize_v_kernel, dim3(lvc_cdiv(gw, 256), gh), dim3(256), 0, st, mid, H, new_w, new_h, yb, yk, kys,
                     out_u8, out_nhwc4, out_nhwc4 ? Hp : new_h, out_nhwc4 ? Wp : new_w, m[0], m[1], m[2], d[0], d[1], d[2]);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}
