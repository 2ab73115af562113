// roi_align.hip -- ROIAlign forward and backward for gfx950 (HBM/L2-bound gather; no matrix work).
//
// Replaces reference detectron2/layers/csrc/ROIAlign/ROIAlign_cuda.cu:65-139 (RoIAlignForward)
// and its CPU twin ROIAlign_cpu.cpp:20-218, reached through detectron2/layers/roi_align.py:63-117
// from ROIPooler.forward (detectron2/modeling/poolers.py:191-246).  Same arithmetic, same operation
// order per output element (this file is compiled with -ffp-contract=off so that no FMA contraction
// changes a rounding relative to the x86-64 reference build):
//   start = x*scale - 0.5 (aligned) ; bin = roi_size/pooled ; grid g = sampling_ratio>0 ? sr : ceil(roi_size/pooled)
//   sample (iy,ix) at start + p*bin + (i+.5)*bin/g ; outside [-1,H]x[-1,W] contributes 0 ; bilinear taps
//   with the <=0 clamp and the top-edge clamp ; average over max(g_h*g_w,1).
// Unlike the reference CUDA host code (ROIAlign_cuda.cu:364) nothing here synchronises the device.
//
// Mapping (NHWC features): one workgroup per (RoI, 256-channel slice); lane = channel, so every
// bilinear tap is one fully coalesced 1 KiB read per wave (256 B per 64 channels) and every output
// bin one coalesced store; tap coordinates are wave-uniform.  The same kernel serves the reference's
// NCHW op signature through explicit strides (lane = channel is then strided; that entry point is a
// drop-in convenience, the engine uses NHWC).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

#define LVC_MAX_LEVELS 8

struct RoiAlignArgs {
  const float* feat[LVC_MAX_LEVELS];
  int H[LVC_MAX_LEVELS], W[LVC_MAX_LEVELS];
  float scale[LVC_MAX_LEVELS];
  long long sb[LVC_MAX_LEVELS];  // batch stride (elements) per level
  long long sc, sh_mul_w;        // channel stride; (unused for NHWC) -- see below
  int nhwc;                      // 1: feature is [B,H,W,C]; 0: [B,C,H,W]
  int C;
  const float* rois;             // [K,5]
  const int* levels;             // [K] or nullptr (level 0)
  const int* num_valid;          // device int: rows >= *num_valid are zero-filled (nullptr = all valid)
  int K, ph, pw, sampling_ratio, aligned;
  float* out;
  long long so_k, so_c, so_h, so_w;  // output strides (elements)
  int* status;                   // device status word (bit 0: negative RoI size with aligned=true)
  const int* order;              // wave form: workgroup i pools RoI order[i] (lvc_roi_work_order; nullptr = i)
};

__global__ __launch_bounds__(256) void roi_align_fwd_kernel(RoiAlignArgs p) {
  const int k = blockIdx.x;
  const int c = blockIdx.y * 256 + threadIdx.x;
  const bool c_ok = c < p.C;
  float* outk = p.out + (long long)k * p.so_k + (long long)c * p.so_c;
  if (p.num_valid && k >= *p.num_valid) {
    if (c_ok)
      for (int a = 0; a < p.ph; ++a)
        for (int b = 0; b < p.pw; ++b) outk[a * p.so_h + b * p.so_w] = 0.f;
    return;
  }
  const float* r = p.rois + (long long)k * 5;
  const int lvl = p.levels ? p.levels[k] : 0;
  const int H = p.H[lvl], W = p.W[lvl];
  const float spatial_scale = p.scale[lvl];
  const int b = (int)r[0];
  const float offset = p.aligned ? 0.5f : 0.0f;
  const float roi_start_w = r[1] * spatial_scale - offset;
  const float roi_start_h = r[2] * spatial_scale - offset;
  const float roi_end_w = r[3] * spatial_scale - offset;
  const float roi_end_h = r[4] * spatial_scale - offset;
  float roi_width = roi_end_w - roi_start_w;
  float roi_height = roi_end_h - roi_start_h;
  if (p.aligned) {
    if (!(roi_width >= 0 && roi_height >= 0)) {
      if (p.status && threadIdx.x == 0 && blockIdx.y == 0) atomicOr(p.status, 1);
    }
  } else {
    roi_width = roi_width > 1.f ? roi_width : 1.f;
    roi_height = roi_height > 1.f ? roi_height : 1.f;
  }
  const float bin_h = roi_height / (float)p.ph;
  const float bin_w = roi_width / (float)p.pw;
  const int gh = p.sampling_ratio > 0 ? p.sampling_ratio : (int)ceilf(roi_height / (float)p.ph);
  const int gw = p.sampling_ratio > 0 ? p.sampling_ratio : (int)ceilf(roi_width / (float)p.pw);
  const int cnt = gh * gw > 1 ? gh * gw : 1;
  const float count = (float)cnt;

  long long s_pix, s_c;
  if (p.nhwc) { s_pix = p.C; s_c = 1; } else { s_pix = 1; s_c = (long long)H * W; }
  const float* in = p.feat[lvl] + (long long)b * p.sb[lvl] + (long long)(c_ok ? c : 0) * s_c;

  for (int ph = 0; ph < p.ph; ++ph) {
    for (int pw = 0; pw < p.pw; ++pw) {
      float acc = 0.f;
      for (int iy = 0; iy < gh; ++iy) {
        const float yy = roi_start_h + ph * bin_h + (float)(iy + .5f) * bin_h / (float)gh;
        for (int ix = 0; ix < gw; ++ix) {
          const float xx = roi_start_w + pw * bin_w + (float)(ix + .5f) * bin_w / (float)gw;
          float x = xx, y = yy;
          if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) continue;
          if (y <= 0) y = 0;
          if (x <= 0) x = 0;
          int y_low = (int)y, x_low = (int)x, y_high, x_high;
          if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
          if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
          const float ly = y - y_low, lx = x - x_low;
          const float hy = 1.f - ly, hx = 1.f - lx;
          const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
          const float v1 = in[(long long)(y_low * W + x_low) * s_pix];
          const float v2 = in[(long long)(y_low * W + x_high) * s_pix];
          const float v3 = in[(long long)(y_high * W + x_low) * s_pix];
          const float v4 = in[(long long)(y_high * W + x_high) * s_pix];
          acc += w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
        }
      }
      if (c_ok) outk[ph * p.so_h + pw * p.so_w] = acc / count;
    }
  }
}

// NHWC fast path: one WAVE per (RoI, output bin) and 256-channel slice, lane = 4 consecutive channels (dwordx4
// taps: 16 B per lane is the coalescing sweet spot and cuts the number of load instructions by 4); sample
// coordinates and bilinear weights are wave-uniform.  Splitting a RoI over its 49 bins bounds the tail: a
// full-image RoI on p5 has 24 samples per bin (4 700 dependent tap groups if one wave walked all 49 bins).
// Same per-element operation order as the scalar kernel above.
__global__ __launch_bounds__(64) void roi_align_fwd_nhwc4_kernel(RoiAlignArgs p) {
  const int lane = threadIdx.x;
  const int nb = p.ph * p.pw;
  const int k = blockIdx.x / nb;
  const int bin = blockIdx.x - k * nb;
  const int ph = bin / p.pw, pw = bin - ph * p.pw;
  const int c = blockIdx.y * 256 + lane * 4;
  const bool c_ok = c < p.C;
  float* outk = p.out + (long long)k * p.so_k + c;
  const float* r = p.rois + (long long)k * 5;
  const int lvl = p.levels ? p.levels[k] : 0;
  const int H = p.H[lvl], W = p.W[lvl];
  const float spatial_scale = p.scale[lvl];
  const int b = (int)r[0];
  const float offset = p.aligned ? 0.5f : 0.0f;
  const float roi_start_w = r[1] * spatial_scale - offset;
  const float roi_start_h = r[2] * spatial_scale - offset;
  const float roi_end_w = r[3] * spatial_scale - offset;
  const float roi_end_h = r[4] * spatial_scale - offset;
  float roi_width = roi_end_w - roi_start_w;
  float roi_height = roi_end_h - roi_start_h;
  if (p.aligned) {
    if (!(roi_width >= 0 && roi_height >= 0)) {
      if (p.status && lane == 0 && blockIdx.y == 0) atomicOr(p.status, 1);
    }
  } else {
    roi_width = roi_width > 1.f ? roi_width : 1.f;
    roi_height = roi_height > 1.f ? roi_height : 1.f;
  }
  const float bin_h = roi_height / (float)p.ph;
  const float bin_w = roi_width / (float)p.pw;
  const int gh = p.sampling_ratio > 0 ? p.sampling_ratio : (int)ceilf(roi_height / (float)p.ph);
  const int gw = p.sampling_ratio > 0 ? p.sampling_ratio : (int)ceilf(roi_width / (float)p.pw);
  const int cnt = gh * gw > 1 ? gh * gw : 1;
  const float count = (float)cnt;
  const float* in = p.feat[lvl] + (long long)b * p.sb[lvl] + (c_ok ? c : 0);
  const long long C = p.C;
  {
    {
      float4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int iy = 0; iy < gh; ++iy) {
        const float yy = roi_start_h + ph * bin_h + (float)(iy + .5f) * bin_h / (float)gh;
        for (int ix = 0; ix < gw; ++ix) {
          const float xx = roi_start_w + pw * bin_w + (float)(ix + .5f) * bin_w / (float)gw;
          float x = xx, y = yy;
          if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) continue;
          if (y <= 0) y = 0;
          if (x <= 0) x = 0;
          int y_low = (int)y, x_low = (int)x, y_high, x_high;
          if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
          if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
          const float ly = y - y_low, lx = x - x_low;
          const float hy = 1.f - ly, hx = 1.f - lx;
          const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
          const float4 v1 = *reinterpret_cast<const float4*>(in + (long long)(y_low * W + x_low) * C);
          const float4 v2 = *reinterpret_cast<const float4*>(in + (long long)(y_low * W + x_high) * C);
          const float4 v3 = *reinterpret_cast<const float4*>(in + (long long)(y_high * W + x_low) * C);
          const float4 v4 = *reinterpret_cast<const float4*>(in + (long long)(y_high * W + x_high) * C);
          acc.x += w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x;
          acc.y += w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y;
          acc.z += w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z;
          acc.w += w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w;
        }
      }
      if (c_ok) {
        float4 o = {acc.x / count, acc.y / count, acc.z / count, acc.w / count};
        *reinterpret_cast<float4*>(outk + ph * p.so_h + pw * p.so_w) = o;
      }
    }
  }
}

// NHWC engine path: one WAVE per (RoI, output row ph) and 256-channel slice -- no LDS, no barrier; a workgroup is the ph waves of a
// RoI.  A bin's value is sum over samples (iy, ix) of sum over the four taps of weight x pixel; a sample is in range when its row
// AND its column are, and a tap's weight is (hy or ly) x (hx or lx): the double sum factors into
//     sum_r sum_c Wy[r] Wx[c] pixel[r][c],   Wy[r] = sum over in-range iy of (hy if y_low == r) + (ly if y_high == r),  Wx alike.
// The wave computes Wy once (one register, lane = window row) and Wx of each of its seven bins (seven registers, lane = window
// column), then reads every pixel of the row's window -- rows [floor(y_first), floor(y_last)+1] x columns [floor(x_first),
// floor(x_last)+1] -- ONCE (1 KiB per pixel and wave, coalesced, ROI_WAVE_MLP pixels in flight) and adds it into the bins whose
// Wx is not zero in that column (weights are read with v_readlane at wave-uniform indices; the test is scalar).  The sum is formed
// in another order than the reference's (rows outer, columns inner; weights pre-summed): the same value to fp32 rounding (the
// tests' 1e-6), not the same bits -- the NCHW drop-in kernel above keeps the reference's order.
// History (profiles/README.md, round 4): the workgroup-per-(RoI, row) form with the window staged in LDS and per-sample taps ran
// 0.60 - 0.66 ms on the bench batch's 8000 proposals (47 % of the windows fitted its 24-pixel buffer, the rest read 10.6 GB of
// taps from L1 / L2); the separable sums in that form cut the L2 reads to 2.5 GB and did not change its time (four items in flight
// per CU, each a chain of dependent latencies); a wave per item with the bins walked one after the other: 0.95 - 0.99 ms (every
// column re-read per bin, L1 far smaller than 28 waves' windows); this form 0.53 - 0.57 ms.
// Items beyond the registers' reach (more than 64 window rows or columns; empty grids) take the per-sample loop.
#define ROI_UNI(x) __builtin_amdgcn_readfirstlane(x)
#ifndef ROI_RB4
#define ROI_RB4 2      // rows per batch of loads, windows of <= 4 columns
#endif
#ifndef ROI_RB8
#define ROI_RB8 1      // rows per batch of loads, chunks of 8 columns
#endif
#ifndef ROI_WAVE_MLP
#define ROI_WAVE_MLP 8
#endif
template <int PW>
__global__ __launch_bounds__(512) void roi_align_fwd_nhwc_wave_kernel(RoiAlignArgs p) {
  const int lane = threadIdx.x & 63;
  const int ph = ROI_UNI((int)(threadIdx.x >> 6));
  const int k = p.order ? p.order[blockIdx.x] : (int)blockIdx.x;
  if (ph >= p.ph) return;
  const int cbase = blockIdx.y * 256;
  const float* r = p.rois + (long long)k * 5;
  const int lvl = p.levels ? p.levels[k] : 0;
  const int H = p.H[lvl], W = p.W[lvl];
  const float spatial_scale = p.scale[lvl];
  const int b = (int)r[0];
  const float offset = p.aligned ? 0.5f : 0.0f;
  const float roi_start_w = r[1] * spatial_scale - offset;
  const float roi_start_h = r[2] * spatial_scale - offset;
  const float roi_end_w = r[3] * spatial_scale - offset;
  const float roi_end_h = r[4] * spatial_scale - offset;
  float roi_width = roi_end_w - roi_start_w;
  float roi_height = roi_end_h - roi_start_h;
  if (p.aligned) {
    if (!(roi_width >= 0 && roi_height >= 0)) {
      if (p.status && lane == 0 && blockIdx.y == 0 && ph == 0) atomicOr(p.status, 1);
    }
  } else {
    roi_width = roi_width > 1.f ? roi_width : 1.f;
    roi_height = roi_height > 1.f ? roi_height : 1.f;
  }
  const float bin_h = roi_height / (float)p.ph;
  const float bin_w = roi_width / (float)p.pw;
  const int gh = ROI_UNI(p.sampling_ratio > 0 ? p.sampling_ratio : (int)ceilf(roi_height / (float)p.ph));
  const int gw = ROI_UNI(p.sampling_ratio > 0 ? p.sampling_ratio : (int)ceilf(roi_width / (float)p.pw));
  const int cnt = gh * gw > 1 ? gh * gw : 1;
  const float count = (float)cnt;
  const long long C = p.C;
  const bool c_ok = cbase + lane * 4 < p.C;
  const float* in = p.feat[lvl] + (long long)b * p.sb[lvl] + (c_ok ? cbase + lane * 4 : 0);
  float* outp = p.out + (long long)k * p.so_k + ph * p.so_h + cbase + lane * 4;

  const float yf = roi_start_h + ph * bin_h, yl = roi_start_h + (ph + 1) * bin_h;
  const int y0 = ROI_UNI(min((int)floorf(fmaxf(yf, 0.f)), H - 1)), y1 = ROI_UNI(min((int)floorf(fmaxf(yl, 0.f)) + 1, H - 1));
  const int x0 = ROI_UNI(min((int)floorf(fmaxf(roi_start_w, 0.f)), W - 1));
  const int x1 = ROI_UNI(min((int)floorf(fmaxf(roi_start_w + roi_width, 0.f)) + 1, W - 1));
  const int nrows = y1 - y0 + 1, ncols = x1 - x0 + 1;
  const bool sep = gh > 0 && gw > 0 && nrows >= 1 && nrows <= 64 && ncols >= 1 && ncols <= 64;
  if (!sep) {
    // the reference's loop, sample by sample (every bin of the row)
    for (int pw = 0; pw < p.pw; ++pw) {
      float4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int iy = 0; iy < gh; ++iy) {
        const float yy = roi_start_h + ph * bin_h + (float)(iy + .5f) * bin_h / (float)gh;
        for (int ix = 0; ix < gw; ++ix) {
          const float xx = roi_start_w + pw * bin_w + (float)(ix + .5f) * bin_w / (float)gw;
          float x = xx, y = yy;
          if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) continue;
          if (y <= 0) y = 0;
          if (x <= 0) x = 0;
          int y_low = (int)y, x_low = (int)x, y_high, x_high;
          if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
          if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
          const float ly = y - y_low, lx = x - x_low;
          const float hy = 1.f - ly, hx = 1.f - lx;
          const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
          const float4 v1 = *reinterpret_cast<const float4*>(in + (long long)(y_low * W + x_low) * C);
          const float4 v2 = *reinterpret_cast<const float4*>(in + (long long)(y_low * W + x_high) * C);
          const float4 v3 = *reinterpret_cast<const float4*>(in + (long long)(y_high * W + x_low) * C);
          const float4 v4 = *reinterpret_cast<const float4*>(in + (long long)(y_high * W + x_high) * C);
          acc.x += w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x;
          acc.y += w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y;
          acc.z += w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z;
          acc.w += w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w;
        }
      }
      if (c_ok) *reinterpret_cast<float4*>(outp + pw * p.so_w) = float4{acc.x / count, acc.y / count, acc.z / count, acc.w / count};
    }
    return;
  }
  // Wy: lane = window row, the samples in the reference's order
  float wy = 0.f;
  {
    const int ymine = y0 + lane;
    for (int iy = 0; iy < gh; ++iy) {
      const float yy = roi_start_h + ph * bin_h + (float)(iy + .5f) * bin_h / (float)gh;
      if (yy < -1.0f || yy > (float)H) continue;
      float y = yy <= 0 ? 0.f : yy;
      int y_low = (int)y, y_high;
      if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
      const float ly = y - y_low, hy = 1.f - ly;
      wy += (y_low == ymine ? hy : 0.f) + (y_high == ymine ? ly : 0.f);
    }
  }
  // Wx of every bin of the row: lane = window column
  float wx[PW];
#pragma unroll
  for (int pw = 0; pw < PW; ++pw) {
    const int xmine = x0 + lane;
    float a = 0.f;
    for (int ix = 0; ix < gw; ++ix) {
      const float xx = roi_start_w + pw * bin_w + (float)(ix + .5f) * bin_w / (float)gw;
      if (xx < -1.0f || xx > (float)W) continue;
      float x = xx <= 0 ? 0.f : xx;
      int x_low = (int)x, x_high;
      if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
      const float lx = x - x_low, hx = 1.f - lx;
      a += (x_low == xmine ? hx : 0.f) + (x_high == xmine ? lx : 0.f);
    }
    wx[pw] = a;
  }
  // Every pixel of the window once.  Wy does not depend on the bin, so the rows are summed FIRST, per column:
  //     T[c] = sum_r Wy[r] pixel[r][c]                      (two packed fmas per pixel, whatever the number of bins that use it)
  //     bin[pw] = sum_c Wx[pw][c] T[c]                      (per column and bin with a weight there, once per column chunk)
  // -- 0.42 -> 0.39 ms on the bench proposals against adding every pixel into each of its (two to seven) bins.  Columns go in chunks of CW (T lives in
  // registers), RB rows of a chunk per batch of loads: 4 x 2 for the narrow windows (<= 4 columns), 8 x 1 otherwise.
  unsigned amask = 0;
#pragma unroll
  for (int pw = 0; pw < PW; ++pw) amask |= wx[pw] != 0.f ? 1u << pw : 0u;
  float4 acc[PW];
#pragma unroll
  for (int pw = 0; pw < PW; ++pw) acc[pw] = float4{0.f, 0.f, 0.f, 0.f};
  const float* base = in + ((long long)y0 * W + x0) * C;
  const int Ci = (int)C, rowstep = W * Ci;      // element offsets inside one image of a level fit 32 bits (checked at launch)
  auto chunk = [&](auto cw_tag, auto rb_tag, int c0) {
    constexpr int CW = decltype(cw_tag)::value, RB = decltype(rb_tag)::value;
    const int ncw = min(CW, ncols - c0);        // columns of this chunk
    float4 T[CW];
#pragma unroll
    for (int u = 0; u < CW; ++u) T[u] = float4{0.f, 0.f, 0.f, 0.f};
    for (int r0 = 0; r0 < nrows; r0 += RB) {
      float4 v[RB][CW];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int u = 0; u < CW; ++u) {
          const bool live = u < ncw && r0 + rb < nrows;      // wave-uniform
          v[rb][u] = *reinterpret_cast<const float4*>(base + (live ? (r0 + rb) * rowstep + (c0 + u) * Ci : 0));
        }
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        if (r0 + rb >= nrows) continue;
        const float wyr = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(wy), r0 + rb));
#pragma unroll
        for (int u = 0; u < CW; ++u) {
          if (u >= ncw) continue;
          T[u].x = __builtin_fmaf(wyr, v[rb][u].x, T[u].x); T[u].y = __builtin_fmaf(wyr, v[rb][u].y, T[u].y);
          T[u].z = __builtin_fmaf(wyr, v[rb][u].z, T[u].z); T[u].w = __builtin_fmaf(wyr, v[rb][u].w, T[u].w);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < CW; ++u) {
      if (u >= ncw) continue;
      const unsigned am = __builtin_amdgcn_readlane(amask, c0 + u);      // the bins with a weight in this column
#pragma unroll
      for (int pw = 0; pw < PW; ++pw) {
        if (am & (1u << pw)) {      // wave-uniform
          const float w = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(wx[pw]), c0 + u));
          acc[pw].x = __builtin_fmaf(w, T[u].x, acc[pw].x); acc[pw].y = __builtin_fmaf(w, T[u].y, acc[pw].y);
          acc[pw].z = __builtin_fmaf(w, T[u].z, acc[pw].z); acc[pw].w = __builtin_fmaf(w, T[u].w, acc[pw].w);
        }
      }
    }
  };
  if (ncols <= 4) {
    chunk(std::integral_constant<int, 4>{}, std::integral_constant<int, ROI_RB4>{}, 0);
  } else {
    for (int c0 = 0; c0 < ncols; c0 += 8) chunk(std::integral_constant<int, 8>{}, std::integral_constant<int, ROI_RB8>{}, c0);
  }
  if (c_ok) {
    const float inv = 1.f / count;      // one division per item instead of 28 (the bins' sums are not the reference's bit for bit anyway)
#pragma unroll
    for (int pw = 0; pw < PW; ++pw)
      *reinterpret_cast<float4*>(outp + pw * p.so_w) = float4{acc[pw].x * inv, acc[pw].y * inv, acc[pw].z * inv, acc[pw].w * inv};
  }
}

static int launch(RoiAlignArgs& a, void* stream) {
  if (a.K == 0) return LVC_OK;
  bool small = true;     // the staged kernel keeps tap offsets inside one image of a level as 32-bit element counts
  for (int l = 0; l < LVC_MAX_LEVELS; ++l)
    if (a.feat[l] && (long long)a.H[l] * a.W[l] * a.C >= (1ll << 31)) small = false;
  if (a.nhwc && (a.C & 3) == 0 && a.ph <= 8 && a.pw == 7 && a.num_valid == nullptr && a.so_c == 1 && small) {
    hipLaunchKernelGGL(roi_align_fwd_nhwc_wave_kernel<7>, dim3(a.K, lvc_cdiv(a.C, 256)), dim3(64 * a.ph), 0, (hipStream_t)stream, a);
    LVC_CHECK_LAUNCH();
    return LVC_OK;
  }
  if (a.nhwc && (a.C & 3) == 0 && a.num_valid == nullptr && a.so_c == 1) {
    dim3 grid4(a.K * a.ph * a.pw, lvc_cdiv(a.C, 256)), block4(64);
    hipLaunchKernelGGL(roi_align_fwd_nhwc4_kernel, grid4, block4, 0, (hipStream_t)stream, a);
    LVC_CHECK_LAUNCH();
    return LVC_OK;
  }
  dim3 grid(a.K, lvc_cdiv(a.C, 256)), block(256);
  hipLaunchKernelGGL(roi_align_fwd_kernel, grid, block, 0, (hipStream_t)stream, a);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// ROIAlign backward.  Replaces ROIAlign_cuda.cu:142-306 (bilinear_interpolate_gradient + RoIAlignBackwardFeature) and
// its CPU twin ROIAlign_cpu.cpp:219-406: every pooled element scatters grad * w / count into the four neighbours of
// each of its gh x gw sampling points (same sample positions, clamps and weights as the forward).  Like the reference
// CUDA kernel the scatter uses fp32 atomic adds, so the summation order -- and the last bits -- are not deterministic;
// the reference CPU kernel (single thread, fixed order) is what oracle.c restates bit-exactly, and tests compare the
// two within 1e-6 of the gradient scale.  Same mapping as the scalar forward: one workgroup per (RoI, 256-channel
// slice), lane = channel (coalesced atomics for NHWC gradients; strided for the NCHW drop-in entry point).
// RoiAlignArgs is reused with feat[] = grad_input (written) and out = grad_output (read).
__global__ __launch_bounds__(256) void roi_align_bwd_kernel(RoiAlignArgs p) {
  const int k = blockIdx.x;
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (p.num_valid && k >= *p.num_valid) return;
  const bool c_ok = c < p.C;
  const float* gk = p.out + (long long)k * p.so_k + (long long)(c_ok ? c : 0) * p.so_c;
  const float* r = p.rois + (long long)k * 5;
  const int lvl = p.levels ? p.levels[k] : 0;
  const int H = p.H[lvl], W = p.W[lvl];
  const float spatial_scale = p.scale[lvl];
  const int b = (int)r[0];
  const float offset = p.aligned ? 0.5f : 0.0f;
  const float roi_start_w = r[1] * spatial_scale - offset;
  const float roi_start_h = r[2] * spatial_scale - offset;
  const float roi_end_w = r[3] * spatial_scale - offset;
  const float roi_end_h = r[4] * spatial_scale - offset;
  float roi_width = roi_end_w - roi_start_w;
  float roi_height = roi_end_h - roi_start_h;
  if (p.aligned) {
    if (!(roi_width >= 0 && roi_height >= 0)) {
      if (p.status && threadIdx.x == 0 && blockIdx.y == 0) atomicOr(p.status, 1);
    }
  } else {
    roi_width = roi_width > 1.f ? roi_width : 1.f;
    roi_height = roi_height > 1.f ? roi_height : 1.f;
  }
  const float bin_h = roi_height / (float)p.ph;
  const float bin_w = roi_width / (float)p.pw;
  const int gh = p.sampling_ratio > 0 ? p.sampling_ratio : (int)ceilf(roi_height / (float)p.ph);
  const int gw = p.sampling_ratio > 0 ? p.sampling_ratio : (int)ceilf(roi_width / (float)p.pw);
  const float count = (float)(gh * gw);

  long long s_pix, s_c;
  if (p.nhwc) { s_pix = p.C; s_c = 1; } else { s_pix = 1; s_c = (long long)H * W; }
  float* gin = const_cast<float*>(p.feat[lvl]) + (long long)b * p.sb[lvl] + (long long)(c_ok ? c : 0) * s_c;

  for (int ph = 0; ph < p.ph; ++ph) {
    for (int pw = 0; pw < p.pw; ++pw) {
      const float g = c_ok ? gk[ph * p.so_h + pw * p.so_w] : 0.f;
      for (int iy = 0; iy < gh; ++iy) {
        const float yy = roi_start_h + ph * bin_h + (float)(iy + .5f) * bin_h / (float)gh;
        for (int ix = 0; ix < gw; ++ix) {
          const float xx = roi_start_w + pw * bin_w + (float)(ix + .5f) * bin_w / (float)gw;
          float x = xx, y = yy;
          if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) continue;
          if (y <= 0) y = 0;
          if (x <= 0) x = 0;
          int y_low = (int)y, x_low = (int)x, y_high, x_high;
          if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
          if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
          const float ly = y - y_low, lx = x - x_low;
          const float hy = 1.f - ly, hx = 1.f - lx;
          const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
          if (c_ok) {
            unsafeAtomicAdd(gin + (long long)(y_low * W + x_low) * s_pix, g * w1 / count);
            unsafeAtomicAdd(gin + (long long)(y_low * W + x_high) * s_pix, g * w2 / count);
            unsafeAtomicAdd(gin + (long long)(y_high * W + x_low) * s_pix, g * w3 / count);
            unsafeAtomicAdd(gin + (long long)(y_high * W + x_high) * s_pix, g * w4 / count);
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// ROIAlign backward, separable form for the engine path (NHWC, 7x7 bins): the bilinear weight of sample (iy, ix) onto
// pixel (y, x) is a product wy(iy, y) * wx(ix, x), and a sample is dropped when EITHER coordinate is out of range, so
//     d feat[y][x][c] = 1/count * sum_ph sum_pw  G[ph][pw][c] * Wy[ph][y] * Wx[pw][x],
//     Wy[ph][y] = sum over the bin's valid sample rows of their weight onto row y (likewise Wx).
// One workgroup per (RoI, 256-channel slice), lane = channel: the two weight tables (7 x footprint rows / columns) are
// built once in LDS, the RoI's 49 gradient vectors live in registers, and every pixel of the RoI's footprint receives ONE
// atomic add per channel -- the sample-by-sample scatter of roi_align_bwd_kernel issues 4 * gh * gw atomics per bin,
// ~3.3x as many (a 14 x 14 px RoI: 784 vs 225), and the L2 atomic rate is what bounds this kernel (6.6 ms of the 87 ms
// batch-8 training step).  Same sample positions, clamps and weights as the reference (ROIAlign_cpu.cpp:219-406); the
// products are associated differently (weights first), which moves the last bits as the atomics' order already does.
// RoIs whose footprint or sample count exceeds the tables take the scatter loop below.
#define RB_MAXF 192    // footprint rows / columns held in the tables
#define RB_MAXS 448    // samples per axis (7 bins x 64)
template <int PH, int PW>
__global__ __launch_bounds__(256) void roi_align_bwd_rows_kernel(RoiAlignArgs p) {
  __shared__ float s_w[2][PH][RB_MAXF];      // [axis][bin][footprint index]
  __shared__ int s_lo[2][RB_MAXS];           // per sample: low pixel index (-1: dropped)
  __shared__ float s_l[2][RB_MAXS];          // per sample: weight of the HIGH pixel (low gets 1 - l)
  __shared__ int s_hi[2][RB_MAXS];
  __shared__ int s_min[2], s_max[2];
  const int k = blockIdx.x;
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (p.num_valid && k >= *p.num_valid) return;
  const bool c_ok = c < p.C;
  const float* gk = p.out + (long long)k * p.so_k + (long long)(c_ok ? c : 0) * p.so_c;
  const float* r = p.rois + (long long)k * 5;
  const int lvl = p.levels ? p.levels[k] : 0;
  const int H = p.H[lvl], W = p.W[lvl];
  const float spatial_scale = p.scale[lvl];
  const int b = (int)r[0];
  const float offset = p.aligned ? 0.5f : 0.0f;
  const float roi_start_w = r[1] * spatial_scale - offset;
  const float roi_start_h = r[2] * spatial_scale - offset;
  const float roi_end_w = r[3] * spatial_scale - offset;
  const float roi_end_h = r[4] * spatial_scale - offset;
  float roi_width = roi_end_w - roi_start_w;
  float roi_height = roi_end_h - roi_start_h;
  if (p.aligned) {
    if (!(roi_width >= 0 && roi_height >= 0)) {
      if (p.status && threadIdx.x == 0 && blockIdx.y == 0) atomicOr(p.status, 1);
    }
  } else {
    roi_width = roi_width > 1.f ? roi_width : 1.f;
    roi_height = roi_height > 1.f ? roi_height : 1.f;
  }
  const float bin_h = roi_height / (float)PH;
  const float bin_w = roi_width / (float)PW;
  const int gh = p.sampling_ratio > 0 ? p.sampling_ratio : (int)ceilf(roi_height / (float)PH);
  const int gw = p.sampling_ratio > 0 ? p.sampling_ratio : (int)ceilf(roi_width / (float)PW);
  const float count = (float)(gh * gw);
  float* gin = const_cast<float*>(p.feat[lvl]) + (long long)b * p.sb[lvl] + (long long)(c_ok ? c : 0);

  bool tables = gh >= 1 && gw >= 1 && PH * gh <= RB_MAXS && PW * gw <= RB_MAXS;
  if (tables) {
    if (threadIdx.x < 2) { s_min[threadIdx.x] = 0x7fffffff; s_max[threadIdx.x] = -1; }
    __syncthreads();
    // ---- per-sample records of both axes (thread = sample), footprint extent by LDS atomics
    for (int axis = 0; axis < 2; ++axis) {
      const int g = axis ? gw : gh, n = (axis ? PW : PH) * g, L = axis ? W : H;
      const float start = axis ? roi_start_w : roi_start_h, bin = axis ? bin_w : bin_h;
      for (int t = threadIdx.x; t < n; t += 256) {
        const int pb = t / g, i = t - pb * g;
        float v = start + pb * bin + (float)(i + .5f) * bin / (float)g;
        int lo = -1, hi = -1;
        float l = 0.f;
        if (!(v < -1.0f || v > (float)L)) {
          if (v <= 0) v = 0;
          lo = (int)v;
          if (lo >= L - 1) { hi = lo = L - 1; v = (float)lo; } else hi = lo + 1;
          l = v - lo;
          atomicMin(&s_min[axis], lo);
          atomicMax(&s_max[axis], hi);
        }
        s_lo[axis][t] = lo; s_hi[axis][t] = hi; s_l[axis][t] = l;
      }
    }
    __syncthreads();
    tables = s_max[0] - s_min[0] < RB_MAXF && s_max[1] - s_min[1] < RB_MAXF;   // uniform over the workgroup
  }
  if (tables && s_max[0] >= 0 && s_max[1] >= 0) {
    const int y0 = s_min[0], x0 = s_min[1];
    const int FH = s_max[0] - y0 + 1, FW = s_max[1] - x0 + 1;
    for (int i = threadIdx.x; i < 2 * PH * RB_MAXF; i += 256) (&s_w[0][0][0])[i] = 0.f;
    __syncthreads();
    // ---- weight tables: one thread per (axis, bin) walks that bin's samples in order (deterministic sums)
    if (threadIdx.x < 2 * PH) {
      const int axis = threadIdx.x / PH, pb = threadIdx.x % PH;
      const int g = axis ? gw : gh, base = axis ? x0 : y0;
      for (int i = 0; i < g; ++i) {
        const int t = pb * g + i;
        const int lo = s_lo[axis][t];
        if (lo < 0) continue;
        const float l = s_l[axis][t];
        s_w[axis][pb][lo - base] += 1.f - l;
        s_w[axis][pb][s_hi[axis][t] - base] += l;
      }
    }
    __syncthreads();
    if (c_ok) {
      float g[PH][PW];
#pragma unroll
      for (int a = 0; a < PH; ++a)
#pragma unroll
        for (int bb = 0; bb < PW; ++bb) g[a][bb] = gk[a * p.so_h + bb * p.so_w] / count;
      for (int fy = 0; fy < FH; ++fy) {
        float row[PW];
#pragma unroll
        for (int bb = 0; bb < PW; ++bb) row[bb] = 0.f;
        bool any = false;
#pragma unroll
        for (int a = 0; a < PH; ++a) {
          const float wy = s_w[0][a][fy];
          if (wy != 0.f) {
            any = true;
#pragma unroll
            for (int bb = 0; bb < PW; ++bb) row[bb] += g[a][bb] * wy;
          }
        }
        if (!any) continue;
        float* grow = gin + ((long long)(y0 + fy) * W + x0) * p.C;
        for (int fx = 0; fx < FW; ++fx) {
          float v = 0.f;
          bool hit = false;
#pragma unroll
          for (int bb = 0; bb < PW; ++bb) {
            const float wx = s_w[1][bb][fx];
            if (wx != 0.f) { hit = true; v += row[bb] * wx; }
          }
          if (hit) unsafeAtomicAdd(grow + (long long)fx * p.C, v);
        }
      }
    }
    return;
  }
  if (tables) return;   // every sample of one axis is out of range: no gradient
  // ---- scatter loop (roi_align_bwd_kernel) for RoIs beyond the tables
  for (int ph = 0; ph < PH; ++ph) {
    for (int pw = 0; pw < PW; ++pw) {
      const float g = c_ok ? gk[ph * p.so_h + pw * p.so_w] : 0.f;
      for (int iy = 0; iy < gh; ++iy) {
        const float yy = roi_start_h + ph * bin_h + (float)(iy + .5f) * bin_h / (float)gh;
        for (int ix = 0; ix < gw; ++ix) {
          const float xx = roi_start_w + pw * bin_w + (float)(ix + .5f) * bin_w / (float)gw;
          float x = xx, y = yy;
          if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) continue;
          if (y <= 0) y = 0;
          if (x <= 0) x = 0;
          int y_low = (int)y, x_low = (int)x, y_high, x_high;
          if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
          if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
          const float ly = y - y_low, lx = x - x_low;
          const float hy = 1.f - ly, hx = 1.f - lx;
          if (c_ok) {
            unsafeAtomicAdd(gin + (long long)(y_low * W + x_low) * p.C, g * (hy * hx) / count);
            unsafeAtomicAdd(gin + (long long)(y_low * W + x_high) * p.C, g * (hy * lx) / count);
            unsafeAtomicAdd(gin + (long long)(y_high * W + x_low) * p.C, g * (ly * hx) / count);
            unsafeAtomicAdd(gin + (long long)(y_high * W + x_high) * p.C, g * (ly * lx) / count);
          }
        }
      }
    }
  }
}

// Reference-shaped op (csrc/vision.cpp:97, ROIAlign.h:88-128): grad [K,C,ph,pw] contiguous -> grad_input [B,C,H,W],
// zeroed here (the reference returns a fresh at::zeros tensor, ROIAlign_cuda.cu:392).
extern "C" int lvc_roi_align_backward_nchw(const float* grad, const float* rois, float* grad_input, int B, int C,
                                           int H, int W, int K, int pooled_h, int pooled_w, float spatial_scale,
                                           int sampling_ratio, int aligned, int* d_status, void* stream) {
  LVC_CHECK_ARG(grad_input && (K == 0 || (grad && rois)), "null pointer");
  LVC_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && pooled_h > 0 && pooled_w > 0 && K >= 0, "bad shape");
  if (hipMemsetAsync(grad_input, 0, (size_t)B * C * H * W * sizeof(float), (hipStream_t)stream) != hipSuccess) {
    lvc_set_error("%s: hipMemsetAsync failed", __func__);
    return LVC_ERR_HIP;
  }
  if (K == 0) return LVC_OK;
  RoiAlignArgs a;
  memset(&a, 0, sizeof a);
  a.feat[0] = grad_input; a.H[0] = H; a.W[0] = W; a.scale[0] = spatial_scale;
  a.sb[0] = (long long)C * H * W;
  a.nhwc = 0; a.C = C; a.rois = rois; a.levels = nullptr; a.num_valid = nullptr;
  a.K = K; a.ph = pooled_h; a.pw = pooled_w; a.sampling_ratio = sampling_ratio; a.aligned = aligned;
  a.out = const_cast<float*>(grad);
  a.so_k = (long long)C * pooled_h * pooled_w; a.so_c = (long long)pooled_h * pooled_w;
  a.so_h = pooled_w; a.so_w = 1;
  a.status = d_status;
  hipLaunchKernelGGL(roi_align_bwd_kernel, dim3(K, lvc_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, a);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// Engine op: gradient of lvc_roi_align_fpn_nhwc.  grad [K, ph, pw, C] channels-last -> grad_feats[l] [B, Hl, Wl, C]
// (zeroed here), one launch over all RoIs: the backward of the per-level gather/scatter loop of
// detectron2/modeling/poolers.py:236-246.
extern "C" int lvc_roi_align_fpn_backward_nhwc(const float* grad, float* const* grad_feats, const int* Hs,
                                               const int* Ws, const float* scales, int L, int B, int C,
                                               const float* rois, const int* levels, const int* d_num_valid, int K,
                                               int pooled_h, int pooled_w, int sampling_ratio, int aligned,
                                               int* d_status, void* stream) {
  LVC_CHECK_ARG(L >= 1 && L <= LVC_MAX_LEVELS, "1..8 levels");
  LVC_CHECK_ARG(grad_feats && Hs && Ws && scales && (K == 0 || (grad && rois)), "null pointer");
  LVC_CHECK_ARG(B > 0 && C > 0 && pooled_h > 0 && pooled_w > 0 && K >= 0, "bad shape");
  LVC_CHECK_ARG(L == 1 || levels, "levels required when L > 1");
  RoiAlignArgs a;
  memset(&a, 0, sizeof a);
  for (int l = 0; l < L; ++l) {
    LVC_CHECK_ARG(grad_feats[l] && Hs[l] > 0 && Ws[l] > 0, "bad level");
    if (hipMemsetAsync(grad_feats[l], 0, (size_t)B * Hs[l] * Ws[l] * C * sizeof(float), (hipStream_t)stream) != hipSuccess) {
      lvc_set_error("%s: hipMemsetAsync failed", __func__);
      return LVC_ERR_HIP;
    }
    a.feat[l] = grad_feats[l]; a.H[l] = Hs[l]; a.W[l] = Ws[l]; a.scale[l] = scales[l];
    a.sb[l] = (long long)C * Hs[l] * Ws[l];
  }
  if (K == 0) return LVC_OK;
  a.nhwc = 1; a.C = C; a.rois = rois; a.levels = levels; a.num_valid = d_num_valid;
  a.K = K; a.ph = pooled_h; a.pw = pooled_w; a.sampling_ratio = sampling_ratio; a.aligned = aligned;
  a.out = const_cast<float*>(grad);
  a.so_k = (long long)C * pooled_h * pooled_w; a.so_c = 1;
  a.so_h = (long long)pooled_w * C; a.so_w = C;
  a.status = d_status;
  if (pooled_h == 7 && pooled_w == 7)      // the separable form: one atomic per footprint pixel
    hipLaunchKernelGGL((roi_align_bwd_rows_kernel<7, 7>), dim3(K, lvc_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(roi_align_bwd_kernel, dim3(K, lvc_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, a);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// Reference-shaped op: NCHW in, [K,C,ph,pw] out (csrc/vision.cpp:96 roi_align_forward).
extern "C" int lvc_roi_align_forward_nchw(const float* input, const float* rois, float* output, int B,
                                          int C, int H, int W, int K, int pooled_h, int pooled_w,
                                          float spatial_scale, int sampling_ratio, int aligned,
                                          int* d_status, void* stream) {
  LVC_CHECK_ARG(K == 0 || (input && rois && output), "null pointer");
  LVC_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && pooled_h > 0 && pooled_w > 0 && K >= 0, "bad shape");
  RoiAlignArgs a;
  memset(&a, 0, sizeof a);
  a.feat[0] = input; a.H[0] = H; a.W[0] = W; a.scale[0] = spatial_scale;
  a.sb[0] = (long long)C * H * W;
  a.nhwc = 0; a.C = C; a.rois = rois; a.levels = nullptr; a.num_valid = nullptr;
  a.K = K; a.ph = pooled_h; a.pw = pooled_w; a.sampling_ratio = sampling_ratio; a.aligned = aligned;
  a.out = output;
  a.so_k = (long long)C * pooled_h * pooled_w; a.so_c = (long long)pooled_h * pooled_w;
  a.so_h = pooled_w; a.so_w = 1;
  a.status = d_status;
  return launch(a, stream);
}

// Engine op: L pyramid levels in NHWC, per-RoI level ids, output [K, ph, pw, C] (channels-last, the
// layout the box-head GEMM consumes).  Replaces the per-level gather/scatter loop of
// detectron2/modeling/poolers.py:236-246 with one launch over all RoIs.
// Order in which the workgroups of the wave-form forward take the RoIs: largest windows first (buckets of half an octave of the
// window area, a counting sort in one workgroup; the order inside a bucket is whatever the atomics give -- it decides which CU
// pools a RoI when, never a value).  The launch's time follows the window area, and in proposal (score) order a few large RoIs
// start last and finish alone: 0.56 - 0.61 ms in proposal order, 0.45 - 0.48 ms largest first (scripts/probe_roi_split.py).
struct RoiOrderArgs {
  const float* rois;
  const int* levels;
  float scale[LVC_MAX_LEVELS];
  int K, ph;
  int* order;
};
__global__ __launch_bounds__(1024) void roi_work_order_kernel(RoiOrderArgs p) {
  __shared__ int hist[64];
  const int tid = threadIdx.x;
  if (tid < 64) hist[tid] = 0;
  __syncthreads();
  auto bucket = [&](int k) {
    const float* r = p.rois + (long long)k * 5;
    const float sc = p.scale[p.levels ? p.levels[k] : 0];
    const float w = fmaxf((r[3] - r[1]) * sc, 0.f) + 2.f, h = fmaxf((r[4] - r[2]) * sc, 0.f) + 2.f * (float)p.ph;
    const float a = w * h;                                     // window pixels over the output rows, roughly
    int b = a == a ? (int)(2.f * log2f(fminf(fmaxf(a, 1.f), 1e9f))) : 0;
    b = b < 0 ? 0 : b > 63 ? 63 : b;
    return 63 - b;                                             // large first
  };
  for (int k = tid; k < p.K; k += 1024) atomicAdd(&hist[bucket(k)], 1);
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int i = 0; i < 64; ++i) { const int n = hist[i]; hist[i] = run; run += n; }
  }
  __syncthreads();
  for (int k = tid; k < p.K; k += 1024) p.order[atomicAdd(&hist[bucket(k)], 1)] = k;
}

extern "C" int lvc_roi_work_order(const float* rois, const int* levels, const float* scales, int L, int K, int pooled_h,
                                  int* d_order, void* stream) {
  LVC_CHECK_ARG(L >= 1 && L <= LVC_MAX_LEVELS && K >= 0 && pooled_h > 0, "bad shape");
  if (K == 0) return LVC_OK;
  LVC_CHECK_ARG(rois && scales && d_order && (L == 1 || levels), "null pointer");
  RoiOrderArgs a;
  memset(&a, 0, sizeof a);
  a.rois = rois; a.levels = levels; a.K = K; a.ph = pooled_h; a.order = d_order;
  for (int l = 0; l < L; ++l) a.scale[l] = scales[l];
  hipLaunchKernelGGL(roi_work_order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// XCD-local work order.  Workgroup i runs on XCD i % 8; the windows of a launch are read through that XCD's 4 MB L2, and since the
// column-sum form the launch is bound by what misses it (3 GB of window reads, 0.7 GB of pyramid).  Here position i of the order holds
// a RoI of image i % B (B images; with B = 8 an XCD pools ONE image's RoIs), and inside an image the RoIs go by (level, 16-pixel band
// of the box centre's row): neighbours in time on an XCD read neighbouring rows of one map.  0.394 -> 0.368 ms on the bench batch
// (scripts/probe_roi_orders.py).  One workgroup, counting sort with B x 256 counters in LDS; B <= 16, else the area order above.
#define ROI_XCD_MAX_B 16
struct RoiOrderXcdArgs {
  const float* rois;
  const int* levels;
  int K, B;
  int* order;
};
__global__ __launch_bounds__(1024) void roi_work_order_xcd_kernel(RoiOrderXcdArgs p) {
  __shared__ int hist[ROI_XCD_MAX_B][256];
  __shared__ int cnt[ROI_XCD_MAX_B], tail0[ROI_XCD_MAX_B], s_min;
  const int tid = threadIdx.x;
  for (int i = tid; i < ROI_XCD_MAX_B * 256; i += 1024) (&hist[0][0])[i] = 0;
  __syncthreads();
  auto key = [&](int k, int& b) {
    const float* r = p.rois + (long long)k * 5;
    const float bf = r[0];
    b = bf >= 0.f && bf < (float)p.B ? (int)bf : 0;
    const int lv = p.levels ? (p.levels[k] & 3) : 0;
    const float yc = (r[2] + r[4]) * 0.5f;
    int band = yc == yc ? (int)(fminf(fmaxf(yc, 0.f), 1e6f) * (1.f / 16.f)) : 0;
    band = band > 63 ? 63 : band;
    return lv * 64 + band;
  };
  for (int k = tid; k < p.K; k += 1024) { int b; const int ky = key(k, b); atomicAdd(&hist[b][ky], 1); }
  __syncthreads();
  if (tid < p.B) {      // exclusive prefix inside the image; its count
    int run = 0;
    for (int i = 0; i < 256; ++i) { const int n = hist[tid][i]; hist[tid][i] = run; run += n; }
    cnt[tid] = run;
  }
  __syncthreads();
  if (tid == 0) {
    int m = cnt[0];
    for (int b = 1; b < p.B; ++b) m = cnt[b] < m ? cnt[b] : m;
    s_min = m;
    int run = 0;
    for (int b = 0; b < p.B; ++b) { tail0[b] = run; run += cnt[b] - m; }
  }
  __syncthreads();
  const int m = s_min;
  for (int k = tid; k < p.K; k += 1024) {
    int b;
    const int ky = key(k, b);
    const int r = atomicAdd(&hist[b][ky], 1);      // rank inside the image (order inside a key: whatever the atomics give)
    p.order[r < m ? r * p.B + b : m * p.B + tail0[b] + (r - m)] = k;
  }
}

extern "C" int lvc_roi_work_order_xcd(const float* rois, const int* levels, int K, int B, int* d_order, void* stream) {
  LVC_CHECK_ARG(K >= 0 && B >= 1 && B <= ROI_XCD_MAX_B, "1..16 images");
  if (K == 0) return LVC_OK;
  LVC_CHECK_ARG(rois && d_order, "null pointer");
  RoiOrderXcdArgs a;
  a.rois = rois; a.levels = levels; a.K = K; a.B = B; a.order = d_order;
  hipLaunchKernelGGL(roi_work_order_xcd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

static int roi_align_fpn_nhwc_impl(const float* const* feats, const int* Hs, const int* Ws,
                                      const float* scales, int L, int B, int C, const float* rois,
                                      const int* levels, const int* d_num_valid, int K, int pooled_h,
                                      int pooled_w, int sampling_ratio, int aligned, float* output,
                                      int* d_status, const int* d_order, void* stream) {
  LVC_CHECK_ARG(L >= 1 && L <= LVC_MAX_LEVELS, "1..8 levels");
  LVC_CHECK_ARG(K == 0 || (feats && Hs && Ws && scales && rois && output), "null pointer");
  LVC_CHECK_ARG(B > 0 && C > 0 && pooled_h > 0 && pooled_w > 0 && K >= 0, "bad shape");
  LVC_CHECK_ARG(L == 1 || levels, "levels required when L > 1");
  RoiAlignArgs a;
  memset(&a, 0, sizeof a);
  for (int l = 0; l < L; ++l) {
    a.feat[l] = feats[l]; a.H[l] = Hs[l]; a.W[l] = Ws[l]; a.scale[l] = scales[l];
    a.sb[l] = (long long)C * Hs[l] * Ws[l];
  }
  a.nhwc = 1; a.C = C; a.rois = rois; a.levels = levels; a.num_valid = d_num_valid;
  a.K = K; a.ph = pooled_h; a.pw = pooled_w; a.sampling_ratio = sampling_ratio; a.aligned = aligned;
  a.out = output;
  a.so_k = (long long)C * pooled_h * pooled_w; a.so_c = 1;
  a.so_h = (long long)pooled_w * C; a.so_w = C;
  a.status = d_status;
  a.order = d_order;
  return launch(a, stream);
}

extern "C" int lvc_roi_align_fpn_nhwc(const float* const* feats, const int* Hs, const int* Ws,
                                      const float* scales, int L, int B, int C, const float* rois,
                                      const int* levels, const int* d_num_valid, int K, int pooled_h,
                                      int pooled_w, int sampling_ratio, int aligned, float* output,
                                      int* d_status, void* stream) {
  return roi_align_fpn_nhwc_impl(feats, Hs, Ws, scales, L, B, C, rois, levels, d_num_valid, K, pooled_h, pooled_w, sampling_ratio, aligned,
                                 output, d_status, nullptr, stream);
}

// The same with a work order from lvc_roi_work_order (d_order [K]: a permutation of 0..K-1; NULL = RoI order).  Outputs are
// identical, row k of the output is RoI k either way.
extern "C" int lvc_roi_align_fpn_nhwc_ordered(const float* const* feats, const int* Hs, const int* Ws,
                                      const float* scales, int L, int B, int C, const float* rois,
                                      const int* levels, const int* d_num_valid, int K, int pooled_h,
                                      int pooled_w, int sampling_ratio, int aligned, float* output,
                                      int* d_status, const int* d_order, void* stream) {
  return roi_align_fpn_nhwc_impl(feats, Hs, Ws, scales, L, B, C, rois, levels, d_num_valid, K, pooled_h, pooled_w, sampling_ratio, aligned,
                                 output, d_status, d_order, stream);
}
