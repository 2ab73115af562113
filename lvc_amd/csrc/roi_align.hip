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
  int xcd_rows;                  // LDS-staged forward: the output rows of a RoI on one XCD (LVC_ROI_XCD_ROWS, default 1)
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

// NHWC LDS-staged path: one 512-thread workgroup per (RoI, output row ph), 256 channels (wave = bin, lane = 4 channels).  The feature window that
// the 7 bins of that output row can touch -- rows [floor(y_first), floor(y_last)+1] x columns [floor(x_first),
// floor(x_last)+1] -- is copied ONCE into LDS with fully coalesced 1-KiB pixel segments (64 lanes x dwordx4), then
// every thread (bin pw = tid/64, 4 channels = tid%64) walks its bin's samples reading the four bilinear taps from
// LDS (a wave reads one contiguous 1-KiB pixel: conflict-free ds_read_b128).  A small RoI touches each
// feature pixel ~10 times (49 bins x g^2 samples x 4 taps over <= 9x9 pixels); after staging it is fetched from
// L2/HBM once per output row (~3 rows each), which cuts the L2 read volume by ~3x.  Windows larger than
// ROI_LDS_MAX_PIX pixels (very elongated boxes) take the direct-from-L2 loop: same arithmetic, same order.
#ifndef ROI_LDS_MAX_PIX
#define ROI_LDS_MAX_PIX 24
#endif
#ifndef ROI_TAB
#define ROI_TAB 32      // samples per pass of a wave's weight / offset table
#endif
__global__ __launch_bounds__(512, 8) void roi_align_fwd_nhwc_lds_kernel(RoiAlignArgs p) {
  __shared__ __attribute__((aligned(16))) float win[ROI_LDS_MAX_PIX * 256];
  __shared__ float4 tab_w[8][ROI_TAB];     // per wave (bin): the four bilinear weights of up to 64 in-range samples ...
  __shared__ int4 tab_o[8][ROI_TAB];       // ... and the element offsets of their four taps (into win when staged, else into the level's image)
  const int tid = threadIdx.x;
  // workgroup -> (RoI, output row).  The hardware deals consecutive workgroups to the 8 XCDs in turn; the rows of one RoI read
  // overlapping feature rows and the same columns, so they are kept on ONE XCD (one fetch of the window into that L2 instead of
  // one per XCD): groups of p.ph consecutive slots of an XCD form a RoI, RoIs are dealt to the XCDs round robin
  int k, ph;
  if (p.xcd_rows) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int g = slot / p.ph;
    ph = slot - g * p.ph;
    k = g * 8 + xcd;
    if (k >= p.K) return;
  } else {
    k = blockIdx.x / p.ph;
    ph = blockIdx.x - k * p.ph;
  }
  const int cbase = blockIdx.y * 256;
  const int cq = tid & 63, bin = tid >> 6;   // bin = pw (8 slots, p.pw <= 8 used)
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
      if (p.status && tid == 0 && blockIdx.y == 0 && ph == 0) atomicOr(p.status, 1);
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
  const long long C = p.C;
  const float* in = p.feat[lvl] + (long long)b * p.sb[lvl] + cbase + cq * 4;

  // window of this output row (conservative: every tap of every in-range sample lies inside)
  const float yf = roi_start_h + ph * bin_h, yl = roi_start_h + (ph + 1) * bin_h;
  int y0 = (int)floorf(fmaxf(yf, 0.f)), y1 = (int)floorf(fmaxf(yl, 0.f)) + 1;
  int x0 = (int)floorf(fmaxf(roi_start_w, 0.f)), x1 = (int)floorf(fmaxf(roi_start_w + roi_width, 0.f)) + 1;
  y0 = min(y0, H - 1); y1 = min(y1, H - 1); x0 = min(x0, W - 1); x1 = min(x1, W - 1);
  const int nrows = y1 - y0 + 1, ncols = x1 - x0 + 1;
  const bool staged = (nrows > 0) && (ncols > 0) && (nrows * ncols <= ROI_LDS_MAX_PIX) && (cbase + 256 <= p.C);
  if (staged) {
    const int npix = nrows * ncols;
    for (int px = bin; px < npix; px += 8) {   // 8 pixels per pass, one wave (64 lanes x 16 B) each
      const int yy = px / ncols, xx = px - yy * ncols;
      const float4 v = *reinterpret_cast<const float4*>(in + (long long)((y0 + yy) * W + x0 + xx) * C);
      *reinterpret_cast<float4*>(win + px * 256 + cq * 4) = v;
    }
    __syncthreads();
  }
  if (bin >= p.pw) return;        // nothing below synchronises across waves
  const int pw = bin;
  float4 acc = {0.f, 0.f, 0.f, 0.f};
  // The sample positions, clamps and bilinear weights depend on (RoI, ph, pw, iy, ix) only -- the same for the 64 lanes of this
  // wave.  64 samples at a time, lane = SAMPLE computes its four weights and tap offsets once (the reference's arithmetic), the
  // in-range samples are compacted in order (iy outer, ix inner) into this wave's LDS table, and the channel loop below reads a
  // sample as two broadcast ds_read_b128: ~25 instead of ~60 vector instructions per sample and wave, the sums term by term as
  // the reference forms them.
  const int nsamp = gh * gw;
  float4* tw = tab_w[bin];
  int4* to = tab_o[bin];
  const bool c_ok = cbase + cq * 4 < p.C;
  for (int s0 = 0; s0 < nsamp; s0 += ROI_TAB) {
    const int smp = s0 + cq;
    bool valid = smp < nsamp && cq < ROI_TAB;
    float4 wv = {0.f, 0.f, 0.f, 0.f};
    int4 ov = {0, 0, 0, 0};
    if (valid) {
      const int iy = smp / gw, ix = smp - iy * gw;
      const float yy = roi_start_h + ph * bin_h + (float)(iy + .5f) * bin_h / (float)gh;
      const float xx = roi_start_w + pw * bin_w + (float)(ix + .5f) * bin_w / (float)gw;
      float x = xx, y = yy;
      if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) valid = false;
      else {
        if (y <= 0) y = 0;
        if (x <= 0) x = 0;
        int y_low = (int)y, x_low = (int)x, y_high, x_high;
        if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
        if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
        const float ly = y - y_low, lx = x - x_low;
        const float hy = 1.f - ly, hx = 1.f - lx;
        wv = float4{hy * hx, hy * lx, ly * hx, ly * lx};
        if (staged)
          ov = int4{((y_low - y0) * ncols + (x_low - x0)) * 256, ((y_low - y0) * ncols + (x_high - x0)) * 256,
                    ((y_high - y0) * ncols + (x_low - x0)) * 256, ((y_high - y0) * ncols + (x_high - x0)) * 256};
        else
          ov = int4{(y_low * W + x_low) * (int)C, (y_low * W + x_high) * (int)C, (y_high * W + x_low) * (int)C, (y_high * W + x_high) * (int)C};
      }
    }
    const unsigned long long m = __ballot(valid);
    const int n = __popcll(m);
    if (valid) {
      const int pos = __popcll(m & ((1ull << cq) - 1ull));
      tw[pos] = wv;
      to[pos] = ov;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (c_ok) {
      // one sample per iteration: rounds of four samples with their sixteen taps in flight together measured slower (0.70 vs 0.66 ms on
      // the bench batch's proposals): the kernel is bound by the taps' LDS / L1 throughput, not by their latency
      auto run = [&](const float* base) {
        for (int e = 0; e < n; ++e) {
          const float4 w = tw[e];
          const int4 o = to[e];
          const float4 v1 = *reinterpret_cast<const float4*>(base + o.x), v2 = *reinterpret_cast<const float4*>(base + o.y);
          const float4 v3 = *reinterpret_cast<const float4*>(base + o.z), v4 = *reinterpret_cast<const float4*>(base + o.w);
          acc.x += w.x * v1.x + w.y * v2.x + w.z * v3.x + w.w * v4.x;
          acc.y += w.x * v1.y + w.y * v2.y + w.z * v3.y + w.w * v4.y;
          acc.z += w.x * v1.z + w.y * v2.z + w.z * v3.z + w.w * v4.z;
          acc.w += w.x * v1.w + w.y * v2.w + w.z * v3.w + w.w * v4.w;
        }
      };
      if (staged) run(win + cq * 4);
      else run(in);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  }
  if (!c_ok) return;
  float4 o = {acc.x / count, acc.y / count, acc.z / count, acc.w / count};
  *reinterpret_cast<float4*>(p.out + (long long)k * p.so_k + ph * p.so_h + pw * p.so_w + cbase + cq * 4) = o;
}

static int launch(RoiAlignArgs& a, void* stream) {
  if (a.K == 0) return LVC_OK;
  bool small = true;     // the staged kernel keeps tap offsets inside one image of a level as 32-bit element counts
  for (int l = 0; l < LVC_MAX_LEVELS; ++l)
    if (a.feat[l] && (long long)a.H[l] * a.W[l] * a.C >= (1ll << 31)) small = false;
  if (a.nhwc && (a.C & 255) == 0 && a.pw <= 8 && a.num_valid == nullptr && a.so_c == 1 && small) {
    constexpr int xcd_rows = 1;
    a.xcd_rows = xcd_rows;
    dim3 gridl(xcd_rows ? lvc_cdiv(a.K, 8) * 8 * a.ph : a.K * a.ph, a.C / 256), blockl(512);
    hipLaunchKernelGGL(roi_align_fwd_nhwc_lds_kernel, gridl, blockl, 0, (hipStream_t)stream, a);
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
extern "C" int lvc_roi_align_fpn_nhwc(const float* const* feats, const int* Hs, const int* Ws,
                                      const float* scales, int L, int B, int C, const float* rois,
                                      const int* levels, const int* d_num_valid, int K, int pooled_h,
                                      int pooled_w, int sampling_ratio, int aligned, float* output,
                                      int* d_status, void* stream) {
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
  return launch(a, stream);
}
