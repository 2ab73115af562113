/*
 * oracle.c -- CPU restatement of the integer/geometry kernels on the
 * Faster-R-CNN hot path.  TEST INFRASTRUCTURE ONLY: nothing under lvc_amd/
 * may link, import or call this file; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, as the checker.
 *
 * Every function cites the reference code it restates (paths relative to
 * /root/reference).  Parity pins:
 *   - orc_roi_align_forward is checked against the reference's own
 *     ROIAlign_cpu.cpp compiled into oracle/_ref (tests/test_oracle_golden.py)
 *     and against golden vectors produced by it (tests/golden/roi_align_*.npz);
 *     orc_roi_align_backward likewise (golden roi_align.npz keys bwd_*).
 *   - orc_nms restates torchvision 0.8.2's nms_cpu_kernel, a third-party
 *     dependency that is NOT under /root/reference (README.md:65-67 pins
 *     torchvision 0.8.2; call sites detectron2/layers/nms.py:6-7,20,25).
 *     "parity unpinned" by the reference for that function: it holds no tests or golden vectors
 *     for NMS; the restatement follows the published algorithm (sort scores
 *     descending, greedy suppress on  inter/(a_i+a_j-inter) > thr  with the
 *     float IoU compared against the DOUBLE threshold).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: no FMA contraction,
 * so that every float op rounds exactly as the x86-64 reference build does).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- ROIAlign forward ---------------------------------------------------
 * Restates detectron2/layers/csrc/ROIAlign/ROIAlign_cpu.cpp:20-114
 * (pre_calc_for_bilinear_interpolate) and :116-218 (ROIAlignForward), T=float.
 * input NCHW [B,C,H,W]; rois [K,5] = (batch_idx,x1,y1,x2,y2); out [K,C,ph,pw].
 * Returns 0, or -1 when aligned and an RoI has negative size
 * (ROIAlign_cpu.cpp:149-152 asserts there).
 */
#define ORC_T float
#define ORC_FN orc_roi_align_forward
#define ORC_TAP orc_tap
#define ORC_CEIL ceilf
#include "roi_align_fwd.inc"
#undef ORC_T
#undef ORC_FN
#undef ORC_TAP
#undef ORC_CEIL
/* T = double: used by the fp64 evaluation of the whole path (tests/test_gpu_chain.py), which measures how far the
 * reference's own fp32 CPU path and this build each are from the exact result. */
#define ORC_T double
#define ORC_FN orc_roi_align_forward_f64
#define ORC_TAP orc_tap64
#define ORC_CEIL ceil
#include "roi_align_fwd.inc"
#undef ORC_T
#undef ORC_FN
#undef ORC_TAP
#undef ORC_CEIL

/* ---- ROIAlign backward --------------------------------------------------
 * Restates ROIAlign_cpu.cpp:219-281 (bilinear_interpolate_gradient) and :288-406 (ROIAlignBackward), T=float:
 * every pooled element (n, c, ph, pw), visited in that index order, scatters grad * w / count into the four
 * neighbours of each of its gh x gw sampling points; samples outside [-1, H] x [-1, W] contribute nothing.
 * grad_output [K,C,ph,pw] contiguous, grad_input [B,C,H,W] (zeroed here, like ROIAlign_backward_cpu's at::zeros).
 * The single-threaded accumulation order of the reference is kept, so results are bit-identical to it.
 */
int orc_roi_align_backward(const float* grad_output, const float* rois, float* grad_input, int K, int B, int C,
                           int H, int W, int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio,
                           int aligned) {
  memset(grad_input, 0, (size_t)B * C * H * W * sizeof(float));
  for (int n = 0; n < K; n++) {
    const float* r = rois + (size_t)n * 5;
    int b = (int)r[0];
    float offset = aligned ? 0.5f : 0.0f;
    float roi_start_w = r[1] * spatial_scale - offset;
    float roi_start_h = r[2] * spatial_scale - offset;
    float roi_end_w = r[3] * spatial_scale - offset;
    float roi_end_h = r[4] * spatial_scale - offset;
    float roi_width = roi_end_w - roi_start_w;
    float roi_height = roi_end_h - roi_start_h;
    if (aligned) {
      if (!(roi_width >= 0 && roi_height >= 0)) return -1;
    } else {
      roi_width = roi_width > 1.f ? roi_width : 1.f;
      roi_height = roi_height > 1.f ? roi_height : 1.f;
    }
    float bin_h = roi_height / (float)pooled_h;
    float bin_w = roi_width / (float)pooled_w;
    int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_height / (float)pooled_h);
    int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_width / (float)pooled_w);
    float count = (float)(gh * gw);
    for (int c = 0; c < C; c++) {
      float* gin = grad_input + ((size_t)b * C + c) * H * W;
      const float* gout = grad_output + ((size_t)n * C + c) * pooled_h * pooled_w;
      for (int ph = 0; ph < pooled_h; ph++)
        for (int pw = 0; pw < pooled_w; pw++) {
          float g = gout[ph * pooled_w + pw];
          for (int iy = 0; iy < gh; iy++) {
            float y0 = roi_start_h + ph * bin_h + (float)(iy + .5f) * bin_h / (float)gh;
            for (int ix = 0; ix < gw; ix++) {
              float x = roi_start_w + pw * bin_w + (float)(ix + .5f) * bin_w / (float)gw;
              float y = y0;
              if (y < -1.0 || y > H || x < -1.0 || x > W) continue;
              if (y <= 0) y = 0;
              if (x <= 0) x = 0;
              int y_low = (int)y, x_low = (int)x, y_high, x_high;
              if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
              if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
              float ly = y - y_low, lx = x - x_low;
              float hy = (float)(1. - ly), hx = (float)(1. - lx);
              float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
              gin[y_low * W + x_low] += g * w1 / count;
              gin[y_low * W + x_high] += g * w2 / count;
              gin[y_high * W + x_low] += g * w3 / count;
              gin[y_high * W + x_high] += g * w4 / count;
            }
          }
        }
    }
  }
  return 0;
}

/* ---- greedy NMS -----------------------------------------------------------
 * Restates torchvision 0.8.2 ops/cpu/nms_cpu.cpp nms_cpu_kernel<float>
 * (third-party; see header).  `order` = indices of `scores` sorted descending
 * (the caller sorts: torch.sort in the shim / stable argsort in oracle/ops.py).
 * keep gets box indices in score order; returns how many.
 */
int64_t orc_nms(const float* boxes, const int64_t* order, int64_t n, double thr, int64_t* keep) {
  if (n == 0) return 0;
  unsigned char* sup = (unsigned char*)calloc((size_t)n, 1);
  float* areas = (float*)malloc((size_t)n * sizeof(float));
  for (int64_t i = 0; i < n; i++)
    areas[i] = (boxes[4 * i + 2] - boxes[4 * i]) * (boxes[4 * i + 3] - boxes[4 * i + 1]);
  int64_t nk = 0;
  for (int64_t _i = 0; _i < n; _i++) {
    int64_t i = order[_i];
    if (sup[i]) continue;
    keep[nk++] = i;
    float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
    float iarea = areas[i];
    for (int64_t _j = _i + 1; _j < n; _j++) {
      int64_t j = order[_j];
      if (sup[j]) continue;
      float xx1 = ix1 < boxes[4 * j] ? boxes[4 * j] : ix1;          /* std::max(a,b) */
      float yy1 = iy1 < boxes[4 * j + 1] ? boxes[4 * j + 1] : iy1;
      float xx2 = boxes[4 * j + 2] < ix2 ? boxes[4 * j + 2] : ix2;  /* std::min(a,b) */
      float yy2 = boxes[4 * j + 3] < iy2 ? boxes[4 * j + 3] : iy2;
      float w = xx2 - xx1; if (!(w > 0.f)) w = 0.f;
      float h = yy2 - yy1; if (!(h > 0.f)) h = 0.f;
      float inter = w * h;
      float ovr = inter / (iarea + areas[j] - inter);
      if ((double)ovr > thr) sup[j] = 1;
    }
  }
  free(sup);
  free(areas);
  return nk;
}
