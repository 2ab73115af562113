// train.hip -- training-time kernels of the fine-tune step (BASELINE config 3: only box_predictor trains).
//   lvc_match_boxes        = pairwise_iou + Matcher.__call__ (+ set_low_quality_matches_)
//       reference detectron2/structures/boxes.py:315-347, detectron2/modeling/matcher.py:61-126; used by
//       RPN.label_and_sample_anchors (rpn.py:269-325, 268 569 anchors x G) and
//       ROIHeads.label_and_sample_proposals (lvc/modeling/roi_heads/roi_heads.py:173-278)
//   lvc_fast_rcnn_losses   = FastRCNNOutputs.losses (lvc/modeling/roi_heads/fast_rcnn.py:267-279 softmax CE "mean",
//       :296-359 box_reg_loss smooth-L1 "sum" / #rows) forward AND the gradients w.r.t. logits / deltas
//   lvc_rpn_losses         = RPN.losses (rpn.py:328-400): BCE-with-logits "sum" over valid anchors and smooth-L1 "sum"
//       over positive anchors of get_deltas(anchor, gt), both / (batch_size_per_image * num_images);
//       lvc_rpn_losses_grad also emits d/dlogits, d/ddeltas (the base / ft_all yamls train the RPN).
// Built with -ffp-contract=off like the other geometry files.
#include "common.h"

__device__ __forceinline__ float iou_ref(float ax1, float ay1, float ax2, float ay2, float bx1, float by1, float bx2,
                                         float by2) {
  // boxes.py:315-347: wh = clamp(min(rb) - max(lt), 0); inter = w*h; iou = inter > 0 ? inter/(a1+a2-inter) : 0
  const float a1 = (ax2 - ax1) * (ay2 - ay1), a2 = (bx2 - bx1) * (by2 - by1);
  float w = fminf(ax2, bx2) - fmaxf(ax1, bx1);
  float h = fminf(ay2, by2) - fmaxf(ay1, by1);
  w = w < 0.f ? 0.f : w;
  h = h < 0.f ? 0.f : h;
  const float inter = w * h;
  return inter > 0.f ? inter / (a1 + a2 - inter) : 0.f;
}

#define MAX_GT 512
// pass 1: per box: max IoU over the G gt boxes + first arg-max; per gt: max IoU over boxes (atomicMax on the bits)
__global__ __launch_bounds__(256) void match_pass1_kernel(const float* __restrict__ gt, int G,
                                                          const float* __restrict__ boxes, int N,
                                                          float* __restrict__ matched_vals,
                                                          long long* __restrict__ matches,
                                                          unsigned int* __restrict__ gt_best) {
  __shared__ float sgt[MAX_GT * 4];
  __shared__ unsigned int sbest[MAX_GT];
  for (int i = threadIdx.x; i < G * 4; i += 256) sgt[i] = gt[i];
  for (int i = threadIdx.x; i < G; i += 256) sbest[i] = 0u;
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n < N) {
    const float4 b = *reinterpret_cast<const float4*>(boxes + (size_t)n * 4);
    float best = -1.f; int bi = 0;
    for (int g = 0; g < G; ++g) {
      const float v = iou_ref(sgt[g * 4], sgt[g * 4 + 1], sgt[g * 4 + 2], sgt[g * 4 + 3], b.x, b.y, b.z, b.w);
      if (v > best) { best = v; bi = g; }
      atomicMax(&sbest[g], __float_as_uint(v));  // v >= 0: bit pattern is monotone
    }
    matched_vals[n] = best;
    matches[n] = bi;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < G; i += 256) atomicMax(&gt_best[i], sbest[i]);
}

// pass 2: labels from thresholds; low-quality matches: any gt whose best IoU equals this box's IoU with it -> 1
__global__ __launch_bounds__(256) void match_pass2_kernel(const float* __restrict__ gt, int G,
                                                          const float* __restrict__ boxes, int N,
                                                          const float* __restrict__ matched_vals,
                                                          const unsigned int* __restrict__ gt_best, float t0, float t1,
                                                          int nthr, int l0, int l1, int l2, int allow_low_quality,
                                                          signed char* __restrict__ labels) {
  __shared__ float sgt[MAX_GT * 4];
  __shared__ float sbest[MAX_GT];
  for (int i = threadIdx.x; i < G * 4; i += 256) sgt[i] = gt[i];
  for (int i = threadIdx.x; i < G; i += 256) sbest[i] = __uint_as_float(gt_best[i]);
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const float v = matched_vals[n];
  // thresholds = [-inf, t0, (t1,) +inf]; labels l0 below t0, l1 in [t0,t1), l2 above (nthr == 1: l0 / l1)
  int lab;
  if (nthr == 1) lab = v < t0 ? l0 : l1;
  else lab = v < t0 ? l0 : (v < t1 ? l1 : l2);
  if (allow_low_quality) {
    const float4 b = *reinterpret_cast<const float4*>(boxes + (size_t)n * 4);
    for (int g = 0; g < G; ++g) {
      const float q = iou_ref(sgt[g * 4], sgt[g * 4 + 1], sgt[g * 4 + 2], sgt[g * 4 + 3], b.x, b.y, b.z, b.w);
      if (q == sbest[g]) { lab = 1; break; }
    }
  }
  labels[n] = (signed char)lab;
}

// gt [G,4], boxes [N,4] -> matches [N] int64 (arg-max gt, first on ties), labels [N] int8, matched_vals [N] fp32.
// thresholds/labels as in Matcher(thresholds, labels): nthr in {1,2}.  d_gt_best: [G] uint32 scratch (zeroed here).
extern "C" int lvc_match_boxes(const float* gt, int G, const float* boxes, int N, float t0, float t1, int nthr, int l0,
                               int l1, int l2, int allow_low_quality, long long* matches, signed char* labels,
                               float* matched_vals, unsigned int* d_gt_best, void* stream) {
  LVC_CHECK_ARG(G > 0 && G <= MAX_GT, "1..512 ground-truth boxes (the G == 0 case is handled by the host)");
  LVC_CHECK_ARG(N >= 0 && (nthr == 1 || nthr == 2), "bad arguments");
  if (N == 0) return LVC_OK;
  LVC_CHECK_ARG(gt && boxes && matches && labels && matched_vals && d_gt_best, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(d_gt_best, 0, sizeof(unsigned int) * G, st);
  hipLaunchKernelGGL(match_pass1_kernel, dim3(lvc_cdiv(N, 256)), dim3(256), 0, st, gt, G, boxes, N, matched_vals,
                     matches, d_gt_best);
  LVC_CHECK_LAUNCH();
  hipLaunchKernelGGL(match_pass2_kernel, dim3(lvc_cdiv(N, 256)), dim3(256), 0, st, gt, G, boxes, N, matched_vals,
                     d_gt_best, t0, t1, nthr, l0, l1, l2, allow_low_quality, labels);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

__device__ __forceinline__ void get_deltas(float sx1, float sy1, float sx2, float sy2, float tx1, float ty1, float tx2,
                                           float ty2, float wx, float wy, float ww, float wh, float* d) {
  // box_regression.py:40-71
  const float sw = sx2 - sx1, sh = sy2 - sy1, scx = sx1 + 0.5f * sw, scy = sy1 + 0.5f * sh;
  const float tw = tx2 - tx1, th = ty2 - ty1, tcx = tx1 + 0.5f * tw, tcy = ty1 + 0.5f * th;
  d[0] = wx * (tcx - scx) / sw; d[1] = wy * (tcy - scy) / sh;
  d[2] = ww * logf(tw / sw); d[3] = wh * logf(th / sh);
}
__device__ __forceinline__ float smooth_l1(float x, float t, float beta, float* grad) {
  const float n = fabsf(x - t);
  if (beta < 1e-5f) { *grad = x > t ? 1.f : (x < t ? -1.f : 0.f); return n; }
  if (n < beta) { *grad = (x - t) / beta; return 0.5f * n * n / beta; }
  *grad = x > t ? 1.f : -1.f;
  return n - 0.5f * beta;
}

// One wave per row (lanes over the K+1 logits and the row's regression columns: coalesced), four rows per workgroup;
// the per-row loss terms go to row_terms [2][R] (fp64) and fast_rcnn_losses_sum_kernel adds them in a fixed order, so
// the losses do not depend on the launch geometry.  (As one 1024-thread workgroup walking strided rows: 0.71 ms for 4096
// rows x 61 classes.)
__global__ __launch_bounds__(256) void fast_rcnn_losses_rows_kernel(
    const float* __restrict__ logits, int ld_cls, const float* __restrict__ deltas, int ld_delta, int K,
    int cls_agnostic, const float* __restrict__ proposals, const float* __restrict__ gt_boxes,
    const long long* __restrict__ gt_classes, int R, float wx, float wy, float ww, float wh, float beta,
    double* __restrict__ row_terms, float* __restrict__ dlogits, float* __restrict__ ddeltas) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const float* lg = logits + (size_t)r * ld_cls;
  const long long c = gt_classes[r];
  float mx = -INFINITY;
  for (int k = lane; k <= K; k += 64) mx = fmaxf(mx, lg[k]);
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
  for (int k = lane; k <= K; k += 64) sum += expf(lg[k] - mx);
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float lse = mx + logf(sum);
  for (int k = lane; k <= K; k += 64) dlogits[(size_t)r * (K + 1) + k] = (expf(lg[k] - lse) - (k == c ? 1.f : 0.f)) / (float)R;
  const int nreg = cls_agnostic ? 4 : 4 * K;
  const bool fg = c >= 0 && c < K;
  const int col = cls_agnostic ? 0 : 4 * (int)c;
  float gr[4] = {0.f, 0.f, 0.f, 0.f};
  double lb = 0.0;
  if (fg) {
    float t[4];
    const float* p = proposals + (size_t)r * 4;
    const float* g = gt_boxes + (size_t)r * 4;
    get_deltas(p[0], p[1], p[2], p[3], g[0], g[1], g[2], g[3], wx, wy, ww, wh, t);
    for (int j = 0; j < 4; ++j) {
      float gj;
      lb += (double)smooth_l1(deltas[(size_t)r * ld_delta + col + j], t[j], beta, &gj);
      gr[j] = gj / (float)R;
    }
  }
  for (int j = lane; j < nreg; j += 64) {
    const int q = j - col;
    ddeltas[(size_t)r * nreg + j] = (fg && q >= 0 && q < 4) ? (q == 0 ? gr[0] : q == 1 ? gr[1] : q == 2 ? gr[2] : gr[3]) : 0.f;
  }
  if (lane == 0) {
    row_terms[r] = (double)(lse - lg[c]);
    row_terms[R + r] = lb;
  }
}

__global__ __launch_bounds__(1024) void fast_rcnn_losses_sum_kernel(const double* __restrict__ row_terms, int R,
                                                                     float* __restrict__ out_losses) {
  __shared__ double red[2][16];
  double lc = 0.0, lb = 0.0;
  for (int r = threadIdx.x; r < R; r += 1024) { lc += row_terms[r]; lb += row_terms[R + r]; }
  for (int o = 32; o > 0; o >>= 1) { lc += __shfl_xor(lc, o); lb += __shfl_xor(lb, o); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = lc; red[1][threadIdx.x >> 6] = lb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, b = 0;
    for (int w = 0; w < 16; ++w) { a += red[0][w]; b += red[1][w]; }
    out_losses[0] = (float)(a / R);
    out_losses[1] = (float)(b / R);
  }
}

// logits [R, ld_cls] (K+1 used), deltas [R, ld_delta], proposals/gt_boxes [R,4], gt_classes [R] int64 (K = background)
// out_losses [2] = (loss_cls, loss_box_reg) ; dlogits [R,K+1], ddeltas [R, 4K | 4] = d(loss)/d(input), dense;
// row_terms [2R] fp64 scratch.
extern "C" int lvc_fast_rcnn_losses(const float* logits, int ld_cls, const float* deltas, int ld_delta, int K,
                                    int cls_agnostic, const float* proposals, const float* gt_boxes,
                                    const long long* gt_classes, int R, float wx, float wy, float ww, float wh,
                                    float smooth_l1_beta, float* out_losses, float* dlogits, float* ddeltas,
                                    double* row_terms, void* stream) {
  LVC_CHECK_ARG(R > 0 && K > 0, "empty batch");
  LVC_CHECK_ARG(logits && deltas && proposals && gt_boxes && gt_classes && out_losses && dlogits && ddeltas && row_terms, "null pointer");
  hipLaunchKernelGGL(fast_rcnn_losses_rows_kernel, dim3(lvc_cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, logits, ld_cls,
                     deltas, ld_delta, K, cls_agnostic, proposals, gt_boxes, gt_classes, R, wx, wy, ww, wh, smooth_l1_beta,
                     row_terms, dlogits, ddeltas);
  LVC_CHECK_LAUNCH();
  hipLaunchKernelGGL(fast_rcnn_losses_sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, row_terms, R, out_losses);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// RPN losses over S sampled anchors (labels in {0,1}); rows gathered by the host side index plumbing.
__global__ __launch_bounds__(1024) void rpn_losses_kernel(const float* __restrict__ logits, const float* __restrict__ deltas,
                                                          const float* __restrict__ anchors,
                                                          const float* __restrict__ gt_boxes,
                                                          const signed char* __restrict__ labels, int S, float beta,
                                                          float normalizer, float* __restrict__ out,
                                                          float* __restrict__ dlogits, float* __restrict__ ddeltas) {
  __shared__ double red[2][16];
  double lc = 0.0, lb = 0.0;
  for (int i = threadIdx.x; i < S; i += 1024) {
    const float x = logits[i], y = (float)labels[i];
    // F.binary_cross_entropy_with_logits: max(x,0) - x*y + log1p(exp(-|x|));  d/dx = sigmoid(x) - y
    lc += (double)(fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x))));
    if (dlogits) dlogits[i] = (1.f / (1.f + expf(-x)) - y) / normalizer;
    float gd[4] = {0.f, 0.f, 0.f, 0.f};
    if (labels[i] == 1) {
      float t[4], gr;
      const float* a = anchors + (size_t)i * 4;
      const float* g = gt_boxes + (size_t)i * 4;
      get_deltas(a[0], a[1], a[2], a[3], g[0], g[1], g[2], g[3], 1.f, 1.f, 1.f, 1.f, t);
      for (int j = 0; j < 4; ++j) {
        lb += (double)smooth_l1(deltas[(size_t)i * 4 + j], t[j], beta, &gr);
        gd[j] = gr / normalizer;
      }
    }
    if (ddeltas)
      for (int j = 0; j < 4; ++j) ddeltas[(size_t)i * 4 + j] = gd[j];
  }
  for (int o = 32; o > 0; o >>= 1) { lc += __shfl_xor(lc, o); lb += __shfl_xor(lb, o); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = lc; red[1][threadIdx.x >> 6] = lb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, b = 0;
    for (int w = 0; w < 16; ++w) { a += red[0][w]; b += red[1][w]; }
    out[0] = (float)(a / normalizer);
    out[1] = (float)(b / normalizer);
  }
}

extern "C" int lvc_rpn_losses(const float* logits, const float* deltas, const float* anchors, const float* gt_boxes,
                              const signed char* labels, int S, float smooth_l1_beta, float normalizer, float* out_losses,
                              void* stream) {
  LVC_CHECK_ARG(S >= 0 && normalizer > 0.f && out_losses, "bad arguments");
  hipLaunchKernelGGL(rpn_losses_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, deltas, anchors, gt_boxes,
                     labels, S, smooth_l1_beta, normalizer, out_losses, (float*)nullptr, (float*)nullptr);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// same losses plus d(loss_cls)/d(logits) [S] and d(loss_loc)/d(deltas) [S,4] (zero rows for non-positive anchors)
extern "C" int lvc_rpn_losses_grad(const float* logits, const float* deltas, const float* anchors, const float* gt_boxes,
                                   const signed char* labels, int S, float smooth_l1_beta, float normalizer,
                                   float* out_losses, float* dlogits, float* ddeltas, void* stream) {
  LVC_CHECK_ARG(S >= 0 && normalizer > 0.f && out_losses && dlogits && ddeltas, "bad arguments");
  hipLaunchKernelGGL(rpn_losses_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, deltas, anchors, gt_boxes,
                     labels, S, smooth_l1_beta, normalizer, out_losses, dlogits, ddeltas);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// =====================================================================================
// Box-corrector regression loss: BoxOnlyLayersCascade.box_reg_loss / BoxOnlyLayers.box_reg_loss
// (lvc/modeling/roi_heads/roi_heads_cascade.py:165-195) = Box2BoxTransform.apply_deltas (box_regression.py:73-110) on
// the foreground rows -> fvcore.nn.giou_loss (third-party, eps = 1e-7; restated in oracle/refshim.py) -> mean, or with
// `iterate` mean(max(loss_after - lambda * loss_before, 0)).  Emits d(loss)/d(deltas) [R,4] (zero on background rows)
// so that the backward of the op is one multiply.  An empty foreground set gives NaN, as .mean() of an empty tensor does.
// =====================================================================================
__device__ __forceinline__ float giou_terms(float x1, float y1, float x2, float y2, float gx1, float gy1, float gx2,
                                            float gy2, float* g /* dL/d(x1,y1,x2,y2) or nullptr */) {
  const float eps = 1e-7f;
  const float xk1 = fmaxf(x1, gx1), yk1 = fmaxf(y1, gy1), xk2 = fminf(x2, gx2), yk2 = fminf(y2, gy2);
  const bool m = (yk2 > yk1) && (xk2 > xk1);
  const float I = m ? (xk2 - xk1) * (yk2 - yk1) : 0.f;
  const float A = (x2 - x1) * (y2 - y1);
  const float U = A + (gx2 - gx1) * (gy2 - gy1) - I;
  const float iou = I / (U + eps);
  const float xc1 = fminf(x1, gx1), yc1 = fminf(y1, gy1), xc2 = fmaxf(x2, gx2), yc2 = fmaxf(y2, gy2);
  const float C = (xc2 - xc1) * (yc2 - yc1);
  const float L = 1.f - (iou - (C - U) / (C + eps));
  if (g) {
    // subgradient of max/min at ties = 1/2, as torch.max / torch.min (binary) backward
    auto gt_ = [](float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); };
    const float dI[4] = {m ? -gt_(x1, gx1) * (yk2 - yk1) : 0.f, m ? -gt_(y1, gy1) * (xk2 - xk1) : 0.f,
                         m ? gt_(gx2, x2) * (yk2 - yk1) : 0.f, m ? gt_(gy2, y2) * (xk2 - xk1) : 0.f};
    const float dA[4] = {-(y2 - y1), -(x2 - x1), (y2 - y1), (x2 - x1)};
    const float dC[4] = {-gt_(gx1, x1) * (yc2 - yc1), -gt_(gy1, y1) * (xc2 - xc1), gt_(x2, gx2) * (yc2 - yc1),
                         gt_(y2, gy2) * (xc2 - xc1)};
    for (int j = 0; j < 4; ++j) {
      const float dU = dA[j] - dI[j];
      const float diou = (dI[j] * (U + eps) - I * dU) / ((U + eps) * (U + eps));
      const float dterm = ((dC[j] - dU) * (C + eps) - (C - U) * dC[j]) / ((C + eps) * (C + eps));
      g[j] = -(diou - dterm);
    }
  }
  return L;
}

__global__ __launch_bounds__(1024) void giou_box_loss_kernel(const float* __restrict__ deltas, int ld_delta,
                                                             const float* __restrict__ proposals,
                                                             const float* __restrict__ gt_boxes,
                                                             const long long* __restrict__ gt_classes, int R, int K,
                                                             float wx, float wy, float ww, float wh, float scale_clamp,
                                                             int iterate, float lambda, float* __restrict__ out_loss,
                                                             float* __restrict__ ddeltas) {
  __shared__ double red[16];
  __shared__ int redn[16];
  __shared__ int s_nfg;
  double ls = 0.0;
  int nfg = 0;
  for (int r = threadIdx.x; r < R; r += 1024) {
    float* dd = ddeltas + (size_t)r * 4;
    dd[0] = dd[1] = dd[2] = dd[3] = 0.f;
    const long long c = gt_classes[r];
    if (!(c >= 0 && c < K)) continue;
    ++nfg;
    const float* p = proposals + (size_t)r * 4;
    const float* g = gt_boxes + (size_t)r * 4;
    const float* d = deltas + (size_t)r * ld_delta;
    const float w = p[2] - p[0], h = p[3] - p[1];
    const float cx = p[0] + 0.5f * w, cy = p[1] + 0.5f * h;
    const float dx = d[0] / wx, dy = d[1] / wy;
    const float dw0 = d[2] / ww, dh0 = d[3] / wh;
    const float dw = fminf(dw0, scale_clamp), dh = fminf(dh0, scale_clamp);
    const float pcx = dx * w + cx, pcy = dy * h + cy;
    const float pw = expf(dw) * w, ph = expf(dh) * h;
    const float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph, x2 = pcx + 0.5f * pw, y2 = pcy + 0.5f * ph;
    float gl[4];
    float La = giou_terms(x1, y1, x2, y2, g[0], g[1], g[2], g[3], gl);
    float pass = 1.f;
    if (iterate) {
      const float Lb = giou_terms(p[0], p[1], p[2], p[3], g[0], g[1], g[2], g[3], nullptr);
      const float diff = La - Lb * lambda;
      pass = diff > 0.f ? 1.f : (diff == 0.f ? 0.5f : 0.f);
      La = fmaxf(diff, 0.f);
    }
    ls += (double)La;
    dd[0] = pass * (gl[0] + gl[2]) * (w / wx);
    dd[1] = pass * (gl[1] + gl[3]) * (h / wy);
    dd[2] = pass * (0.5f * (gl[2] - gl[0])) * (dw0 <= scale_clamp ? pw / ww : 0.f);
    dd[3] = pass * (0.5f * (gl[3] - gl[1])) * (dh0 <= scale_clamp ? ph / wh : 0.f);
  }
  for (int o = 32; o > 0; o >>= 1) { ls += __shfl_xor(ls, o); nfg += __shfl_xor(nfg, o); }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = ls; redn[threadIdx.x >> 6] = nfg; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0;
    int n = 0;
    for (int w = 0; w < 16; ++w) { a += red[w]; n += redn[w]; }
    s_nfg = n;
    out_loss[0] = n > 0 ? (float)(a / n) : NAN;
  }
  __syncthreads();
  const float inv = s_nfg > 0 ? 1.f / (float)s_nfg : 0.f;
  for (int r = threadIdx.x; r < R; r += 1024)
    for (int j = 0; j < 4; ++j) ddeltas[(size_t)r * 4 + j] *= inv;
}

// deltas [R, ld_delta] (first 4 columns: class-agnostic regression), proposals / gt_boxes [R,4], gt_classes [R] int64
// (foreground = [0, K)); out_loss [1]; ddeltas [R,4] = d(loss)/d(deltas).
extern "C" int lvc_giou_box_loss(const float* deltas, int ld_delta, const float* proposals, const float* gt_boxes,
                                 const long long* gt_classes, int R, int K, float wx, float wy, float ww, float wh,
                                 float scale_clamp, int iterate, float lambda, float* out_loss, float* ddeltas,
                                 void* stream) {
  LVC_CHECK_ARG(R > 0 && K > 0 && ld_delta >= 4, "empty batch");
  LVC_CHECK_ARG(deltas && proposals && gt_boxes && gt_classes && out_loss && ddeltas, "null pointer");
  hipLaunchKernelGGL(giou_box_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, deltas, ld_delta, proposals,
                     gt_boxes, gt_classes, R, K, wx, wy, ww, wh, scale_clamp, iterate, lambda, out_loss, ddeltas);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// =====================================================================================
// Backward helpers of the fused Linear(+bias)(+ReLU) layers of the box head (FastRCNNConvFCHead, box_head.py:82-91):
// masked upstream gradient dz = dy * (y > 0) and the bias gradient = column sums of dz (fixed row order per column ->
// deterministic).  dX and dW are GEMMs on the conv/GEMM kernel.
// =====================================================================================
__global__ void relu_backward_kernel(const float* __restrict__ dy, const float* __restrict__ y, long long n,
                                     float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// four elements per lane: the trunk's ReLU masks are 12 B/element HBM streams (3.1 ms of the batch-8 training step as
// scalar loads)
__global__ __launch_bounds__(256) void relu_backward4_kernel(const float4* __restrict__ dy, const float4* __restrict__ y,
                                                             long long n4, float4* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 g = dy[i], v = y[i];
  out[i] = float4{v.x > 0.f ? g.x : 0.f, v.y > 0.f ? g.y : 0.f, v.z > 0.f ? g.z : 0.f, v.w > 0.f ? g.w : 0.f};
}

extern "C" int lvc_relu_backward(const float* dy, const float* y, long long n, float* out, void* stream) {
  LVC_CHECK_ARG(n >= 0 && (n == 0 || (dy && y && out)), "bad arguments");
  if (n == 0) return LVC_OK;
  if ((n & 3) == 0 && ((((uintptr_t)dy | (uintptr_t)y | (uintptr_t)out) & 15) == 0)) {
    hipLaunchKernelGGL(relu_backward4_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(dy), reinterpret_cast<const float4*>(y), n / 4,
                       reinterpret_cast<float4*>(out));
    LVC_CHECK_LAUNCH();
    return LVC_OK;
  }
  hipLaunchKernelGGL(relu_backward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, y, n, out);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

__global__ void colsum_kernel(const float* __restrict__ x, int M, int N, int ldx, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  float s = 0.f;
  for (int r = 0; r < M; ++r) s += x[(size_t)r * ldx + c];
  out[c] = s;
}

extern "C" int lvc_colsum(const float* x, int M, int N, int ldx, float* out, void* stream) {
  LVC_CHECK_ARG(M >= 0 && N > 0 && ldx >= N && x && out, "bad arguments");
  hipLaunchKernelGGL(colsum_kernel, dim3(lvc_cdiv(N, 64)), dim3(64), 0, (hipStream_t)stream, x, M, N, ldx, out);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}
