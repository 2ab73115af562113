// boxes.hip -- the box pipeline between the conv trunk and the heads, entirely on device, no host sync:
//   * lvc_rpn_proposals          = RPN.predict_proposals + find_top_rpn_proposals
//       reference detectron2/modeling/proposal_generator/rpn.py:455-508,
//                 detectron2/modeling/proposal_generator/proposal_utils.py:13-118,
//                 detectron2/modeling/anchor_generator.py:157-178 (grid anchors = shift + cell anchor),
//                 detectron2/modeling/box_regression.py:73-110 (apply_deltas)
//   * lvc_assign_levels_rois     = assign_boxes_to_levels + convert_boxes_to_pooler_format
//       reference detectron2/modeling/poolers.py:23-59, 69-96
//   * lvc_fast_rcnn_inference    = FastRCNNOutputs.predict_boxes/predict_probs + fast_rcnn_inference
//       (+ optional detector_postprocess)
//       reference lvc/modeling/roi_heads/fast_rcnn.py:95-137, 440-468; detectron2/modeling/postprocessing.py:10-79
// Built with -ffp-contract=off: each fp32 operation of the decode/clip chain rounds once, in the
// reference's association, so boxes entering NMS equal the CPU path's up to the exp()/log2()
// library difference (<= 1 ulp).  All orderings that the reference defines by position
// (level-major candidate order, row-major (roi, class) order, score-sorted keep) are reproduced by
// ORDERED compaction (block scan), never by atomics.
#include "common.h"
#include <stdlib.h>

typedef unsigned long long u64;

extern "C" int lvc_batched_nms(const float*, const float*, const int*, const int*, int, int, double, int,
                               int*, int*, void*, long long, void*);
extern "C" long long lvc_batched_nms_workspace_bytes(int, int);

#define MAXL 8

__device__ __forceinline__ unsigned int desc_key(float f) {
  if (f == 0.f) f = 0.f;
  unsigned int u = __float_as_uint(f);
  u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
  return ~u;
}

// block-wide exclusive scan of one int per thread (1024 threads = 16 waves); returns exclusive prefix,
// *total gets the block sum.  `sh` must hold >= 17 ints.
__device__ __forceinline__ int block_excl_scan_1024(int v, int* sh, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(x, o);
    if (lane >= o) x += t;
  }
  __syncthreads();  // protect sh from a previous use
  if (lane == 63) sh[wave] = x;
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int w = 0; w < 16; ++w) { int t = sh[w]; sh[w] = s; s += t; }
    sh[16] = s;
  }
  __syncthreads();
  *total = sh[16];
  return sh[wave] + x - v;
}

template <typename T>
__device__ __forceinline__ void bitonic_sort_lds(T* keys, int npad) {
  for (int k = 2; k <= npad; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < npad / 2; t += blockDim.x) {
        int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        int hi = lo | j;
        bool up = (lo & k) == 0;
        T a = keys[lo], b = keys[hi];
        if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
      }
      __syncthreads();
    }
  }
}

// =====================================================================================
// RPN step 1: per (image, level) top-k of the objectness logits, sorted descending (ties: lower index).
// Radix select (4 x 8 bit) on the order-preserving key, ordered take of the threshold ties, bitonic sort.
// =====================================================================================
struct RpnLevels {
  const float* logits[MAXL];  // [B, HW, ld_logit] : logit of anchor a at pixel p = base[(b*HW+p)*ld + a]
  const float* deltas[MAXL];  // [B, HW, ld_delta] : delta c of anchor a = base[(b*HW+p)*ld + a*4 + c]
  int ld_logit[MAXL], ld_delta[MAXL];
  int H[MAXL], W[MAXL], stride[MAXL];
  int cand_off[MAXL + 1];     // prefix of min(topk, H*W*A)
  long long key_off[MAXL + 1];// prefix of H*W*A (workspace offsets, per image)
  const float* cell_anchors[MAXL];  // [A,4]
  int L, A;
};

// first 256 threads: digit d with  sum(h[0..d-1]) < krem <= sum(h[0..d]);  returns through sh[8] = d, sh[9] = sum(h[0..d-1])
__device__ __forceinline__ void find_digit(const int* __restrict__ h, int krem, int* sh) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int x = 0, incl = 0;
  if (tid < 256) {
    x = h[tid];
    incl = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane == 63) sh[wave] = incl;
  }
  __syncthreads();
  if (tid < 256) {
    int off = 0;
    for (int w = 0; w < wave; ++w) off += sh[w];
    incl += off;
    if (incl - x < krem && krem <= incl) { sh[8] = tid; sh[9] = incl - x; }
  }
  __syncthreads();
}

#define TOPK_PAD 2048
__global__ __launch_bounds__(1024) void rpn_topk_kernel(RpnLevels lv, int topk, unsigned int* __restrict__ wkeys,
                                                        float* __restrict__ cand_score,
                                                        int* __restrict__ cand_idx, int Ntot, int small_only) {
  __shared__ u64 sortbuf[TOPK_PAD];
  __shared__ int hist[256];
  __shared__ int sh[20];
  __shared__ int s_nlt;
  const int l = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int A = lv.A, HW = lv.H[l] * lv.W[l], n = HW * A;
  if (n > small_only) return;   // handled by the multi-workgroup phases
  const int k = topk < n ? topk : n;
  const int ld = lv.ld_logit[l];
  const float* lg = lv.logits[l] + (size_t)b * HW * ld;
  unsigned int* keys = wkeys + (size_t)b * lv.key_off[lv.L] + lv.key_off[l];
  const int npad = k <= 1024 ? 1024 : TOPK_PAD;

  if (tid < 256) hist[tid] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += 1024) {
    int p = i / A, a = i - p * A;
    unsigned int key = desc_key(lg[(size_t)p * ld + a]);
    keys[i] = key;
    atomicAdd(&hist[key >> 24], 1);
  }
  __syncthreads();
  unsigned int prefix = 0, pmask = 0;
  int krem = k;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (pass > 0) {
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      for (int i = tid; i < n; i += 1024) {
        unsigned int key = keys[i];
        if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1);
      }
      __syncthreads();
    }
    find_digit(hist, krem, sh);   // parallel scan of the 256 bins (a serial scan by one thread cost ~12 us per pass)
    prefix |= (unsigned int)sh[8] << shift;
    pmask |= 255u << shift;
    krem -= sh[9];
    __syncthreads();
  }
  // prefix = threshold key T; take every key < T, and the first `krem` (by index) with key == T
  const unsigned int T = prefix;
  const int need_eq = krem, n_lt = k - krem;
  for (int i = tid; i < npad; i += 1024) sortbuf[i] = ~0ull;
  if (tid == 0) s_nlt = 0;
  __syncthreads();
  int eq_base = 0;
  for (int i0 = 0; i0 < n; i0 += 1024) {
    const int i = i0 + tid;
    unsigned int key = i < n ? keys[i] : 0xFFFFFFFFu;
    bool is_lt = i < n && key < T;
    bool is_eq = i < n && key == T;
    if (is_lt) {
      int slot = atomicAdd(&s_nlt, 1);
      sortbuf[slot] = ((u64)key << 32) | (unsigned)i;
    }
    if (eq_base < need_eq) {  // uniform
      int tot;
      int rank = eq_base + block_excl_scan_1024(is_eq ? 1 : 0, sh, &tot);
      if (is_eq && rank < need_eq) sortbuf[n_lt + rank] = ((u64)key << 32) | (unsigned)i;
      eq_base += tot;
    }
  }
  __syncthreads();
  bitonic_sort_lds(sortbuf, npad);
  for (int r = tid; r < k; r += 1024) {
    int i = (int)(sortbuf[r] & 0xFFFFFFFFu);
    int p = i / A, a = i - p * A;
    cand_idx[(size_t)b * Ntot + lv.cand_off[l] + r] = i;
    cand_score[(size_t)b * Ntot + lv.cand_off[l] + r] = lg[(size_t)p * ld + a];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Multi-workgroup form of the same selection for the large levels (p2 has 201 600 anchors per image: one workgroup per
// (image, level) sweeping them six times was 0.43 ms per batch on 40 workgroups).  Same radix select, same tie rule,
// same output; the sweeps are spread over slices of TK_SLICE anchors:
//   phase 0..3  every slice adds its 256-bin digit histogram (of the keys matching the prefix found so far) to the
//               level's global histogram; phase 0 also materialises the keys
//   phase 4     threshold T known: keys < T are appended (atomic slot, order irrelevant: they are sorted afterwards),
//               the indices of keys == T go to a tie list (first TOPK_PAD of them)
//   final       one workgroup per (image, level): ties ordered by index (sorted tie list, or -- if there were more
//               than TOPK_PAD -- the ordered scan of the single-workgroup kernel), bitonic sort, output
// Integer atomics only: every count, and therefore the result, is deterministic.
#define TK_SLICE 8192
#define TK_HSTRIDE 1040          // ints per (image, level): 4 x 256 histogram bins, [1024] = #(key < T), [1025] = #(key == T)

// prefix / remaining count after `npass` digit passes (uniform over the workgroup)
__device__ __forceinline__ void topk_prefix(const int* __restrict__ H, int npass, int k, int* sh, unsigned int* prefix,
                                            int* krem) {
  unsigned int pf = 0;
  int kr = k;
  for (int ps = 0; ps < npass; ++ps) {
    find_digit(H + ps * 256, kr, sh);
    pf |= (unsigned int)sh[8] << (24 - 8 * ps);
    kr -= sh[9];
    __syncthreads();
  }
  *prefix = pf;
  *krem = kr;
}

__global__ __launch_bounds__(1024) void rpn_topk_phase_kernel(RpnLevels lv, int topk, unsigned int* __restrict__ wkeys,
                                                              int* __restrict__ hist_all, u64* __restrict__ ckeys,
                                                              int* __restrict__ ties, int Ntot, int phase) {
  __shared__ int hist[256];
  __shared__ int sh[16];
  const int slice = blockIdx.x, l = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, lane = tid & 63;
  const int A = lv.A, HW = lv.H[l] * lv.W[l], n = HW * A;
  if (n <= TK_SLICE) return;                 // small levels: rpn_topk_kernel
  const int i0 = slice * TK_SLICE;
  if (i0 >= n) return;
  const int i1 = min(n, i0 + TK_SLICE);
  const int k = topk < n ? topk : n;
  const int ld = lv.ld_logit[l];
  const float* lg = lv.logits[l] + (size_t)b * HW * ld;
  unsigned int* keys = wkeys + (size_t)b * lv.key_off[lv.L] + lv.key_off[l];
  int* H = hist_all + ((size_t)b * lv.L + l) * TK_HSTRIDE;
  unsigned int prefix;
  int krem;
  topk_prefix(H, phase < 4 ? phase : 4, k, sh, &prefix, &krem);
  if (phase < 4) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned int pmask = phase ? 0xFFFFFFFFu << (32 - 8 * phase) : 0u;
    const int shift = 24 - 8 * phase;
    for (int ib = i0; ib < i1; ib += 1024) {
      const int i = ib + tid;
      unsigned int key = 0;
      bool act = false;
      if (i < i1) {
        if (phase == 0) {
          const int pp = i / A, a = i - pp * A;
          key = desc_key(lg[(size_t)pp * ld + a]);
          keys[i] = key;
        } else {
          key = keys[i];
        }
        act = (key & pmask) == prefix;
      }
      // wave-aggregated LDS histogram: objectness logits cluster in a few exponent bins, so a plain atomicAdd per
      // lane would serialise most of the wave on one address
      const unsigned int d = (key >> shift) & 255u;
      u64 m = __ballot(act);
      while (m) {
        const int first = __ffsll((long long)m) - 1;
        const unsigned int dd = (unsigned int)__shfl((int)d, first);
        const u64 same = __ballot(act && d == dd);
        if (lane == first) atomicAdd(&hist[dd], __popcll(same));
        m &= ~same;
      }
    }
    __syncthreads();
    if (tid < 256 && hist[tid]) atomicAdd(&H[phase * 256 + tid], hist[tid]);
  } else {
    const unsigned int T = prefix;
    u64* ck = ckeys + (size_t)b * Ntot + lv.cand_off[l];
    int* tl = ties + ((size_t)b * lv.L + l) * TOPK_PAD;
    for (int i = i0 + tid; i < i1; i += 1024) {
      const unsigned int key = keys[i];
      if (key < T) {
        const int slot = atomicAdd(&H[1024], 1);
        ck[slot] = ((u64)key << 32) | (unsigned)i;
      } else if (key == T) {
        const int slot = atomicAdd(&H[1025], 1);
        if (slot < TOPK_PAD) tl[slot] = i;
      }
    }
  }
}

__global__ __launch_bounds__(1024) void rpn_topk_final_kernel(RpnLevels lv, int topk, const unsigned int* __restrict__ wkeys,
                                                              const int* __restrict__ hist_all,
                                                              const u64* __restrict__ ckeys, const int* __restrict__ ties,
                                                              float* __restrict__ cand_score, int* __restrict__ cand_idx,
                                                              int Ntot) {
  __shared__ u64 sortbuf[TOPK_PAD];
  __shared__ unsigned int tiebuf[TOPK_PAD];
  __shared__ int sh[20];
  const int l = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int A = lv.A, HW = lv.H[l] * lv.W[l], n = HW * A;
  if (n <= TK_SLICE) return;
  const int k = topk < n ? topk : n;
  const int ld = lv.ld_logit[l];
  const float* lg = lv.logits[l] + (size_t)b * HW * ld;
  const unsigned int* keys = wkeys + (size_t)b * lv.key_off[lv.L] + lv.key_off[l];
  const int* H = hist_all + ((size_t)b * lv.L + l) * TK_HSTRIDE;
  unsigned int T;
  int need_eq;
  topk_prefix(H, 4, k, sh, &T, &need_eq);
  const int n_lt = k - need_eq;
  const int npad = k <= 1024 ? 1024 : TOPK_PAD;
  const u64* ck = ckeys + (size_t)b * Ntot + lv.cand_off[l];
  for (int i = tid; i < npad; i += 1024) sortbuf[i] = i < n_lt ? ck[i] : ~0ull;
  const int tie_total = H[1025];
  __syncthreads();
  if (tie_total <= TOPK_PAD) {
    // ties at the threshold: the `need_eq` lowest indices
    const int* tl = ties + ((size_t)b * lv.L + l) * TOPK_PAD;
    int tp = 64;
    while (tp < tie_total) tp <<= 1;
    for (int i = tid; i < tp; i += 1024) tiebuf[i] = i < tie_total ? (unsigned int)tl[i] : 0xFFFFFFFFu;
    __syncthreads();
    bitonic_sort_lds(tiebuf, tp);
    for (int r = tid; r < need_eq; r += 1024) sortbuf[n_lt + r] = ((u64)T << 32) | tiebuf[r];
  } else {
    // more equal keys than the tie list holds (e.g. constant logits): ordered scan, stops once need_eq are found
    int eq_base = 0;
    for (int i0 = 0; i0 < n && eq_base < need_eq; i0 += 1024) {
      const int i = i0 + tid;
      const bool is_eq = i < n && keys[i] == T;
      int tot;
      const int rank = eq_base + block_excl_scan_1024(is_eq ? 1 : 0, sh, &tot);
      if (is_eq && rank < need_eq) sortbuf[n_lt + rank] = ((u64)T << 32) | (unsigned)i;
      eq_base += tot;
    }
  }
  __syncthreads();
  bitonic_sort_lds(sortbuf, npad);
  for (int r = tid; r < k; r += 1024) {
    const int i = (int)(sortbuf[r] & 0xFFFFFFFFu);
    const int pp = i / A, a = i - pp * A;
    cand_idx[(size_t)b * Ntot + lv.cand_off[l] + r] = i;
    cand_score[(size_t)b * Ntot + lv.cand_off[l] + r] = lg[(size_t)pp * ld + a];
  }
}

// =====================================================================================
// shared decode (Box2BoxTransform.apply_deltas, box_regression.py:73-110)
// =====================================================================================
__device__ __forceinline__ void apply_deltas(float bx1, float by1, float bx2, float by2, float d0, float d1,
                                             float d2, float d3, float wx, float wy, float ww, float wh,
                                             float scale_clamp, float* o) {
  const float widths = bx2 - bx1, heights = by2 - by1;
  const float ctr_x = bx1 + 0.5f * widths, ctr_y = by1 + 0.5f * heights;
  const float dx = d0 / wx, dy = d1 / wy;
  float dw = d2 / ww, dh = d3 / wh;
  dw = dw > scale_clamp ? scale_clamp : dw;  // torch.clamp(max=): NaN stays NaN
  dh = dh > scale_clamp ? scale_clamp : dh;
  const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;
  const float pw = expf(dw) * widths, ph = expf(dh) * heights;
  o[0] = pcx - 0.5f * pw; o[1] = pcy - 0.5f * ph; o[2] = pcx + 0.5f * pw; o[3] = pcy + 0.5f * ph;
}
__device__ __forceinline__ float clampf(float v, float lo, float hi) {  // torch clamp_(min,max)
  v = v < lo ? lo : v;
  return v > hi ? hi : v;
}


// RPN steps 2-4, per-level form.  `batched_nms` with the level as group id (proposal_utils.py:104) never lets boxes of
// different levels suppress each other, so NMS is L independent problems per image: one segment per (image, level) --
// B*L chains of <= pre_nms_topk / 64 chunks instead of B chains over the concatenation (40 x 16 instead of 8 x 76 chunks,
// and 136 instead of 2 926 mask blocks per image), then a merge of the L kept lists by (score desc, level asc, position
// asc) = exactly the order the concatenated formulation keeps, first post_nms_topk.
__global__ __launch_bounds__(1024) void rpn_decode_seg_kernel(RpnLevels lv, const float* __restrict__ cand_score,
                                                              const int* __restrict__ cand_idx, int Ntot,
                                                              const int* __restrict__ image_sizes, float scale_clamp,
                                                              float min_box_size, int SN, float* __restrict__ sboxes,
                                                              float* __restrict__ sscores, int* __restrict__ scount) {
  __shared__ int sh[20];
  const int l = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int seg = b * lv.L + l;
  const float img_h = (float)image_sizes[b * 2 + 0], img_w = (float)image_sizes[b * 2 + 1];
  const int c_begin = lv.cand_off[l], c_end = lv.cand_off[l + 1];
  const int A = lv.A, W = lv.W[l], HW = lv.H[l] * W;
  int base = 0;
  for (int c0 = c_begin; c0 < c_end; c0 += 1024) {
    const int c = c0 + tid;
    bool ok = false;
    float box[4] = {0, 0, 0, 0};
    float score = 0.f;
    if (c < c_end) {
      const int i = cand_idx[(size_t)b * Ntot + c];
      score = cand_score[(size_t)b * Ntot + c];
      const int p = i / A, a = i - p * A;
      const int y = p / W, x = p - y * W;
      const float sx = (float)(x * lv.stride[l]), sy = (float)(y * lv.stride[l]);
      const float* ca = lv.cell_anchors[l] + a * 4;
      const float ax1 = sx + ca[0], ay1 = sy + ca[1], ax2 = sx + ca[2], ay2 = sy + ca[3];
      const float* d = lv.deltas[l] + ((size_t)b * HW + p) * lv.ld_delta[l] + a * 4;
      apply_deltas(ax1, ay1, ax2, ay2, d[0], d[1], d[2], d[3], 1.f, 1.f, 1.f, 1.f, scale_clamp, box);
      ok = isfinite(box[0]) && isfinite(box[1]) && isfinite(box[2]) && isfinite(box[3]) && isfinite(score);
      box[0] = clampf(box[0], 0.f, img_w); box[1] = clampf(box[1], 0.f, img_h);
      box[2] = clampf(box[2], 0.f, img_w); box[3] = clampf(box[3], 0.f, img_h);
      ok = ok && (box[2] - box[0] > min_box_size) && (box[3] - box[1] > min_box_size);
    }
    int tot;
    const int pos = base + block_excl_scan_1024(ok ? 1 : 0, sh, &tot);
    if (ok) {
      float* o = sboxes + ((size_t)seg * SN + pos) * 4;
      o[0] = box[0]; o[1] = box[1]; o[2] = box[2]; o[3] = box[3];
      sscores[(size_t)seg * SN + pos] = score;
    }
    base += tot;
  }
  if (tid == 0) scount[seg] = base;
}

#define MERGE_LDS_SCORES 8192
__global__ __launch_bounds__(1024) void rpn_merge_levels_kernel(const float* __restrict__ sboxes,
                                                                const float* __restrict__ sscores,
                                                                const int* __restrict__ keep,
                                                                const int* __restrict__ num_keep, int L, int SN,
                                                                int post_topk, float* __restrict__ pboxes,
                                                                float* __restrict__ plogits, int* __restrict__ out_count) {
  __shared__ int s_nk[MAXL + 1];            // prefix of the kept counts
  __shared__ float s_sc[MERGE_LDS_SCORES];  // kept scores of all levels, list after list (when they fit)
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) {
    int acc = 0;
    for (int l = 0; l < L; ++l) { s_nk[l] = acc; acc += num_keep[b * L + l]; }
    s_nk[L] = acc;
  }
  __syncthreads();
  const int total = s_nk[L];
  const int nout = total < post_topk ? total : post_topk;
  const bool in_lds = total <= MERGE_LDS_SCORES;
  if (in_lds) {
    for (int e = tid; e < total; e += 1024) {
      int l = 0;
      while (e >= s_nk[l + 1]) ++l;
      const int seg = b * L + l;
      s_sc[e] = sscores[(size_t)seg * SN + keep[(size_t)seg * SN + (e - s_nk[l])]];
    }
    __syncthreads();
  }
  for (int e = tid; e < total; e += 1024) {
    int l = 0;
    while (e >= s_nk[l + 1]) ++l;
    const int t = e - s_nk[l];
    const int seg = b * L + l;
    const int j = keep[(size_t)seg * SN + t];
    const float s = sscores[(size_t)seg * SN + j];
    int rank = t;
    for (int m = 0; m < L; ++m) {
      if (m == l) continue;
      const int segm = b * L + m, n = s_nk[m + 1] - s_nk[m];
      int lo = 0, hi = n;   // first u whose element does NOT precede (s, l)
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const float sm = in_lds ? s_sc[s_nk[m] + mid] : sscores[(size_t)segm * SN + keep[(size_t)segm * SN + mid]];
        if (sm > s || (sm == s && m < l)) lo = mid + 1; else hi = mid;
      }
      rank += lo;
    }
    if (rank < post_topk) {
      *reinterpret_cast<float4*>(pboxes + ((size_t)b * post_topk + rank) * 4) =
          *reinterpret_cast<const float4*>(sboxes + ((size_t)seg * SN + j) * 4);
      plogits[(size_t)b * post_topk + rank] = s;
    }
  }
  for (int r = nout + tid; r < post_topk; r += 1024) {   // rows past count = 0
    *reinterpret_cast<float4*>(pboxes + ((size_t)b * post_topk + r) * 4) = float4{0.f, 0.f, 0.f, 0.f};
    plogits[(size_t)b * post_topk + r] = 0.f;
  }
  if (tid == 0) out_count[b] = nout;
}


static long long align16(long long x) { return (x + 15) & ~15ll; }

struct RpnPlan {
  int Ntot;
  long long nkeys;  // per image
  long long off_keys, off_cscore, off_cidx, off_cboxes, off_cscores2, off_clevels, off_ccount, off_keep, off_hist, off_ckeys, off_ties, off_nms, total;
  long long off_sboxes, off_sscores, off_scount, off_skeep, off_snk, off_snms, snms_bytes;   // per-(image, level) segments
  int max_slices, SN;
};
static RpnPlan rpn_plan(int B, int L, int A, const int* Hs, const int* Ws, int pre_topk) {
  RpnPlan p;
  p.Ntot = 0; p.nkeys = 0; p.max_slices = 0;
  for (int l = 0; l < L; ++l) {
    long long n = (long long)Hs[l] * Ws[l] * A;
    p.nkeys += n;
    p.Ntot += (int)(n < pre_topk ? n : pre_topk);
    if (n > TK_SLICE) { int sl = (int)((n + TK_SLICE - 1) / TK_SLICE); if (sl > p.max_slices) p.max_slices = sl; }
  }
  long long o = 0;
  p.off_keys = o; o = align16(o + (long long)B * p.nkeys * 4);
  p.off_cscore = o; o = align16(o + (long long)B * p.Ntot * 4);
  p.off_cidx = o; o = align16(o + (long long)B * p.Ntot * 4);
  p.off_cboxes = o; o = align16(o + (long long)B * p.Ntot * 16);
  p.off_cscores2 = o; o = align16(o + (long long)B * p.Ntot * 4);
  p.off_clevels = o; o = align16(o + (long long)B * p.Ntot * 4);
  p.off_ccount = o; o = align16(o + (long long)B * 4);
  p.off_keep = o; o = align16(o + (long long)B * p.Ntot * 4);
  p.off_hist = o; o = align16(o + (long long)B * L * TK_HSTRIDE * 4);
  p.off_ckeys = o; o = align16(o + (long long)B * p.Ntot * 8);
  p.off_ties = o; o = align16(o + (long long)B * L * TOPK_PAD * 4);
  p.off_nms = o; o = align16(o + lvc_batched_nms_workspace_bytes(B, p.Ntot));
  p.SN = 0;
  for (int l = 0; l < L; ++l) {
    long long n = (long long)Hs[l] * Ws[l] * A;
    const int c = (int)(n < pre_topk ? n : pre_topk);
    if (c > p.SN) p.SN = c;
  }
  const long long segs = (long long)B * L;
  p.off_sboxes = o; o = align16(o + segs * p.SN * 16);
  p.off_sscores = o; o = align16(o + segs * p.SN * 4);
  p.off_scount = o; o = align16(o + segs * 4);
  p.off_skeep = o; o = align16(o + segs * p.SN * 4);
  p.off_snk = o; o = align16(o + segs * 4);
  p.snms_bytes = lvc_batched_nms_workspace_bytes((int)segs, p.SN);
  p.off_snms = o; o = align16(o + p.snms_bytes);
  p.total = o;
  return p;
}

extern "C" long long lvc_rpn_proposals_workspace_bytes(int B, int L, int A, const int* Hs, const int* Ws,
                                                       int pre_nms_topk) {
  if (L < 1 || L > MAXL) return -1;
  return rpn_plan(B, L, A, Hs, Ws, pre_nms_topk).total;
}

extern "C" int lvc_rpn_proposals(const float* const* logits, const int* ld_logit, const float* const* deltas,
                                 const int* ld_delta, const float* const* cell_anchors, const int* Hs,
                                 const int* Ws, const int* strides, int L, int A, int B,
                                 const int* d_image_sizes, int pre_nms_topk, int post_nms_topk,
                                 double nms_thresh, float min_box_size, float scale_clamp,
                                 float* out_boxes, float* out_logits, int* d_out_count, void* workspace,
                                 long long workspace_bytes, void* stream) {
  LVC_CHECK_ARG(L >= 1 && L <= MAXL, "1..8 levels");
  LVC_CHECK_ARG(B > 0 && A > 0, "bad B/A");
  LVC_CHECK_ARG(pre_nms_topk > 0 && pre_nms_topk <= TOPK_PAD, "pre_nms_topk must be in 1..2048");
  LVC_CHECK_ARG(post_nms_topk > 0, "post_nms_topk must be positive");
  LVC_CHECK_ARG(logits && deltas && cell_anchors && Hs && Ws && strides && d_image_sizes && out_boxes &&
                    out_logits && d_out_count && workspace, "null pointer");
  RpnPlan p = rpn_plan(B, L, A, Hs, Ws, pre_nms_topk);
  LVC_CHECK_ARG(workspace_bytes >= p.total, "workspace too small");
  RpnLevels lv;
  memset(&lv, 0, sizeof lv);
  lv.L = L; lv.A = A;
  for (int l = 0; l < L; ++l) {
    lv.logits[l] = logits[l]; lv.deltas[l] = deltas[l];
    lv.ld_logit[l] = ld_logit[l]; lv.ld_delta[l] = ld_delta[l];
    lv.H[l] = Hs[l]; lv.W[l] = Ws[l]; lv.stride[l] = strides[l];
    lv.cell_anchors[l] = cell_anchors[l];
    long long n = (long long)Hs[l] * Ws[l] * A;
    lv.key_off[l + 1] = lv.key_off[l] + n;
    lv.cand_off[l + 1] = lv.cand_off[l] + (int)(n < pre_nms_topk ? n : pre_nms_topk);
  }
  char* ws = (char*)workspace;
  hipStream_t st = (hipStream_t)stream;
  unsigned int* keys = (unsigned int*)(ws + p.off_keys);
  float* cand_score = (float*)(ws + p.off_cscore);
  int* cand_idx = (int*)(ws + p.off_cidx);
  const bool multi = p.max_slices > 0;
  if (multi) {
    int* hist = (int*)(ws + p.off_hist);
    u64* ckeys = (u64*)(ws + p.off_ckeys);
    int* ties = (int*)(ws + p.off_ties);
    if (hipMemsetAsync(hist, 0, (size_t)B * L * TK_HSTRIDE * 4, st) != hipSuccess) {
      lvc_set_error("%s: hipMemsetAsync failed", __func__);
      return LVC_ERR_HIP;
    }
    for (int phase = 0; phase <= 4; ++phase)
      hipLaunchKernelGGL(rpn_topk_phase_kernel, dim3(p.max_slices, L, B), dim3(1024), 0, st, lv, pre_nms_topk, keys,
                         hist, ckeys, ties, p.Ntot, phase);
    hipLaunchKernelGGL(rpn_topk_final_kernel, dim3(L, B), dim3(1024), 0, st, lv, pre_nms_topk, keys, hist, ckeys, ties,
                       cand_score, cand_idx, p.Ntot);
  }
  hipLaunchKernelGGL(rpn_topk_kernel, dim3(L, B), dim3(1024), 0, st, lv, pre_nms_topk, keys, cand_score,
                     cand_idx, p.Ntot, multi ? TK_SLICE : 0x7FFFFFFF);
  LVC_CHECK_LAUNCH();
  // NMS per (image, level) segment -- levels never interact under `batched_nms` -- and a rank merge of the kept lists in the
  // order the concatenated form keeps them (136 instead of 2 926 mask blocks per image; the concatenated form lost the A/B in
  // round 1 and was removed in round 4)
  float* sboxes = (float*)(ws + p.off_sboxes);
  float* sscores = (float*)(ws + p.off_sscores);
  int* scount = (int*)(ws + p.off_scount);
  int* skeep = (int*)(ws + p.off_skeep);
  int* snk = (int*)(ws + p.off_snk);
  hipLaunchKernelGGL(rpn_decode_seg_kernel, dim3(L, B), dim3(1024), 0, st, lv, cand_score, cand_idx, p.Ntot,
                     d_image_sizes, scale_clamp, min_box_size, p.SN, sboxes, sscores, scount);
  LVC_CHECK_LAUNCH();
  const int keep_per_level = post_nms_topk < p.SN ? post_nms_topk : p.SN;
  int rc = lvc_batched_nms(sboxes, sscores, nullptr, scount, B * L, p.SN, nms_thresh, keep_per_level, skeep, snk,
                           ws + p.off_snms, p.snms_bytes, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(rpn_merge_levels_kernel, dim3(B), dim3(1024), 0, st, sboxes, sscores, skeep, snk, L, p.SN,
                     post_nms_topk, out_boxes, out_logits, d_out_count);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// =====================================================================================
// FPN level assignment + pooler-format rois
// =====================================================================================
__global__ void assign_levels_kernel(const float* __restrict__ boxes, int R, int B, int min_level, int max_level,
                                     float canonical_box_size, float canonical_level, int* __restrict__ levels,
                                     float* __restrict__ rois) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * B) return;
  const float4 bx = *reinterpret_cast<const float4*>(boxes + (size_t)i * 4);
  const float area = (bx.z - bx.x) * (bx.w - bx.y);
  const float size = sqrtf(area);
  float lvl = floorf(canonical_level + log2f(size / canonical_box_size + 1e-8f));
  lvl = lvl < (float)min_level ? (float)min_level : lvl;
  lvl = lvl > (float)max_level ? (float)max_level : lvl;
  levels[i] = (int)lvl - min_level;
  if (rois) {
    float* r = rois + (size_t)i * 5;
    r[0] = (float)(i / R); r[1] = bx.x; r[2] = bx.y; r[3] = bx.z; r[4] = bx.w;
  }
}

// boxes [B,R,4] -> levels [B*R] int32 (offset from min_level) and rois [B*R,5] (batch index = image)
extern "C" int lvc_assign_levels_rois(const float* boxes, int B, int R, int min_level, int max_level,
                                      int canonical_box_size, int canonical_level, int* levels, float* rois,
                                      void* stream) {
  LVC_CHECK_ARG(B >= 0 && R >= 0, "negative size");
  if (B * R == 0) return LVC_OK;
  LVC_CHECK_ARG(boxes && levels, "null pointer");
  hipLaunchKernelGGL(assign_levels_kernel, dim3(lvc_cdiv(B * R, 256)), dim3(256), 0, (hipStream_t)stream, boxes, R,
                     B, min_level, max_level, (float)canonical_box_size, (float)canonical_level, levels, rois);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// =====================================================================================
// Fast R-CNN inference: softmax, per-class decode, clip, threshold -> candidates (ordered) -> NMS -> top-k
// =====================================================================================
// Row statistics of the softmax (max, 1 / sum, the classes above the threshold as a bit mask) for every RoI row, 64 rows per
// workgroup: the logits tile is read coalesced into LDS and every thread then walks ITS row in class order -- the same
// operations in the same order as the one-thread-per-row loops of det_candidates_kernel (bit-identical scores), which
// read 3 x (K+1) strided words per thread from HBM with eight workgroups on the whole chip (0.19 ms of the step).
#define DET_TILE_ROWS 64
#define DET_MAX_COLS 97
__global__ __launch_bounds__(DET_TILE_ROWS) void det_row_stats_kernel(const float* __restrict__ cls_logits, int ld_cls, int K,
                                                                     int rows, float score_thresh, float* __restrict__ st_mx,
                                                                     float* __restrict__ st_inv, int* __restrict__ st_cnt,
                                                                     unsigned long long* __restrict__ st_mask) {
  __shared__ float tile[DET_TILE_ROWS * DET_MAX_COLS];
  const int r0 = blockIdx.x * DET_TILE_ROWS, tid = threadIdx.x;
  const int ncol = K + 1, pitch = ncol | 1;    // odd pitch: the per-thread row walks hit 64 different banks
  for (int i = tid; i < DET_TILE_ROWS * ncol; i += DET_TILE_ROWS) {
    const int rr = i / ncol, cc = i - rr * ncol;
    const int r = r0 + rr;
    tile[rr * pitch + cc] = r < rows ? cls_logits[(size_t)r * ld_cls + cc] : 0.f;
  }
  __syncthreads();
  const int r = r0 + tid;
  if (r >= rows) return;
  const float* lg = tile + tid * pitch;
  float mx = -INFINITY;
  for (int k = 0; k <= K; ++k) { float v = lg[k]; mx = v > mx ? v : mx; }
  float sum = 0.f;
  for (int k = 0; k <= K; ++k) sum += expf(lg[k] - mx);
  const float inv = 1.f / sum;
  int cnt = 0;
  unsigned long long m0 = 0, m1 = 0;
  for (int k = 0; k < K; ++k) {
    const bool hit = expf(lg[k] - mx) * inv > score_thresh;
    cnt += hit ? 1 : 0;
    if (hit) { if (k < 64) m0 |= 1ull << k; else m1 |= 1ull << (k - 64); }
  }
  st_mx[r] = mx; st_inv[r] = inv; st_cnt[r] = cnt;
  st_mask[2 * (size_t)r] = m0; st_mask[2 * (size_t)r + 1] = m1;
}

__global__ __launch_bounds__(1024) void det_candidates_kernel(
    const float* __restrict__ cls_logits, int ld_cls, const float* __restrict__ deltas, int ld_delta,
    int K, int cls_agnostic, const float* __restrict__ proposals, const int* __restrict__ prop_count, int R,
    const int* __restrict__ image_sizes, float wx, float wy, float ww, float wh, float scale_clamp,
    float score_thresh, int Nmax, float* __restrict__ cboxes, float* __restrict__ cscores,
    int* __restrict__ cclass, int* __restrict__ crow, int* __restrict__ ccount, int* __restrict__ status,
    const float* __restrict__ st_mx, const float* __restrict__ st_inv, const int* __restrict__ st_cnt,
    const unsigned long long* __restrict__ st_mask) {
  __shared__ int sh[20];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int nrow = prop_count ? min(prop_count[b], R) : R;
  const float img_h = (float)image_sizes[b * 2 + 0], img_w = (float)image_sizes[b * 2 + 1];
  int base = 0;
  for (int r0 = 0; r0 < nrow; r0 += 1024) {
    const int r = r0 + tid;
    int cnt = 0;
    float mx = -INFINITY, inv = 0.f;
    const float* lg = cls_logits + ((size_t)b * R + (r < nrow ? r : 0)) * ld_cls;
    if (r < nrow && st_cnt) {       // row statistics precomputed by det_row_stats_kernel
      const size_t gr = (size_t)b * R + r;
      mx = st_mx[gr]; inv = st_inv[gr]; cnt = st_cnt[gr];
    } else if (r < nrow) {
      for (int k = 0; k <= K; ++k) { float v = lg[k]; mx = v > mx ? v : mx; }
      float sum = 0.f;
      for (int k = 0; k <= K; ++k) sum += expf(lg[k] - mx);
      inv = 1.f / sum;
      for (int k = 0; k < K; ++k) cnt += (expf(lg[k] - mx) * inv > score_thresh) ? 1 : 0;
    }
    int tot;
    int pos = base + block_excl_scan_1024(cnt, sh, &tot);
    if (r < nrow && cnt > 0) {
      const float4 pb = *reinterpret_cast<const float4*>(proposals + ((size_t)b * R + r) * 4);
      const float* dl = deltas + ((size_t)b * R + r) * ld_delta;
      auto emit = [&](int k) {
        const float pr = expf(lg[k] - mx) * inv;
        if (pr > score_thresh) {
          if (pos < Nmax) {
            const float* d = dl + (cls_agnostic ? 0 : k * 4);
            float box[4];
            apply_deltas(pb.x, pb.y, pb.z, pb.w, d[0], d[1], d[2], d[3], wx, wy, ww, wh, scale_clamp, box);
            float* o = cboxes + ((size_t)b * Nmax + pos) * 4;
            o[0] = clampf(box[0], 0.f, img_w); o[1] = clampf(box[1], 0.f, img_h);
            o[2] = clampf(box[2], 0.f, img_w); o[3] = clampf(box[3], 0.f, img_h);
            cscores[(size_t)b * Nmax + pos] = pr;
            cclass[(size_t)b * Nmax + pos] = k;
            crow[(size_t)b * Nmax + pos] = r;
          }
          ++pos;
        }
      };
      if (st_mask) {       // only the classes det_row_stats_kernel found above the threshold, in class order
        for (int w = 0; w < 2; ++w) {
          unsigned long long m = st_mask[2 * ((size_t)b * R + r) + w];
          while (m) {
            emit(w * 64 + __ffsll((long long)m) - 1);
            m &= m - 1;
          }
        }
      } else {
        for (int k = 0; k < K; ++k) emit(k);
      }
    }
    base += tot;
  }
  if (tid == 0) {
    if (base > Nmax) { atomicOr(status, 2); base = Nmax; }
    ccount[b] = base;
  }
}

// gather the kept detections, optionally detector_postprocess (scale, clip, drop empty; ordered)
__global__ __launch_bounds__(256) void det_gather_kernel(
    const float* __restrict__ cboxes, const float* __restrict__ cscores, const int* __restrict__ cclass,
    const int* __restrict__ crow, const int* __restrict__ keep, const int* __restrict__ num_keep, int Nmax,
    int topk, const float* __restrict__ post, float* __restrict__ oboxes, float* __restrict__ oscores,
    int* __restrict__ oclasses, int* __restrict__ orows, int* __restrict__ ocount) {
  __shared__ int wsum[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int nk = min(num_keep[b], topk);
  int base = 0;
  for (int r0 = 0; r0 < topk; r0 += 256) {
    const int r = r0 + tid;
    bool ok = false;
    float4 bx = {0, 0, 0, 0};
    float s = 0.f; int c = 0, row = 0;
    if (r < nk) {
      const int i = keep[(size_t)b * Nmax + r];
      bx = *reinterpret_cast<const float4*>(cboxes + ((size_t)b * Nmax + i) * 4);
      s = cscores[(size_t)b * Nmax + i]; c = cclass[(size_t)b * Nmax + i]; row = crow[(size_t)b * Nmax + i];
      ok = true;
      if (post) {  // post[b] = (scale_x, scale_y, out_h, out_w)
        const float sx = post[b * 4 + 0], sy = post[b * 4 + 1], oh = post[b * 4 + 2], ow = post[b * 4 + 3];
        bx.x = clampf(bx.x * sx, 0.f, ow); bx.y = clampf(bx.y * sy, 0.f, oh);
        bx.z = clampf(bx.z * sx, 0.f, ow); bx.w = clampf(bx.w * sy, 0.f, oh);
        ok = (bx.z - bx.x > 0.f) && (bx.w - bx.y > 0.f);
      }
    }
    const u64 bal = __ballot(ok);
    const int lane = tid & 63, wave = tid >> 6;
    __syncthreads();
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int wbase = 0, tot = 0;
    for (int w = 0; w < 4; ++w) { if (w < wave) wbase += wsum[w]; tot += wsum[w]; }
    const int pos = base + wbase + __popcll(bal & ((1ull << lane) - 1ull));
    if (ok) {
      *reinterpret_cast<float4*>(oboxes + ((size_t)b * topk + pos) * 4) = bx;
      oscores[(size_t)b * topk + pos] = s; oclasses[(size_t)b * topk + pos] = c; orows[(size_t)b * topk + pos] = row;
    }
    base += tot;
  }
  __syncthreads();
  // zero-fill the tail
  for (int r = base + tid; r < topk; r += 256) {
    float4 z = {0, 0, 0, 0};
    *reinterpret_cast<float4*>(oboxes + ((size_t)b * topk + r) * 4) = z;
    oscores[(size_t)b * topk + r] = 0.f; oclasses[(size_t)b * topk + r] = 0; orows[(size_t)b * topk + r] = 0;
  }
  if (tid == 0) ocount[b] = base;
}

struct DetPlan { long long off_cboxes, off_cscores, off_cclass, off_crow, off_ccount, off_keep, off_nk, off_nms, total; };
static DetPlan det_plan(int B, int Nmax) {
  DetPlan p; long long o = 0;
  p.off_cboxes = o; o = align16(o + (long long)B * Nmax * 16);
  p.off_cscores = o; o = align16(o + (long long)B * Nmax * 4);
  p.off_cclass = o; o = align16(o + (long long)B * Nmax * 4);
  p.off_crow = o; o = align16(o + (long long)B * Nmax * 4);
  p.off_ccount = o; o = align16(o + (long long)B * 4);
  p.off_keep = o; o = align16(o + (long long)B * Nmax * 4);
  p.off_nk = o; o = align16(o + (long long)B * 4);
  p.off_nms = o; o = align16(o + lvc_batched_nms_workspace_bytes(B, Nmax));
  p.total = o;
  return p;
}
extern "C" long long lvc_fast_rcnn_inference_workspace_bytes(int B, int max_candidates) {
  return det_plan(B, max_candidates).total;
}

// cls_logits [B*R, ld_cls] (K+1 used), deltas [B*R, ld_delta] (4K or 4 used), proposals [B,R,4],
// d_prop_count [B] or NULL, d_image_sizes [B,2] (h,w) int32, d_post [B,4] (sx,sy,out_h,out_w) or NULL.
// Outputs are fixed-size [B, topk, ...] with d_out_count [B]; d_status bit 1 (value 2) = candidate overflow.
extern "C" int lvc_fast_rcnn_inference(const float* cls_logits, int ld_cls, const float* deltas, int ld_delta,
                                       int K, int cls_agnostic, const float* proposals,
                                       const int* d_prop_count, int B, int R, const int* d_image_sizes,
                                       float wx, float wy, float ww, float wh, float scale_clamp,
                                       float score_thresh, double nms_thresh, int topk, int max_candidates,
                                       const float* d_post, float* out_boxes, float* out_scores,
                                       int* out_classes, int* out_rows, int* d_out_count, int* d_status,
                                       void* workspace, long long workspace_bytes, void* stream) {
  LVC_CHECK_ARG(B > 0 && R > 0 && K > 0 && topk > 0, "bad sizes");
  LVC_CHECK_ARG(max_candidates > 0, "max_candidates must be positive");
  LVC_CHECK_ARG(cls_logits && deltas && proposals && d_image_sizes && out_boxes && out_scores && out_classes &&
                    out_rows && d_out_count && d_status && workspace, "null pointer");
  DetPlan p = det_plan(B, max_candidates);
  LVC_CHECK_ARG(workspace_bytes >= p.total, "workspace too small");
  char* ws = (char*)workspace;
  hipStream_t st = (hipStream_t)stream;
  float* cboxes = (float*)(ws + p.off_cboxes);
  float* cscores = (float*)(ws + p.off_cscores);
  int* cclass = (int*)(ws + p.off_cclass);
  int* crow = (int*)(ws + p.off_crow);
  int* ccount = (int*)(ws + p.off_ccount);
  int* keep = (int*)(ws + p.off_keep);
  int* nk = (int*)(ws + p.off_nk);
  // row statistics live in the NMS scratch (not in use before lvc_batched_nms below)
  const long long rows = (long long)B * R;
  float* st_mx = nullptr; float* st_inv = nullptr; int* st_cnt = nullptr;
  unsigned long long* st_mask = nullptr;
  if (K + 1 < DET_MAX_COLS && rows * 28 + 64 <= p.total - p.off_nms) {
    st_mask = (unsigned long long*)(ws + p.off_nms);
    st_mx = (float*)(st_mask + 2 * rows);
    st_inv = st_mx + rows;
    st_cnt = (int*)(st_inv + rows);
    hipLaunchKernelGGL(det_row_stats_kernel, dim3((unsigned)lvc_cdiv64(rows, DET_TILE_ROWS)), dim3(DET_TILE_ROWS), 0, st,
                       cls_logits, ld_cls, K, (int)rows, score_thresh, st_mx, st_inv, st_cnt, st_mask);
    LVC_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(det_candidates_kernel, dim3(B), dim3(1024), 0, st, cls_logits, ld_cls, deltas, ld_delta, K,
                     cls_agnostic, proposals, d_prop_count, R, d_image_sizes, wx, wy, ww, wh, scale_clamp,
                     score_thresh, max_candidates, cboxes, cscores, cclass, crow, ccount, d_status, st_mx, st_inv, st_cnt, st_mask);
  LVC_CHECK_LAUNCH();
  int rc = lvc_batched_nms(cboxes, cscores, cclass, ccount, B, max_candidates, nms_thresh, topk, keep, nk,
                           ws + p.off_nms, p.total - p.off_nms, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(det_gather_kernel, dim3(B), dim3(256), 0, st, cboxes, cscores, cclass, crow, keep, nk,
                     max_candidates, topk, d_post, out_boxes, out_scores, out_classes, out_rows, d_out_count);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// =====================================================================================
// Class-agnostic decode + clip (one cascade stage of the box corrector):
// BoxOnlyLayersCascade.predict_boxes (lvc/modeling/roi_heads/roi_heads_cascade.py:197-211) followed by the clip of
// CascadeROIHeads._create_proposals_from_boxes (cascade_rcnn.py:348-369) / fast_rcnn_inference_single_image.
// =====================================================================================
__global__ void decode_boxes_kernel(const float* __restrict__ deltas, int ld, const float* __restrict__ boxes, int M,
                                    int R, const int* __restrict__ image_sizes, float wx, float wy, float ww, float wh,
                                    float scale_clamp, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const float4 b = *reinterpret_cast<const float4*>(boxes + (size_t)i * 4);
  const float* d = deltas + (size_t)i * ld;
  float o[4];
  apply_deltas(b.x, b.y, b.z, b.w, d[0], d[1], d[2], d[3], wx, wy, ww, wh, scale_clamp, o);
  if (image_sizes) {
    const int img = i / R;
    const float h = (float)image_sizes[img * 2], w = (float)image_sizes[img * 2 + 1];
    o[0] = clampf(o[0], 0.f, w); o[1] = clampf(o[1], 0.f, h); o[2] = clampf(o[2], 0.f, w); o[3] = clampf(o[3], 0.f, h);
  }
  float4 r = {o[0], o[1], o[2], o[3]};
  *reinterpret_cast<float4*>(out + (size_t)i * 4) = r;
}

// deltas [M, ld] (first 4 columns used), boxes [M,4] = B images x R rows, d_image_sizes [B,2] (h,w) or NULL (no clip).
extern "C" int lvc_decode_boxes(const float* deltas, int ld, const float* boxes, int M, int R, const int* d_image_sizes,
                                float wx, float wy, float ww, float wh, float scale_clamp, float* out, void* stream) {
  LVC_CHECK_ARG(M >= 0 && R > 0 && ld >= 4, "bad shape");
  if (M == 0) return LVC_OK;
  LVC_CHECK_ARG(deltas && boxes && out, "null pointer");
  hipLaunchKernelGGL(decode_boxes_kernel, dim3(lvc_cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, deltas, ld, boxes, M, R,
                     d_image_sizes, wx, wy, ww, wh, scale_clamp, out);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}
