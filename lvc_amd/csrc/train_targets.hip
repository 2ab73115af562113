// train_targets.hip -- the label-and-sample steps of a training forward for the WHOLE batch in a handful of launches (round 5).
//
// Reference: RPN.label_and_sample_anchors (detectron2/modeling/proposal_generator/rpn.py:269-325: Matcher with low-quality matches
// over 268 569 anchors per image + subsample_labels, sampling.py:10-54) and ROIHeads.label_and_sample_proposals
// (lvc/modeling/roi_heads/roi_heads.py:173-278: add_ground_truth_to_proposals, Matcher, subsample_labels).  The reference walks the
// images one by one, each with two device->host reads and two randperm sorts; round 4 did the same with ~70 small launches per image
// around lvc_match_boxes (profiles/r05_train_cfg3_kernel_stats.csv: 576 launches of library kernels per step, 2.8 ms of GPU time and a
// host-bound tail behind the trunk).  Here:
//   lvc_match_boxes_batched   pairwise IoU + Matcher for B images in two launches (ragged ground truth, shared or per-image boxes)
//   lvc_subsample_batched     subsample_labels for B rows in one launch: per row the num_pos positives / num_neg negatives with the
//                             SMALLEST random keys (keys = one torch.randperm(B * N): any subset of a random permutation is in
//                             uniformly random order, and with randperm patched to arange -- the parity tests -- it is the
//                             reference's choice exactly: the first positives / negatives), by an 11-bit radix select in LDS
//   lvc_rpn_gather_sampled    logits / deltas / anchors (rebuilt from the cell anchors) / matched gt of the sampled anchors straight
//                             from the head's per-level outputs: the [B, R] / [B, R, 4] concatenations are never formed
//   lvc_roi_build_table       proposals + ground truth of an image in one padded [B, W] table (add_ground_truth_to_proposals)
//   lvc_roi_gather_sampled    the sampled rows of that table with their classes (gt class of the match, or K = background)
// Built with -ffp-contract=off like the other geometry files (the IoU arithmetic is lvc_match_boxes').
#include "common.h"

__device__ __forceinline__ float iou_ref_b(float ax1, float ay1, float ax2, float ay2, float bx1, float by1, float bx2, float by2) {
  const float a1 = (ax2 - ax1) * (ay2 - ay1), a2 = (bx2 - bx1) * (by2 - by1);
  float w = fminf(ax2, bx2) - fmaxf(ax1, bx1);
  float h = fminf(ay2, by2) - fmaxf(ay1, by1);
  w = w < 0.f ? 0.f : w;
  h = h < 0.f ? 0.f : h;
  const float inter = w * h;
  return inter > 0.f ? inter / (a1 + a2 - inter) : 0.f;
}

#define MAX_GT 512

// pass 1: per box: max IoU over the image's gt boxes + first arg-max; per gt: max IoU over the image's boxes
__global__ __launch_bounds__(256) void match_b_pass1_kernel(const float* __restrict__ gt, const int* __restrict__ gt_off,
                                                            const float* __restrict__ boxes, long long box_img_stride,
                                                            const int* __restrict__ nbox, int N, float* __restrict__ vals,
                                                            int* __restrict__ matches, unsigned int* __restrict__ gt_best) {
  __shared__ float sgt[MAX_GT * 4];
  __shared__ unsigned int sbest[MAX_GT];
  const int b = blockIdx.y;
  const int g0 = gt_off[b], G = min(gt_off[b + 1] - g0, MAX_GT);
  for (int i = threadIdx.x; i < G * 4; i += 256) sgt[i] = gt[(size_t)g0 * 4 + i];
  for (int i = threadIdx.x; i < G; i += 256) sbest[i] = 0u;
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int nb = nbox ? nbox[b] : N;
  if (n < nb) {
    const float4 bx = *reinterpret_cast<const float4*>(boxes + (size_t)b * box_img_stride + (size_t)n * 4);
    float best = G ? -1.f : 0.f;
    int bi = 0;
    for (int g = 0; g < G; ++g) {
      const float v = iou_ref_b(sgt[g * 4], sgt[g * 4 + 1], sgt[g * 4 + 2], sgt[g * 4 + 3], bx.x, bx.y, bx.z, bx.w);
      if (v > best) { best = v; bi = g; }
      if (v > 0.f) atomicMax(&sbest[g], __float_as_uint(v));      // the table starts at 0: only overlapping pairs can raise it
    }
    vals[(size_t)b * N + n] = best;
    matches[(size_t)b * N + n] = bi;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < G; i += 256) atomicMax(&gt_best[g0 + i], sbest[i]);
}

// pass 2: labels from the thresholds; low-quality matches; rows past the image's box count: label -1, match 0.
// An image without ground truth: every label = l0, every match 0 (matcher.py:76-87)
__global__ __launch_bounds__(256) void match_b_pass2_kernel(const float* __restrict__ gt, const int* __restrict__ gt_off,
                                                            const float* __restrict__ boxes, long long box_img_stride,
                                                            const int* __restrict__ nbox, int N, const float* __restrict__ vals,
                                                            const unsigned int* __restrict__ gt_best, float t0, float t1, int nthr,
                                                            int l0, int l1, int l2, int allow_low_quality, int* __restrict__ matches,
                                                            signed char* __restrict__ labels) {
  __shared__ float sgt[MAX_GT * 4];
  __shared__ float sbest[MAX_GT];
  const int b = blockIdx.y;
  const int g0 = gt_off[b], G = min(gt_off[b + 1] - g0, MAX_GT);
  for (int i = threadIdx.x; i < G * 4; i += 256) sgt[i] = gt[(size_t)g0 * 4 + i];
  for (int i = threadIdx.x; i < G; i += 256) sbest[i] = __uint_as_float(gt_best[g0 + i]);
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const int nb = nbox ? nbox[b] : N;
  if (n >= nb) {
    labels[(size_t)b * N + n] = -1;
    matches[(size_t)b * N + n] = 0;
    return;
  }
  if (G == 0) {
    labels[(size_t)b * N + n] = (signed char)l0;
    return;
  }
  const float v = vals[(size_t)b * N + n];
  int lab;
  if (nthr == 1) lab = v < t0 ? l0 : l1;
  else lab = v < t0 ? l0 : (v < t1 ? l1 : l2);
  if (allow_low_quality) {
    const float4 bx = *reinterpret_cast<const float4*>(boxes + (size_t)b * box_img_stride + (size_t)n * 4);
    for (int g = 0; g < G; ++g) {
      const float q = iou_ref_b(sgt[g * 4], sgt[g * 4 + 1], sgt[g * 4 + 2], sgt[g * 4 + 3], bx.x, bx.y, bx.z, bx.w);
      if (q == sbest[g]) { lab = 1; break; }
    }
  }
  labels[(size_t)b * N + n] = (signed char)lab;
}

// gt [Gtot,4] (the images' boxes one after the other), gt_off [B+1] int32 DEVICE (prefix of the per-image counts, each <= 512),
// boxes: box_img_stride == 0: [N,4] shared by the images (the anchors); else [B][N][4] with nbox [B] valid rows per image (device; null = N).
// -> matches int32 [B,N] (arg-max gt of the image, first on ties), labels int8 [B,N], vals fp32 [B,N].  gt_best: [Gtot] uint32 scratch.
extern "C" int lvc_match_boxes_batched(const float* gt, const int* gt_off, int Gtot, int B, const float* boxes, long long box_img_stride,
                                       const int* nbox, int N, float t0, float t1, int nthr, int l0, int l1, int l2,
                                       int allow_low_quality, int* matches, signed char* labels, float* vals, unsigned int* gt_best,
                                       void* stream) {
  LVC_CHECK_ARG(B > 0 && N >= 0 && Gtot >= 0 && (nthr == 1 || nthr == 2), "bad arguments");
  if (N == 0) return LVC_OK;
  LVC_CHECK_ARG(gt_off && boxes && matches && labels && vals && gt_best && (gt || Gtot == 0), "null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (Gtot) (void)hipMemsetAsync(gt_best, 0, sizeof(unsigned int) * Gtot, st);
  const dim3 grid(lvc_cdiv(N, 256), B);
  hipLaunchKernelGGL(match_b_pass1_kernel, grid, dim3(256), 0, st, gt, gt_off, boxes, box_img_stride, nbox, N, vals, matches, gt_best);
  LVC_CHECK_LAUNCH();
  hipLaunchKernelGGL(match_b_pass2_kernel, grid, dim3(256), 0, st, gt, gt_off, boxes, box_img_stride, nbox, N, vals, gt_best, t0, t1,
                     nthr, l0, l1, l2, allow_low_quality, matches, labels);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// ------------------------------------------------------------------------------------------------ subsample_labels, batched
#define SS_NT 1024
#define SS_BINS 2048
#define SS_CAP 1024      // most rows a class contributes to one image's sample
#define SS_CHUNK 8192    // elements per workgroup of the streaming passes

// Per (row, class) state of the radix select: {prefix, have, need, n}.  labels: 1 = positive, 0 = negative, anything else ignored.
// keys: distinct non-negative integers below 2^nbits (int64, as torch.randperm makes them).  Selected: the num_pos = min(#pos, cap_pos)
// positives with the smallest keys, then the num_neg = min(#neg, bs - num_pos) negatives with the smallest keys, each group in increasing
// key order.  The 268 569-anchor rows are streamed by MANY workgroups per pass (one per 8192 elements: a single workgroup per row spent
// 300 us on its three passes); the histograms and candidate lists of a row meet in global memory.
struct SsState { unsigned prefix, have; int need, n; };

// The key of element i of row b: the caller's (torch.randperm) or, with keys == NULL, a pseudo-random BIJECTION of [0, 2^(2 hb)) applied
// to b * N + i -- a four-round Feistel network whose round function is murmur3's finaliser keyed by the seed: distinct keys by
// construction, no 17 MB key tensor to sort (torch.randperm of 2.1 M keys: 0.3 ms) or to read in the three passes.
__device__ __forceinline__ unsigned ss_key(const long long* __restrict__ key, int i, unsigned base, unsigned long long seed, int hb) {
  if (key) return (unsigned)key[i];
  const unsigned mask = (1u << hb) - 1u;
  const unsigned x = base + (unsigned)i;
  unsigned L = x >> hb, R = x & mask;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    unsigned f = R ^ ((unsigned)(seed >> (16 * r)) * 0x9E3779B1u + (unsigned)r * 0x7F4A7C15u);
    f ^= f >> 16; f *= 0x85EBCA6Bu; f ^= f >> 13; f *= 0xC2B2AE35u; f ^= f >> 16;
    const unsigned t = L ^ (f & mask);
    L = R; R = t;
  }
  return (L << hb) | R;
}

// pass `pass` of the select: histogram of the 11-bit digit at `shift` over the elements whose leading digits equal the class's prefix
__global__ __launch_bounds__(256) void ss_hist_kernel(const signed char* __restrict__ labels, const long long* __restrict__ keys, int N, int pass,
                                                      int shift, const SsState* __restrict__ state, int* __restrict__ ghist,
                                                      unsigned long long seed, int hb) {
  __shared__ int hist[2][SS_BINS];
  const int b = blockIdx.y, tid = threadIdx.x;
  for (int i = tid; i < 2 * SS_BINS; i += 256) (&hist[0][0])[i] = 0;
  __syncthreads();
  unsigned pre[2] = {0u, 0u};
  bool live[2] = {true, true};
  if (pass > 0) {
    for (int c = 0; c < 2; ++c) {
      const SsState st = state[b * 2 + c];
      pre[c] = st.prefix;
      live[c] = st.need > 0 && st.need < st.n;      // otherwise the class needs no threshold: nothing or everything is taken
    }
  }
  const signed char* lab = labels + (size_t)b * N;
  const long long* key = keys ? keys + (size_t)b * N : nullptr;
  const unsigned kbase = (unsigned)b * (unsigned)N;
  const int i0 = blockIdx.x * SS_CHUNK, i1 = min(N, i0 + SS_CHUNK);
  for (int i = i0 + tid; i < i1; i += 256) {
    const int l = lab[i];
    if ((l == 0 || l == 1) && live[l]) {
      const unsigned k = ss_key(key, i, kbase, seed, hb);
      if (pass == 0 || (k >> (shift + 11)) == pre[l]) atomicAdd(&hist[l][(k >> shift) & (SS_BINS - 1)], 1);
    }
  }
  __syncthreads();
  int* gh = ghist + (size_t)b * 2 * SS_BINS;
  for (int i = tid; i < 2 * SS_BINS; i += 256) {
    const int v = (&hist[0][0])[i];
    if (v) atomicAdd(gh + i, v);
  }
}

// one workgroup per row: class totals and wanted counts (pass 0), the bin of the need-th smallest key; clears the histogram
__global__ __launch_bounds__(256) void ss_find_kernel(int pass, int cap_pos, int bs, SsState* __restrict__ state, int* __restrict__ ghist,
                                                      int* __restrict__ cand_count) {
  __shared__ int part[2][256];
  __shared__ int s_bin[2], s_below[2];
  const int b = blockIdx.x, tid = threadIdx.x;
  int* gh = ghist + (size_t)b * 2 * SS_BINS;
  int h[2][8];
  for (int c = 0; c < 2; ++c) {
    int sum = 0;
    for (int q = 0; q < 8; ++q) { h[c][q] = gh[c * SS_BINS + tid * 8 + q]; sum += h[c][q]; }
    part[c][tid] = sum;
  }
  __syncthreads();
  for (int i = tid; i < 2 * SS_BINS; i += 256) gh[i] = 0;
  if (tid < 2) {
    // exclusive prefix over the 256 partial sums by one thread per class (256 LDS reads), then the bin inside the group of eight
    const int c = tid;
    SsState st = state[b * 2 + c];
    if (pass == 0) {
      int n = 0;
      for (int i = 0; i < 256; ++i) n += part[c][i];
      st.n = n; st.prefix = 0u; st.have = 0u; st.need = 0;
      state[b * 2 + c] = st;
    }
  }
  __syncthreads();
  if (pass == 0 && tid == 0) {
    const int npos = min(state[b * 2 + 1].n, cap_pos);
    state[b * 2 + 1].need = npos;
    state[b * 2 + 0].need = min(state[b * 2 + 0].n, bs - npos);
    cand_count[b * 2] = 0; cand_count[b * 2 + 1] = 0;
  }
  __syncthreads();
  if (tid < 2) {
    const int c = tid;
    const SsState st = state[b * 2 + c];
    s_bin[c] = -1;
    if (st.need > 0 && st.need < st.n) {
      const int want = st.need - (int)st.have;
      int acc = 0, g = 0;
      for (; g < 256; ++g) {
        if (acc + part[c][g] >= want) break;
        acc += part[c][g];
      }
      s_bin[c] = g; s_below[c] = acc;
    }
  }
  __syncthreads();
  for (int c = 0; c < 2; ++c) {
    if (s_bin[c] == tid) {      // the thread that holds the group's eight bins
      SsState st = state[b * 2 + c];
      const int want = st.need - (int)st.have;
      int acc = s_below[c], q = 0;
      for (; q < 8; ++q) {
        if (acc + h[c][q] >= want) break;
        acc += h[c][q];
      }
      st.have += (unsigned)acc;
      st.prefix = (st.prefix << 11) | (unsigned)(tid * 8 + q);
      state[b * 2 + c] = st;
    }
  }
}

// elements at or below the class's threshold key -> the row's candidate list (exactly `need` of them: keys are distinct)
__global__ __launch_bounds__(256) void ss_collect_kernel(const signed char* __restrict__ labels, const long long* __restrict__ keys, int N,
                                                         const SsState* __restrict__ state, unsigned long long* __restrict__ cand,
                                                         int* __restrict__ cand_count, unsigned long long seed, int hb) {
  const int b = blockIdx.y, tid = threadIdx.x;
  unsigned thr[2];
  int need[2];
  for (int c = 0; c < 2; ++c) {
    const SsState st = state[b * 2 + c];
    need[c] = st.need;
    thr[c] = st.need <= 0 ? 0u : (st.need >= st.n ? 0xffffffffu : st.prefix);
  }
  const signed char* lab = labels + (size_t)b * N;
  const long long* key = keys ? keys + (size_t)b * N : nullptr;
  const unsigned kbase = (unsigned)b * (unsigned)N;
  const int i0 = blockIdx.x * SS_CHUNK, i1 = min(N, i0 + SS_CHUNK);
  for (int i = i0 + tid; i < i1; i += 256) {
    const int l = lab[i];
    if (l == 0 || l == 1) {
      const unsigned k = ss_key(key, i, kbase, seed, hb);
      if (need[l] > 0 && k <= thr[l]) {
        const int pos = atomicAdd(&cand_count[b * 2 + l], 1);
        if (pos < SS_CAP) cand[((size_t)b * 2 + l) * SS_CAP + pos] = ((unsigned long long)k << 32) | (unsigned)i;
      }
    }
  }
}

// one workgroup per row: both candidate lists sorted by key (bitonic over SS_CAP entries) and written positives first
__global__ __launch_bounds__(SS_NT) void ss_emit_kernel(const SsState* __restrict__ state, const unsigned long long* __restrict__ cand, int bs,
                                                        int* __restrict__ sel, int* __restrict__ counts) {
  __shared__ unsigned long long list[2][SS_CAP];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int np = min(state[b * 2 + 1].need, SS_CAP), nn = min(state[b * 2 + 0].need, SS_CAP);
  list[1][tid] = tid < np ? cand[((size_t)b * 2 + 1) * SS_CAP + tid] : ~0ull;
  list[0][tid] = tid < nn ? cand[((size_t)b * 2 + 0) * SS_CAP + tid] : ~0ull;
  __syncthreads();
  for (int k = 2; k <= SS_CAP; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int ixj = tid ^ j;
      if (ixj > tid) {
        for (int c = 0; c < 2; ++c) {
          const unsigned long long a = list[c][tid], d = list[c][ixj];
          const bool up = (tid & k) == 0;
          if ((a > d) == up) { list[c][tid] = d; list[c][ixj] = a; }
        }
      }
      __syncthreads();
    }
  for (int j = tid; j < bs; j += SS_NT) {
    int v = -1;
    if (j < np) v = (int)(unsigned)list[1][j];
    else if (j < np + nn) v = (int)(unsigned)list[0][j - np];
    sel[(size_t)b * bs + j] = v;
  }
  if (tid == 0) { counts[b * 2] = np; counts[b * 2 + 1] = nn; }
}

extern "C" long long lvc_subsample_workspace_bytes(int B) {
  return (long long)B * (2 * SS_BINS * 4 + 2 * sizeof(SsState) + 2 * 4 + 2 * SS_CAP * 8) + 256;
}

// labels int8 [B,N]; keys int64 [B,N] distinct in [0, 2^nbits), or NULL: keys generated from `seed` (a pseudo-random bijection of
// b N + i; nbits is then rounded up to an even count) -> sel int32 [B,bs] (positives first, -1 padded), counts int32 [B,2].
// workspace: lvc_subsample_workspace_bytes(B) bytes, ZEROED by the caller before the first use (the launches leave it zeroed).
extern "C" int lvc_subsample_batched(const signed char* labels, const long long* keys, unsigned long long seed, int B, int N, int nbits,
                                     int cap_pos, int bs, int* sel, int* counts, void* workspace, void* stream) {
  LVC_CHECK_ARG(labels && sel && counts && workspace, "null pointer");
  LVC_CHECK_ARG(B > 0 && N > 0 && nbits > 0 && nbits <= 32 && cap_pos >= 0 && bs > 0 && bs <= SS_CAP && cap_pos <= bs, "bad arguments");
  LVC_CHECK_ARG((long long)B * N <= (1ll << 32), "more than 2^32 elements");
  int hb = 0;
  if (!keys) {
    hb = (nbits + 1) / 2;
    if (hb < 1) hb = 1;
    nbits = 2 * hb;
  }
  char* w = (char*)workspace;
  int* ghist = (int*)w;                         w += (size_t)B * 2 * SS_BINS * 4;
  unsigned long long* cand = (unsigned long long*)w;  w += (size_t)B * 2 * SS_CAP * 8;
  SsState* state = (SsState*)w;                 w += (size_t)B * 2 * sizeof(SsState);
  int* cand_count = (int*)w;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(lvc_cdiv(N, SS_CHUNK), B);
  const int npass = (nbits + 10) / 11;
  for (int pass = 0; pass < npass; ++pass) {
    const int shift = (npass - 1 - pass) * 11;
    hipLaunchKernelGGL(ss_hist_kernel, grid, dim3(256), 0, st, labels, keys, N, pass, shift, state, ghist, seed, hb);
    hipLaunchKernelGGL(ss_find_kernel, dim3(B), dim3(256), 0, st, pass, cap_pos, bs, state, ghist, cand_count);
  }
  hipLaunchKernelGGL(ss_collect_kernel, grid, dim3(256), 0, st, labels, keys, N, state, cand, cand_count, seed, hb);
  hipLaunchKernelGGL(ss_emit_kernel, dim3(B), dim3(SS_NT), 0, st, state, cand, bs, sel, counts);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// ------------------------------------------------------------------------------------------------ RPN: rows of the sampled anchors
#define TT_MAXL 8
struct RpnGather {
  const float* fused[TT_MAXL];         // [B, H*W, ld]: channel a = objectness of anchor a, A + 4 a + c = delta c
  const float* cell[TT_MAXL];          // [A,4]
  int ld[TT_MAXL], H[TT_MAXL], W[TT_MAXL], stride[TT_MAXL];
  long long off[TT_MAXL + 1];          // prefix of H*W*A
  int L, A;
};

__global__ __launch_bounds__(256) void rpn_gather_kernel(RpnGather lv, const int* __restrict__ sel, const int* __restrict__ counts,
                                                         const int* __restrict__ matches, long long R, const float* __restrict__ gt,
                                                         const int* __restrict__ gt_off, int B, int bs, float* __restrict__ logits,
                                                         float* __restrict__ deltas, float* __restrict__ anchors, float* __restrict__ gtb,
                                                         signed char* __restrict__ labels) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= B * bs) return;
  const int b = t / bs, j = t - b * bs;
  const int np = counts[b * 2], nn = counts[b * 2 + 1];
  const int r = sel[t];
  float lg = 0.f, d[4] = {0.f, 0.f, 0.f, 0.f}, an[4] = {0.f, 0.f, 1.f, 1.f}, g[4] = {0.f, 0.f, 1.f, 1.f};
  signed char lab = -1;
  if (j < np + nn && r >= 0) {
    lab = j < np ? 1 : 0;
    int l = 0;
    while (l + 1 < lv.L && (long long)r >= lv.off[l + 1]) ++l;
    const int q = r - (int)lv.off[l];
    const int pix = q / lv.A, a = q - pix * lv.A;
    const int y = pix / lv.W[l], x = pix - y * lv.W[l];
    const float* row = lv.fused[l] + ((size_t)b * lv.H[l] * lv.W[l] + pix) * lv.ld[l];
    lg = row[a];
    const float sx = (float)(x * lv.stride[l]), sy = (float)(y * lv.stride[l]);
    const float* ca = lv.cell[l] + a * 4;
    an[0] = sx + ca[0]; an[1] = sy + ca[1]; an[2] = sx + ca[2]; an[3] = sy + ca[3];
#pragma unroll
    for (int c = 0; c < 4; ++c) d[c] = row[lv.A + a * 4 + c];
    if (lab == 1 && gt_off[b + 1] > gt_off[b]) {
      const float* gp = gt + (size_t)(gt_off[b] + matches[(size_t)b * R + r]) * 4;
#pragma unroll
      for (int c = 0; c < 4; ++c) g[c] = gp[c];
    }
  }
  logits[t] = lg;
  labels[t] = lab;
#pragma unroll
  for (int c = 0; c < 4; ++c) { deltas[(size_t)t * 4 + c] = d[c]; anchors[(size_t)t * 4 + c] = an[c]; gtb[(size_t)t * 4 + c] = g[c]; }
}

// sel / counts of lvc_subsample_batched over the R = sum_l H_l W_l A anchors of an image -> the rows RPN.losses needs, [B * bs] each:
// logits, deltas [.,4], anchors [.,4] (grid anchors rebuilt as shift + cell anchor, anchor_generator.py:161-185), the matched gt box of
// positives [.,4], labels int8 (1 / 0 / -1 = padding, which lvc_rpn_losses ignores).
extern "C" int lvc_rpn_gather_sampled(const void* const* fused, const int* ld, const void* const* cell_anchors, const int* H, const int* W,
                                      const int* strides, int L, int A, int B, int bs, const int* sel, const int* counts,
                                      const int* matches, const float* gt, const int* gt_off, float* logits, float* deltas, float* anchors,
                                      float* gt_boxes, signed char* labels, void* stream) {
  LVC_CHECK_ARG(L > 0 && L <= TT_MAXL && A > 0 && B > 0 && bs > 0, "bad arguments");
  LVC_CHECK_ARG(fused && ld && cell_anchors && H && W && strides && sel && counts && matches && gt_off && logits && deltas && anchors &&
                    gt_boxes && labels, "null pointer");
  RpnGather lv;
  lv.L = L; lv.A = A;
  lv.off[0] = 0;
  for (int l = 0; l < L; ++l) {
    lv.fused[l] = (const float*)fused[l]; lv.cell[l] = (const float*)cell_anchors[l];
    lv.ld[l] = ld[l]; lv.H[l] = H[l]; lv.W[l] = W[l]; lv.stride[l] = strides[l];
    lv.off[l + 1] = lv.off[l] + (long long)H[l] * W[l] * A;
  }
  hipLaunchKernelGGL(rpn_gather_kernel, dim3(lvc_cdiv(B * bs, 256)), dim3(256), 0, (hipStream_t)stream, lv, sel, counts, matches, lv.off[L],
                     gt, gt_off, B, bs, logits, deltas, anchors, gt_boxes, labels);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// ------------------------------------------------------------------------------------------------ ROI heads: table and sampled rows
__global__ __launch_bounds__(256) void roi_table_kernel(const float* __restrict__ pboxes, const float* __restrict__ plogits,
                                                        const int* __restrict__ pcount, int P, const float* __restrict__ gt,
                                                        const int* __restrict__ gt_off, float gt_logit, int B, int Wt,
                                                        float* __restrict__ boxes, float* __restrict__ logits, int* __restrict__ nrow) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= B * Wt) return;
  const int b = t / Wt, w = t - b * Wt;
  const int np = min(pcount[b], P), G = gt_off[b + 1] - gt_off[b];
  float4 bx = {0.f, 0.f, 0.f, 0.f};
  float lg = 0.f;
  if (w < np) {
    bx = *reinterpret_cast<const float4*>(pboxes + ((size_t)b * P + w) * 4);
    lg = plogits[(size_t)b * P + w];
  } else if (w < np + G) {
    bx = *reinterpret_cast<const float4*>(gt + (size_t)(gt_off[b] + w - np) * 4);
    lg = gt_logit;
  }
  *reinterpret_cast<float4*>(boxes + (size_t)t * 4) = bx;
  logits[t] = lg;
  if (w == 0) nrow[b] = np + G;
}

// proposals [B,P,4] / logits [B,P] / count [B] (lvc_rpn_proposals) + the images' gt boxes -> boxes [B,Wt,4], logits [B,Wt]: the image's
// proposals followed by its gt boxes (add_ground_truth_to_proposals, proposal_utils.py:121-162: logit log((1 - 1e-10) / 1e-10)), zero
// rows behind; nrow [B] = rows in use.  Wt >= P + max G.
extern "C" int lvc_roi_build_table(const float* pboxes, const float* plogits, const int* pcount, int B, int P, const float* gt,
                                   const int* gt_off, float gt_logit, int Wt, float* boxes, float* logits, int* nrow, void* stream) {
  LVC_CHECK_ARG(pboxes && plogits && pcount && gt_off && boxes && logits && nrow && B > 0 && P >= 0 && Wt > 0, "bad arguments");
  hipLaunchKernelGGL(roi_table_kernel, dim3(lvc_cdiv(B * Wt, 256)), dim3(256), 0, (hipStream_t)stream, pboxes, plogits, pcount, P, gt, gt_off,
                     gt_logit, B, Wt, boxes, logits, nrow);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

__global__ __launch_bounds__(256) void roi_gather_kernel(const float* __restrict__ boxes, const float* __restrict__ logits,
                                                         const int* __restrict__ matches, const int* __restrict__ sel,
                                                         const int* __restrict__ counts, const long long* __restrict__ gt_classes,
                                                         const int* __restrict__ gt_off, int B, int Wt, int bs, int K,
                                                         float* __restrict__ s_boxes, float* __restrict__ s_logits,
                                                         long long* __restrict__ s_cls, long long* __restrict__ s_match) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= B * bs) return;
  const int b = t / bs, j = t - b * bs;
  const int np = counts[b * 2], nn = counts[b * 2 + 1];
  float4 bx = {0.f, 0.f, 0.f, 0.f};
  float lg = 0.f;
  long long cls = K, m = 0;      // padding rows: background (callers cut them off or check the counts)
  if (j < np + nn) {
    const int r = sel[t];
    bx = *reinterpret_cast<const float4*>(boxes + ((size_t)b * Wt + r) * 4);
    lg = logits[(size_t)b * Wt + r];
    m = matches[(size_t)b * Wt + r];
    cls = j < np ? gt_classes[gt_off[b] + m] : (long long)K;
  }
  *reinterpret_cast<float4*>(s_boxes + (size_t)t * 4) = bx;
  s_logits[t] = lg;
  s_cls[t] = cls;
  s_match[t] = m;
}

// rows sel [B,bs] of the table -> s_boxes [B,bs,4], s_logits [B,bs], s_cls int64 [B,bs] (the matched gt's class for the foreground rows,
// K for background and for padding rows), s_match int64 [B,bs] (index of the matched gt inside the image)
extern "C" int lvc_roi_gather_sampled(const float* boxes, const float* logits, const int* matches, const int* sel, const int* counts,
                                      const long long* gt_classes, const int* gt_off, int B, int Wt, int bs, int K, float* s_boxes,
                                      float* s_logits, long long* s_cls, long long* s_match, void* stream) {
  LVC_CHECK_ARG(boxes && logits && matches && sel && counts && gt_classes && gt_off && s_boxes && s_logits && s_cls && s_match, "null pointer");
  LVC_CHECK_ARG(B > 0 && Wt > 0 && bs > 0 && K > 0, "bad arguments");
  hipLaunchKernelGGL(roi_gather_kernel, dim3(lvc_cdiv(B * bs, 256)), dim3(256), 0, (hipStream_t)stream, boxes, logits, matches, sel, counts,
                     gt_classes, gt_off, B, Wt, bs, K, s_boxes, s_logits, s_cls, s_match);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}
