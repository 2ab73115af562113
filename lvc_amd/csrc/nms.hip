// nms.hip -- batched greedy NMS with bit-exact keep indices, for gfx950 (wave64).
//
// Replaces (for the hot path) torchvision.ops.boxes.batched_nms + torchvision.ops.nms as called from
// reference detectron2/layers/nms.py:10-29, consumers detectron2/modeling/proposal_generator/
// proposal_utils.py:104 (RPN, thr 0.7, idx = FPN level) and lvc/modeling/roi_heads/fast_rcnn.py:128
// (detections, thr 0.5, idx = class).  torchvision is a third-party dependency that is not vendored in
// the reference (README.md:65-67 pins 0.8.2); the algorithm restated here is its CPU kernel:
//     off   = float(idx) * (max(boxes) + 1)            (batched_nms: one fp32 multiply)
//     b'    = boxes + off                              (one fp32 add per coordinate)
//     order = argsort(scores, descending, stable)
//     for i in order: if alive(i): keep i; for j after i: if inter/(a_i + a_j - inter) > thr: kill j
// with areas and IoU evaluated in fp32 in exactly that association and the fp32 IoU compared against
// the DOUBLE threshold.  This file is compiled with -ffp-contract=off (an FMA in `a_i+a_j-w*h` would
// change keep decisions) and fp32 division is IEEE (hipcc default).
//
// Three kernels per call, B images at once (blockIdx = image), no host sync, counts stay on device:
//   1. nms_prep   one workgroup/image: max-coordinate reduce, 64-bit key bitonic sort in LDS
//                 (key = ~ordered(score) << 32 | index  => score descending, ties by lower index),
//                 writes order[] and the offset boxes gathered into sorted order.
//   2. nms_mask   one WAVE per (64-row chunk, 64-column word): lane = column box j, the row box i is
//                 wave-uniform (v_readlane), the predicate of 64 columns is collected with one
//                 __ballot per row -> 64-bit suppression word mask[i][word].  Upper triangle only;
//                 pairs with different idx are skipped (their IoU is exactly 0 after the offset).
//   3. nms_reduce one wave/image: walks the chunks in order; resolves the 64x64 diagonal block in
//                 registers (ffs over the alive word + readlane of the row's diagonal word), then ORs
//                 the mask rows of the kept boxes into the removed-words (lane = word, coalesced rows).
#include "common.h"

typedef unsigned long long u64;

__device__ __forceinline__ unsigned int ordered_desc_key(float f) {
  if (f == 0.f) f = 0.f;  // -0.0 and +0.0 compare equal in the reference sort
  unsigned int u = __float_as_uint(f);
  u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;  // ascending-as-unsigned
  return ~u;                                   // descending
}

// ---------------------------------------------------------------- 1. prep: max, offsets, sort, gather
template <int NPAD>
__global__ __launch_bounds__(1024) void nms_prep_kernel(const float* __restrict__ boxes,
                                                        const float* __restrict__ scores,
                                                        const int* __restrict__ idxs,
                                                        const int* __restrict__ counts, int Nmax,
                                                        int* __restrict__ order,
                                                        float* __restrict__ sboxes,
                                                        int* __restrict__ sidx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  u64* keys = reinterpret_cast<u64*>(dsm);
  __shared__ float red[16];
  const int img = blockIdx.x;
  const int tid = threadIdx.x;
  int n = counts ? counts[img] : Nmax;
  if (n > Nmax) n = Nmax;
  const float* bx = boxes + (size_t)img * Nmax * 4;
  const float* sc = scores + (size_t)img * Nmax;
  const int* ix = idxs ? idxs + (size_t)img * Nmax : nullptr;

  // max coordinate over the n boxes (torch: boxes.max())
  float m = -INFINITY;
  for (int i = tid; i < n * 4; i += 1024) {
    float v = bx[i];
    m = v > m ? v : m;
  }
  for (int o = 32; o > 0; o >>= 1) {
    float t = __shfl_xor(m, o);
    m = t > m ? t : m;
  }
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  if (tid < 64) {
    float v = tid < 16 ? red[tid] : -INFINITY;
    for (int o = 8; o > 0; o >>= 1) {
      float t = __shfl_xor(v, o);
      v = t > v ? t : v;
    }
    if (tid == 0) red[0] = v;
  }
  __syncthreads();
  const float maxp1 = red[0] + 1.0f;

  // sort only the power of two that covers this image's n (the detection stage allocates for 16 384 candidates
  // per image and typically has a few thousand)
  int npad = 1024;
  while (npad < n) npad <<= 1;
  if (npad > NPAD) npad = NPAD;
  for (int i = tid; i < npad; i += 1024)
    keys[i] = i < n ? (((u64)ordered_desc_key(sc[i]) << 32) | (unsigned)i) : ~0ull;
  __syncthreads();
  // bitonic sort ascending on 64-bit keys
  for (int k = 2; k <= npad; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < npad / 2; t += 1024) {
        int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        int hi = lo | j;
        bool up = (lo & k) == 0;
        u64 a = keys[lo], b = keys[hi];
        if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
      }
      __syncthreads();
    }
  }
  int* ord = order + (size_t)img * Nmax;
  float* sb = sboxes + (size_t)img * Nmax * 4;
  int* si = sidx + (size_t)img * Nmax;
  for (int r = tid; r < n; r += 1024) {
    int i = (int)(keys[r] & 0xFFFFFFFFu);
    ord[r] = i;
    int id = ix ? ix[i] : 0;
    float off = (float)id * maxp1;
    sb[r * 4 + 0] = bx[i * 4 + 0] + off;
    sb[r * 4 + 1] = bx[i * 4 + 1] + off;
    sb[r * 4 + 2] = bx[i * 4 + 2] + off;
    sb[r * 4 + 3] = bx[i * 4 + 3] + off;
    si[r] = id;
  }
}

// ---------------------------------------------------------------- 2. suppression mask via ballot
#define NMS_MASK_STRIDE 96
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ sboxes,
                                                      const int* __restrict__ sidx,
                                                      const int* __restrict__ counts, int Nmax,
                                                      int nwords, double thr, u64* __restrict__ mask) {
  // grid (NMS_MASK_STRIDE, nwords, B): workgroup (x, ci) walks the column words wj = ci + x, ci + x + STRIDE, ... of its
  // 64-row chunk (upper triangle only).  The grid is sized for Nmax, the loop for this image's n: the detection stage
  // (Nmax = 16 384, a few thousand real candidates) no longer dispatches 256 x 256 mostly empty workgroups per image.
  const int ci = blockIdx.y, img = blockIdx.z;
  int n = counts ? counts[img] : Nmax;
  if (n > Nmax) n = Nmax;
  if (ci * 64 >= n) return;
  const int nchunks = (n + 63) >> 6;
  const int lane = threadIdx.x;
  const float* sb = sboxes + (size_t)img * Nmax * 4;
  const int* si = sidx + (size_t)img * Nmax;
  const int i_me = ci * 64 + lane;
  // row box held by lane i (broadcast later), column box held by lane j
  float ix1 = 0, iy1 = 0, ix2 = 0, iy2 = 0; int iid = -1;
  if (i_me < n) {
    const float4 b = *reinterpret_cast<const float4*>(sb + (size_t)i_me * 4);
    ix1 = b.x; iy1 = b.y; ix2 = b.z; iy2 = b.w; iid = si[i_me];
  }
  const float iarea_me = (ix2 - ix1) * (iy2 - iy1);
  const int rows = min(64, n - ci * 64);
  for (int wj = ci + blockIdx.x; wj < nchunks; wj += gridDim.x) {
    const int j_me = wj * 64 + lane;
    float jx1 = 0, jy1 = 0, jx2 = 0, jy2 = 0; int jid = -2;
    if (j_me < n) {
      const float4 b = *reinterpret_cast<const float4*>(sb + (size_t)j_me * 4);
      jx1 = b.x; jy1 = b.y; jx2 = b.z; jy2 = b.w; jid = si[j_me];
    }
    const float jarea = (jx2 - jx1) * (jy2 - jy1);
    u64 my_word = 0;
    for (int r = 0; r < rows; ++r) {
      const float ax1 = __shfl(ix1, r), ay1 = __shfl(iy1, r), ax2 = __shfl(ix2, r), ay2 = __shfl(iy2, r);
      const float aarea = __shfl(iarea_me, r);
      const int aid = __shfl(iid, r);
      const float xx1 = ax1 < jx1 ? jx1 : ax1;   // std::max(a, b)
      const float yy1 = ay1 < jy1 ? jy1 : ay1;
      const float xx2 = jx2 < ax2 ? jx2 : ax2;   // std::min(a, b)
      const float yy2 = jy2 < ay2 ? jy2 : ay2;
      float w = xx2 - xx1; if (!(w > 0.f)) w = 0.f;
      float h = yy2 - yy1; if (!(h > 0.f)) h = 0.f;
      const float inter = w * h;
      const float ovr = inter / (aarea + jarea - inter);
      const bool hit = (j_me < n) && (j_me > ci * 64 + r) && (aid == jid) && ((double)ovr > thr);
      const u64 word = __ballot(hit);
      if (lane == r) my_word = word;
    }
    if (i_me < n) mask[((size_t)img * Nmax + i_me) * nwords + wj] = my_word;
  }
}

// ---------------------------------------------------------------- 3. ordered reduce
#define NMS_MAX_WORDS 256
__global__ __launch_bounds__(64) void nms_reduce_kernel(const u64* __restrict__ mask,
                                                        const int* __restrict__ order,
                                                        const int* __restrict__ counts, int Nmax,
                                                        int nwords, int max_keep,
                                                        int* __restrict__ keep,
                                                        int* __restrict__ num_keep) {
  __shared__ u64 removed[NMS_MAX_WORDS];
  const int img = blockIdx.x, lane = threadIdx.x;
  int n = counts ? counts[img] : Nmax;
  if (n > Nmax) n = Nmax;
  const u64* mk = mask + (size_t)img * Nmax * nwords;
  const int* ord = order + (size_t)img * Nmax;
  int* kp = keep + (size_t)img * Nmax;
  for (int w = lane; w < nwords; w += 64) removed[w] = 0;
  __syncthreads();
  const int nchunks = (n + 63) >> 6;
  int nk = 0;
  for (int c = 0; c < nchunks && nk < max_keep; ++c) {
    const int row = c * 64 + lane;
    const u64 diag = row < n ? mk[(size_t)row * nwords + c] : 0ull;
    u64 alive = ~removed[c];
    const int valid = n - c * 64;
    if (valid < 64) alive &= ((1ull << valid) - 1ull);
    u64 kept = 0;
    const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
    while (alive) {
      const int i = __ffsll((long long)alive) - 1;
      kept |= 1ull << i;
      const u64 d = ((u64)(unsigned)__shfl((int)dhi, i) << 32) | (unsigned)__shfl((int)dlo, i);
      alive &= ~d;
      alive &= ~(1ull << i);
    }
    // emit kept boxes (original indices) in order
    if ((kept >> lane) & 1ull) {
      const int pos = nk + __popcll(kept & ((1ull << lane) - 1ull));
      if (pos < max_keep) kp[pos] = ord[row];
    }
    nk += __popcll(kept);
    // OR the rows of the kept boxes into the removed words of later chunks
    for (int w0 = c + 1; w0 < nchunks; w0 += 64) {   // words of chunks past n are never read
      const int w = w0 + lane;
      const int wc = w < nwords ? w : nwords - 1;  // clamped: out-of-range lanes load a valid word
      u64 acc = 0;
      u64 k = kept;
      while (k) {  // 8 independent row loads in flight per trip
        u64 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = k ? __ffsll((long long)k) - 1 : -1;
          k &= k - 1;  // 0 & anything stays 0
          const u64 x = mk[(size_t)(c * 64 + (i < 0 ? 0 : i)) * nwords + wc];
          v[u] = i < 0 ? 0ull : x;
        }
        acc |= (v[0] | v[1]) | (v[2] | v[3]) | (v[4] | v[5]) | (v[6] | v[7]);
      }
      if (w < nwords) removed[w] |= acc;
    }
    __syncthreads();
  }
  if (lane == 0) num_keep[img] = nk < max_keep ? nk : max_keep;
}

static int pad_pow2(int n) {
  int p = 1024;
  while (p < n) p <<= 1;
  return p;
}

extern "C" long long lvc_batched_nms_workspace_bytes(int B, int Nmax) {
  long long nwords = (Nmax + 63) / 64;
  long long per = (long long)Nmax * 4 * 4 /*sboxes*/ + (long long)Nmax * 4 /*order*/ +
                  (long long)Nmax * 4 /*sidx*/ + (long long)Nmax * nwords * 8 /*mask*/;
  return (long long)B * per + 256;
}

// boxes [B,Nmax,4] fp32 xyxy, scores [B,Nmax] fp32, idxs [B,Nmax] int32 or NULL, d_counts [B] device int32
// or NULL (= Nmax each).  keep [B,Nmax] int32 (indices into the image's Nmax rows, score-descending),
// d_num_keep [B] device int32.  At most max_keep (<=0: all) indices are produced per image.
extern "C" int lvc_batched_nms(const float* boxes, const float* scores, const int* idxs,
                               const int* d_counts, int B, int Nmax, double iou_threshold,
                               int max_keep, int* keep, int* d_num_keep, void* workspace,
                               long long workspace_bytes, void* stream) {
  LVC_CHECK_ARG(B >= 0 && Nmax >= 0, "negative size");
  if (B == 0) return LVC_OK;
  LVC_CHECK_ARG(d_num_keep, "null d_num_keep");
  hipStream_t st = (hipStream_t)stream;
  if (Nmax == 0) {
    (void)hipMemsetAsync(d_num_keep, 0, sizeof(int) * B, st);
    return LVC_OK;
  }
  LVC_CHECK_ARG(boxes && scores && keep && workspace, "null pointer");
  LVC_CHECK_ARG(Nmax <= 16384, "Nmax > 16384 boxes per image is not supported");
  LVC_CHECK_ARG(workspace_bytes >= lvc_batched_nms_workspace_bytes(B, Nmax), "workspace too small");
  LVC_CHECK_ARG(((uintptr_t)workspace & 15) == 0, "workspace must be 16-byte aligned");
  const int nwords = (Nmax + 63) / 64;
  char* ws = (char*)workspace;
  float* sboxes = (float*)ws; ws += (size_t)B * Nmax * 16;
  int* order = (int*)ws; ws += (size_t)B * Nmax * 4;
  int* sidx = (int*)ws; ws += (size_t)B * Nmax * 4;
  ws = (char*)(((uintptr_t)ws + 15) & ~(uintptr_t)15);
  u64* mask = (u64*)ws;
  if (max_keep <= 0) max_keep = Nmax;

  const int npad = pad_pow2(Nmax);
  const size_t lds = (size_t)npad * 8;
#define LAUNCH_PREP(NP)                                                                              \
  {                                                                                                  \
    (void)hipFuncSetAttribute((const void*)nms_prep_kernel<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                        (int)lds);                                                                   \
    hipLaunchKernelGGL(nms_prep_kernel<NP>, dim3(B), dim3(1024), lds, st, boxes, scores, idxs,       \
                       d_counts, Nmax, order, sboxes, sidx);                                         \
  }
  switch (npad) {
    case 1024: LAUNCH_PREP(1024); break;
    case 2048: LAUNCH_PREP(2048); break;
    case 4096: LAUNCH_PREP(4096); break;
    case 8192: LAUNCH_PREP(8192); break;
    default: LAUNCH_PREP(16384); break;
  }
#undef LAUNCH_PREP
  LVC_CHECK_LAUNCH();
  hipLaunchKernelGGL(nms_mask_kernel, dim3(nwords < NMS_MASK_STRIDE ? nwords : NMS_MASK_STRIDE, nwords, B), dim3(64), 0, st, sboxes, sidx, d_counts,
                     Nmax, nwords, iou_threshold, mask);
  LVC_CHECK_LAUNCH();
  hipLaunchKernelGGL(nms_reduce_kernel, dim3(B), dim3(64), 0, st, mask, order, d_counts, Nmax, nwords,
                     max_keep, keep, d_num_keep);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}
