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
#define NMS_TILE 16384   // rows of one block of the blocked form = keys of one LDS sort tile
#define NMS_HEAD_MIN 2048   // lists longer than this with max_keep << Nmax run the greedy pass on a HEAD block of sorted rows first
// Rows / columns are the sorted positions [row0, min(n, row0 + rows_cap)) of the image (row0 = 0 and rows_cap >= n:
// the whole image, the usual case); mask row (i - row0) holds nwords column words relative to the block.
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ sboxes,
                                                      const int* __restrict__ sidx,
                                                      const int* __restrict__ counts, int Nmax,
                                                      int nwords, double thr, u64* __restrict__ mask, int row0,
                                                      int rows_cap, const int* __restrict__ num_keep, int max_keep) {
  // grid (NMS_MASK_STRIDE, nwords, B): workgroup (x, ci) walks the column words wj = ci + x, ci + x + STRIDE, ... of its
  // 64-row chunk (upper triangle only).  The grid is sized for Nmax, the loop for this image's n: the detection stage
  // (Nmax = 16 384, a few thousand real candidates) no longer dispatches 256 x 256 mostly empty workgroups per image.
  const int img = blockIdx.z;
  int n = counts ? counts[img] : Nmax;
  if (n > Nmax) n = Nmax;
  const int nend = n < row0 + rows_cap ? n : row0 + rows_cap;
  if (num_keep && num_keep[img] >= max_keep) return;   // later blocks of the blocked form: the image is done
  const int nchunks = nend > row0 ? (nend - row0 + 63) >> 6 : 0;
  const int lane = threadIdx.x;
  const float* sb = sboxes + (size_t)img * Nmax * 4;
  const int* si = sidx + (size_t)img * Nmax;
  // gridDim.y may be smaller than the number of 64-row chunks (the second block of the head-block form is launched for the worst
  // case and mostly returns at once: 184 000 empty workgroups cost 39 us): a workgroup walks chunks ci, ci + gridDim.y, ...
  for (int ci = blockIdx.y; ci < nchunks; ci += gridDim.y) {
  const int i_me = row0 + ci * 64 + lane;
  // row box held by lane i (broadcast later), column box held by lane j
  float ix1 = 0, iy1 = 0, ix2 = 0, iy2 = 0; int iid = -1;
  if (i_me < nend) {
    const float4 b = *reinterpret_cast<const float4*>(sb + (size_t)i_me * 4);
    ix1 = b.x; iy1 = b.y; ix2 = b.z; iy2 = b.w; iid = si[i_me];
  }
  const float iarea_me = (ix2 - ix1) * (iy2 - iy1);
  const int rows = min(64, nend - (row0 + ci * 64));
  for (int wj = ci + blockIdx.x; wj < nchunks; wj += gridDim.x) {
    const int j_me = row0 + wj * 64 + lane;
    float jx1 = 0, jy1 = 0, jx2 = 0, jy2 = 0; int jid = -2;
    if (j_me < nend) {
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
      // every pair but a box with itself: the diagonal block (wj == ci) is the SYMMETRIC overlap matrix of its 64 rows (IoU is symmetric
      // in fp32 too: max / min / sum commute) -- row j's low bits are the earlier rows that suppress j (nms_reduce_lds_kernel); in the
      // blocks right of it every column is later than every row anyway
      const bool hit = (j_me < nend) && (j_me != row0 + ci * 64 + r) && (aid == jid) && ((double)ovr > thr);
      const u64 word = __ballot(hit);
      if (lane == r) my_word = word;
    }
    if (i_me < nend) mask[((size_t)img * rows_cap + (i_me - row0)) * nwords + wj] = my_word;
  }
  }
}

// Blocked form, blocks after the first: which boxes of the block [row0, row0 + rows_cap) are suppressed by a box KEPT in
// an earlier block?  One wave per 64-column word; the kept boxes (sorted positions kept_pos[0 .. num_keep)) are walked by
// the whole wave (every lane loads the same address: a broadcast), the same fp32 IoU as the mask kernel with the kept
// box as the row box.  Writes the block's initial removed-words.
__global__ __launch_bounds__(256) void nms_cross_kernel(const float* __restrict__ sboxes, const int* __restrict__ sidx,
                                                        const int* __restrict__ counts, int Nmax, int nwords, double thr,
                                                        const int* __restrict__ kept_pos, const int* __restrict__ num_keep,
                                                        int max_keep, int row0, int rows_cap, u64* __restrict__ removed_init) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, img = blockIdx.y;
  if (w >= nwords) return;
  int n = counts ? counts[img] : Nmax;
  if (n > Nmax) n = Nmax;
  const int nend = n < row0 + rows_cap ? n : row0 + rows_cap;
  const int nk = num_keep[img];
  u64* dst = removed_init + (size_t)img * nwords + w;
  if (row0 + w * 64 >= nend || nk >= max_keep) {
    if (lane == 0) *dst = 0ull;
    return;
  }
  const float* sb = sboxes + (size_t)img * Nmax * 4;
  const int* si = sidx + (size_t)img * Nmax;
  const int* kp = kept_pos + (size_t)img * Nmax;
  const int j_me = row0 + w * 64 + lane;
  float jx1 = 0, jy1 = 0, jx2 = 0, jy2 = 0; int jid = -2;
  if (j_me < nend) {
    const float4 b = *reinterpret_cast<const float4*>(sb + (size_t)j_me * 4);
    jx1 = b.x; jy1 = b.y; jx2 = b.z; jy2 = b.w; jid = si[j_me];
  }
  const float jarea = (jx2 - jx1) * (jy2 - jy1);
  bool dead = false;
  for (int i = 0; i < nk; ++i) {
    const int p = kp[i];
    const float4 a = *reinterpret_cast<const float4*>(sb + (size_t)p * 4);
    const int aid = si[p];
    const float aarea = (a.z - a.x) * (a.w - a.y);
    const float xx1 = a.x < jx1 ? jx1 : a.x;
    const float yy1 = a.y < jy1 ? jy1 : a.y;
    const float xx2 = jx2 < a.z ? jx2 : a.z;
    const float yy2 = jy2 < a.w ? jy2 : a.w;
    float ww = xx2 - xx1; if (!(ww > 0.f)) ww = 0.f;
    float hh = yy2 - yy1; if (!(hh > 0.f)) hh = 0.f;
    const float inter = ww * hh;
    const float ovr = inter / (aarea + jarea - inter);
    dead = dead || ((aid == jid) && ((double)ovr > thr));
  }
  const u64 word = __ballot(dead && j_me < nend);
  if (lane == 0) *dst = word;
}

// ---------------------------------------------------------------- 3. ordered reduce
#define NMS_MAX_WORDS 256
// accumulate = 0: the whole image in one pass (row0 = 0).  accumulate = 1 (blocked form): continues the keep list at
// num_keep[img], starts from removed_init's words and records the kept boxes' sorted positions for nms_cross_kernel.
__global__ __launch_bounds__(64) void nms_reduce_kernel(const u64* __restrict__ mask,
                                                        const int* __restrict__ order,
                                                        const int* __restrict__ counts, int Nmax,
                                                        int nwords, int max_keep,
                                                        int* __restrict__ keep,
                                                        int* __restrict__ num_keep, int row0, int rows_cap,
                                                        const u64* __restrict__ removed_init, int* __restrict__ kept_pos,
                                                        int accumulate) {
  __shared__ u64 removed[NMS_MAX_WORDS];
  const int img = blockIdx.x, lane = threadIdx.x;
  int n = counts ? counts[img] : Nmax;
  if (n > Nmax) n = Nmax;
  const int nend = n < row0 + rows_cap ? n : row0 + rows_cap;
  int nk = accumulate ? num_keep[img] : 0;
  if (accumulate && (row0 >= nend || nk >= max_keep)) return;
  const u64* mk = mask + (size_t)img * rows_cap * nwords;
  const int* ord = order + (size_t)img * Nmax;
  int* kp = keep + (size_t)img * Nmax;
  int* kpos = kept_pos ? kept_pos + (size_t)img * Nmax : nullptr;
  for (int w = lane; w < nwords; w += 64) removed[w] = (removed_init && row0 > 0) ? removed_init[(size_t)img * nwords + w] : 0ull;
  __syncthreads();
  const int nchunks = nend > row0 ? (nend - row0 + 63) >> 6 : 0;
  for (int c = 0; c < nchunks && nk < max_keep; ++c) {
    const int row = row0 + c * 64 + lane;
    const u64 diag = row < nend ? mk[(size_t)(row - row0) * nwords + c] : 0ull;
    u64 alive = ~removed[c];
    const int valid = nend - (row0 + c * 64);
    if (valid < 64) alive &= ((1ull << valid) - 1ull);
    // the chunk's greedy pass in rounds (see nms_reduce_lds_kernel: the diagonal block is symmetric, col = the earlier rows of the
    // chunk that overlap this lane's row); `alive` -- the undecided rows -- is the same in every lane
    alive = ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(alive >> 32)) << 32) |
            (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)alive);
    const u64 col = diag & ((1ull << lane) - 1ull);
    u64 kept = 0;
    while (alive) {
      const bool und = (alive >> lane) & 1ull;
      const bool rem = und && (col & kept) != 0ull;
      const bool kp1 = und && !rem && (col & alive) == 0ull;
      const u64 bk = __ballot(kp1), br = __ballot(rem);
      kept |= bk;
      alive &= ~(bk | br);
    }
    // emit kept boxes (original indices) in order
    if ((kept >> lane) & 1ull) {
      const int pos = nk + __popcll(kept & ((1ull << lane) - 1ull));
      if (pos < max_keep) {
        kp[pos] = ord[row];
        if (kpos) kpos[pos] = row;
      }
    }
    nk += __popcll(kept);
    // OR the rows of the kept boxes into the removed words of later chunks
    for (int w0 = c + 1; w0 < nchunks; w0 += 64) {   // words of chunks past n are never read
      const int w = w0 + lane;
      const int wc = w < nwords ? w : nwords - 1;  // clamped: out-of-range lanes load a valid word
      u64 acc = 0;
      u64 k = kept;
      while (k) {  // 24 independent row loads in flight per trip: a chunk's kept boxes (~40 of 64 at threshold 0.7) in two trips
        u64 v[24];
#pragma unroll
        for (int u = 0; u < 24; ++u) {
          const int i = k ? __ffsll((long long)k) - 1 : -1;
          k &= k - 1;  // 0 & anything stays 0
          const u64 x = mk[(size_t)(c * 64 + (i < 0 ? 0 : i)) * nwords + wc];
          v[u] = i < 0 ? 0ull : x;
        }
#pragma unroll
        for (int u = 0; u < 24; ++u) acc |= v[u];
      }
      if (w < nwords) removed[w] |= acc;
    }
    __syncthreads();
  }
  if (lane == 0) num_keep[img] = nk < max_keep ? nk : max_keep;
}

// The same pass for lists of at most 1024 sorted rows (the RPN's per-level lists: 1000 candidates, 40 lists per batch), with the
// bit matrix in LDS.  nms_reduce_kernel walks a list in chunks of 64 rows and pays, per chunk, the global-memory latency of the diagonal
// block and of two batches of kept rows (6.4 us per chunk, 103 us for the RPN lists of a batch -- a third of the proposal stage).  Here
// 256 threads copy the upper triangle (<= 128 KB) into LDS once, then ONE wave runs the chunks: diagonal word and kept rows come from
// LDS, the removed words live in registers (lane w holds word w), the kept rows of a chunk are OR-ed by four lane groups of sixteen
// words.  Same decisions, same output order.
#define NMS_LDS_ROWS 1024
#define NMS_LDS_PITCH 17      // u64 per row: 16 words + 1 (rows 136 B apart: the diagonal read, a row per lane, spreads over the banks)
__global__ __launch_bounds__(256) void nms_reduce_lds_kernel(const u64* __restrict__ mask, const int* __restrict__ order,
                                                             const int* __restrict__ counts, int Nmax, int nwords, int max_keep,
                                                             int* __restrict__ keep, int* __restrict__ num_keep, int rows_cap) {
  __shared__ u64 m[NMS_LDS_ROWS * NMS_LDS_PITCH];      // 139,264 B
  __shared__ int sord[NMS_LDS_ROWS];                   // the rows' original indices: no global load on the serial chain below
  const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  int n = counts ? counts[img] : Nmax;
  if (n > Nmax) n = Nmax;
  if (n > rows_cap) n = rows_cap;
  const u64* mk = mask + (size_t)img * rows_cap * nwords;
  const int* ord = order + (size_t)img * Nmax;
  const int nchunks = (n + 63) >> 6;
  for (int r = tid; r < n; r += 256) sord[r] = ord[r];
  // eight loads in flight per thread; words left of the diagonal (never read) are skipped
  const int total = n * nwords;
  for (int base = tid; base < total; base += 256 * 8) {
    u64 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int idx = base + j * 256;
      v[j] = idx < total ? mk[idx] : 0ull;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int idx = base + j * 256;
      const int r = idx / nwords, w = idx - r * nwords;
      if (idx < total && w >= (r >> 6) && w < nchunks) m[r * NMS_LDS_PITCH + w] = v[j];
    }
  }
  __syncthreads();
  if (tid >= 64) return;
  int* kp = keep + (size_t)img * Nmax;
  u64 removed = 0;      // lane w: the removed word of chunk w
  int nk = 0;
  const int w16 = lane & 15, g4 = lane >> 4;
  for (int c = 0; c < nchunks && nk < max_keep; ++c) {
    const int row = c * 64 + lane;
    const u64 diag = row < n ? m[row * NMS_LDS_PITCH + c] : 0ull;
    const u64 rem_c = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(removed >> 32), c) << 32) |
                      (unsigned)__builtin_amdgcn_readlane((int)(unsigned)removed, c);
    u64 alive = ~rem_c;
    const int valid = n - c * 64;
    if (valid < 64) alive &= ((1ull << valid) - 1ull);
    // The greedy pass over the chunk in ROUNDS instead of row by row (64 dependent steps of ~70 cycles each when most rows survive, as
    // on the RPN's lists).  col = the earlier rows of the chunk that overlap this lane's row (the diagonal block is symmetric).  A row
    // still undecided is removed once a kept row is in col, and kept once col holds neither a kept nor an undecided row; the first
    // undecided row is always decided, so the loop ends, and the decisions are the sequential ones.  Rounds = the longest chain of
    // dependent decisions in the chunk (2 - 4 on scattered boxes), ~15 instructions each.
    const u64 col = diag & ((1ull << lane) - 1ull);
    u64 kept = 0;
    while (alive) {      // `alive` (the undecided rows) and `kept` are wave-uniform
      const bool und = (alive >> lane) & 1ull;
      const bool rem = und && (col & kept) != 0ull;
      const bool kp1 = und && !rem && (col & alive) == 0ull;
      const u64 bk = __ballot(kp1), br = __ballot(rem);
      kept |= bk;
      alive &= ~(bk | br);
    }
    if ((kept >> lane) & 1ull) {
      const int pos = nk + __popcll(kept & ((1ull << lane) - 1ull));
      if (pos < max_keep) kp[pos] = sord[row];
    }
    nk += __popcll(kept);
    if (c + 1 < nchunks) {
      // lane = (row group g4, word w16): group g4 ORs the kept rows g4, g4 + 4, ... of this chunk; rows past n hold stale LDS, their
      // bits are not in `kept`
      u64 acc = 0;
#pragma unroll
      for (int i = 0; i < 64; i += 4) {
        const int r = i + g4;
        const u64 x = m[(c * 64 + r) * NMS_LDS_PITCH + w16];
        acc |= ((kept >> r) & 1ull) ? x : 0ull;
      }
      acc |= (u64)__shfl_xor((long long)acc, 16);
      acc |= (u64)__shfl_xor((long long)acc, 32);
      if (lane < 16 && lane > c) removed |= acc;
    }
  }
  if (lane == 0) num_keep[img] = nk < max_keep ? nk : max_keep;
}

// ---------------------------------------------------------------- large images (Nmax > 16 384): global sort
// The one-workgroup LDS sort above covers 16 384 keys.  Beyond that the same 64-bit keys are sorted with a bitonic
// network split the usual way: compare-exchange distances below NMS_TILE stay inside an LDS tile (one launch per merge
// level), the larger distances are one global pass each.  Keys of absent rows are ~0 and sort to the end, so a stage
// whose lower half already holds every real key of the image has nothing to do and returns.
__device__ __forceinline__ unsigned int ordered_asc_key(float f) {
  unsigned int u = __float_as_uint(f);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float ordered_asc_value(unsigned int u) {
  u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
  return __uint_as_float(u);
}

__global__ __launch_bounds__(256) void nms_max_kernel(const float* __restrict__ boxes, const int* __restrict__ counts,
                                                      int Nmax, unsigned int* __restrict__ maxkey) {
  const int img = blockIdx.y;
  int n = counts ? counts[img] : Nmax;
  if (n > Nmax) n = Nmax;
  const float* bx = boxes + (size_t)img * Nmax * 4;
  float m = -INFINITY;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n * 4; i += gridDim.x * 256) {
    const float v = bx[i];
    m = v > m ? v : m;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float t = __shfl_xor(m, o);
    m = t > m ? t : m;
  }
  if ((threadIdx.x & 63) == 0) atomicMax(maxkey + img, ordered_asc_key(m));
}

// mode 0: build the tile's keys and run every stage k = 2 .. NMS_TILE; mode 1: the distances below NMS_TILE of level k
__global__ __launch_bounds__(1024) void nms_sort_tile_kernel(const float* __restrict__ scores, const int* __restrict__ counts,
                                                             int Nmax, int npad, u64* __restrict__ keys, int mode, int klevel) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  u64* tk = reinterpret_cast<u64*>(dsm);
  const int img = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
  int n = counts ? counts[img] : Nmax;
  if (n > Nmax) n = Nmax;
  if (mode == 1 && n <= (klevel >> 1)) return;
  const int base = tile * NMS_TILE;
  u64* gk = keys + (size_t)img * npad + base;
  if (mode == 0) {
    const float* sc = scores + (size_t)img * Nmax;
    for (int i = tid; i < NMS_TILE; i += 1024) {
      const int gi = base + i;
      tk[i] = gi < n ? (((u64)ordered_desc_key(sc[gi]) << 32) | (unsigned)gi) : ~0ull;
    }
  } else {
    for (int i = tid; i < NMS_TILE; i += 1024) tk[i] = gk[i];
  }
  __syncthreads();
  const int k0 = mode == 0 ? 2 : klevel, k1 = mode == 0 ? NMS_TILE : klevel;
  for (int k = k0; k <= k1; k <<= 1) {
    for (int j = (k >> 1) < NMS_TILE ? (k >> 1) : (NMS_TILE >> 1); j > 0; j >>= 1) {
      for (int t = tid; t < NMS_TILE / 2; t += 1024) {
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int hi = lo | j;
        const bool up = ((base + lo) & k) == 0;
        const u64 a = tk[lo], b = tk[hi];
        if ((a > b) == up) { tk[lo] = b; tk[hi] = a; }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < NMS_TILE; i += 1024) gk[i] = tk[i];
}

__global__ __launch_bounds__(256) void nms_sort_global_kernel(const int* __restrict__ counts, int Nmax, int npad,
                                                              u64* __restrict__ keys, int k, int j) {
  const int img = blockIdx.y;
  int n = counts ? counts[img] : Nmax;
  if (n > Nmax) n = Nmax;
  if (n <= (k >> 1)) return;
  u64* gk = keys + (size_t)img * npad;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < npad / 2; t += gridDim.x * 256) {
    const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
    const int hi = lo | j;
    const bool up = (lo & k) == 0;
    const u64 a = gk[lo], b = gk[hi];
    if ((a > b) == up) { gk[lo] = b; gk[hi] = a; }
  }
}

__global__ __launch_bounds__(256) void nms_gather_kernel(const float* __restrict__ boxes, const int* __restrict__ idxs,
                                                         const int* __restrict__ counts, int Nmax, int npad,
                                                         const u64* __restrict__ keys, const unsigned int* __restrict__ maxkey,
                                                         int* __restrict__ order, float* __restrict__ sboxes,
                                                         int* __restrict__ sidx) {
  const int img = blockIdx.y;
  int n = counts ? counts[img] : Nmax;
  if (n > Nmax) n = Nmax;
  const float maxp1 = ordered_asc_value(maxkey[img]) + 1.0f;
  const float* bx = boxes + (size_t)img * Nmax * 4;
  const int* ix = idxs ? idxs + (size_t)img * Nmax : nullptr;
  const u64* gk = keys + (size_t)img * npad;
  int* ord = order + (size_t)img * Nmax;
  float* sb = sboxes + (size_t)img * Nmax * 4;
  int* si = sidx + (size_t)img * Nmax;
  for (int r = blockIdx.x * 256 + threadIdx.x; r < n; r += gridDim.x * 256) {
    const int i = (int)(gk[r] & 0xFFFFFFFFu);
    ord[r] = i;
    const int id = ix ? ix[i] : 0;
    const float off = (float)id * maxp1;
    sb[(size_t)r * 4 + 0] = bx[(size_t)i * 4 + 0] + off;
    sb[(size_t)r * 4 + 1] = bx[(size_t)i * 4 + 1] + off;
    sb[(size_t)r * 4 + 2] = bx[(size_t)i * 4 + 2] + off;
    sb[(size_t)r * 4 + 3] = bx[(size_t)i * 4 + 3] + off;
    si[r] = id;
  }
}

static int pad_pow2(int n) {
  int p = 1024;
  while (p < n) p <<= 1;
  return p;
}

static long long pad_pow2_ll(long long n) {
  long long p = NMS_TILE;
  while (p < n) p <<= 1;
  return p;
}

extern "C" long long lvc_batched_nms_workspace_bytes(int B, int Nmax) {
  const long long rows = Nmax < NMS_TILE ? Nmax : NMS_TILE;   // rows of one mask block
  const long long nwords = (rows + 63) / 64;
  long long per = (long long)Nmax * 4 * 4 /*sboxes*/ + (long long)Nmax * 4 /*order*/ +
                  (long long)Nmax * 4 /*sidx*/ + rows * nwords * 8 /*mask (one block)*/;
  if (Nmax > NMS_TILE)   // blocked form: kept positions, initial removed words, global sort keys, max key
    per += (long long)Nmax * 4 + NMS_MAX_WORDS * 8 + pad_pow2_ll(Nmax) * 8 + 16;
  else if (Nmax > NMS_HEAD_MIN)   // head-block form (max_keep << Nmax): kept positions, initial removed words
    per += (long long)Nmax * 4 + NMS_MAX_WORDS * 8 + 16;
  return (long long)B * per + 512;
}

// boxes [B,Nmax,4] fp32 xyxy, scores [B,Nmax] fp32, idxs [B,Nmax] int32 or NULL, d_counts [B] device int32
// or NULL (= Nmax each).  keep [B,Nmax] int32 (indices into the image's Nmax rows, score-descending),
// d_num_keep [B] device int32.  At most max_keep (<=0: all) indices are produced per image.
// Nmax <= 16 384: one sort workgroup, one mask launch, one reduce launch per call.  Larger Nmax (no limit; the
// reference switches to a per-class loop at 40 000 boxes, detectron2/layers/nms.py:22-29, with the same result): global
// bitonic sort, then the greedy pass in blocks of 16 384 sorted rows -- suppression by boxes kept in earlier blocks
// (nms_cross_kernel), mask + reduce inside the block; images whose count fits the first block skip the rest.
// 1: lists of <= NMS_LDS_ROWS rows on nms_reduce_kernel (bit matrix read from global memory) instead of nms_reduce_lds_kernel -- the A/B
// form of tests/test_gpu_kernels.py::test_nms_reduce_from_lds_equals_the_global_form_and_the_oracle (no environment lookup per launch)
static int g_nms_reduce_global = 0;
extern "C" void lvc_set_nms_reduce_global(int on) { g_nms_reduce_global = on; }

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
  LVC_CHECK_ARG(workspace_bytes >= lvc_batched_nms_workspace_bytes(B, Nmax), "workspace too small");
  LVC_CHECK_ARG(((uintptr_t)workspace & 15) == 0, "workspace must be 16-byte aligned");
  const int rows_cap = Nmax < NMS_TILE ? Nmax : NMS_TILE;
  const int nwords = (rows_cap + 63) / 64;
  char* ws = (char*)workspace;
  float* sboxes = (float*)ws; ws += (size_t)B * Nmax * 16;
  int* order = (int*)ws; ws += (size_t)B * Nmax * 4;
  int* sidx = (int*)ws; ws += (size_t)B * Nmax * 4;
  ws = (char*)(((uintptr_t)ws + 15) & ~(uintptr_t)15);
  u64* mask = (u64*)ws; ws += (size_t)B * rows_cap * nwords * 8;
  if (max_keep <= 0) max_keep = Nmax;
  const dim3 mask_grid(nwords < NMS_MASK_STRIDE ? nwords : NMS_MASK_STRIDE, nwords, B);

  if (Nmax <= NMS_TILE) {
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
    // Head-block form.  The detection stage asks for the 100 best of ~10 000 sorted candidates per image: the greedy pass ends
    // as soon as max_keep boxes are kept, typically inside the first few hundred rows, but the mask kernel used to evaluate the
    // whole upper triangle (thousands of 64 x 64 blocks per image).  Here the pass runs on the first `head` sorted rows alone
    // (mask + reduce over a 1024-row block); the rest of the list follows as a second block of the blocked form below --
    // suppression by the boxes kept so far (nms_cross_kernel), mask, reduce -- whose kernels return at once for every image
    // that already has its max_keep boxes.  Same kernels, same decisions as the one-pass form.
    int head = ((8 * max_keep + 63) / 64) * 64;
    if (head < 1024) head = 1024;
    if (Nmax > NMS_HEAD_MIN && head * 2 <= Nmax) {
      int* kept_pos = (int*)ws; ws += (size_t)B * Nmax * 4;
      ws = (char*)(((uintptr_t)ws + 15) & ~(uintptr_t)15);
      u64* removed_init = (u64*)ws;
      if (hipMemsetAsync(d_num_keep, 0, sizeof(int) * B, st) != hipSuccess) {
        lvc_set_error("%s: hipMemsetAsync failed", __func__);
        return LVC_ERR_HIP;
      }
      const int hw = head / 64;
      hipLaunchKernelGGL(nms_mask_kernel, dim3(hw < NMS_MASK_STRIDE ? hw : NMS_MASK_STRIDE, hw, B), dim3(64), 0, st, sboxes, sidx,
                         d_counts, Nmax, hw, iou_threshold, mask, 0, head, (const int*)nullptr, max_keep);
      LVC_CHECK_LAUNCH();
      hipLaunchKernelGGL(nms_reduce_kernel, dim3(B), dim3(64), 0, st, mask, order, d_counts, Nmax, hw, max_keep, keep,
                         d_num_keep, 0, head, removed_init, kept_pos, 1);
      LVC_CHECK_LAUNCH();
      const int rest = Nmax - head, rw = (rest + 63) / 64;
      hipLaunchKernelGGL(nms_cross_kernel, dim3(lvc_cdiv(rw, 4), B), dim3(256), 0, st, sboxes, sidx, d_counts, Nmax, rw,
                         iou_threshold, kept_pos, d_num_keep, max_keep, head, rest, removed_init);
      LVC_CHECK_LAUNCH();
      hipLaunchKernelGGL(nms_mask_kernel, dim3(rw < NMS_MASK_STRIDE ? rw : NMS_MASK_STRIDE, rw < 16 ? rw : 16, B), dim3(64), 0, st, sboxes, sidx,
                         d_counts, Nmax, rw, iou_threshold, mask, head, rest, (const int*)d_num_keep, max_keep);
      LVC_CHECK_LAUNCH();
      hipLaunchKernelGGL(nms_reduce_kernel, dim3(B), dim3(64), 0, st, mask, order, d_counts, Nmax, rw, max_keep, keep,
                         d_num_keep, head, rest, removed_init, kept_pos, 1);
      LVC_CHECK_LAUNCH();
      return LVC_OK;
    }
    hipLaunchKernelGGL(nms_mask_kernel, mask_grid, dim3(64), 0, st, sboxes, sidx, d_counts, Nmax, nwords, iou_threshold,
                       mask, 0, rows_cap, (const int*)nullptr, max_keep);
    LVC_CHECK_LAUNCH();
    if (rows_cap <= NMS_LDS_ROWS && !g_nms_reduce_global)
      hipLaunchKernelGGL(nms_reduce_lds_kernel, dim3(B), dim3(256), 0, st, mask, order, d_counts, Nmax, nwords, max_keep, keep, d_num_keep,
                         rows_cap);
    else
      hipLaunchKernelGGL(nms_reduce_kernel, dim3(B), dim3(64), 0, st, mask, order, d_counts, Nmax, nwords, max_keep, keep,
                         d_num_keep, 0, rows_cap, (const u64*)nullptr, (int*)nullptr, 0);
    LVC_CHECK_LAUNCH();
    return LVC_OK;
  }

  // ---- Nmax > 16 384
  int* kept_pos = (int*)ws; ws += (size_t)B * Nmax * 4;
  ws = (char*)(((uintptr_t)ws + 15) & ~(uintptr_t)15);
  u64* removed_init = (u64*)ws; ws += (size_t)B * NMS_MAX_WORDS * 8;
  const long long npad = pad_pow2_ll(Nmax);
  LVC_CHECK_ARG(npad <= (1ll << 30), "more than 2^30 boxes per image");
  u64* keys = (u64*)ws; ws += (size_t)B * npad * 8;
  unsigned int* maxkey = (unsigned int*)ws;
  if (hipMemsetAsync(maxkey, 0, sizeof(unsigned int) * B, st) != hipSuccess ||
      hipMemsetAsync(d_num_keep, 0, sizeof(int) * B, st) != hipSuccess) {
    lvc_set_error("%s: hipMemsetAsync failed", __func__);
    return LVC_ERR_HIP;
  }
  hipLaunchKernelGGL(nms_max_kernel, dim3(64, B), dim3(256), 0, st, boxes, d_counts, Nmax, maxkey);
  LVC_CHECK_LAUNCH();
  const size_t lds = (size_t)NMS_TILE * 8;
  (void)hipFuncSetAttribute((const void*)nms_sort_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int ntiles = (int)(npad / NMS_TILE);
  hipLaunchKernelGGL(nms_sort_tile_kernel, dim3(ntiles, B), dim3(1024), lds, st, scores, d_counts, Nmax, (int)npad, keys, 0, 0);
  LVC_CHECK_LAUNCH();
  for (long long k = 2ll * NMS_TILE; k <= npad; k <<= 1) {
    for (long long j = k >> 1; j >= NMS_TILE; j >>= 1) {
      hipLaunchKernelGGL(nms_sort_global_kernel, dim3(256, B), dim3(256), 0, st, d_counts, Nmax, (int)npad, keys, (int)k, (int)j);
      LVC_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(nms_sort_tile_kernel, dim3(ntiles, B), dim3(1024), lds, st, scores, d_counts, Nmax, (int)npad, keys, 1, (int)k);
    LVC_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(nms_gather_kernel, dim3(64, B), dim3(256), 0, st, boxes, idxs, d_counts, Nmax, (int)npad, keys, maxkey,
                     order, sboxes, sidx);
  LVC_CHECK_LAUNCH();
  for (int row0 = 0; row0 < Nmax; row0 += NMS_TILE) {
    if (row0 > 0) {
      hipLaunchKernelGGL(nms_cross_kernel, dim3(lvc_cdiv(nwords, 4), B), dim3(256), 0, st, sboxes, sidx, d_counts, Nmax,
                         nwords, iou_threshold, kept_pos, d_num_keep, max_keep, row0, rows_cap, removed_init);
      LVC_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(nms_mask_kernel, mask_grid, dim3(64), 0, st, sboxes, sidx, d_counts, Nmax, nwords, iou_threshold, mask,
                       row0, rows_cap, (const int*)d_num_keep, max_keep);
    LVC_CHECK_LAUNCH();
    hipLaunchKernelGGL(nms_reduce_kernel, dim3(B), dim3(64), 0, st, mask, order, d_counts, Nmax, nwords, max_keep, keep,
                       d_num_keep, row0, rows_cap, removed_init, kept_pos, 1);
    LVC_CHECK_LAUNCH();
  }
  return LVC_OK;
}
