// knn.hip -- label-verification kNN (reference tools/run_nearest_neighbours.py:142-162 + :214-227).
//
// The reference loops over query images on the CPU and materialises a [q_i, S, D] broadcast per image for
// F.cosine_similarity, then topk(10), class gather and torch.mode.  Here the whole sweep is three kinds of
// launches over all queries at once:
//   1. lvc_colmean + lvc_rownorm (elementwise.hip): mu = shots.mean(0); rows (x - mu) / max(|x - mu|, 1e-8)
//   2. sims = Qn . Sn^T on the fp32-MFMA GEMM (conv_igemm.hip; 589.8 GFLOP for 120k x 2400 x 1024)
//   3. knn_topk_vote_kernel (this file): one wave per query row, the S similarities in registers: the 10th largest
//      lane maximum bounds the answer from below, the few values above it are compacted into LDS and ranked against
//      each other (ties -> lower shot index), no cross-lane reduction chains; shot_classes gather, majority vote
//      with torch.mode's tie rule (smallest class id), keep = (vote == detector class).
#include "common.h"

// mu[d] = mean over rows.  One workgroup sums a 64-row slab for 256 columns; slabs are combined with fp32 atomics
// into the zeroed mu, already divided by M (the serial one-thread-per-column form took 0.9 ms for 2400 x 1024).
__global__ __launch_bounds__(256) void colmean_kernel(const float* __restrict__ x, float* __restrict__ mu, int M, int D,
                                                      int ld, float inv_m) {
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D) return;
  const int r0 = blockIdx.y * 64;
  int r1 = r0 + 64;
  if (r1 > M) r1 = M;
  float s = 0.f;
  for (int m = r0; m < r1; ++m) s += x[(size_t)m * ld + d];
  unsafeAtomicAdd(mu + d, s * inv_m);
}

extern "C" int lvc_colmean(const float* x, float* mu, int M, int D, int ld, void* stream) {
  LVC_CHECK_ARG(x && mu && M > 0 && D > 0, "bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(mu, 0, (size_t)D * sizeof(float), st) != hipSuccess) {
    lvc_set_error("%s: hipMemsetAsync failed", __func__);
    return LVC_ERR_HIP;
  }
  hipLaunchKernelGGL(colmean_kernel, dim3(lvc_cdiv(D, 256), lvc_cdiv(M, 64)), dim3(256), 0, st, x, mu, M, D, ld > 0 ? ld : D,
                     1.f / (float)M);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

#define KNN_MAX_PER_LANE 64  // S <= 4096
// One wave per query row, the row's S similarities in PER registers per lane.  No cross-lane reduction chains:
//   1. every lane publishes its maximum; each lane RANKS its maximum among the 64 (a loop of broadcast LDS reads);
//      the maximum of rank KTOP-1 is T, a lower bound of the row's KTOP-th largest value;
//   2. the values >= T (a few dozen of S) are compacted with ballots into (value, index) pairs in LDS;
//   3. every candidate is ranked among the candidates (value descending, ties -> lower shot index); ranks < KTOP
//      write their shot's class straight to the output slot of that rank;
//   4. lane 0 takes the torch.mode vote over the first kvote classes (ties -> smallest class id).
// More than 256 candidates (a row of near-constant similarities) falls back to KTOP arg-max rounds over the row.
#define KNN_MAX_CAND 256
template <int KTOP, int PER>
__global__ __launch_bounds__(256) void knn_topk_vote_kernel(const float* __restrict__ sims, int ld, int Q, int S,
                                                            const long long* __restrict__ shot_classes,
                                                            const long long* __restrict__ det_classes, int kvote,
                                                            long long* __restrict__ top_classes,
                                                            long long* __restrict__ keep) {
  __shared__ float s_lmax[4][64];
  __shared__ float s_cval[4][KNN_MAX_CAND];
  __shared__ int s_cidx[4][KNN_MAX_CAND];
  __shared__ float s_T[4];
  __shared__ long long s_cls[4][KTOP];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + w;
  if (row >= Q) return;     // whole waves leave; nothing below synchronises across waves
  const float* sr = sims + (size_t)row * ld;
  float v[PER];
  float lmax = -INFINITY;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = j * 64 + lane;
    v[j] = i < S ? sr[i] : -INFINITY;
    if (v[j] != v[j]) v[j] = INFINITY;   // NaN similarity: torch.topk ranks NaN above every number; ties -> lower index
    lmax = fmaxf(lmax, v[j]);
  }
  // ---- 1. T = the lane maximum of rank KTOP-1
  s_lmax[w][lane] = lmax;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  int rank = 0;
#pragma unroll 8
  for (int l = 0; l < 64; ++l) {
    const float o = s_lmax[w][l];
    rank += (o > lmax || (o == lmax && l < lane)) ? 1 : 0;
  }
  if (rank == KTOP - 1) s_T[w] = lmax;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  const float T = s_T[w];
  // ---- 2. compact the candidates
  int total = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const bool is_c = v[j] >= T && v[j] > -INFINITY;
    const unsigned long long m = __ballot(is_c);
    if (m) {
      const int pos = total + __popcll(m & ((1ull << lane) - 1ull));
      if (is_c && pos < KNN_MAX_CAND) { s_cval[w][pos] = v[j]; s_cidx[w][pos] = j * 64 + lane; }
      total += __popcll(m);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  if (total <= KNN_MAX_CAND && T > -INFINITY) {
    // ---- 3. rank every candidate among the candidates
    for (int c0 = 0; c0 < total; c0 += 64) {
      const int c = c0 + lane;
      const float mv = c < total ? s_cval[w][c] : -INFINITY;
      const int mi = c < total ? s_cidx[w][c] : 0x7fffffff;
      int r = 0;
      for (int l = 0; l < total; ++l) {
        const float o = s_cval[w][l];
        const int oi = s_cidx[w][l];
        r += (o > mv || (o == mv && oi < mi)) ? 1 : 0;
      }
      if (c < total && r < KTOP) {
        const long long cl = shot_classes[mi];
        s_cls[w][r] = cl;
        top_classes[(size_t)row * KTOP + r] = cl;
      }
    }
  } else {
    for (int r = 0; r < KTOP; ++r) {
      float best = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const int i = j * 64 + lane;
        if (v[j] > best) { best = v[j]; bi = i; }  // ascending i within a lane: first max = lowest index
      }
      for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
#pragma unroll
      for (int j = 0; j < PER; ++j)
        if (j * 64 + lane == bi) v[j] = -INFINITY;   // owner lane retires the winner
      if (lane == 0) {
        const long long cl = (bi < S) ? shot_classes[bi] : -1;
        s_cls[w][r] = cl;
        top_classes[(size_t)row * KTOP + r] = cl;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  if (lane == 0 && keep) {
    // torch.mode over the first kvote votes: most frequent value, ties -> smallest value
    long long mode = -1; int mcount = 0;
    for (int a = 0; a < kvote; ++a) {
      int c = 0;
      for (int b = 0; b < kvote; ++b) c += (s_cls[w][b] == s_cls[w][a]);
      if (c > mcount || (c == mcount && s_cls[w][a] < mode)) { mcount = c; mode = s_cls[w][a]; }
    }
    keep[row] = (det_classes && det_classes[row] == mode) ? 1 : 0;
  }
}

// sims [Q, ld] fp32 (S used columns); shot_classes [S] int64; det_classes [Q] int64 or NULL;
// top_classes [Q,10] int64 (class ids of the 10 most similar shots, most similar first); keep [Q] int64 or NULL.
extern "C" int lvc_knn_topk_vote(const float* sims, int ld, int Q, int S, const long long* shot_classes,
                                 const long long* det_classes, int kvote, long long* top_classes, long long* keep,
                                 void* stream) {
  LVC_CHECK_ARG(Q >= 0 && S >= 10, "need at least 10 shots");
  if (Q == 0) return LVC_OK;
  LVC_CHECK_ARG(sims && shot_classes && top_classes, "null pointer");
  LVC_CHECK_ARG(S <= 64 * KNN_MAX_PER_LANE, "at most 4096 shots per call");
  LVC_CHECK_ARG(kvote >= 1 && kvote <= 10, "k must be in 1..10 (the reference stores top-10)");
  const int per = lvc_cdiv(S, 64);
  const dim3 grid(lvc_cdiv(Q, 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int ldd = ld > 0 ? ld : S;
#define KNN_LAUNCH(P) hipLaunchKernelGGL((knn_topk_vote_kernel<10, P>), grid, block, 0, st, sims, ldd, Q, S, shot_classes, \
                                         det_classes, kvote, top_classes, keep)
  if (per <= 8) KNN_LAUNCH(8);
  else if (per <= 16) KNN_LAUNCH(16);
  else if (per <= 24) KNN_LAUNCH(24);
  else if (per <= 32) KNN_LAUNCH(32);
  else if (per <= 40) KNN_LAUNCH(40);
  else if (per <= 48) KNN_LAUNCH(48);
  else KNN_LAUNCH(64);
#undef KNN_LAUNCH
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}
