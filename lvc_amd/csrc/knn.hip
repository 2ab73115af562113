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


// ---------------------------------------------------------------------------------------------------------------------
// Two-stage exact top-10: the [Q, S] matrix handed in is an APPROXIMATION of the similarities (lvc_gemm_f16_hi_dma: every
// operand rounded to fp16, one MFMA per block instead of three) with |approx - exact| <= eps for unit-norm rows
// (eps = 2^-10 * sum |q_i s_i| <= 2^-10 by Cauchy-Schwarz, plus the fp32 accumulation, ~1e-6).  Let A10 be the 10th largest
// approximate value of a row.  The ten best approximate shots have exact similarity >= A10 - eps, so the exact 10th best
// x10 >= A10 - eps, and every shot of the exact top ten has approx >= x10 - eps >= A10 - 2 eps: the shots with
// approx >= A10 - margin (margin >= 2 eps) CONTAIN the exact top ten.  Only those (a dozen or so on uncorrelated
// descriptors) are re-evaluated in fp32 -- lane l sums elements l*4 .. l*4+3 of every 256-element slice in order, then a
// butterfly over the lanes: a fixed order -- and ranked (value descending, ties -> lower shot index); classes and the vote as
// in knn_topk_vote_kernel.  A row with more than KV_MAX_CAND candidates (a cloud of near-identical shots) evaluates ALL
// shots exactly, KV_MAX_CAND at a time, keeping the running ten best.
#define KV_MAX_CAND 192
template <int KTOP, int PER>
__global__ __launch_bounds__(256) void knn_verify_topk_vote_kernel(const float* __restrict__ approx, int ld, int Q, int S,
                                                                   const float* __restrict__ qn, const float* __restrict__ sn,
                                                                   int D, float margin, const long long* __restrict__ shot_classes,
                                                                   const long long* __restrict__ det_classes, int kvote,
                                                                   long long* __restrict__ top_classes, long long* __restrict__ keep) {
  __shared__ float s_lmax[4][64];
  __shared__ float s_val[4][KV_MAX_CAND + KTOP];     // exact similarity of the candidates (+ the running best of the slow path)
  __shared__ int s_idx[4][KV_MAX_CAND + KTOP];
  __shared__ float s_T[4];
  __shared__ long long s_cls[4][KTOP];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + w;
  if (row >= Q) return;     // whole waves leave; nothing below synchronises across waves
  const float* ar = approx + (size_t)row * ld;
  float v[PER];
  float lmax = -INFINITY;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = j * 64 + lane;
    v[j] = i < S ? ar[i] : -INFINITY;
    if (v[j] != v[j]) v[j] = INFINITY;
    lmax = fmaxf(lmax, v[j]);
  }
  // ---- A10: exact 10th largest approximate value.  First a lower bound T (the lane maximum of rank KTOP-1), then the rank of
  // every value >= T among those values
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
  int total = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const bool is_c = v[j] >= T;
    const unsigned long long m = __ballot(is_c);
    if (m) {
      const int pos = total + __popcll(m & ((1ull << lane) - 1ull));
      if (is_c && pos < KV_MAX_CAND) s_val[w][pos] = v[j];
      total += __popcll(m);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  float A10 = T;      // more than KV_MAX_CAND values >= T (near-constant row): T itself is a valid lower bound of A10
  if (total <= KV_MAX_CAND) {
    for (int c0 = 0; c0 < total; c0 += 64) {
      const int c = c0 + lane;
      const float mv = c < total ? s_val[w][c] : -INFINITY;
      int r = 0;
      for (int l = 0; l < total; ++l) {
        const float o = s_val[w][l];
        r += (o > mv || (o == mv && l < c)) ? 1 : 0;
      }
      if (c < total && r == KTOP - 1) s_T[w] = mv;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    A10 = s_T[w];
  }
  const float Tv = A10 - margin;
  // the query row in registers: lane l holds elements s*256 + l*4 .. +3 of every 256-element slice
  const float* qr = qn + (size_t)row * D;
  float4 qv[8];
  // D % 4 == 0, D <= 2048: lane l owns elements sl*256 + l*4 .. +3 of slice sl (where below D)
#pragma unroll
  for (int sl = 0; sl < 8; ++sl)
    qv[sl] = sl * 256 + lane * 4 < D ? *reinterpret_cast<const float4*>(qr + sl * 256 + lane * 4) : float4{0.f, 0.f, 0.f, 0.f};
  auto exact_dot = [&](int si) {
    const float* sr = sn + (size_t)si * D;
    float acc = 0.f;
#pragma unroll
    for (int sl = 0; sl < 8; ++sl)
      if (sl * 256 + lane * 4 < D) {
        const float4 sv = *reinterpret_cast<const float4*>(sr + sl * 256 + lane * 4);
        acc += qv[sl].x * sv.x; acc += qv[sl].y * sv.y; acc += qv[sl].z * sv.z; acc += qv[sl].w * sv.w;
      }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    return acc;
  };
  // rank the first n entries of (s_val, s_idx) (value descending, ties -> lower shot index) and keep the KTOP best in place
  auto keep_best = [&](int n) {
    for (int c0 = 0; c0 < n; c0 += 64) {
      const int c = c0 + lane;
      const float mv = c < n ? s_val[w][c] : -INFINITY;
      const int mi = c < n ? s_idx[w][c] : 0x7fffffff;
      int r = 0;
      for (int l = 0; l < n; ++l) {
        const float o = s_val[w][l];
        const int oi = s_idx[w][l];
        r += (o > mv || (o == mv && oi < mi)) ? 1 : 0;
      }
      // winners park in the tail slots [KV_MAX_CAND + rank] (ranks are unique, the loops above never read the tail)
      if (c < n && r < KTOP) { s_val[w][KV_MAX_CAND + r] = mv; s_idx[w][KV_MAX_CAND + r] = mi; }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    const int kept = n < KTOP ? n : KTOP;
    if (lane < kept) { s_val[w][lane] = s_val[w][KV_MAX_CAND + lane]; s_idx[w][lane] = s_idx[w][KV_MAX_CAND + lane]; }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    return kept;
  };
  // ---- candidates: approx >= A10 - margin
  int ncand = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const bool is_c = v[j] >= Tv && v[j] > -INFINITY;
    const unsigned long long m = __ballot(is_c);
    if (m) {
      const int pos = ncand + __popcll(m & ((1ull << lane) - 1ull));
      if (is_c && pos < KV_MAX_CAND) s_idx[w][pos] = j * 64 + lane;
      ncand += __popcll(m);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  int nbest;
  if (ncand <= KV_MAX_CAND) {
    for (int c = 0; c < ncand; ++c) {
      const float e = exact_dot(s_idx[w][c]);
      if (lane == 0) s_val[w][c] = e;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    nbest = keep_best(ncand);
  } else {
    // slow path: every shot exactly, in blocks of KV_MAX_CAND - KTOP next to the running best
    nbest = 0;
    for (int s0 = 0; s0 < S; s0 += KV_MAX_CAND - KTOP) {
      const int nb = min(KV_MAX_CAND - KTOP, S - s0);
      for (int c = 0; c < nb; ++c) {
        const float e = exact_dot(s0 + c);
        if (lane == 0) { s_val[w][nbest + c] = e; s_idx[w][nbest + c] = s0 + c; }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      nbest = keep_best(nbest + nb);
    }
  }
  if (lane < KTOP) {
    const long long cl = lane < nbest ? shot_classes[s_idx[w][lane]] : -1;
    s_cls[w][lane] = cl;
    top_classes[(size_t)row * KTOP + lane] = cl;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  if (lane == 0 && keep) {
    long long mode = -1; int mcount = 0;
    for (int a = 0; a < kvote; ++a) {
      int c = 0;
      for (int b = 0; b < kvote; ++b) c += (s_cls[w][b] == s_cls[w][a]);
      if (c > mcount || (c == mcount && s_cls[w][a] < mode)) { mcount = c; mode = s_cls[w][a]; }
    }
    keep[row] = (det_classes && det_classes[row] == mode) ? 1 : 0;
  }
}

// approx [Q, ld] (S used columns) from lvc_gemm_f16_hi_dma over the SAME normalised rows qn [Q, D] / sn [S, D] (fp32, D % 4 == 0,
// D <= 2048); margin >= 2 x the approximation error bound (2^-9 for unit-norm rows is what the host passes).
extern "C" int lvc_knn_verify_topk_vote(const float* approx, int ld, int Q, int S, const float* qn, const float* sn, int D,
                                        float margin, const long long* shot_classes, const long long* det_classes, int kvote,
                                        long long* top_classes, long long* keep, void* stream) {
  LVC_CHECK_ARG(Q >= 0 && S >= 10, "need at least 10 shots");
  if (Q == 0) return LVC_OK;
  LVC_CHECK_ARG(approx && qn && sn && shot_classes && top_classes, "null pointer");
  LVC_CHECK_ARG(S <= 64 * KNN_MAX_PER_LANE, "at most 4096 shots per call");
  LVC_CHECK_ARG(D % 4 == 0 && D <= 2048 && D > 0, "descriptor length must be a multiple of 4, at most 2048");
  LVC_CHECK_ARG(kvote >= 1 && kvote <= 10 && margin >= 0.f, "k must be in 1..10, margin >= 0");
  LVC_CHECK_ARG((((uintptr_t)qn | (uintptr_t)sn) & 15) == 0, "descriptor rows must be 16-byte aligned");
  const int per = lvc_cdiv(S, 64);
  const dim3 grid(lvc_cdiv(Q, 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int ldd = ld > 0 ? ld : S;
#define KV_LAUNCH(P) hipLaunchKernelGGL((knn_verify_topk_vote_kernel<10, P>), grid, block, 0, st, approx, ldd, Q, S, qn, sn, D, margin, \
                                        shot_classes, det_classes, kvote, top_classes, keep)
  if (per <= 8) KV_LAUNCH(8);
  else if (per <= 16) KV_LAUNCH(16);
  else if (per <= 24) KV_LAUNCH(24);
  else if (per <= 32) KV_LAUNCH(32);
  else if (per <= 40) KV_LAUNCH(40);
  else if (per <= 48) KV_LAUNCH(48);
  else KV_LAUNCH(64);
#undef KV_LAUNCH
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}
