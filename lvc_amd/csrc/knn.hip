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
#include <stdlib.h>

// mu[d] = mean over rows, DETERMINISTIC (the reference's shots.mean(0) is; an atomic combine made mu -- and with it near-tied
// ranks -- vary in the last bits from run to run).  A workgroup of 1024 threads owns 32 columns: thread (g, c) adds rows g, g + 32,
// g + 64, ... of column c in that order (128-byte row segments per 32 lanes; five loads in flight per thread), the 32 partial sums
// of a column are then added in the fixed order g = 0..31 by one thread.  2400 x 1024: 32 workgroups, ~10 us.
__global__ __launch_bounds__(1024) void colmean_kernel(const float* __restrict__ x, float* __restrict__ mu, int M, int D,
                                                       int ld, float inv_m) {
  __shared__ float part[32][33];
  const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int d = blockIdx.x * 32 + c;
  float s = 0.f;
  if (d < D) {
    int m = g;
    for (; m + 128 < M; m += 160) {
      float t[5];
#pragma unroll
      for (int i = 0; i < 5; ++i) t[i] = x[(size_t)(m + 32 * i) * ld + d];
#pragma unroll
      for (int i = 0; i < 5; ++i) s += t[i];
    }
    for (; m < M; m += 32) s += x[(size_t)m * ld + d];
  }
  part[g][c] = s;
  __syncthreads();
  if (g == 0 && d < D) {
    float t = part[0][c];
#pragma unroll
    for (int i = 1; i < 32; ++i) t += part[i][c];
    mu[d] = t * inv_m;
  }
}

extern "C" int lvc_colmean(const float* x, float* mu, int M, int D, int ld, void* stream) {
  LVC_CHECK_ARG(x && mu && M > 0 && D > 0, "bad arguments");
  hipLaunchKernelGGL(colmean_kernel, dim3(lvc_cdiv(D, 32)), dim3(1024), 0, (hipStream_t)stream, x, mu, M, D, ld > 0 ? ld : D,
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
                                                            long long* __restrict__ keep, float* __restrict__ cand_val,
                                                            int* __restrict__ cand_idx, int idx_base) {
  // cand_val / cand_idx != NULL (shot sets beyond one launch: lvc_knn_topk_candidates): the KTOP best (value, idx_base + index)
  // pairs of this column block are written instead of classes; lvc_knn_merge_vote ranks the blocks' lists against each other
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
        if (cand_val) {
          cand_val[(size_t)row * KTOP + r] = mv;
          cand_idx[(size_t)row * KTOP + r] = idx_base + mi;
        } else {
          const long long cl = shot_classes[mi];
          s_cls[w][r] = cl;
          top_classes[(size_t)row * KTOP + r] = cl;
        }
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
        if (cand_val) {
          cand_val[(size_t)row * KTOP + r] = bi < S ? best : -INFINITY;
          cand_idx[(size_t)row * KTOP + r] = bi < S ? idx_base + bi : 0x7fffffff;
        } else {
          const long long cl = (bi < S) ? shot_classes[bi] : -1;
          s_cls[w][r] = cl;
          top_classes[(size_t)row * KTOP + r] = cl;
        }
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  if (lane == 0 && keep && !cand_val) {
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

static int knn_topk_launch(const float* sims, int ld, int Q, int S, const long long* shot_classes, const long long* det_classes,
                           int kvote, long long* top_classes, long long* keep, float* cand_val, int* cand_idx, int idx_base,
                           void* stream) {
  const int per = lvc_cdiv(S, 64);
  const dim3 grid(lvc_cdiv(Q, 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int ldd = ld > 0 ? ld : S;
#define KNN_LAUNCH(P) hipLaunchKernelGGL((knn_topk_vote_kernel<10, P>), grid, block, 0, st, sims, ldd, Q, S, shot_classes, \
                                         det_classes, kvote, top_classes, keep, cand_val, cand_idx, idx_base)
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
  return knn_topk_launch(sims, ld, Q, S, shot_classes, det_classes, kvote, top_classes, keep, nullptr, nullptr, 0, stream);
}

// Shot sets of any size (LVIS: 1230 classes x 10 shots): the caller cuts the columns into blocks of <= 4096 shots, this call
// writes the ten best (similarity, idx_base + column) pairs of ONE block per row (value descending, ties -> lower index; rows of
// a block with fewer than ten columns are padded with (-inf, INT_MAX)), and lvc_knn_merge_vote ranks the blocks' lists:
// the exact top ten of the whole row are among the per-block top tens.
extern "C" int lvc_knn_topk_candidates(const float* sims, int ld, int Q, int S, int idx_base, float* cand_val, int* cand_idx,
                                       void* stream) {
  LVC_CHECK_ARG(Q >= 0 && S >= 1, "bad sizes");
  if (Q == 0) return LVC_OK;
  LVC_CHECK_ARG(sims && cand_val && cand_idx, "null pointer");
  LVC_CHECK_ARG(S <= 64 * KNN_MAX_PER_LANE, "at most 4096 shots per call");
  return knn_topk_launch(sims, ld, Q, S, nullptr, nullptr, 10, nullptr, nullptr, cand_val, cand_idx, idx_base, stream);
}

// cand_val / cand_idx [nlists][Q][10]: per row the 10 x nlists candidates are ranked (value descending, ties -> lower shot index),
// the ten best give top_classes [Q,10] and the vote / keep exactly as lvc_knn_topk_vote does.  One wave per row.
#define KM_MAX 640
__global__ __launch_bounds__(256) void knn_merge_vote_kernel(const float* __restrict__ cand_val, const int* __restrict__ cand_idx,
                                                             int nlists, int Q, const long long* __restrict__ shot_classes,
                                                             const long long* __restrict__ det_classes, int kvote,
                                                             long long* __restrict__ top_classes, long long* __restrict__ keep) {
  __shared__ float s_v[4][KM_MAX];
  __shared__ int s_i[4][KM_MAX];
  __shared__ long long s_cls[4][10];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + w;
  if (row >= Q) return;
  const int total = nlists * 10;
  for (int c = lane; c < total; c += 64) {
    const int l = c / 10, r = c - l * 10;
    s_v[w][c] = cand_val[((size_t)l * Q + row) * 10 + r];
    s_i[w][c] = cand_idx[((size_t)l * Q + row) * 10 + r];
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  for (int c0 = 0; c0 < total; c0 += 64) {
    const int c = c0 + lane;
    const float mv = c < total ? s_v[w][c] : -INFINITY;
    const int mi = c < total ? s_i[w][c] : 0x7fffffff;
    int r = 0;
    for (int l = 0; l < total; ++l) {
      const float o = s_v[w][l];
      const int oi = s_i[w][l];
      r += (o > mv || (o == mv && oi < mi)) ? 1 : 0;
    }
    if (c < total && r < 10 && mi != 0x7fffffff) {
      const long long cl = shot_classes[mi];
      s_cls[w][r] = cl;
      top_classes[(size_t)row * 10 + r] = cl;
    }
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

extern "C" int lvc_knn_merge_vote(const float* cand_val, const int* cand_idx, int nlists, int Q, const long long* shot_classes,
                                  const long long* det_classes, int kvote, long long* top_classes, long long* keep, void* stream) {
  LVC_CHECK_ARG(Q >= 0 && nlists >= 1 && nlists * 10 <= KM_MAX, "between 1 and 64 candidate lists");
  if (Q == 0) return LVC_OK;
  LVC_CHECK_ARG(cand_val && cand_idx && shot_classes && top_classes, "null pointer");
  LVC_CHECK_ARG(kvote >= 1 && kvote <= 10, "k must be in 1..10 (the reference stores top-10)");
  hipLaunchKernelGGL(knn_merge_vote_kernel, dim3(lvc_cdiv(Q, 4)), dim3(256), 0, (hipStream_t)stream, cand_val, cand_idx, nlists, Q,
                     shot_classes, det_classes, kvote, top_classes, keep);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// out[0] = max over x[0 .. n) (-inf for n = 0; NaN entries are ignored).  One workgroup: for the handful of per-shot statistics
// the sweep's margins need.
__global__ __launch_bounds__(256) void max_f32_kernel(const float* __restrict__ x, long long n, float* __restrict__ out) {
  __shared__ float red[256];
  float m = -INFINITY;
  for (long long i = threadIdx.x; i < n; i += 256) m = fmaxf(m, x[i]);
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

extern "C" int lvc_max_f32(const float* x, long long n, float* out, void* stream) {
  LVC_CHECK_ARG(x && out && n >= 0, "bad arguments");
  hipLaunchKernelGGL(max_f32_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, n, out);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// margins[i] = 2 (1 + 1e-4) (qres[i] (1 + 1e-6 + sres_max) + sres_max (1 + 1e-5) + acc) + extra: the per-row pre-filter margin of the
// two-stage sweep (lvc_amd.label_verification.pre_filter_margins states the bound) in one launch instead of five elementwise
// ones, in the same fp32 operation order.  d_sres_max: device scalar (lvc_max_f32 over the shots' residual norms).
__global__ __launch_bounds__(256) void knn_margins_kernel(const float* __restrict__ qres, const float* __restrict__ d_sres_max, float acc,
                                                          float extra, int Q, float* __restrict__ margins) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Q) return;
  const float sres_max = d_sres_max[0];
  const float sh_max = 1.0f + 1e-6f + sres_max;
  const float eps = qres[i] * sh_max + sres_max * (1.0f + 1e-5f) + acc;
  margins[i] = (2.0f * (1.0f + 1e-4f)) * eps + extra;
}

extern "C" int lvc_knn_margins(const float* qres, const float* d_sres_max, float acc, float extra, int Q, float* margins, void* stream) {
  LVC_CHECK_ARG(Q >= 0 && (Q == 0 || (qres && d_sres_max && margins)), "bad arguments");
  if (Q == 0) return LVC_OK;
  hipLaunchKernelGGL(knn_margins_kernel, dim3(lvc_cdiv(Q, 256)), dim3(256), 0, (hipStream_t)stream, qres, d_sres_max, acc, extra, Q, margins);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Two-stage exact top-10: the [Q, S] matrix handed in is an APPROXIMATION of the similarities (lvc_gemm_f16, gemm_h.hip: every
// operand rounded to fp16, one MFMA per block instead of three) with |approx - exact| <= eps for unit-norm rows
// (eps = 2^-10 * sum |q_i s_i| <= 2^-10 by Cauchy-Schwarz, plus the fp32 accumulation, ~1e-6).
//   1. Containment.  Let A10 be the 10th largest approximate value of a row.  The ten best approximate shots have exact
//      similarity >= A10 - eps, so the exact 10th best x10 >= A10 - eps, and every shot of the exact top ten has
//      approx >= x10 - eps >= A10 - 2 eps: the candidates {approx >= A10 - margin}, margin >= 2 eps, contain the exact top ten
//      (a dozen or so shots on uncorrelated descriptors).
//   2. What has to be exact.  The outputs are CLASS ids in rank order.  Call a candidate flagged when another candidate of a
//      DIFFERENT class lies within margin of it; flagged candidates are re-evaluated in fp32 (lane l sums elements l*4 .. l*4+3
//      of every 256-element slice in order with fmas, then a butterfly over the lanes: a fixed order), and the candidates are
//      sorted by key = exact value where flagged, approximate value otherwise (descending, ties -> lower shot index).  Two
//      candidates of different classes are ordered as by the exact values: both flagged -> both keys exact; one of them
//      unflagged -> their approximate values differ by more than margin >= 2 eps, and moving either value by eps cannot swap
//      them.  Two total orders that agree on every cross-class pair put each class on the same set of positions, so the class
//      sequence -- and with it top_classes and the vote -- equals the one of the exact ranking with the reference's tie rule.
//   3. A row with more than KV_MAX_CAND candidates (a cloud of near-identical shots) evaluates ALL shots exactly,
//      KV_MAX_CAND - 10 at a time next to the running ten best.
// The query row is normalised on the fly from the raw descriptors exactly as lvc_rownorm does it ((q - mu) / den, den from
// lvc_rownorm_h), and only for rows that have a flagged candidate.
#define KV_MAX_CAND 192

template <int KTOP>
struct KvLds {
  float ap[KV_MAX_CAND];             // approximate value of a candidate
  float val[KV_MAX_CAND + KTOP];     // sort key: exact similarity where evaluated, else the approximate one
  int idx[KV_MAX_CAND + KTOP];
  long long ccl[KV_MAX_CAND];        // class of a candidate
  short alist[KV_MAX_CAND];          // positions of the flagged candidates
  long long cls[KTOP];
  float T;
};

// v + (the same value NSTEP lanes round the 16-lane row): one v_add_f32 with a DPP operand
template <int CTRL>
__device__ __forceinline__ float kv_add_ror(float v) {
  return v + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ float kv_min_ror(float v) {
  return fminf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, false)));
}
// minimum over the 64 lanes (every lane gets it)
__device__ __forceinline__ float kv_wave_min(float v) {
  v = kv_min_ror<0x128>(v); v = kv_min_ror<0x124>(v); v = kv_min_ror<0x122>(v); v = kv_min_ror<0x121>(v);
  const float r0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 0));
  const float r1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 16));
  const float r2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 32));
  const float r3 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 48));
  return fminf(fminf(r0, r1), fminf(r2, r3));
}

// Steps 2 and 3 of the comment above for one row (one wave): the candidates sit in L.ap / L.idx [0, ncand) when
// ncand <= KV_MAX_CAND; a larger ncand means "evaluate every shot".  Writes top_classes[row] and keep[row].
// NSL = 256-element slices of a descriptor row (D <= 256 NSL).
template <int KTOP, int NSL>
__device__ __forceinline__ void kv_finish(KvLds<KTOP>& L, const int lane, const int row, const int ncand, const int S,
                                          const float* __restrict__ q, const int ldq, const float* __restrict__ mu,
                                          const float* __restrict__ den, const float* __restrict__ sn, const int D, const float margin,
                                          const long long* __restrict__ shot_classes, const long long* __restrict__ det_classes,
                                          const int kvote, long long* __restrict__ top_classes, long long* __restrict__ keep) {
  // Exact similarities of the n shots listed through pos_of(k) -> position in (L.idx, L.val), FOUR shot rows per round with all
  // their loads in flight at once (one memory round trip per four shots).  Lane l owns elements sl*256 + l*4 .. +3 of every
  // 256-element slice sl (where below D) and sums them in that order with fmas; the 64 partial sums of the four shots are added in
  // a fixed tree: lanes l and l + 32 (v_permlane32_swap hands each half-wave two of the four shots), then the 16-lane rows two
  // by two (v_permlane16_swap: row t now holds shot t), then inside the row.  The query row is normalised ((q - mu) / den, as
  // lvc_rownorm does it) once per row, and only for rows that have a flagged candidate.
  constexpr int CH = NSL < 4 ? NSL : 4;
  float4 xq[NSL];
  bool have_q = false;
  auto load_q = [&]() {
    const float* qr = q + (size_t)row * ldq;
    const float dn = den ? den[row] : 1.f;
    // every slice of the row (and of mu) is requested before the first is used: one memory round trip, not one per slice (the
    // bounds test per slice made the compiler wait slice by slice); lanes past D read element 0 and are zeroed
    float4 m4[NSL];
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) {
      const int d = sl * 256 + lane * 4;
      const int dd = d < D ? d : 0;
      xq[sl] = *reinterpret_cast<const float4*>(qr + dd);
      m4[sl] = mu ? *reinterpret_cast<const float4*>(mu + dd) : float4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) {
      const int d = sl * 256 + lane * 4;
      float4 x = xq[sl];
      if (mu) { x.x -= m4[sl].x; x.y -= m4[sl].y; x.z -= m4[sl].z; x.w -= m4[sl].w; }
      if (den) { x.x /= dn; x.y /= dn; x.z /= dn; x.w /= dn; }
      if (!(d < D)) x = float4{0.f, 0.f, 0.f, 0.f};
      xq[sl] = x;
    }
    have_q = true;
  };
  auto exact_dots = [&](int n, auto pos_of) {
    if (!have_q) load_q();
    const int t_mine = lane >> 4;
    for (int k0 = 0; k0 < n; k0 += 4) {
      const float* sr[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) sr[t] = sn + (size_t)L.idx[pos_of(min(k0 + t, n - 1))] * D + lane * 4;
      const int mypos = pos_of(min(k0 + t_mine, n - 1));
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c0 = 0; c0 < NSL; c0 += CH) {
        float4 sv[4][CH];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int c = 0; c < CH; ++c)
            sv[t][c] = (c0 + c) * 256 + lane * 4 < D ? *reinterpret_cast<const float4*>(sr[t] + (c0 + c) * 256) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            // explicit fmas: every row in flight must round the same way (identical shots tie exactly)
            const float4 x = xq[c0 + c];
            acc[t] = __builtin_fmaf(x.x, sv[t][c].x, acc[t]); acc[t] = __builtin_fmaf(x.y, sv[t][c].y, acc[t]);
            acc[t] = __builtin_fmaf(x.z, sv[t][c].z, acc[t]); acc[t] = __builtin_fmaf(x.w, sv[t][c].w, acc[t]);
          }
      }
      const auto p02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[0]), __float_as_uint(acc[2]), false, false);
      const auto p13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[1]), __float_as_uint(acc[3]), false, false);
      const float s02 = __uint_as_float(p02[0]) + __uint_as_float(p02[1]);     // lanes 0..31: shot 0, lanes 32..63: shot 2
      const float s13 = __uint_as_float(p13[0]) + __uint_as_float(p13[1]);     //              shot 1,               shot 3
      const auto pr = __builtin_amdgcn_permlane16_swap(__float_as_uint(s02), __float_as_uint(s13), false, false);
      float u = __uint_as_float(pr[0]) + __uint_as_float(pr[1]);               // 16-lane row t: shot t
      u = kv_add_ror<0x128>(u); u = kv_add_ror<0x124>(u); u = kv_add_ror<0x122>(u); u = kv_add_ror<0x121>(u);
      if ((lane & 15) == 0) L.val[mypos] = u;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  };
  // rank the first n entries by (val descending, idx ascending) and keep the KTOP best in place.  Winners park in the tail
  // slots [KV_MAX_CAND + rank] (ranks are unique, the loops never read the tail) and move to the front afterwards.
  auto keep_best = [&](int n) {
    for (int c0 = 0; c0 < n; c0 += 64) {
      const int c = c0 + lane;
      if (c < n) {
        const float mv = L.val[c];
        const int mi = L.idx[c];
        int r = 0;
        for (int l = 0; l < n; ++l) {
          const float o = L.val[l];
          r += (o > mv || (o == mv && L.idx[l] < mi)) ? 1 : 0;
        }
        if (r < KTOP) { L.val[KV_MAX_CAND + r] = mv; L.idx[KV_MAX_CAND + r] = mi; }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    const int kept = n < KTOP ? n : KTOP;
    if (lane < kept) { L.val[lane] = L.val[KV_MAX_CAND + lane]; L.idx[lane] = L.idx[KV_MAX_CAND + lane]; }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    return kept;
  };
  int nbest;
  if (ncand <= KV_MAX_CAND) {
    for (int c = lane; c < ncand; c += 64) L.ccl[c] = shot_classes[L.idx[c]];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    // flagged candidates (a candidate of another class within margin), their positions compacted; key = approx for the others
    int nflag = 0;
    for (int c0 = 0; c0 < ncand; c0 += 64) {
      const int c = c0 + lane;
      bool fl = false;
      if (c < ncand) {
        const float mv = L.ap[c];
        const long long mc = L.ccl[c];
        for (int l = 0; l < ncand; ++l) fl = fl || (L.ccl[l] != mc && !(fabsf(L.ap[l] - mv) > margin));
        L.val[c] = mv;
      }
      const unsigned long long m = __ballot(fl);
      if (fl) L.alist[nflag + __popcll(m & ((1ull << lane) - 1ull))] = (short)c;
      nflag += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (nflag) exact_dots(nflag, [&](int k) { return (int)L.alist[k]; });
    nbest = keep_best(ncand);
  } else {
    // every shot exactly, in blocks of KV_MAX_CAND - KTOP next to the running best
    nbest = 0;
    for (int s0 = 0; s0 < S; s0 += KV_MAX_CAND - KTOP) {
      const int nb = min(KV_MAX_CAND - KTOP, S - s0);
      for (int c = lane; c < nb; c += 64) L.idx[nbest + c] = s0 + c;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      const int base = nbest;
      exact_dots(nb, [&](int k) { return base + k; });
      nbest = keep_best(nbest + nb);
    }
  }
  long long mycl = -1;
  if (lane < KTOP) {
    mycl = lane < nbest ? shot_classes[L.idx[lane]] : -1;
    L.cls[lane] = mycl;
    top_classes[(size_t)row * KTOP + lane] = mycl;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  if (keep) {
    // mode of the first kvote classes, ties -> smallest class id: lane a counts its class, then the best (count, -class)
    int cnt = 0;
    if (lane < kvote)
      for (int b = 0; b < kvote; ++b) cnt += (L.cls[b] == mycl);
    int best = cnt;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o));
    long long cand = (lane < kvote && cnt == best) ? mycl : 0x7fffffffffffffffll;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      const long long other = __shfl_xor(cand, o);
      cand = other < cand ? other : cand;
    }
    if (lane == 0) keep[row] = (det_classes && det_classes[row] == cand) ? 1 : 0;
  }
}

// Form A: the pre-filter similarities as a dense [Q, ld] matrix (lvc_gemm_f16).  One wave per row, the row's S values in PER
// registers per lane:
//   1. T = the KTOP-th largest of the 64 lane maxima (every lane counts the maxima above its own through 64 lane broadcasts; the
//      smallest maximum with fewer than KTOP above it is T): a lower bound of A10, the row's KTOP-th largest value;
//   2. ONE pass over the registers compacts the values >= T - margin into LDS (a superset of the candidates);
//   3. A10 = the KTOP-th largest of those (every value >= T is among them); the entries below A10 - margin are dropped.
// (The 16-bit fixed-point matrix of lvc_gemm_f16_q15 has a kernel of its own: knn_verify_q15_kernel below.)
template <int KTOP, int PER, int NSL, bool VEC>
__global__ __launch_bounds__(256) void knn_verify_topk_vote_kernel(const void* __restrict__ approx_, int ld, int Q, int S,
                                                                   const float* __restrict__ q, int ldq, const float* __restrict__ mu,
                                                                   const float* __restrict__ den, const float* __restrict__ sn,
                                                                   int D, float margin_all, const float* __restrict__ margins,
                                                                   const long long* __restrict__ shot_classes,
                                                                   const long long* __restrict__ det_classes, int kvote,
                                                                   long long* __restrict__ top_classes, long long* __restrict__ keep) {
  __shared__ KvLds<KTOP> s_L[4];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + w;
  if (row >= Q) return;     // whole waves leave; nothing below synchronises across waves
  const float margin = margins ? margins[row] : margin_all;
  KvLds<KTOP>& L = s_L[w];
  const float* ar = reinterpret_cast<const float*>(approx_) + (size_t)row * ld;
  float v[PER];
  float lmax = -INFINITY;
  // register j of lane l holds column col_of(j): one dwordx4 per lane and 256 columns when the rows are 16-byte aligned
  auto col_of = [&](int j) { return VEC ? (j >> 2) * 256 + lane * 4 + (j & 3) : j * 64 + lane; };
  if constexpr (VEC) {
#pragma unroll
    for (int jj = 0; jj < PER / 4; ++jj) {
      const int i = jj * 256 + lane * 4;
      float4 x = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      if (i < S) x = *reinterpret_cast<const float4*>(ar + i);
      v[4 * jj + 0] = x.x; v[4 * jj + 1] = i + 1 < S ? x.y : -INFINITY; v[4 * jj + 2] = i + 2 < S ? x.z : -INFINITY; v[4 * jj + 3] = i + 3 < S ? x.w : -INFINITY;
    }
  } else {
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = j * 64 + lane;
      v[j] = i < S ? ar[i] : -INFINITY;
    }
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    if (v[j] != v[j]) v[j] = INFINITY;
    lmax = fmaxf(lmax, v[j]);
  }
  // ---- T: the lanes with fewer than KTOP lane maxima strictly above their own hold values >= the KTOP-th largest maximum
  // (counted with multiplicity), which is itself one of them: T is their minimum
  int above = 0;
#pragma unroll
  for (int l = 0; l < 64; ++l) above += __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(lmax), l)) > lmax ? 1 : 0;
  const float T = kv_wave_min(above < KTOP ? lmax : INFINITY);
  const float Tv0 = T - margin;
  int total = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const bool is_c = v[j] >= Tv0 && v[j] > -INFINITY;
    const unsigned long long m = __ballot(is_c);
    if (m) {
      const int pos = total + __popcll(m & ((1ull << lane) - 1ull));
      if (is_c && pos < KV_MAX_CAND) { L.idx[pos] = col_of(j); L.ap[pos] = v[j]; }
      total += __popcll(m);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  int ncand = total;       // more than KV_MAX_CAND values >= T - margin (a near-constant row): every shot is evaluated exactly
  if (total <= KV_MAX_CAND) {
    // A10 among the listed values, then only the entries >= A10 - margin stay (in order)
    float a10 = INFINITY;
    for (int c0 = 0; c0 < total; c0 += 64) {
      const int c = c0 + lane;
      const float mv = c < total ? L.ap[c] : -INFINITY;
      int r = 0;
      for (int l = 0; l < total; ++l) r += L.ap[l] > mv ? 1 : 0;
      a10 = fminf(a10, r < KTOP && c < total ? mv : INFINITY);
    }
    const float Tv = kv_wave_min(a10) - margin;
    ncand = 0;
    for (int c0 = 0; c0 < total; c0 += 64) {
      const int c = c0 + lane;
      const float mv = c < total ? L.ap[c] : -INFINITY;
      const int mi = c < total ? L.idx[c] : 0;
      const bool live = mv >= Tv;
      const unsigned long long m = __ballot(live);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      if (live) { const int pos = ncand + __popcll(m & ((1ull << lane) - 1ull)); L.ap[pos] = mv; L.idx[pos] = mi; }
      ncand += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  }
  kv_finish<KTOP, NSL>(L, lane, row, ncand, S, q, ldq, mu, den, sn, D, margin, shot_classes, det_classes, kvote, top_classes, keep);
}

#define KV_CHECKS()                                                                                                              \
  LVC_CHECK_ARG(Q >= 0 && S >= 10, "need at least 10 shots");                                                                    \
  if (Q == 0) return LVC_OK;                                                                                                     \
  LVC_CHECK_ARG(q && sn && shot_classes && top_classes, "null pointer");                                                         \
  LVC_CHECK_ARG(D % 4 == 0 && D <= 2048 && D > 0, "descriptor length must be a multiple of 4, at most 2048");                    \
  LVC_CHECK_ARG(kvote >= 1 && kvote <= 10 && margin >= 0.f, "k must be in 1..10, margin >= 0");                                  \
  const int ldqq = ldq > 0 ? ldq : D;                                                                                            \
  LVC_CHECK_ARG((((uintptr_t)q | (uintptr_t)sn | (uintptr_t)mu) & 15) == 0 && ldqq % 4 == 0, "descriptor rows must be 16-byte aligned")

// approx [Q, ld] (S used columns) from lvc_gemm_f16 over the fp16 roundings of the normalised rows; q [Q, ldq] the RAW query
// descriptors with mu [D] (or NULL) and den [Q] (or NULL) as lvc_rownorm_h used them (NULL, NULL: q already holds the normalised
// rows); sn [S, D] the normalised shots (fp32).  D % 4 == 0, D <= 2048; margin >= 2 x the approximation error bound (2^-9 covers
// unit-norm rows in the worst case); margins [Q] (or NULL) replaces it row by row (the host passes 2 x the row's own bound from
// the rounding residual norms lvc_rownorm_h measures: about half the worst case, so fewer exact re-evaluations).
extern "C" int lvc_knn_verify_topk_vote(const float* approx, int ld, int Q, int S, const float* q, int ldq, const float* mu,
                                        const float* den, const float* sn, int D, float margin, const float* margins,
                                        const long long* shot_classes, const long long* det_classes, int kvote,
                                        long long* top_classes, long long* keep, void* stream) {
  KV_CHECKS();
  LVC_CHECK_ARG(approx, "null pointer");
  LVC_CHECK_ARG(S <= 64 * KNN_MAX_PER_LANE, "at most 4096 shots per call");
  const int per = lvc_cdiv(S, 64);
  const dim3 grid(lvc_cdiv(Q, 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int ldd = ld > 0 ? ld : S;
#define KV_LAUNCH_V(P, N, V) hipLaunchKernelGGL((knn_verify_topk_vote_kernel<10, P, N, V>), grid, block, 0, st, (const void*)approx, ldd, Q, S, q, ldqq, mu, \
                                                den, sn, D, margin, margins, shot_classes, det_classes, kvote, top_classes, keep)
#define KV_LAUNCH_N(P, N) do { if (vec) KV_LAUNCH_V(P, N, true); else KV_LAUNCH_V(P, N, false); } while (0)
#define KV_LAUNCH(P) do { if (D <= 512) KV_LAUNCH_N(P, 2); else if (D <= 1024) KV_LAUNCH_N(P, 4); else KV_LAUNCH_N(P, 8); } while (0)
  const bool vec = ldd % 4 == 0 && (((uintptr_t)approx) & 15) == 0;     // rows that dwordx4 loads can walk
  if (per <= 8) KV_LAUNCH(8);
  else if (per <= 16) KV_LAUNCH(16);
  else if (per <= 24) KV_LAUNCH(24);
  else if (per <= 32) KV_LAUNCH(32);
  else if (per <= 40) KV_LAUNCH(40);
  else if (per <= 48) KV_LAUNCH(48);
  else KV_LAUNCH(64);
#undef KV_LAUNCH_N
#undef KV_LAUNCH_V
#undef KV_LAUNCH
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The same verification over lvc_gemm_f16_q15's 16-bit fixed-point matrix (gemm_h.hip: y = rint(32766 * dot) clamped to +-32766,
// NaN -> 32767 so that it sorts first as in torch.topk; -32768 is never written and stands for "no value" here).  The float kernel
// above is bound by VALU issue, not by memory (0.61 ms for 576 MB on the cfg-4 sweep, 0.37 ms of it before the first exact dot
// product: scripts/probe_knn_verify.py): it unpacks and tests every one of the S values of a row in fp32.  Here
//   * the scan stays in the 16-bit domain: lane maxima with v_pk_max_i16 on the packed words as loaded (eight values per dwordx4),
//     T = the KTOP-th largest lane maximum by a bisection over the value range with one ballot per step (scalar work), the
//     compaction tests each packed word with two integer compares against T - margin (margin rounded UP to steps of 1 / 32766: a
//     superset of the float test) and converts nothing but the candidates;
//   * a candidate is one packed key (value << 16 | 65535 - column: signed order = value descending, column ascending) next to its
//     class, read back as ONE 64-bit broadcast per step of the ranking loops;
//   * the exact re-evaluation works on q - mu (not normalised) with two-wide fmas and divides the finished dot product by the row's
//     norm once -- sixteen divisions per row become one per dot product; the sum order is fixed (lane l: the x/y and the z/w halves
//     of its float4 slices as two chains, added, then the lane tree of the float kernel), identical shots still tie exactly;
//   * the query row is requested at the top of the kernel, under the scan.
// The containment / flagging argument is the float kernel's (header above); candidates carrying the NaN code are never flagged
// (they keep +inf and order by shot index).  More than KQ_MAX_CAND candidates: every shot exactly, as there.
#define KQ_MAX_CAND 192
#define KQ_SCALE 32766.f
#ifndef KQ_WAVES
#define KQ_WAVES
#endif
#ifndef KQ_CH
#define KQ_CH 4     // 256-element slices of four shot rows in flight per round of exact dot products
#endif
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef float kq_f32x2 __attribute__((ext_vector_type(2)));

struct KqLds {
  int2 ent[KQ_MAX_CAND];      // x: approximate value in steps (32767 = NaN code), y: class
  int2 rk[KQ_MAX_CAND];       // x: bits of the sort key (exact similarity where evaluated, else the approximate one), y: shot
  short alist[KQ_MAX_CAND];   // positions of the flagged candidates
  int cls[16];
};

__device__ __forceinline__ int kq_prefix(unsigned long long m) {     // set bits of m below this lane
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

__device__ __forceinline__ int kq_count(bool p) {     // lanes with p, as a 32-bit scalar (the compiler's own 64-bit count is compared on the VALU)
  int n;
  const unsigned long long m = __ballot(p);
  asm("s_bcnt1_i32_b64 %0, %1" : "=s"(n) : "s"(m) : "scc");
  return n;
}

template <int KTOP, int NJ, int NSL>
__global__ __launch_bounds__(256) KQ_WAVES void knn_verify_q15_kernel(const short* __restrict__ approx, int ld, int Q, int S,
                                                             const float* __restrict__ q, int ldq, const float* __restrict__ mu,
                                                             const float* __restrict__ den, const float* __restrict__ sn, int D,
                                                             float margin_all, const float* __restrict__ margins,
                                                             const long long* __restrict__ shot_classes,
                                                             const long long* __restrict__ det_classes, int kvote,
                                                             long long* __restrict__ top_classes, long long* __restrict__ keep) {
  __shared__ KqLds s_L[4];
  // wave-uniform values are made scalar by hand (readfirstlane): row pointers and shot rows then address as SGPR base + lane offset
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + w;
  if (row >= Q) return;     // whole waves leave; nothing below synchronises across waves
  KqLds& L = s_L[w];
  constexpr int PADW = (int)0x80008000u;      // two "no value" halves
  constexpr int NW = 4 * NJ;
  // ---- the row's packed values: word d of lane l holds columns (d / 4) * 512 + l * 8 + (d % 4) * 2 and + 1
  const short* as = approx + (size_t)row * ld;
  int xw[NW];
  // (NJ = ceil(S / 512): only the last group of 512 columns can be partial.  Its loads are clamped to the row and masked without
  // branches -- rows are padded to a multiple of 8 columns, so a dwordx4 that starts below S stays inside the row.)
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    const int i = jj * 512 + lane * 8;
    if (jj < NJ - 1) {
      const int4 x = *reinterpret_cast<const int4*>(as + i);
      xw[4 * jj + 0] = x.x; xw[4 * jj + 1] = x.y; xw[4 * jj + 2] = x.z; xw[4 * jj + 3] = x.w;
    } else {
      const int4 x = *reinterpret_cast<const int4*>(as + (i < S ? i : 0));
      const int t[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int rem = S - (i + 2 * k);      // valid halves of this word: >= 2 both, 1 the low one, <= 0 none
        xw[4 * jj + k] = rem >= 2 ? t[k] : rem == 1 ? ((t[k] & 0xffff) | (int)0x80000000u) : PADW;
      }
    }
  }
  // ---- the raw query row, requested now and used by the first exact dot product
  float4 xq[NSL];
  {
    const float* qr = q + (size_t)row * ldq;
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) {
      const int d = sl * 256 + lane * 4;
      xq[sl] = *reinterpret_cast<const float4*>(qr + (d < D ? d : 0));
    }
  }
  const float dn = den ? den[row] : 1.f;
  const float margin = margins ? margins[row] : margin_all;
  const int mq = (int)fminf(ceilf(margin * KQ_SCALE), 65535.f);      // margin in steps, rounded up

  // largest t in [-32768, 32767] with at least KTOP of the listed values >= t (there are at least KTOP values): built bit by bit from
  // the top in offset binary, one ballot per bit -- sixteen steps of scalar work, no loop, whatever the values
  auto kth = [&](auto count_ge_biased) {
    unsigned u = 0;
#pragma unroll
    for (int b = 15; b >= 0; --b) {
      const unsigned c = u | (1u << b);
      if (count_ge_biased(c) >= KTOP) u = c;
    }
    return (int)u - 32768;
  };
  // ---- T: the KTOP-th largest of the 64 lane maxima (with multiplicity), a lower bound of the row's KTOP-th largest value
  s16x2 pm = __builtin_bit_cast(s16x2, PADW);
#pragma unroll
  for (int d = 0; d < NW; ++d) pm = __builtin_elementwise_max(pm, __builtin_bit_cast(s16x2, xw[d]));
  const int lmax = max((int)pm.x, (int)pm.y);
  const unsigned lmax_b = (unsigned)(lmax + 32768);
  const int T = kth([&](unsigned t) { return kq_count(lmax_b >= t); });
  const int Tm = max(T - mq, -32767);
  // ---- candidates: the values >= T - margin, as (steps, shot) in LDS
  const int TmH = Tm * 65536;
  int total = 0;
#pragma unroll
  for (int d = 0; d < NW; ++d) {
    const int x = xw[d];
    const bool hl = (int)((unsigned)x << 16) >= TmH;
    const bool hh = x >= TmH;
    const unsigned long long ml = __ballot(hl), mh = __ballot(hh);
    if (ml | mh) {
      const int col = (d >> 2) * 512 + lane * 8 + (d & 3) * 2;
      if (ml) {
        const int pos = total + kq_prefix(ml);
        if (hl && pos < KQ_MAX_CAND) L.ent[pos] = int2{(int)(short)(x & 0xffff), col};
        total += __popcll(ml);
      }
      if (mh) {
        const int pos = total + kq_prefix(mh);
        if (hh && pos < KQ_MAX_CAND) L.ent[pos] = int2{x >> 16, col + 1};
        total += __popcll(mh);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");

  // exact similarities of the n shots listed through pos_of(k) -> position in L.rk, four shot rows per round (see the float kernel)
  bool have_q = false;
  auto exact_dots = [&](int n, auto pos_of) {
    if (!have_q) {
      if (mu) {       // every slice of mu is requested before the first is used
        float4 m4[NSL];
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl) m4[sl] = *reinterpret_cast<const float4*>(mu + (sl * 256 + lane * 4 < D ? sl * 256 + lane * 4 : 0));
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl) { xq[sl].x -= m4[sl].x; xq[sl].y -= m4[sl].y; xq[sl].z -= m4[sl].z; xq[sl].w -= m4[sl].w; }
      }
#pragma unroll
      for (int sl = 0; sl < NSL; ++sl)
        if (!(sl * 256 + lane * 4 < D)) xq[sl] = float4{0.f, 0.f, 0.f, 0.f};
      have_q = true;
    }
    constexpr int CH = NSL < KQ_CH ? NSL : KQ_CH;
    const int t_mine = lane >> 4;
    for (int k0 = 0; k0 < n; k0 += 4) {
      const float* sr[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) sr[t] = sn + (size_t)__builtin_amdgcn_readfirstlane(L.rk[pos_of(min(k0 + t, n - 1))].y) * D + lane * 4;
      const int mypos = pos_of(min(k0 + t_mine, n - 1));
      kq_f32x2 acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
      for (int c0 = 0; c0 < NSL; c0 += CH) {
        if (c0) __builtin_amdgcn_sched_barrier(0);      // the next slices' loads stay behind this round's fmas (register budget)
        float4 sv[4][CH];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int c = 0; c < CH; ++c)
            sv[t][c] = (c0 + c) * 256 + lane * 4 < D ? *reinterpret_cast<const float4*>(sr[t] + (c0 + c) * 256) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float4 x = xq[c0 + c];
            acc[t] = __builtin_elementwise_fma(kq_f32x2{x.x, x.y}, kq_f32x2{sv[t][c].x, sv[t][c].y}, acc[t]);
            acc[t] = __builtin_elementwise_fma(kq_f32x2{x.z, x.w}, kq_f32x2{sv[t][c].z, sv[t][c].w}, acc[t]);
          }
      }
      float a1[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) a1[t] = acc[t].x + acc[t].y;
      const auto p02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a1[0]), __float_as_uint(a1[2]), false, false);
      const auto p13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a1[1]), __float_as_uint(a1[3]), false, false);
      const float s02 = __uint_as_float(p02[0]) + __uint_as_float(p02[1]);     // lanes 0..31: shot 0, lanes 32..63: shot 2
      const float s13 = __uint_as_float(p13[0]) + __uint_as_float(p13[1]);     //              shot 1,               shot 3
      const auto pr = __builtin_amdgcn_permlane16_swap(__float_as_uint(s02), __float_as_uint(s13), false, false);
      float u = __uint_as_float(pr[0]) + __uint_as_float(pr[1]);               // 16-lane row t: shot t
      u = kv_add_ror<0x128>(u); u = kv_add_ror<0x124>(u); u = kv_add_ror<0x122>(u); u = kv_add_ror<0x121>(u);
      if ((lane & 15) == 0) L.rk[mypos].x = (int)__float_as_uint(den ? u / dn : u);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  };
  // rank of entry c among the first n of L.rk by (key descending, shot ascending)
  auto rank_of = [&](int c, int n) {
    const int2 me = L.rk[c];
    const float mv = __uint_as_float((unsigned)me.x);
    int r = 0;
#pragma unroll 2
    for (int l = 0; l < n; ++l) {
      const int2 o = L.rk[l];
      const float ov = __uint_as_float((unsigned)o.x);
      r += (ov > mv || (ov == mv && o.y < me.y)) ? 1 : 0;
    }
    return r;
  };

  if (total <= KQ_MAX_CAND) {
    // ---- A10 = the KTOP-th largest candidate value; the candidates below A10 - margin go
    constexpr int NC = KQ_MAX_CAND / 64;
    int2 e[NC];
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) e[ch] = ch * 64 + lane < total ? L.ent[ch * 64 + lane] : int2{-32768, 0};
    unsigned eb[NC];
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) eb[ch] = (unsigned)(e[ch].x + 32768);
    const int A10 = total <= 64 ? kth([&](unsigned t) { return kq_count(eb[0] >= t); })
                                : kth([&](unsigned t) {
                                    int n = 0;
#pragma unroll
                                    for (int ch = 0; ch < NC; ++ch) n += kq_count(eb[ch] >= t);
                                    return n;
                                  });
    const int Tv = max(A10 - mq, -32767);
    int ncand = 0;
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) {
      if (ch * 64 < total) {
        const bool live = e[ch].x >= Tv;
        const unsigned long long m = __ballot(live);
        if (live) {
          const int pos = ncand + kq_prefix(m);
          const int cl = (int)shot_classes[e[ch].y];
          L.ent[pos] = int2{e[ch].x, cl};
          L.rk[pos] = int2{(int)__float_as_uint(e[ch].x == 32767 ? INFINITY : (float)e[ch].x * (1.f / KQ_SCALE)), e[ch].y};
        }
        ncand += __popcll(m);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    // ---- flagged: a candidate of another class within margin (NaN codes excepted); their positions compacted
    int nflag = 0;
    for (int c0 = 0; c0 < ncand; c0 += 64) {
      const int c = c0 + lane;
      bool fl = false;
      if (c < ncand) {
        const int2 me = L.ent[c];
        const int base = me.x - mq;
        const unsigned span = 2u * (unsigned)mq;
#pragma unroll 2
        for (int l = 0; l < ncand; ++l) {
          const int2 o = L.ent[l];
          fl = fl || ((unsigned)(o.x - base) <= span && o.y != me.y);
        }
        fl = fl && me.x != 32767;
      }
      const unsigned long long m = __ballot(fl);
      if (fl) L.alist[nflag + kq_prefix(m)] = (short)c;
      nflag += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (nflag) exact_dots(nflag, [&](int k) { return (int)L.alist[k]; });
    // ---- ranks below KTOP write their class to the output slot of that rank
    for (int c0 = 0; c0 < ncand; c0 += 64) {
      const int c = c0 + lane;
      if (c < ncand) {
        const int r = rank_of(c, ncand);
        if (r < KTOP) {
          const int cl = L.ent[c].y;
          L.cls[r] = cl;
          top_classes[(size_t)row * KTOP + r] = cl;
        }
      }
    }
  } else {
    // every shot exactly, in blocks of KQ_MAX_CAND - KTOP next to the running best (kept in slots 0 .. nbest - 1)
    int nbest = 0;
    for (int s0 = 0; s0 < S; s0 += KQ_MAX_CAND - KTOP) {
      const int nb = min(KQ_MAX_CAND - KTOP, S - s0);
      for (int c = lane; c < nb; c += 64) L.rk[nbest + c].y = s0 + c;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      const int base = nbest;
      exact_dots(nb, [&](int k) { return base + k; });
      const int n = nbest + nb;
      for (int c0 = 0; c0 < n; c0 += 64) {     // winners park in L.ent (unused on this path), then move to the front
        const int c = c0 + lane;
        if (c < n) {
          const int r = rank_of(c, n);
          if (r < KTOP) L.ent[r] = L.rk[c];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      nbest = n < KTOP ? n : KTOP;
      if (lane < nbest) L.rk[lane] = L.ent[lane];
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    if (lane < KTOP) {
      const int cl = (int)shot_classes[L.rk[lane].y];
      L.cls[lane] = cl;
      top_classes[(size_t)row * KTOP + lane] = cl;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  if (keep) {
    // mode of the first kvote classes, ties -> smallest class id: lane a counts its class, then the best (count, -class)
    const int mycl = lane < KTOP ? L.cls[lane] : -1;
    int cnt = 0;
    if (lane < kvote)
      for (int b = 0; b < kvote; ++b) cnt += (L.cls[b] == mycl);
    int best = cnt;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o));
    int cand = (lane < kvote && cnt == best) ? mycl : 0x7fffffff;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o));
    if (lane == 0) keep[row] = (det_classes && det_classes[row] == (long long)cand) ? 1 : 0;
  }
}

// approx [Q, ld] int16 from lvc_gemm_f16_q15 (value = steps / 32766, 32767 = NaN; ld % 8 == 0, rows 16-byte aligned); `margin` /
// `margins` must include 2 x 1.6e-5 for the quantisation (lvc_amd.label_verification.Q15_MARGIN).  Other arguments as above.
extern "C" int lvc_knn_verify_topk_vote_q15(const short* approx, int ld, int Q, int S, const float* q, int ldq, const float* mu,
                                        const float* den, const float* sn, int D, float margin, const float* margins,
                                        const long long* shot_classes, const long long* det_classes, int kvote,
                                        long long* top_classes, long long* keep, void* stream) {
  KV_CHECKS();
  LVC_CHECK_ARG(approx, "null pointer");
  LVC_CHECK_ARG(S <= 4096, "at most 4096 shots per call");
  const int nj = lvc_cdiv(S, 512);
  const dim3 grid(lvc_cdiv(Q, 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int ldd = ld > 0 ? ld : S;
  LVC_CHECK_ARG(ldd % 8 == 0 && ldd >= S && (((uintptr_t)approx) & 15) == 0, "16-bit rows must be 16-byte aligned (ld % 8 == 0)");
#define KQ_LAUNCH_N(J, N) hipLaunchKernelGGL((knn_verify_q15_kernel<10, J, N>), grid, block, 0, st, approx, ldd, Q, S, q, ldqq, mu, den, sn, D, \
                                             margin, margins, shot_classes, det_classes, kvote, top_classes, keep)
#define KQ_LAUNCH(J) do { if (D <= 512) KQ_LAUNCH_N(J, 2); else if (D <= 1024) KQ_LAUNCH_N(J, 4); else KQ_LAUNCH_N(J, 8); } while (0)
  switch (nj) {      // exactly ceil(S / 512): the kernel masks only its last group of columns
    case 1: KQ_LAUNCH(1); break;
    case 2: KQ_LAUNCH(2); break;
    case 3: KQ_LAUNCH(3); break;
    case 4: KQ_LAUNCH(4); break;
    case 5: KQ_LAUNCH(5); break;
    case 6: KQ_LAUNCH(6); break;
    case 7: KQ_LAUNCH(7); break;
    default: KQ_LAUNCH(8); break;
  }
#undef KQ_LAUNCH_N
#undef KQ_LAUNCH
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}
