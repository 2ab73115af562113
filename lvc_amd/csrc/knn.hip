// knn.hip -- label-verification kNN (reference tools/run_nearest_neighbours.py:142-162 + :214-227).
//
// The reference loops over query images on the CPU and materialises a [q_i, S, D] broadcast per image for
// F.cosine_similarity, then topk(10), class gather and torch.mode.  Here the whole sweep is three kinds of
// launches over all queries at once:
//   1. lvc_colmean + lvc_rownorm (elementwise.hip): mu = shots.mean(0); rows (x - mu) / max(|x - mu|, 1e-8)
//   2. sims = Qn . Sn^T on the fp32-MFMA GEMM (conv_igemm.hip; 589.8 GFLOP for 120k x 2400 x 1024)
//   3. knn_topk_vote_kernel (this file): one wave per query row: 10 rounds of wave-wide arg-max over the S
//      similarities held in registers (ties -> lower shot index), shot_classes gather, majority vote with
//      torch.mode's tie rule (smallest class id), keep = (vote == detector class).
#include "common.h"

__global__ void colmean_kernel(const float* __restrict__ x, float* __restrict__ mu, int M, int D, int ld) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  float s = 0.f;
  for (int m = 0; m < M; ++m) s += x[(size_t)m * ld + d];
  mu[d] = s / (float)M;
}

extern "C" int lvc_colmean(const float* x, float* mu, int M, int D, int ld, void* stream) {
  LVC_CHECK_ARG(x && mu && M > 0 && D > 0, "bad arguments");
  hipLaunchKernelGGL(colmean_kernel, dim3(lvc_cdiv(D, 64)), dim3(64), 0, (hipStream_t)stream, x, mu, M, D, ld > 0 ? ld : D);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

#define KNN_MAX_PER_LANE 64  // S <= 4096
template <int KTOP>
__global__ __launch_bounds__(256) void knn_topk_vote_kernel(const float* __restrict__ sims, int ld, int Q, int S,
                                                            const long long* __restrict__ shot_classes,
                                                            const long long* __restrict__ det_classes, int kvote,
                                                            long long* __restrict__ top_classes,
                                                            long long* __restrict__ keep) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= Q) return;
  const float* sr = sims + (size_t)row * ld;
  float v[KNN_MAX_PER_LANE];
  const int per = (S + 63) / 64;
#pragma unroll
  for (int j = 0; j < KNN_MAX_PER_LANE; ++j) {
    const int i = j * 64 + lane;
    v[j] = (j < per && i < S) ? sr[i] : -INFINITY;
  }
  long long cls[KTOP];
  for (int r = 0; r < KTOP; ++r) {
    float best = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < KNN_MAX_PER_LANE; ++j) {
      if (j < per) {
        const int i = j * 64 + lane;
        if (v[j] > best) { best = v[j]; bi = i; }  // ascending i within a lane: first max = lowest index
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o);
      const int oi = __shfl_xor(bi, o);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    // owner lane retires the winner
#pragma unroll
    for (int j = 0; j < KNN_MAX_PER_LANE; ++j)
      if (j < per && j * 64 + lane == bi) v[j] = -INFINITY;
    cls[r] = (bi < S) ? shot_classes[bi] : -1;
  }
  if (lane == 0) {
    for (int r = 0; r < KTOP; ++r) top_classes[(size_t)row * KTOP + r] = cls[r];
    if (keep) {
      // torch.mode over the first kvote votes: most frequent value, ties -> smallest value
      long long mode = -1; int mcount = 0;
      for (int a = 0; a < kvote; ++a) {
        int c = 0;
        for (int b = 0; b < kvote; ++b) c += (cls[b] == cls[a]);
        if (c > mcount || (c == mcount && cls[a] < mode)) { mcount = c; mode = cls[a]; }
      }
      keep[row] = (det_classes && det_classes[row] == mode) ? 1 : 0;
    }
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
  hipLaunchKernelGGL(knn_topk_vote_kernel<10>, dim3(lvc_cdiv(Q, 4)), dim3(256), 0, (hipStream_t)stream, sims,
                     ld > 0 ? ld : S, Q, S, shot_classes, det_classes, kvote, top_classes, keep);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}
