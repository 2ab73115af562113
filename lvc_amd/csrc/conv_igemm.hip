// conv_igemm.hip -- NHWC fp32 convolution / GEMM as an implicit GEMM on the CDNA4 matrix cores.
//
// Replaces, for the Faster-R-CNN hot path, what the reference runs through ATen/cuDNN:
//   detectron2/layers/wrappers.py:41-99 (Conv2d = conv + norm + activation),
//   detectron2/layers/batch_norm.py:45-65 (FrozenBatchNorm2d affine, folded into the epilogue),
//   detectron2/modeling/backbone/resnet.py:195-211 (bottleneck residual add + ReLU),
//   detectron2/modeling/backbone/fpn.py:131-133 (nearest x2 upsample + add, folded into the lateral conv),
//   lvc/modeling/roi_heads/box_head.py:82-91 and fast_rcnn.py:583-598 (Linear layers = 1x1 conv on M x 1 x 1 x K).
//
// GEMM view:  D[m][n] = sum_k A[m][k] * B[n][k]
//   m = output pixel (n_img, ho, wo)  -- NHWC, so m is exactly the row index of the output tensor
//   n = output channel
//   k = (c/32, r, s, c%32)            -- a BK=32 chunk of k is 128 contiguous bytes of one input pixel; the 9 taps
//                                        of a channel chunk are consecutive chunks (L2 reuse of the shifted lines)
// Arithmetic: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain; 64 FLOP/clk/SIMD = 157 TF chip peak).
// The parity bar of the path needs true fp32; bf16 MFMA would not hold it.
//
// Work decomposition (stream-K): the launch is a set of PERSISTENT workers (<= 2 per CU, all co-resident).
// The (tile, k-chunk) iteration space of the layer is cut into equal contiguous ranges, one per worker, so
// every CU gets the same number of MFMA chunks whatever the layer's tile count is (the 50x84 and 25x42
// pyramid levels have 1.03 or 2.05 tiles per CU: a tile-per-workgroup launch leaves the chip 1/3 idle there).
// A tile whose k-range is split over several workers is finished by the worker holding its k=0 piece: the
// others store their 64 KB partial accumulators (their FIRST work item) and raise a flag with an agent-scope
// release; the finisher (for whom this tile is the LAST work item) polls relaxed, acquires once, adds the
// partials in worker order (deterministic) and runs the epilogue.  Flags are reset by their consumer.
//
// Tile: 128(M) x BN(N) x 32(K) per 256-thread workgroup, BN = 128 or 64; 4 waves as 2x2, each wave a
// 64 x BN/2 sub-tile = 2 x NI MFMA 32x32 accumulators.  LDS: A and B tiles, k-contiguous rows padded to 36
// floats (144 B) so that the 16-lane groups of ds_read_b128 hit 16 distinct 16-B slots; double buffered.
// Global->LDS staging goes through registers with a two-deep software pipeline (chunk k+1 is written to the
// idle LDS buffer and chunk k+2 is requested while chunk k's 64 MFMAs issue); loads are branch-free
// buffer_load_dwordx4 whose out-of-range lanes (padding taps, rows past M) return 0; one barrier per chunk.
// Epilogue: accumulators -> LDS (row-major C tile) -> each lane handles float4 column groups: 512-B
// coalesced stores, residual / scale / shift read as float4 and all issued before use.
//
// Lane/fragment map for 32x32x2 (A: lane l holds A[i=l&31][k=l>>5]; B: B[k=l>>5][j=l&31]):
// lane (i,h) ds_read_b128's 4 consecutive k (= 8*kk + 4*h + t, t=0..3) from row i; MFMA step t then
// contracts k in {8kk+t, 8kk+4+t}; A and B use the same permutation so every k is used exactly once.
#include "common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define BM 128
#define BK 32
#define LDS_STRIDE 36   // floats per staged tile row (BK + 4 pad)
#define SPIN_LIMIT (1 << 24)

struct ConvArgs {
  const float* x;      // input  [N,H,W,C]   (C = physical channel count, multiple of 4)
  const float* w;      // packed weights [Kpad][Kg]  (Kpad multiple of 128, Kg multiple of BK)
  const float* scale;  // per out-channel multiplier or nullptr (=1)
  const float* shift;  // per out-channel addend or nullptr (=0)
  const float* res;    // residual tensor or nullptr
  float* y;            // output [M][ldy]
  float* partials;     // [workers][256 threads][64 floats]
  int* flags;          // [workers] (+ [workers] = error word)
  int N, H, W, C;
  int K;               // real out channels
  int R, S, stride, pad;
  int Ho, Wo, M;
  int Kg;              // padded gemm-K
  int relu;
  int res_mode;        // 0 none | 1 same shape [M][ldr] | 2 nearest-x2-upsampled: res is [N,Ho/2,Wo/2,ldr]
  int ldy, ldr;
  int tiles_n, nk, total_units, units_per_worker, nworkers, err_index;
  int x_bytes, w_bytes;
};

template <int MODE, int NI>
__global__ __launch_bounds__(256, 2) void conv_igemm_f32_kernel(ConvArgs p) {
  constexpr int BN_ = 64 * NI;
  constexpr int STAGE_FLOATS = 2 * (BM + BN_) * LDS_STRIDE;
  constexpr int CS_STRIDE = BN_ + 4;  // floats per C-staging row
  constexpr int CS_FLOATS = BM * CS_STRIDE;
  constexpr int SMEM_FLOATS = STAGE_FLOATS > CS_FLOATS ? STAGE_FLOATS : CS_FLOATS;
  __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
  float* As = smem;                          // [2][BM][LDS_STRIDE]
  float* Bs = smem + 2 * BM * LDS_STRIDE;    // [2][BN_][LDS_STRIDE]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fi = lane & 31, fh = lane >> 5;
  const int q = tid & 7;        // float4 slot inside a 32-float chunk
  const int row0 = tid >> 3;    // staged rows row0 + 32*j

  const int lw = lvc_xcd_remap(blockIdx.x, p.nworkers);  // logical worker id
  int u = lw * p.units_per_worker;
  const int u_end = min(u + p.units_per_worker, p.total_units);

  const int a_frag_off = (wm * 64 + fi) * LDS_STRIDE + fh * 4;
  const int b_frag_off = (wn * (BN_ / 2) + fi) * LDS_STRIDE + fh * 4;

  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

  while (u < u_end) {
    const int tile = u / p.nk;
    const int kc0 = u - tile * p.nk;
    const int kc1 = min(p.nk, kc0 + (u_end - u));
    const int tile_n = tile % p.tiles_n;
    const int tile_m = tile / p.tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN_;

    // ---- per-row load descriptors.  A is read through a raw buffer resource: lane offsets are 32-bit and an
    // offset >= num_records returns 0, which is how padding taps / rows past M are produced (no branches).
    unsigned a_off[4];   // byte offset of (n, ho*stride-pad, wo*stride-pad, q*4)   [mode 1: pixel +q]
    unsigned a_msk[4];   // bit rs set <=> tap (r,s) of this row is inside the image
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + row0 + 32 * j;
      const bool okm = m < p.M;
      const int mm = okm ? m : 0;
      const int n = mm / (p.Ho * p.Wo);
      const int rem = mm - n * (p.Ho * p.Wo);
      const int ho = rem / p.Wo;
      const int wo = rem - ho * p.Wo;
      const int bh = ho * p.stride - p.pad, bw = wo * p.stride - p.pad;
      a_off[j] = (unsigned)(((n * p.H + bh) * p.W + bw) * p.C + (MODE == 0 ? q * 4 : q * 4)) * 4u;
      unsigned msk = 0;
      if (okm) {
        if (MODE == 0) {
          for (int r = 0; r < p.R; ++r)
            for (int s2 = 0; s2 < p.S; ++s2)
              if (bh + r >= 0 && bh + r < p.H && bw + s2 >= 0 && bw + s2 < p.W) msk |= 1u << (r * p.S + s2);
        } else {
          for (int r = 0; r < p.R; ++r)
            if (bh + r >= 0 && bh + r < p.H && bw + q >= 0 && bw + q < p.W) msk |= 1u << r;
        }
      }
      a_msk[j] = msk;
    }
    unsigned b_off[NI * 2];
#pragma unroll
    for (int j = 0; j < NI * 2; ++j) b_off[j] = (unsigned)((n0 + row0 + 32 * j) * p.Kg + q * 4) * 4u;

    f32x4 areg[4], breg[NI * 2];
    // running (r, s, c-chunk) position of the NEXT chunk to request: advanced incrementally (no divisions in
    // the MFMA loop); requests past the work item's range re-load its last chunk (harmless, branch-free).
    int ld_kc = kc0, ld_c, ld_r, ld_s;
    if (MODE == 0) {
      // k order is (channel chunk, r, s) with s fastest: consecutive chunks of one tile re-read the same input
      // lines shifted by one pixel / one row, so the taps hit L2 instead of going back to the fabric
      const int RS = p.R * p.S;
      ld_c = kc0 / RS;
      const int rs0 = kc0 - ld_c * RS;
      ld_r = rs0 / p.S;
      ld_s = rs0 - ld_r * p.S;
    } else {
      ld_c = 0; ld_r = kc0; ld_s = 0;
    }
    auto load_next = [&]() {
      const int rs = (MODE == 0) ? ld_r * p.S + ld_s : ld_r;
      const int coff = (MODE == 0) ? ((ld_r * p.W + ld_s) * p.C + ld_c * BK) * 4 : ld_r * p.W * p.C * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned vo = ((a_msk[j] >> rs) & 1u) ? a_off[j] + (unsigned)coff : 0x80000000u;
        areg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, vo, 0, 0));
      }
#pragma unroll
      for (int j = 0; j < NI * 2; ++j)
        breg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, b_off[j], ld_kc * (BK * 4), 0));
      if (ld_kc + 1 < kc1) {   // uniform scalar bookkeeping
        ++ld_kc;
        if (MODE == 0) {
          if (++ld_s == p.S) { ld_s = 0; if (++ld_r == p.R) { ld_r = 0; ++ld_c; } }
        } else {
          ++ld_r;
        }
      }
    };
    auto store_chunk = [&](int buf) {
      float* a = As + buf * BM * LDS_STRIDE;
      float* b = Bs + buf * BN_ * LDS_STRIDE;
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(a + (row0 + 32 * j) * LDS_STRIDE + q * 4) = areg[j];
#pragma unroll
      for (int j = 0; j < NI * 2; ++j) *reinterpret_cast<f32x4*>(b + (row0 + 32 * j) * LDS_STRIDE + q * 4) = breg[j];
    };

    f32x16 acc[2][NI];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < NI; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    // Software pipeline, one barrier per chunk, everything else issued in the shadow of the 64-cycle MFMAs:
    //   group kk=0: ds_read frags(kk=1) | 16 MFMA(kk=0) interleaved with the 8 ds_write of chunk kc+1 (regs -> idle buffer)
    //   group kk=1: ds_read frags(kk=2) | 16 MFMA(kk=1) interleaved with the 8 buffer_load of chunk kc+2 (-> the same regs)
    //   group kk=2: ds_read frags(kk=3) | 16 MFMA(kk=2)
    //   barrier (all reads of this buffer issued and waited, all writes of the other buffer visible)
    //   group kk=3: ds_read frags(kk=0 of chunk kc+1, other buffer) | 16 MFMA(kk=3)
    // The loop body is branch-free (loads of chunks past the range are clamped re-loads, stores past the range
    // land in the idle buffer) so that it is ONE scheduling region and the sched_group_barrier interleave holds.
    f32x4 fa[2][2], fb[2][NI];
    auto read_frags = [&](int sel, const float* a, const float* b, int kk) {
      fa[sel][0] = *reinterpret_cast<const f32x4*>(a + kk * 8);
      fa[sel][1] = *reinterpret_cast<const f32x4*>(a + 32 * LDS_STRIDE + kk * 8);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) fb[sel][ni] = *reinterpret_cast<const f32x4*>(b + ni * 32 * LDS_STRIDE + kk * 8);
    };
    auto mfma_group = [&](int sel) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sel][mi][t], fb[sel][ni][t], acc[mi][ni], 0, 0, 0);
    };
    constexpr int NMF = 8 * NI;       // MFMAs per kk group
    constexpr int NST = 4 + 2 * NI;   // staged float4 per thread per chunk (A: 4, B: 2*NI)

    load_next();
    store_chunk(0);
    load_next();
    __syncthreads();
    int cur = 0;
    read_frags(0, As + a_frag_off, Bs + b_frag_off, 0);
    for (int kc = kc0; kc < kc1; ++kc) {
      const float* a = As + cur * BM * LDS_STRIDE + a_frag_off;
      const float* b = Bs + cur * BN_ * LDS_STRIDE + b_frag_off;
      const float* an = As + (cur ^ 1) * BM * LDS_STRIDE + a_frag_off;
      const float* bn = Bs + (cur ^ 1) * BN_ * LDS_STRIDE + b_frag_off;
      // ---- group 0
      read_frags(1, a, b, 1);
      store_chunk(cur ^ 1);
      mfma_group(0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2 + NI, 0);   // DS reads first
#pragma unroll
      for (int i = 0; i < NST; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // 1 DS write
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NMF - NST, 0);
      __builtin_amdgcn_sched_barrier(0);
      // ---- group 1
      read_frags(0, a, b, 2);
      load_next();
      mfma_group(1);
      __builtin_amdgcn_sched_group_barrier(0x100, 2 + NI, 1);
#pragma unroll
      for (int i = 0; i < NST; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);      // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 1);      // 1 VMEM read
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NMF - NST, 1);
      __builtin_amdgcn_sched_barrier(0);
      // ---- group 2
      read_frags(1, a, b, 3);
      mfma_group(0);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      // ---- group 3 (frags of the next chunk come from the other buffer, complete behind these MFMAs)
      read_frags(0, an, bn, 0);
      mfma_group(1);
      __builtin_amdgcn_sched_barrier(0);
      cur ^= 1;
    }
    __syncthreads();  // the speculative frag reads / idle-buffer stores of the last iteration are done
    u += kc1 - kc0;

    // ------------------------------------------------------------ split tiles: publish or combine partials
    if (kc0 != 0) {
      // not the k=0 piece: hand the accumulators to the finisher.  Layout [e4][tid] float4 -> coalesced.
      float* dst = p.partials + (size_t)lw * (256 * 32 * NI);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {
            f32x4 v = {acc[mi][ni][e4 * 4 + 0], acc[mi][ni][e4 * 4 + 1], acc[mi][ni][e4 * 4 + 2], acc[mi][ni][e4 * 4 + 3]};
            *reinterpret_cast<f32x4*>(dst + ((size_t)((mi * NI + ni) * 4 + e4) * 256 + tid) * 4) = v;
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(p.flags + lw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      continue;  // the finisher owns the epilogue
    }
    if (kc1 < p.nk) {
      // k=0 piece of a tile that continues in the following workers: wait for them, add in worker order
      const int last_unit = tile * p.nk + p.nk - 1;
      const int last_worker = last_unit / p.units_per_worker;
      for (int pw = lw + 1; pw <= last_worker; ++pw) {
        if (tid == 0) {
          int spins = 0;
          while (__hip_atomic_load(p.flags + pw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > SPIN_LIMIT) { atomicOr(p.flags + p.err_index, 1); break; }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        const float* src = p.partials + (size_t)pw * (256 * 32 * NI);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(src + ((size_t)((mi * NI + ni) * 4 + e4) * 256 + tid) * 4);
              acc[mi][ni][e4 * 4 + 0] += v[0]; acc[mi][ni][e4 * 4 + 1] += v[1];
              acc[mi][ni][e4 * 4 + 2] += v[2]; acc[mi][ni][e4 * 4 + 3] += v[3];
            }
        __syncthreads();
        if (tid == 0) __hip_atomic_store(p.flags + pw, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }

    // ------------------------------------------------------------ epilogue: y = act(acc*scale + shift + residual)
    // C/D map of 32x32x2: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    float* Cs = smem;  // staging tiles are dead (barrier at the end of the k loop)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
          const int col = wn * (BN_ / 2) + ni * 32 + fi;
          Cs[row * CS_STRIDE + col] = acc[mi][ni][e];
        }
    __syncthreads();
    constexpr int C4 = BN_ / 4;          // float4 groups per tile row
    constexpr int RPI = 256 / C4;        // rows handled per iteration
    const int c4 = tid % C4, rsub = tid / C4;
    const int col = n0 + c4 * 4;
    const bool vec_ok = ((p.K & 3) == 0) && ((p.ldy & 3) == 0) && (p.res_mode == 0 || (p.ldr & 3) == 0);
    if (vec_ok) {
      if (col < p.K) {
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + col);
        if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + col);
#pragma unroll 4
        for (int it = 0; it < BM / RPI; ++it) {
          const int r = it * RPI + rsub;
          const int row = m0 + r;
          if (row < p.M) {
            f32x4 v = *reinterpret_cast<const f32x4*>(Cs + r * CS_STRIDE + c4 * 4);
            v = v * sc + sh;
            if (p.res_mode == 1) {
              v += *reinterpret_cast<const f32x4*>(p.res + (size_t)row * p.ldr + col);
            } else if (p.res_mode == 2) {
              int n = row / (p.Ho * p.Wo);
              int rem = row - n * (p.Ho * p.Wo);
              int ho = rem / p.Wo;
              int wo = rem - ho * p.Wo;
              size_t ro = ((size_t)(n * (p.Ho >> 1) + (ho >> 1)) * (p.Wo >> 1) + (wo >> 1));
              v += *reinterpret_cast<const f32x4*>(p.res + ro * p.ldr + col);
            }
            if (p.relu) {
              v[0] = v[0] > 0.f ? v[0] : 0.f; v[1] = v[1] > 0.f ? v[1] : 0.f;
              v[2] = v[2] > 0.f ? v[2] : 0.f; v[3] = v[3] > 0.f ? v[3] : 0.f;
            }
            *reinterpret_cast<f32x4*>(p.y + (size_t)row * p.ldy + col) = v;
          }
        }
      }
    } else {
      for (int it = 0; it < BM / RPI; ++it) {
        const int r = it * RPI + rsub;
        const int row = m0 + r;
        if (row >= p.M) continue;
        size_t ro = 0;
        if (p.res_mode == 2) {
          int n = row / (p.Ho * p.Wo);
          int rem = row - n * (p.Ho * p.Wo);
          int ho = rem / p.Wo;
          int wo = rem - ho * p.Wo;
          ro = ((size_t)(n * (p.Ho >> 1) + (ho >> 1)) * (p.Wo >> 1) + (wo >> 1));
        }
        for (int cc = 0; cc < 4; ++cc) {
          const int c = col + cc;
          if (c >= p.K) break;
          float v = Cs[r * CS_STRIDE + c4 * 4 + cc] * (p.scale ? p.scale[c] : 1.f) + (p.shift ? p.shift[c] : 0.f);
          if (p.res_mode == 1) v += p.res[(size_t)row * p.ldr + c];
          else if (p.res_mode == 2) v += p.res[ro * p.ldr + c];
          if (p.relu) v = v > 0.f ? v : 0.f;
          p.y[(size_t)row * p.ldy + c] = v;
        }
      }
    }
    __syncthreads();  // Cs is overwritten by the next work item's staging stores
  }
}

static int g_capacity = 0;  // co-resident workers (2 per CU)
static int worker_capacity() {
  if (g_capacity == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    const int per_cu = 2;     // the LDS / VGPR residency limit
    g_capacity = per_cu * cus;
  }
  return g_capacity;
}
#define LVC_MAX_WORKERS 1024

// Persistent per-device scratch for split tiles: partial accumulators + flags (+1 error word).  Must be
// zero-initialised once by the caller and must not be shared by launches that run concurrently on
// different streams.
extern "C" long long lvc_conv_workspace_bytes(void) {
  return (long long)LVC_MAX_WORKERS * 256 * 128 * 4 + (LVC_MAX_WORKERS + 1024) * 4 + 256;   // worker flags, then 1024 range / error words
}

// C ABI -- see include/lvc_amd.h for the contract of each argument.
extern "C" int lvc_conv2d_nhwc_f32(const float* x, const float* w_packed, const float* scale,
                                   const float* shift, const float* residual, float* y, int N, int H,
                                   int W, int C, int K, int R, int S, int stride, int pad, int Kg,
                                   int relu, int res_mode, int ldy, int ldr, int mode, void* workspace,
                                   void* stream) {
  LVC_CHECK_ARG(x && w_packed && y && workspace, "null pointer");
  LVC_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && K > 0, "non-positive dimension");
  LVC_CHECK_ARG(mode == 0 || mode == 1, "mode must be 0 or 1");
  LVC_CHECK_ARG(Kg % BK == 0, "Kg must be a multiple of 32");
  if (mode == 0) {
    LVC_CHECK_ARG(C % BK == 0, "mode 0 needs C % 32 == 0");
    LVC_CHECK_ARG(Kg == R * S * C, "mode 0 needs Kg == R*S*C");
  } else {
    LVC_CHECK_ARG(C == 4 && S <= 8 && Kg == R * BK, "mode 1 needs C == 4, S <= 8, Kg == R*32");
  }
  LVC_CHECK_ARG(res_mode >= 0 && res_mode <= 2, "res_mode must be 0..2");
  LVC_CHECK_ARG(res_mode == 0 || residual, "residual pointer missing");
  LVC_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)w_packed & 15) == 0 && ((uintptr_t)workspace & 15) == 0,
                "x / w_packed / workspace must be 16-byte aligned");
  int Ho = (H + 2 * pad - R) / stride + 1;
  int Wo = (W + 2 * pad - S) / stride + 1;
  LVC_CHECK_ARG(Ho > 0 && Wo > 0, "empty output");
  if (res_mode == 2) LVC_CHECK_ARG(Ho % 2 == 0 && Wo % 2 == 0, "upsample-add needs even output size");
  long long Mll = (long long)N * Ho * Wo;
  LVC_CHECK_ARG(Mll < (1ll << 31) && (long long)N * H * W < (1ll << 31), "tensor too large for int32 pixel index");
  ConvArgs a;
  a.x = x; a.w = w_packed; a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
  a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.R = R; a.S = S; a.stride = stride; a.pad = pad;
  a.Ho = Ho; a.Wo = Wo; a.M = (int)Mll; a.Kg = Kg; a.relu = relu; a.res_mode = res_mode;
  a.ldy = ldy > 0 ? ldy : K; a.ldr = ldr > 0 ? ldr : K;
  if ((a.K & 3) == 0 && (a.ldy & 3) == 0) LVC_CHECK_ARG(((uintptr_t)y & 15) == 0, "y must be 16-byte aligned");
  const int ni = (K <= 64) ? 1 : 2;  // BN = 64 tile for the 64-channel layers, else 128
  // (a 128x256 tile / 1 worker per CU was measured: 256 VGPRs + SGPR spills, 120 vs 125 TF on the p2 3x3 -> dropped)
  const int bn = 64 * ni;
  a.tiles_n = lvc_cdiv(K, bn);
  const int tiles_m = lvc_cdiv(a.M, BM);
  a.nk = Kg / BK;
  long long units = (long long)tiles_m * a.tiles_n * a.nk;
  LVC_CHECK_ARG(units < (1ll << 31), "iteration space too large");
  a.total_units = (int)units;
  int cap = worker_capacity();
  if (cap > LVC_MAX_WORKERS) cap = LVC_MAX_WORKERS;
  const int min_units = 4;  // do not cut below 4 chunks per worker: the split-tile hand-off costs microseconds
  int workers = (int)((units + min_units - 1) / min_units);
  if (workers > cap) workers = cap;
  if (workers < 1) workers = 1;
  a.units_per_worker = (int)((units + workers - 1) / workers);
  a.nworkers = (int)((units + a.units_per_worker - 1) / a.units_per_worker);
  {
    const long long xb = (long long)N * H * W * C * 4, wb = (long long)(a.tiles_n * bn) * Kg * 4;
    LVC_CHECK_ARG(xb < (1ll << 31) && wb < (1ll << 31), "input / weight tensor must be smaller than 2 GiB (32-bit buffer offsets)");
    a.x_bytes = (int)xb; a.w_bytes = (int)wb;
  }
  a.partials = (float*)workspace;
  a.flags = (int*)((char*)workspace + (size_t)LVC_MAX_WORKERS * 256 * 128 * 4);
  a.err_index = LVC_MAX_WORKERS;
  dim3 grid(a.nworkers), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) {
    if (ni == 1) hipLaunchKernelGGL((conv_igemm_f32_kernel<0, 1>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((conv_igemm_f32_kernel<0, 2>), grid, block, 0, st, a);
  } else {
    if (ni == 1) hipLaunchKernelGGL((conv_igemm_f32_kernel<1, 1>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((conv_igemm_f32_kernel<1, 2>), grid, block, 0, st, a);
  }
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}
