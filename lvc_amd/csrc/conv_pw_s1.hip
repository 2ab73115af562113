// conv_pw_s1.hip -- pointwise (1x1 / FC) layers with a long contraction on the software-pipelined loop of conv3x3_halo_s1.hip
// (round 3): the two-way fp16 operand split, a 256-row x 64 NI-channel tile per workgroup, 8 waves as 4 x 2 (wave tile 64 x 32 NI),
// per 32-channel chunk two k16 steps with the workgroup barrier BETWEEN them, fragments rotating through the registers earlier MFMA
// groups vacate, weight planes by LDS-DMA into a ring of three chunk buffers.
//
// Why next to conv_pw_dma.hip.  That kernel streams RAW fp32 activation rows into LDS and splits them at fragment time; a wave
// owns 32 rows x all 128 channels (so that an element is split once), reads ten fragments per twelve MFMAs, and every wave issues
// six DMA instructions, waits, reads and converts at the top of every chunk: its timeline (scripts/probe_pw_timeline.py) shows
// 2.2 - 3.6 us per chunk against 0.7 - 1.0 us of matrix work -- fine for the memory-bound layers (they run at 5 TB/s of HBM traffic),
// far from the matrix pipe on the layers with a long contraction (box-head fc1: 12544 -> 1024, res4 / res5 conv1, the ViT
// linears).  Here the activation chunk goes HBM -> registers (issued three chunks ahead) -> split ONCE per tile -> two fp16
// planes in a double-buffered LDS image, after which the tap loop of the 3x3 kernel applies unchanged with one tap per chunk.
//
// Numerics: ONEACC = false -- a main (a1 b1) and a cross (a1 b2 + a2 b1, x 2^-11) accumulator, the planes of lvc_split_weights,
// |a| <= 65504: the same arithmetic as conv_pw_dma.hip / conv_f16x2.hip.  ONEACC = true -- one accumulator, row-scaled weight
// planes and activations x 2^4 as in conv3x3_halo_s1.hip (|a| <= 4094).
// The asynchronous loads are written as inline asm and waited for with counted vmcnt waits that are TIED to the destination
// registers (the compiler's own bookkeeping would wait for everything in flight, LDS-DMA included, at the first use).
#include "common.h"
#include <type_traits>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define PM 256            // rows per tile
#define AROW 64           // bytes per activation row and plane in LDS (32 halves, unpadded: the 16-byte granules are XOR-swizzled
                          // by (row >> 2) & 3 exactly like the weight rows, conflict-free for the ds_read_b128 lane groups)
#define NJ 4              // float4 slots per thread and chunk (256 rows x 8 slots / 512 threads)
#define NT 512
#define SPIN_LIMIT (1 << 24)
#define ACT_SCALE 16.f
#define ACT_MAX 4094.f

struct PwArgsS {
  const float* x;
  const unsigned short* w;   // [2][Kpad][Kg] fp16 planes
  const float* scale;
  const float* shift;
  const float* res;
  float* y;
  float* partials;
  int* flags;
  int H, W, C, K, stride, Ho, Wo, M, relu, res_mode, ldy, ldr;
  int tiles_n, nk, total_units, units_per_worker, nworkers, err_index, ngroup;
  int x_bytes;
  long long w_plane_elems;
  // PLANES instance only (lvc_conv1x1_qkv_planes_f16s1): the output goes to the attention kernel's fp16 operand planes instead of y
  unsigned short* planes;    // [6][B*H][Npad][64]: q hi, q lo, k hi, k lo, v hi, v lo
  int* pl_err;               // bit 1 (value 2) when an operand leaves fp16's range
  long long pl_PS;           // elements per plane
  int pl_N, pl_Npad, pl_H;
  float pl_qscale;           // multiplied into q (softmax scale x log2 e)
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
}
// buffer_load_dwordx4 the compiler does not track: out-of-range offsets return zeros; completion through wait_tied
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 load_untracked(u32x4 rsrc, unsigned voff, unsigned soff) {
  f32x4 v;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  return v;
}
// at most N vector-memory operations outstanding; the four registers become usable only behind it
template <int N> __device__ __forceinline__ void wait_tied(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// PRE: x holds the pre-split planes of lvc_conv3x3_nhwc_f16s1_presplit (per row and 32-channel chunk 64 B of hi halves, 64 B of lo halves)
template <int NI, bool ONEACC, bool PLANES = false, bool PRE = false>
__global__ __launch_bounds__(NT, 2) void conv_pw_s1_kernel(PwArgsS p) {
  constexpr int HN = 64 * NI;
  constexpr int PLANE_A = PM * AROW;               // bytes
  constexpr int A_BUF = 2 * PLANE_A;               // bytes per activation buffer (two planes)
  constexpr int A_BYTES = 3 * A_BUF;               // THREE buffers (98,304 B): chunk i+1 is written during phase A of chunk i, while
                                                   // slower waves may still read chunk i-1 in their phase B of chunk i-1
  constexpr int PLANE_B = HN * 64;                 // bytes
  constexpr int B_BUF = 2 * PLANE_B;
  constexpr int RING_BYTES = A_BYTES + 3 * B_BUF;
  constexpr int CS_STRIDE = HN + 4;
  constexpr int CS_BYTES = PM * CS_STRIDE * 4;
  constexpr int SMEM_BYTES = RING_BYTES > CS_BYTES ? RING_BYTES : CS_BYTES;
  constexpr int RBLK = HN / 16;
  __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[SMEM_BYTES];
  unsigned char* sA = smem_raw;
  unsigned char* sB = smem_raw + A_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fi = lane & 31, fh = lane >> 5;
  // PRE: the 16 lanes of a ds_write_b128 pass take ONE plane of four consecutive rows (4 x 64 B = the 64 banks once); with the split
  // path's lane order a row's two planes (16 KB apart: the same banks) would meet in one pass
  const int q = PRE ? ((tid >> 4) & 1) * 4 + (tid & 3) : tid & 7;
  const int arid = PRE ? (tid >> 5) * 4 + ((tid >> 2) & 3) : tid >> 3;
  const int hrow = arid;    // tile rows hrow + 64 j: 32 lanes of a ds_write_b64 cover four consecutive 64-byte rows = the 64 banks once
                            // (the 3x3 kernel's row order, made for its 80-byte pitch, put rows r and r + 4 -- the same banks -- into one pass:
                            // SQ_LDS_BANK_CONFLICT was 25 % of the LDS-active cycles on fc1)

  const int lw = lvc_xcd_remap(blockIdx.x, p.nworkers);
  const int wq = p.ngroup > 1 ? lw / p.ngroup : lw;
  const int wsel = p.ngroup > 1 ? lw - wq * p.ngroup : 0;
  int u = wq * p.units_per_worker;
  const int u_end = min(u + p.units_per_worker, p.total_units);

  // buffer descriptor of x by hand (the asm load takes four SGPRs): base, stride 0, num_records = bytes, raw dword format
  const unsigned long long xbase = (unsigned long long)p.x;
  const u32x4 xres = {(unsigned)__builtin_amdgcn_readfirstlane((unsigned)xbase), (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(xbase >> 32) & 0xffffu),
                      (unsigned)p.x_bytes, 0x00020000u};
  const int fx3 = (fi >> 2) & 3;
  const int a_row[2] = {(wm * 64 + fi) * AROW, (wm * 64 + 32 + fi) * AROW};
  const int b_row = (wn * 32 * NI + fi) * 64;
  const int b_g[2] = {((0 + fh) ^ fx3) * 16, ((2 + fh) ^ fx3) * 16};
  int a_lds[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int row = hrow + 64 * j;
    a_lds[j] = row * AROW + (((q >> 1) ^ ((row >> 2) & 3)) << 4) + (q & 1) * 8;     // bytes
  }
  float big = 0.f;
  // PRE: this thread's 16 B piece q of a row's chunk is granule q % 4 of plane q / 4 -- stored as it comes
  int a_pre[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int row = hrow + 64 * j;
    a_pre[j] = (q >> 2) * (PM * AROW) + row * AROW + (((q & 3) ^ ((row >> 2) & 3)) << 4);
  }
#ifdef PW_TIMELINE
  // diagnostics build only (scripts/probe_pw_s1_timeline.py): cycles of this wave in a tile's prologue, chunk loop, hand-off, epilogue
  unsigned long long tl_pro = 0, tl_loop = 0, tl_hand = 0, tl_cs = 0, tl_rows = 0, tl_t0 = __builtin_readcyclecounter();
  unsigned long long tl_wa = 0, tl_wv = 0, tl_wb = 0;   // inside the chunk loop: wait for the activation registers, vmcnt wait, barrier
  unsigned tl_tiles = 0, tl_chunks = 0;
#endif

  while (u < u_end) {
#ifdef PW_TIMELINE
    const unsigned long long tl_a = __builtin_readcyclecounter();
#endif
    const int tile = u / p.nk;
    const int cc0 = u - tile * p.nk;
    const int cc1 = min(p.nk, cc0 + (u_end - u));
    const int tile_n = p.ngroup > 1 ? wsel : tile % p.tiles_n;
    const int tile_m = p.ngroup > 1 ? tile : tile / p.tiles_n;
    const int m0 = tile_m * PM;
    const int n0 = tile_n * HN;

    unsigned a_off[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int m = m0 + hrow + 64 * j;
      const int n = m / (p.Ho * p.Wo);
      const int rem = m - n * (p.Ho * p.Wo);
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      a_off[j] = m < p.M ? (unsigned)(((n * p.H + ho * p.stride) * p.W + wo * p.stride) * p.C + q * 4) * 4u : 0x80000000u;
    }
    const unsigned short* bsrc[NI];
    int bdst[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int idx = wave * NI + j;
      const int pl = idx / RBLK, rb = idx - pl * RBLK;
      const int row = rb * 16 + (lane >> 2);
      const int G = (lane & 3) ^ ((row >> 2) & 3);
      bsrc[j] = p.w + (size_t)pl * p.w_plane_elems + (size_t)(n0 + row) * p.C + G * 8;
      bdst[j] = pl * PLANE_B + rb * 16 * 64;
    }
    auto dma_B = [&](int cc, int buf) {
#pragma unroll
      for (int j = 0; j < NI; ++j) glds16(bsrc[j] + cc * 32, sB + buf * B_BUF + bdst[j]);
    };
    // three activation register sets: chunk cc lives in set cc % 3 from its load (issued at chunk cc - 3 + ...) to its split
    f32x4 ar[3][NJ];
    auto load_A = [&](int set, int cc) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) ar[set][j] = load_untracked(xres, a_off[j], (unsigned)cc * 128u);
    };
    auto store_A = [&](int set, unsigned char* dstA) {
      if (PRE) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          *reinterpret_cast<f32x4*>(dstA + a_pre[j]) = ar[set][j];
        }
        return;
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        f16x4 h, m;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = ONEACC ? ar[set][j][e] * ACT_SCALE : ar[set][j][e];
          const f16 hh = (f16)a;
          h[e] = hh;
          m[e] = ONEACC ? (f16)(a - (float)hh) : (f16)((a - (float)hh) * 2048.f);
          big = fmaxf(big, fabsf(ar[set][j][e]));
        }
        *reinterpret_cast<f16x4*>(dstA + a_lds[j]) = h;
        *reinterpret_cast<f16x4*>(dstA + PLANE_A + a_lds[j]) = m;
      }
    };

    f32x16 acc[2][NI], accx[ONEACC ? 1 : 2][ONEACC ? 1 : NI];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < NI; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          acc[a][b][e] = 0.f;
          if (!ONEACC) accx[ONEACC ? 0 : a][ONEACC ? 0 : b][e] = 0.f;
        }

    f16x8 ahi[2], alo[2], bhi[NI], blo[NI];
    auto rdA = [&](const unsigned char* A, int pl, int mi, int s2) {
      return *reinterpret_cast<const f16x8*>(A + pl * PLANE_A + a_row[mi] + b_g[s2]);
    };
    auto rdB = [&](const unsigned char* B, int pl, int ni, int s2) {
      return *reinterpret_cast<const f16x8*>(B + pl * PLANE_B + b_row + ni * 32 * 64 + b_g[s2]);
    };
    // one k16 step (see conv3x3_halo_s1.hip `step_body`): late reads of this step under G1, early reads of the next under G2
    auto step_body = [&](const unsigned char* A, int s2, const unsigned char* B, const unsigned char* An, int s2n, const unsigned char* Bn) {
#ifndef PW_DIAG_SKIP_LATE_READS      // diagnostics build only (wrong results): half of the fragment reads gone -- is the chunk bound by the LDS port?
      alo[0] = rdA(A, 1, 0, s2);
      bhi[0] = rdB(B, 0, 0, s2);
      alo[1] = rdA(A, 1, 1, s2);
      if (NI == 2) bhi[NI - 1] = rdB(B, 0, NI - 1, s2);
#endif
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          f32x16& c = ONEACC ? acc[mi][ni] : accx[ONEACC ? 0 : mi][ONEACC ? 0 : ni];
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[mi], blo[ni], c, 0, 0, 0);
        }
      f16x8 blo_n[NI], ahi_n[2];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) blo_n[ni] = rdB(Bn, 1, ni, s2n);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          f32x16& c = ONEACC ? acc[mi][ni] : accx[ONEACC ? 0 : mi][ONEACC ? 0 : ni];
#ifndef PW_DIAG_DROP_CROSS       // gate check (scripts/perturbed_build_check.sh): without this term every pointwise layer is a 2^-11 product
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[mi], bhi[ni], c, 0, 0, 0);
#endif
        }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) ahi_n[mi] = rdA(An, 0, mi, s2n);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[mi], bhi[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) ahi[mi] = ahi_n[mi];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) blo[ni] = blo_n[ni];
    };

    // ---- prologue.  VMEM order: A(c0), A(c0+1), A(c0+2), B(c0), B(c0+1); chunks past the unit's end re-fetch its last chunk
    const int last = cc1 - 1;
    load_A(0, cc0);
    load_A(1, min(cc0 + 1, last));
    load_A(2, min(cc0 + 2, last));
    dma_B(cc0, 0);
    dma_B(min(cc0 + 1, last), 1);
    wait_tied<2 * NJ + 2 * NI>(ar[0][0], ar[0][1], ar[0][2], ar[0][3]);
    store_A(0, sA);
    wait_vm<0>();
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) ahi[mi] = rdA(sA, 0, mi, 0);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) blo[ni] = rdB(sB, 1, ni, 0);
#ifdef PW_DIAG_SKIP_LATE_READS
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) alo[mi] = rdA(sA, 1, mi, 0);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) bhi[ni] = rdB(sB, 0, ni, 0);
#endif
    // At the top of chunk i (i = cc - cc0) the VMEM queue holds, oldest first:  A(i+1), A(i+2) | B(i+1) issued at i-1  [i = 0: the
    // prologue's order, everything landed but nothing that matters is missing].  Steady state per chunk, program order:
    //   phase A: split + store A(i+1) -> needs A(i+1): everything issued after it may stay outstanding
    //   barrier: needs B(i+1)
    //   phase B: issue B(i+2), A(i+3)
    // so before the store at most  [A(i+2)] + [B(i+1)] = NJ + NI  newer operations exist (issue order: ..., B(i+1), A(i+2) at chunk
    // i-1) and before the barrier at most A(i+2) = NJ.
#ifdef PW_TIMELINE
    const unsigned long long tl_b = __builtin_readcyclecounter();
    tl_pro += tl_b - tl_a;
#endif
    auto chunk = [&](int i, int k) {      // k = i % 3, compile-time: activation buffer, weight slot and register set of chunk i
      const int cc = cc0 + i;
      const int kn = (k + 1) % 3;
      const unsigned char* Acur = sA + k * A_BUF;
      unsigned char* Anext = sA + kn * A_BUF;
      const unsigned char* Bcur = sB + k * B_BUF;
      const unsigned char* Bnext = sB + kn * B_BUF;
      // ---- phase A: chunk i+1 (set / buffer kn: last read in phase B of chunk i-2, before the previous barrier) is split and stored
#ifdef PW_TIMELINE
      const unsigned long long tl_1 = __builtin_readcyclecounter();
#endif
      wait_tied<NJ + NI>(ar[kn][0], ar[kn][1], ar[kn][2], ar[kn][3]);
#ifdef PW_TIMELINE
      tl_wa += __builtin_readcyclecounter() - tl_1;
#endif
      step_body(Acur, 0, Bcur, Acur, 1, Bcur);
      store_A(kn, Anext);
      __builtin_amdgcn_sched_barrier(0);
#ifdef PW_TIMELINE
      const unsigned long long tl_2 = __builtin_readcyclecounter();
      wait_vm<NJ>();
      const unsigned long long tl_3 = __builtin_readcyclecounter();
      __builtin_amdgcn_s_barrier();
      tl_wv += tl_3 - tl_2; tl_wb += __builtin_readcyclecounter() - tl_3;
#else
      wait_vm<NJ>();
      __builtin_amdgcn_s_barrier();
#endif
      __builtin_amdgcn_sched_barrier(0);
      // ---- phase B: weights of chunk i+2 into chunk i-1's slot, activations of chunk i+3 into chunk i's registers
      dma_B(min(cc + 2, last), (k + 2) % 3);
      load_A(k, min(cc + 3, last));
      step_body(Acur, 1, Bcur, Anext, 0, Bnext);
      __builtin_amdgcn_sched_barrier(0);
    };
    const int nchunks = cc1 - cc0;
#pragma unroll 1
    for (int i = 0; i < nchunks; i += 3) {
      chunk(i, 0);
      if (i + 1 < nchunks) chunk(i + 1, 1);
      if (i + 2 < nchunks) chunk(i + 2, 2);
    }
    wait_vm<0>();
    __syncthreads();
#ifdef PW_TIMELINE
    const unsigned long long tl_c = __builtin_readcyclecounter();
    tl_loop += tl_c - tl_b; tl_chunks += nchunks; ++tl_tiles;
#endif
    u += nchunks;
    if (!ONEACC) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[mi][ni][e] += accx[ONEACC ? 0 : mi][ONEACC ? 0 : ni][e] * (1.f / 2048.f);
    }

    // ---- split tiles (the protocol of the other stream-K kernels)
    if (cc0 != 0) {
      // The lane offset is made opaque HERE: left visible, the sixteen store addresses are loop invariants the compiler forms at the
      // kernel's top and -- in the two-accumulator instances -- spills; every reload is a `vmcnt(0)` that waits for the store before it
      // to be acknowledged (sixteen round trips in series on the hand-off the tile's owner is waiting for).
      unsigned toff = (unsigned)tid * 16u;
      asm volatile("" : "+v"(toff));
      char* dst = reinterpret_cast<char*>(p.partials + (size_t)lw * (NT * 32 * NI)) + toff;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {
            f32x4 v = {acc[mi][ni][e4 * 4 + 0], acc[mi][ni][e4 * 4 + 1], acc[mi][ni][e4 * 4 + 2], acc[mi][ni][e4 * 4 + 3]};
            *reinterpret_cast<f32x4*>(dst + (size_t)((mi * NI + ni) * 4 + e4) * NT * 16) = v;
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(p.flags + lw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      continue;
    }
    if (cc1 < p.nk) {
      const int last_unit = tile * p.nk + p.nk - 1;
      const int wstep = p.ngroup > 1 ? p.ngroup : 1;
      const int last_worker = (last_unit / p.units_per_worker) * wstep + wsel;
      for (int pw = lw + wstep; pw <= last_worker; pw += wstep) {
        if (tid == 0) {
          int spins = 0;
          while (__hip_atomic_load(p.flags + pw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > SPIN_LIMIT) { atomicOr(p.flags + p.err_index, 1); break; }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        const float* src = p.partials + (size_t)pw * (NT * 32 * NI);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(src + ((size_t)((mi * NI + ni) * 4 + e4) * NT + tid) * 4);
              acc[mi][ni][e4 * 4 + 0] += v[0]; acc[mi][ni][e4 * 4 + 1] += v[1];
              acc[mi][ni][e4 * 4 + 2] += v[2]; acc[mi][ni][e4 * 4 + 3] += v[3];
            }
        __syncthreads();
        if (tid == 0) __hip_atomic_store(p.flags + pw, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }

    // ---- epilogue through LDS: affine, residual (same shape, or the nearest-x2 upsample of the half-size map), ReLU / GELU.
    // The residual rows are requested one group of four output rows AHEAD (the first group before the accumulators go through
    // LDS): not one exposed memory round trip per group.
#ifdef PW_TIMELINE
    const unsigned long long tl_d = __builtin_readcyclecounter();
    tl_hand += tl_d - tl_c;
#endif
#ifdef PW_TIMELINE
    unsigned long long tl_e = 0;
#endif
    if constexpr (ONEACC) {
      constexpr int C4 = HN / 4;
      constexpr int RPI = NT / C4;
      constexpr int NIT = PM / RPI;
      constexpr int NG = 4;
      constexpr int NGR = NIT / NG;
      const int c4 = tid % C4, rsub = tid / C4;
      const int col = n0 + c4 * 4;
      const bool col_ok = col < p.K;
      const bool has_res = p.res_mode != 0 && col_ok;
      // Residual rows: requested one group of four output rows AHEAD of their use into two register sets that alternate, the groups
      // unrolled: the compiler's own counted wait then leaves the group just requested in flight.  (With one register set copied
      // forward per group it waited for everything at the copy -- a memory round trip per group, a third to a half of the time of
      // the layers with a residual: scripts/probe_pw_s1_timeline.py.)  Rows past the end re-read the last row.
      // Residual rows are requested in row order (it = 0, 1, ...): the row -> (image, y, x) decomposition the upsample-add needs is
      // formed once and then STEPPED by RPI rows (a compare and a subtract instead of two integer divisions per row: those cost the
      // FPN laterals ~9 k cycles per tile).  Rows past the end re-read the last valid row.
      int r_m = m0 + rsub;
      r_m = r_m < p.M ? r_m : p.M - 1;
      int r_n = 0, r_ho = 0, r_wo = 0;
      if (p.res_mode == 2) {
        r_n = r_m / (p.Ho * p.Wo);
        const int rem = r_m - r_n * (p.Ho * p.Wo);
        r_ho = rem / p.Wo;
        r_wo = rem - r_ho * p.Wo;
      }
      auto res_next = [&]() {
        const size_t ro = p.res_mode == 2 ? ((size_t)(r_n * (p.Ho >> 1) + (r_ho >> 1)) * (p.Wo >> 1) + (r_wo >> 1)) : (size_t)r_m;
        const float* a = p.res + ro * p.ldr + col;
        if (r_m + RPI < p.M) {
          r_m += RPI;
          r_wo += RPI;
          while (r_wo >= p.Wo) {
            r_wo -= p.Wo;
            if (++r_ho == p.Ho) { r_ho = 0; ++r_n; }
          }
        }
        return a;
      };
      f32x4 rr[2][NG];
      auto res_issue = [&](int set, int g) {
#pragma unroll
        for (int i = 0; i < NG; ++i) rr[set][i] = *reinterpret_cast<const f32x4*>(res_next());
      };
      if (has_res) res_issue(0, 0);
      float* Cs = reinterpret_cast<float*>(smem_raw);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
            const int colc = wn * 32 * NI + ni * 32 + fi;
            Cs[row * CS_STRIDE + colc] = acc[mi][ni][e];
          }
      __syncthreads();
#ifdef PW_TIMELINE
      tl_e = __builtin_readcyclecounter();
      tl_cs += tl_e - tl_d;
#endif
      if (col_ok) {
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + col);
        if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + col);
        const float* cs = Cs + rsub * CS_STRIDE + c4 * 4;
        float* const yb = p.y + (size_t)(m0 + rsub) * p.ldy + col;
        // one copy of the row loop per (residual?, activation): with the modes tested inside, every row ended in the compiler's
        // vmcnt(0) lgkmcnt(0) -- its LDS read and the acknowledgement of the previous row's store, one after the other
        auto rows = [&](auto res_tag, auto act_tag) {
          constexpr bool RES = decltype(res_tag)::value;
          constexpr int ACT = decltype(act_tag)::value;
          auto group = [&](int g, const f32x4* rcur) {
            f32x4 v[NG];
#pragma unroll
            for (int i = 0; i < NG; ++i) v[i] = *reinterpret_cast<const f32x4*>(cs + (g + i) * RPI * CS_STRIDE);
#pragma unroll
            for (int i = 0; i < NG; ++i) {
              const int m = m0 + (g + i) * RPI + rsub;
              if (m < p.M) {
                f32x4 o = v[i] * sc + sh;
                if (RES) o += rcur[i];
                if (ACT == 1) {
                  o[0] = o[0] > 0.f ? o[0] : 0.f; o[1] = o[1] > 0.f ? o[1] : 0.f;
                  o[2] = o[2] > 0.f ? o[2] : 0.f; o[3] = o[3] > 0.f ? o[3] : 0.f;
                } else if (ACT == 2) {     // torch.nn.GELU(), the expression of lvc_gelu
#pragma unroll
                  for (int e = 0; e < 4; ++e) o[e] = o[e] * 0.5f * (1.f + erff(o[e] * 0.70710678118654752440f));
                }
                *reinterpret_cast<f32x4*>(yb + (size_t)((g + i) * RPI) * p.ldy) = o;
              }
            }
          };
          if constexpr (RES) {
#pragma unroll
            for (int gi = 0; gi < NGR; ++gi) {
              if (gi + 1 < NGR) res_issue((gi + 1) & 1, (gi + 1) * NG);
              group(gi * NG, rr[gi & 1]);
            }
          } else {
#pragma unroll 1
            for (int g = 0; g < NIT; g += NG) group(g, rr[0]);
          }
        };
        using A0 = std::integral_constant<int, 0>; using A1 = std::integral_constant<int, 1>; using A2 = std::integral_constant<int, 2>;
        if constexpr (PLANES) {
          // y = qkv of a ViT block (row m = image b, token n; column = (q | k | v, head, d)): what lvc_mha_mfma's first pass (mha_split_kernel)
          // makes of the fp32 tensor -- q * qscale, k, v as fp16 (hi, lo) pairs in [B*H][Npad][64] planes -- written here, the fp32 tensor never
          // exists.  Same arithmetic in the same order (affine, one fp32 rounding, x qscale, split): the planes are bit-identical.
          const int hd = p.pl_H * 64;
          const int which = col / hd, hh = (col - which * hd) >> 6, d = col & 63;
          const float f = which == 0 ? p.pl_qscale : 1.f;
          unsigned short* const ph = p.planes + (size_t)(2 * which) * p.pl_PS + d;
          float bigp = 0.f;
#pragma unroll 1
          for (int g = 0; g < NIT; ++g) {
            const int m = m0 + g * RPI + rsub;
            if (m < p.M) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(cs + g * RPI * CS_STRIDE);
              f32x4 o = v * sc + sh;
              o *= f;
              const int b = m / p.pl_N, n = m - b * p.pl_N;
              f16x4 hi, lo;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float a = o[e];
                asm volatile("" : "+v"(a));      // the rounded value, as mha_split_kernel (vit.hip) splits it
                bigp = fmaxf(bigp, fabsf(a));
                if (a != a) bigp = INFINITY;
                hi[e] = (f16)a;
                lo[e] = (f16)(a - (float)hi[e]);
              }
              unsigned short* dst = ph + ((size_t)(b * p.pl_H + hh) * p.pl_Npad + n) * 64;
              *reinterpret_cast<f16x4*>(dst) = hi;
              *reinterpret_cast<f16x4*>(dst + p.pl_PS) = lo;
            }
          }
          if (p.pl_err && !(bigp <= 65504.f)) atomicOr(p.pl_err, 2);
        } else if (has_res) {
          if (p.relu == 1) rows(std::true_type{}, A1{}); else if (p.relu == 2) rows(std::true_type{}, A2{}); else rows(std::true_type{}, A0{});
        } else {
          if (p.relu == 1) rows(std::false_type{}, A1{}); else if (p.relu == 2) rows(std::false_type{}, A2{}); else rows(std::false_type{}, A0{});
        }
      }
    } else {
      // The two-accumulator form has no registers to spare: with the row groups unrolled (as above) the compiler spills 60 - 90
      // registers and reloads them inside the loop (res2 conv3 0.257 -> 0.378 ms).  Rolled group loop, one residual register set
      // copied forward per group, at the price of the compiler's full wait at the copy.
      constexpr int C4 = HN / 4;
      constexpr int RPI = NT / C4;
      constexpr int NIT = PM / RPI;
      constexpr int NG = 4;
      const int c4 = tid % C4, rsub = tid / C4;
      const int col = n0 + c4 * 4;
      const bool col_ok = col < p.K;
      const bool has_res = p.res_mode != 0 && col_ok;
      auto res_load = [&](int it) {
        const int m = m0 + it * RPI + rsub;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < p.M) {
          size_t ro = (size_t)m;
          if (p.res_mode == 2) {
            const int n = m / (p.Ho * p.Wo);
            const int rem = m - n * (p.Ho * p.Wo);
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            ro = ((size_t)(n * (p.Ho >> 1) + (ho >> 1)) * (p.Wo >> 1) + (wo >> 1));
          }
          v = *reinterpret_cast<const f32x4*>(p.res + ro * p.ldr + col);
        }
        return v;
      };
      f32x4 rv[NG];
#pragma unroll
      for (int i = 0; i < NG; ++i) rv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (has_res) {
#pragma unroll
        for (int i = 0; i < NG; ++i) rv[i] = res_load(i);
      }
      float* Cs = reinterpret_cast<float*>(smem_raw);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
            const int colc = wn * 32 * NI + ni * 32 + fi;
            Cs[row * CS_STRIDE + colc] = acc[mi][ni][e];
          }
      __syncthreads();
#ifdef PW_TIMELINE
      tl_e = __builtin_readcyclecounter();
      tl_cs += tl_e - tl_d;
#endif
      if (col_ok) {
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + col);
        if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + col);
        const float* cs = Cs + rsub * CS_STRIDE + c4 * 4;
        float* const yb = p.y + (size_t)(m0 + rsub) * p.ldy + col;
        // one copy of the row loop per (residual?, activation): with the modes tested inside, every row ended in the compiler's
        // vmcnt(0) lgkmcnt(0) -- its LDS read and the acknowledgement of the previous row's store, one after the other
        auto rows = [&](auto res_tag, auto act_tag) {
          constexpr bool RES = decltype(res_tag)::value;
          constexpr int ACT = decltype(act_tag)::value;
#pragma unroll 1
          for (int g = 0; g < NIT; g += NG) {
            f32x4 rn[NG];
            if (RES) {
#pragma unroll
              for (int i = 0; i < NG; ++i) rn[i] = f32x4{0.f, 0.f, 0.f, 0.f};
              if (g + NG < NIT) {
#pragma unroll
                for (int i = 0; i < NG; ++i) rn[i] = res_load(g + NG + i);
              }
            }
            f32x4 v[NG];
#pragma unroll
            for (int i = 0; i < NG; ++i) v[i] = *reinterpret_cast<const f32x4*>(cs + (g + i) * RPI * CS_STRIDE);
#pragma unroll
            for (int i = 0; i < NG; ++i) {
              const int m = m0 + (g + i) * RPI + rsub;
              if (m < p.M) {
                f32x4 o = v[i] * sc + sh;
                if (RES) o += rv[i];
                if (ACT == 1) {
                  o[0] = o[0] > 0.f ? o[0] : 0.f; o[1] = o[1] > 0.f ? o[1] : 0.f;
                  o[2] = o[2] > 0.f ? o[2] : 0.f; o[3] = o[3] > 0.f ? o[3] : 0.f;
                } else if (ACT == 2) {     // torch.nn.GELU(), the expression of lvc_gelu
#pragma unroll
                  for (int e = 0; e < 4; ++e) o[e] = o[e] * 0.5f * (1.f + erff(o[e] * 0.70710678118654752440f));
                }
                *reinterpret_cast<f32x4*>(yb + (size_t)((g + i) * RPI) * p.ldy) = o;
              }
            }
            if (RES) {
#pragma unroll
              for (int i = 0; i < NG; ++i) rv[i] = rn[i];
            }
          }
        };
        using A0 = std::integral_constant<int, 0>; using A1 = std::integral_constant<int, 1>; using A2 = std::integral_constant<int, 2>;
        if (has_res) {
          if (p.relu == 1) rows(std::true_type{}, A1{}); else if (p.relu == 2) rows(std::true_type{}, A2{}); else rows(std::true_type{}, A0{});
        } else {
          if (p.relu == 1) rows(std::false_type{}, A1{}); else if (p.relu == 2) rows(std::false_type{}, A2{}); else rows(std::false_type{}, A0{});
        }
      }
    }
    __syncthreads();
#ifdef PW_TIMELINE
    tl_rows += __builtin_readcyclecounter() - tl_e;
#endif
  }
#ifdef PW_TIMELINE
  if (lane == 0 && (blockIdx.x & 15) == 0) {
    unsigned long long* d = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(p.partials) + (size_t)512 * 256 * 128 * 4) + ((blockIdx.x >> 4) * 8 + wave) * 16;
    d[0] = tl_pro; d[1] = tl_loop; d[2] = tl_hand; d[3] = tl_cs; d[4] = tl_rows; d[5] = tl_tiles; d[6] = tl_chunks; d[7] = __builtin_readcyclecounter() - tl_t0; d[8] = 1; d[9] = tl_wa; d[10] = tl_wv; d[11] = tl_wb;
  }
#endif
  if (PRE) return;      // the producer raised this layer's range word where its output x 2^4 left fp16's range
  if (!(big <= (ONEACC ? ACT_MAX : 65504.f))) atomicOr(p.flags + p.err_index, big < INFINITY ? 2 : 4);      // finite / non-finite: see conv3x3_halo_s1.hip
}

#define LVC_MAX_WORKERS 1024
static int g_cus_pw_s = 0;

struct PwPlanes { unsigned short* planes; int* err; long long PS; int N, Npad, H; float qscale; };

static int pw_s1_launch(bool oneacc, const float* x, const unsigned short* w_split, const float* scale, const float* shift,
                        const float* residual, float* y, int N, int H, int W, int C, int K, int stride, int relu, int res_mode,
                        int ldy, int ldr, void* workspace, void* stream, const PwPlanes* pl = nullptr, bool pre = false) {
  LVC_CHECK_ARG(x && w_split && y && workspace, "null pointer");
  LVC_CHECK_ARG(!oneacc || scale, "the single-accumulator form needs the row factors in `scale`");
  LVC_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && K > 0 && stride >= 1, "non-positive dimension");
  LVC_CHECK_ARG(C % 32 == 0, "needs C % 32 == 0");
  LVC_CHECK_ARG(relu >= 0 && relu <= 2, "relu: 0 none, 1 ReLU, 2 GELU");
  LVC_CHECK_ARG(res_mode >= 0 && res_mode <= 2 && (res_mode == 0 || residual), "bad residual");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  if (res_mode == 2) LVC_CHECK_ARG(Ho % 2 == 0 && Wo % 2 == 0, "upsample-add needs even output size");
  PwArgsS a;
  a.x = x; a.w = w_split; a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
  a.H = H; a.W = W; a.C = C; a.K = K; a.stride = stride; a.Ho = Ho; a.Wo = Wo;
  const long long Mll = (long long)N * Ho * Wo;
  LVC_CHECK_ARG(Mll < (1ll << 31), "too many output pixels");
  a.M = (int)Mll; a.relu = relu; a.res_mode = res_mode;
  a.ldy = ldy > 0 ? ldy : K; a.ldr = ldr > 0 ? ldr : K;
  LVC_CHECK_ARG((K & 3) == 0 && (a.ldy & 3) == 0 && (res_mode == 0 || (a.ldr & 3) == 0), "K, ldy, ldr must be multiples of 4");
  LVC_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)w_split & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)workspace & 15) == 0 &&
                    ((uintptr_t)scale & 15) == 0 && ((uintptr_t)shift & 15) == 0 && ((uintptr_t)residual & 15) == 0,
                "pointers must be 16-byte aligned");
  const long long xb = (long long)N * H * W * C * 4;
  LVC_CHECK_ARG(xb < (1ll << 31), "input tensor must be smaller than 2 GiB");
  a.x_bytes = (int)xb;
  const int ni = K <= 64 ? 1 : 2;
  const int HN = 64 * ni;
  a.tiles_n = lvc_cdiv(K, HN);
  a.nk = C / 32;
  const int tiles_m = lvc_cdiv(a.M, PM);
  long long units = (long long)tiles_m * a.tiles_n * a.nk;
  LVC_CHECK_ARG(units < (1ll << 31), "iteration space too large");
  a.total_units = (int)units;
  a.w_plane_elems = (long long)(lvc_cdiv(K, 128) * 128) * C;
  if (g_cus_pw_s == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    g_cus_pw_s = cus;
  }
  int cap = g_cus_pw_s;
  if (cap > LVC_MAX_WORKERS) cap = LVC_MAX_WORKERS;
  a.ngroup = 1;
  if ((a.tiles_n == 2 || a.tiles_n == 4 || a.tiles_n == 8 || a.tiles_n == 16) && cap % a.tiles_n == 0 && units / a.tiles_n >= (long long)(cap / a.tiles_n) * 4) {
    a.ngroup = a.tiles_n;       // the workers of one row tile's channel tiles are neighbours on one XCD: the rows come from L2
    units /= a.tiles_n;
    cap /= a.tiles_n;
    a.total_units = (int)units;
  }
  const int min_units = 4;     // a worker's pipeline restarts per tile segment: keep segments >= 4 chunks
  int workers = (int)((units + min_units - 1) / min_units);
  if (workers > cap) workers = cap;
  a.units_per_worker = (int)((units + workers - 1) / workers);
  a.nworkers = (int)((units + a.units_per_worker - 1) / a.units_per_worker) * a.ngroup;
  a.partials = (float*)workspace;
  a.flags = (int*)((char*)workspace + (size_t)LVC_MAX_WORKERS * 256 * 128 * 4);
  a.err_index = LVC_MAX_WORKERS + lvc_range_slot();   // the layer's own range word (common.cpp)
  hipStream_t st = (hipStream_t)stream;
  a.planes = nullptr; a.pl_err = nullptr; a.pl_PS = 0; a.pl_N = a.pl_Npad = a.pl_H = 0; a.pl_qscale = 1.f;
  if (pl) {
    LVC_CHECK_ARG(oneacc && ni == 2 && res_mode == 0 && relu == 0, "the planes epilogue exists for the single-accumulator 128-column tile");
    a.planes = pl->planes; a.pl_err = pl->err; a.pl_PS = pl->PS; a.pl_N = pl->N; a.pl_Npad = pl->Npad; a.pl_H = pl->H; a.pl_qscale = pl->qscale;
    hipLaunchKernelGGL((conv_pw_s1_kernel<2, true, true>), dim3(a.nworkers), dim3(NT), 0, st, a);
    LVC_CHECK_LAUNCH();
    return LVC_OK;
  }
  if (pre) {
    LVC_CHECK_ARG(oneacc && ni == 2 && stride == 1, "pre-split input: the single-accumulator 128-column tile, stride 1");
    hipLaunchKernelGGL((conv_pw_s1_kernel<2, true, false, true>), dim3(a.nworkers), dim3(NT), 0, st, a);
    LVC_CHECK_LAUNCH();
    return LVC_OK;
  }
  if (oneacc) {
    if (ni == 1) hipLaunchKernelGGL((conv_pw_s1_kernel<1, true>), dim3(a.nworkers), dim3(NT), 0, st, a);
    else hipLaunchKernelGGL((conv_pw_s1_kernel<2, true>), dim3(a.nworkers), dim3(NT), 0, st, a);
  } else {
    if (ni == 1) hipLaunchKernelGGL((conv_pw_s1_kernel<1, false>), dim3(a.nworkers), dim3(NT), 0, st, a);
    else hipLaunchKernelGGL((conv_pw_s1_kernel<2, false>), dim3(a.nworkers), dim3(NT), 0, st, a);
  }
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// Pointwise (R = S = 1, pad 0) layer y = act(conv(x, w) * scale + shift (+ residual)), x [N,H,W,C] fp32 NHWC (C % 32 == 0), two-way
// fp16 split with the numerics and weight planes of lvc_conv2d_nhwc_f16x2 (lvc_split_weights; |a| <= 65504).  relu: 0 none,
// 1 ReLU, 2 exact GELU.  Meant for layers with a long contraction (C >= 256); any C % 32 == 0 is accepted.
extern "C" int lvc_conv1x1_nhwc_f16x2_pipe(const float* x, const unsigned short* w_split, const float* scale, const float* shift,
                                            const float* residual, float* y, int N, int H, int W, int C, int K, int stride, int relu,
                                            int res_mode, int ldy, int ldr, void* workspace, void* stream) {
  return pw_s1_launch(false, x, w_split, scale, shift, residual, y, N, H, W, C, K, stride, relu, res_mode, ldy, ldr, workspace, stream);
}

// The single-accumulator form: w_split / scale from lvc_split_weights_rowscaled as for lvc_conv3x3_nhwc_f16s1 (|a| <= 4094).
extern "C" int lvc_conv1x1_nhwc_f16s1(const float* x, const unsigned short* w_split, const float* scale, const float* shift,
                                       const float* residual, float* y, int N, int H, int W, int C, int K, int stride, int relu,
                                       int res_mode, int ldy, int ldr, void* workspace, void* stream) {
  return pw_s1_launch(true, x, w_split, scale, shift, residual, y, N, H, W, C, K, stride, relu, res_mode, ldy, ldr, workspace, stream);
}

// lvc_conv1x1_nhwc_f16s1 on the pre-split planes lvc_conv3x3_nhwc_f16s1_presplit wrote (x [N,H,W,C]: per pixel and 32-channel chunk 32 hi
// halves, 32 lo halves; C % 32 == 0, K > 64, stride 1): the same products in the same order -- results bit-identical to the fp32 hand-over.
// The layer's range word is raised by the PRODUCER (its `next_slot`).  Reference: resnet.py:200-212 (conv2 -> conv3 of a bottleneck).
extern "C" int lvc_conv1x1_nhwc_f16s1_presplit(const void* x, const unsigned short* w_split, const float* scale, const float* shift,
                                                const float* residual, float* y, int N, int H, int W, int C, int K, int relu,
                                                int res_mode, int ldy, int ldr, void* workspace, void* stream) {
  return pw_s1_launch(true, (const float*)x, w_split, scale, shift, residual, y, N, H, W, C, K, 1, relu, res_mode, ldy, ldr, workspace, stream, nullptr, true);
}

// The qkv layer of a ViT block straight into the attention kernel's operand planes (round 5): lvc_conv1x1_nhwc_f16s1 on x [B*N][C] with
// K = 3 * H * 64 output columns (q | k | v, head, d) whose epilogue writes what lvc_mha_mfma's first pass would make of the fp32 result --
// q * qscale, k, v as fp16 (hi, lo) pairs, planes [6][B*H][Npad][64] (Npad = N rounded up to 128; rows N..Npad-1 are NOT written: the
// caller keeps them zero) -- bit-identical to lvc_conv1x1_nhwc_f16s1 + the split.  scale = the softmax scale (q is multiplied by scale * log2(e)).  An operand beyond
// fp16's range (or NaN) raises bit 1 of *err_word.  Reference: tools/run_nearest_neighbours.py:102-128 (the DINO ViT's attention.qkv).
extern "C" int lvc_conv1x1_qkv_planes_f16s1(const float* x, const unsigned short* w_split, const float* scale, const float* shift, void* planes,
                                            int B, int N, int C, int H, float softmax_scale, int* err_word, void* workspace, void* stream) {
  LVC_CHECK_ARG(planes && B > 0 && N > 0 && H > 0, "bad arguments");
  const int Npad = (N + 127) / 128 * 128;
  PwPlanes pl;
  pl.planes = (unsigned short*)planes; pl.err = err_word; pl.N = N; pl.Npad = Npad; pl.H = H; pl.qscale = softmax_scale * 1.44269504088896340736f;      // the product lvc_mha_mfma forms (in float)
  pl.PS = (long long)B * H * Npad * 64;
  // y is only a non-null placeholder for the shared argument checks (never written)
  return pw_s1_launch(true, x, w_split, scale, shift, nullptr, (float*)planes, B * N, 1, 1, C, 3 * H * 64, 1, 0, 0, 0, 0, workspace, stream, &pl);
}
