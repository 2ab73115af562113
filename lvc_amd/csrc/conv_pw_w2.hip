// conv_pw_w2.hip -- pointwise (1x1 / FC) layers with >= 256 input AND output channels on a 256-row x 256-channel workgroup tile
// (round 5).  Single-accumulator two-way fp16 split (the numerics of conv_pw_s1.hip's ONEACC form: row-scaled weight planes of
// lvc_split_weights_rowscaled, activations x 2^4, |a| <= 4094 or bit 1 of the layer's range word is raised).
//
// Why next to conv_pw_s1.hip.  That kernel's 256 x 128 tile (wave tile 64 x 64) takes in, per 32-deep chunk and CU, 32 KB of
// activations + 16 KB of weight planes for 24 MFMAs per wave and reads 8 fragments per 12 MFMAs from LDS.  The layers it runs with a
// short contraction and wide outputs (res4 / res5 conv1 and conv3, the FPN laterals) are bound by what a CU takes in per second through
// its vector-memory path (profiles/r03_pw_s1_timeline.txt, profiles/r04_mfma_ingest_ceiling.txt: 22 - 27 GB/s per CU against a
// skeleton's 40), not by the matrix pipe.  Here a wave owns 64 rows x 128 channels on ONE accumulator set (128 registers; two
// accumulators would need 256): per k16 step a wave issues 24 MFMAs for 12 fragment reads (0.5 per MFMA instead of 0.67), and a
// workgroup takes in 16 KB of activations + 16 KB of weights per 48 MFMAs per wave pair -- 0.67x the bytes per MFMA, the
// activation rows read once per 256 output channels instead of once per 128.
//
// Pipeline.  A stage = ONE k16 step (16 input channels): A_hi, A_lo [256 rows x 32 B] and B_hi, B_lo [256 channel rows x 32 B] =
// 32 KB; ring of four stages (128 KB).  Stage i, every wave:
//     top      : DMA of B(i+3) into the slot stage i-1 vacated (its last fragment reads precede the barrier that ended stage i-1)
//     G1 G2 G3 : the stage's 24 MFMAs (small products first: hi x lo', lo x hi', then hi x hi), the late fragments of stage i read
//                under G1, the early fragments of stage i+1 under G2 / G3 (both were published by the barrier that ended stage i-1)
//     store    : activations of stage i+2 (registers, loaded three stages ago) split and written to LDS; their registers reloaded
//                with stage i+5
//     barrier  : behind `vmcnt` = B(i+2) landed, `lgkmcnt(0)` = this wave's LDS writes done
// 16-byte granules of a 32-byte row are XOR-swizzled by (row >> 3) & 1: the sixteen lanes of a ds_read_b128 group (rows r .. r+15 of
// one k half) then touch all 64 banks once.
#include "common.h"
#include <type_traits>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define PM 256            // rows per tile
#define PN 256            // channels per tile
#define KD 16             // contraction depth of a stage
#define ROWB 32           // bytes per row and plane in LDS
#define NJ 2              // float4 slots per thread and stage (256 rows x 4 slots / 512 threads)
#define NSET 2            // activation register sets (stages i+2, i+3 in flight)
#define NSLOT 4           // ring stages
#define DPW 2             // DMA instructions per wave and stage (256 rows x 2 planes x 32 B / 1 KB / 8 waves)
#define NT 512
#define NI 4
#define SPIN_LIMIT (1 << 24)
#define ACT_SCALE 16.f
#define ACT_MAX 4094.f

struct PwArgsW {
  const float* x;
  const unsigned short* w;   // [2][Kpad][C] fp16 planes
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
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// buffer_load_dwordx4 the compiler does not track: out-of-range offsets return zeros; completion through wait_tied
__device__ __forceinline__ f32x4 load_untracked(u32x4 rsrc, unsigned voff, unsigned soff) {
  f32x4 v;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  return v;
}
template <int N> __device__ __forceinline__ void wait_tied(f32x4& a, f32x4& b) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// behind it: at most N vector-memory operations outstanding and every LDS operation of this wave complete
template <int N> __device__ __forceinline__ void wait_vm_lds() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(NT, 2) void conv_pw_w2_kernel(PwArgsW p) {
  constexpr int PLANE = PM * ROWB;                 // 8 KB: one plane of one operand of one stage
  constexpr int STAGE = 4 * PLANE;                 // A_hi, A_lo, B_hi, B_lo
  constexpr int RING_BYTES = NSLOT * STAGE;        // 131,072 B
  constexpr int SMEM_BYTES = RING_BYTES;
  __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[SMEM_BYTES];

  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int lw = lvc_xcd_remap(blockIdx.x, p.nworkers);
  const int wq = p.ngroup > 1 ? lw / p.ngroup : lw;
  const int wsel = p.ngroup > 1 ? lw - wq * p.ngroup : 0;
  int u = wq * p.units_per_worker;
  const int u_end = min(u + p.units_per_worker, p.total_units);

  const unsigned long long xbase = (unsigned long long)p.x;
  const u32x4 xres = {(unsigned)__builtin_amdgcn_readfirstlane((unsigned)xbase), (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(xbase >> 32) & 0xffffu),
                      (unsigned)p.x_bytes, 0x00020000u};

  while (u < u_end) {
    // The thread's constants of the stage loop are formed HERE, per tile segment, from an opaque copy of the thread id: kept live
    // across the whole kernel they overlap the tail's register peak (accumulators + hand-off / epilogue rows) and are spilled -- then
    // reloaded inside every stage behind a vmcnt(0) that drains the loads in flight.
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int fi = lane & 31, fh = lane >> 5;
    const int q = tid & 3;           // float4 slot of a row's 16 channels
    const int hrow = tid >> 2;       // tile rows hrow + 128 j
    // fragment addresses (bytes inside a plane): row fi of a 32-row block, k half fh, granule swizzled by (row >> 3) & 1
    const int fsw = (fh ^ ((fi >> 3) & 1)) * 16;
    const int a_frag0 = (wm * 64 + fi) * ROWB + fsw;        // + mi * 32 * ROWB
    const int b_frag = (wn * 128 + fi) * ROWB + fsw;        // + ni * 32 * ROWB
    // LDS position of this thread's slot of row hrow; row hrow + 128 sits 128 * ROWB bytes further (same swizzle: 128 % 16 == 0)
    const int a_lds0 = hrow * ROWB + (((q >> 1) ^ ((hrow >> 3) & 1)) << 4) + (q & 1) * 8;
    float big = 0.f;
    const int tile = u / p.nk;
    const int cc0 = u - tile * p.nk;
    const int cc1 = min(p.nk, cc0 + (u_end - u));
    const int tile_n = p.ngroup > 1 ? wsel : tile % p.tiles_n;
    const int tile_m = p.ngroup > 1 ? tile : tile / p.tiles_n;
    const int m0 = tile_m * PM;
    const int n0 = tile_n * PN;

    unsigned a_off[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int m = m0 + hrow + 128 * j;
      const int n = m / (p.Ho * p.Wo);
      const int rem = m - n * (p.Ho * p.Wo);
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      a_off[j] = m < p.M ? (unsigned)(((n * p.H + ho * p.stride) * p.W + wo * p.stride) * p.C + q * 4) * 4u : 0x80000000u;
    }
    // weight DMA: piece idx = wave * DPW + j of 16: plane = idx / 8, 32-row block rb = idx % 8; lane l fills LDS slot l of the block
    // (row l >> 1, position l & 1), which holds the row's k granule (l & 1) ^ ((row >> 3) & 1).  Source = a wave-uniform base
    // (scalar registers) + ONE 32-bit lane offset shared by the wave's pieces.
    const unsigned short* bsrc[DPW];
    int bdst[DPW];
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
      const int idx = wave * DPW + j;
      const int pl = idx >> 3, rb = idx & 7;
      bsrc[j] = p.w + (size_t)pl * p.w_plane_elems + (size_t)(n0 + rb * 32) * p.C;
      bdst[j] = (2 + pl) * PLANE + rb * 32 * ROWB;
    }
    const unsigned b_lane = (unsigned)((lane >> 1) * p.C + (((lane & 1) ^ ((lane >> 4) & 1)) << 3)) * 2u;     // bytes
    auto dma_B = [&](int cc, int slot) {
#pragma unroll
      for (int j = 0; j < DPW; ++j)
        glds16(reinterpret_cast<const char*>(bsrc[j] + cc * KD) + b_lane, smem_raw + slot * STAGE + bdst[j]);
    };
    f32x4 ar[NSET][NJ];
    auto load_A = [&](int set, int cc) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) ar[set][j] = load_untracked(xres, a_off[j], (unsigned)cc * (KD * 4u));
    };
    auto store_A = [&](int set, unsigned char* dst) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        f16x4 h, m;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = ar[set][j][e] * ACT_SCALE;
          const f16 hh = (f16)a;
          h[e] = hh;
          m[e] = (f16)(a - (float)hh);
          big = fmaxf(big, fabsf(ar[set][j][e]));
        }
        *reinterpret_cast<f16x4*>(dst + a_lds0 + j * 128 * ROWB) = h;
        *reinterpret_cast<f16x4*>(dst + a_lds0 + j * 128 * ROWB + PLANE) = m;
      }
    };

    f32x16 acc[2][NI];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < NI; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    f16x8 ahi[2], alo[2], bh[2], bl[2];      // B fragments: two {hi, lo} pairs that alternate over the four 32-channel blocks
    // S = a stage's base ADDED to this lane's fragment offset in one opaque register per operand (`frag_base`): every read below is
    // that register + an immediate (< 64 KB).  Left to itself the compiler forms a fresh address register per read for the slots
    // above 64 KB -- twelve registers per stage that the accumulators then pay for in spills.
    struct FragBase { const unsigned char* a; const unsigned char* b; };
    auto frag_base = [&](int slot) {
      unsigned oa = (unsigned)(slot * STAGE) + (unsigned)a_frag0, ob = (unsigned)(slot * STAGE + 2 * PLANE) + (unsigned)b_frag;
      asm volatile("" : "+v"(oa), "+v"(ob));
      return FragBase{smem_raw + oa, smem_raw + ob};
    };
    auto rdA = [&](const FragBase& S, int pl, int mi) { return *reinterpret_cast<const f16x8*>(S.a + pl * PLANE + mi * 32 * ROWB); };
    auto rdB = [&](const FragBase& S, int pl, int ni) { return *reinterpret_cast<const f16x8*>(S.b + pl * PLANE + ni * 32 * ROWB); };
    // One stage = four groups, one per 32-channel block ni of the wave's 128: six MFMAs (both row blocks x {hi lo', hi hi', lo hi'}) on
    // the pair bh / bl [ni & 1] while the other pair is read for block ni + 1 (block 0 of the NEXT stage under group 3).  The next
    // stage's A fragments are read under group 3 as their registers fall free (ahi after its four MFMAs, alo at the end).  Fragment
    // registers live: 4 + 2 x 2 = 8 (32 VGPRs) instead of the 12 of a stage-wide read -- with 128 accumulator registers the
    // difference decides whether the loop fits the 256-register budget of two waves per SIMD.  Group order pinned by sched_barrier.
    auto group = [&](int ni, int pr) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[mi], bl[pr], acc[mi][ni], 0, 0, 0);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[mi], bh[pr], acc[mi][ni], 0, 0, 0);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[mi], bh[pr], acc[mi][ni], 0, 0, 0);
    };
    auto rd_pair = [&](const FragBase& S, int ni, int pr) {
      bl[pr] = rdB(S, 1, ni);
      bh[pr] = rdB(S, 0, ni);
    };

    // ---- prologue: stages 0, 1 complete in LDS, stage 2's weights in flight or landed, activations of stages 2, 3, 4 in registers.
    // Everything is drained before the loop (one memory round trip per tile segment), so the counted waits of the first stages hold.
    const int last = cc1 - 1;
    auto cl = [&](int c) { return min(c, last); };
    load_A(0, cc0);
    load_A(1, cl(cc0 + 1));
    dma_B(cc0, 0);
    dma_B(cl(cc0 + 1), 1);
    dma_B(cl(cc0 + 2), 2);
    wait_tied<3 * DPW + NJ>(ar[0][0], ar[0][1]);
    store_A(0, smem_raw);
    wait_tied<3 * DPW>(ar[1][0], ar[1][1]);
    store_A(1, smem_raw + STAGE);
    load_A(0, cl(cc0 + 2));
    load_A(1, cl(cc0 + 3));
    wait_vm<0>();
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      ahi[mi] = rdA(frag_base(0), 0, mi);
      alo[mi] = rdA(frag_base(0), 1, mi);
    }
    rd_pair(frag_base(0), 0, 0);

    // Vector-memory operations of a wave per stage, program order:  D = DMA B(i+3) [DPW]  ...  L = loads A(i+4) [NJ].
    //   store of A(i+2): loaded as L of stage i-2; behind it D, L of stage i-1 and D of stage i   ->  2 DPW + NJ may stay outstanding
    //   barrier: B(i+2) = D of stage i-1; behind it L(i-1), D(i), L(i)                            ->  DPW + 2 NJ
    auto stage = [&](int i, int k) {      // k = i % NSLOT = compile-time inside the unrolled body; register set i % 2 = k % 2
      const int cc = cc0 + i;
      const FragBase S = frag_base(k), Sn = frag_base((k + 1) % NSLOT);
      unsigned char* S2 = smem_raw + ((k + 2) % NSLOT) * STAGE;
      dma_B(cl(cc + 3), (k + 3) % NSLOT);
      rd_pair(S, 1, 1);
      group(0, 0);
      __builtin_amdgcn_sched_barrier(0);
      rd_pair(S, 2, 0);
      group(1, 1);
      wait_tied<2 * DPW + NJ>(ar[k % 2][0], ar[k % 2][1]);
      store_A(k % 2, S2);
      __builtin_amdgcn_sched_barrier(0);
      rd_pair(S, 3, 1);
      group(2, 0);
      load_A(k % 2, cl(cc + 4));
      __builtin_amdgcn_sched_barrier(0);
      rd_pair(Sn, 0, 0);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) acc[mi][3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[mi], bl[1], acc[mi][3], 0, 0, 0);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) acc[mi][3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[mi], bh[1], acc[mi][3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) ahi[mi] = rdA(Sn, 0, mi);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) acc[mi][3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[mi], bh[1], acc[mi][3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) alo[mi] = rdA(Sn, 1, mi);
      __builtin_amdgcn_sched_barrier(0);
      wait_vm_lds<DPW + 2 * NJ>();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    const int nst = cc1 - cc0;
#pragma unroll 1
    for (int i = 0; i < nst; i += NSLOT) {
      stage(i, 0);
      if (i + 1 < nst) stage(i + 1, 1);
      if (i + 2 < nst) stage(i + 2, 2);
      if (i + 3 < nst) stage(i + 3, 3);
    }
    wait_vm<0>();
    __syncthreads();
    u += nst;
    if (!(big <= ACT_MAX)) atomicOr(p.flags + p.err_index, big < INFINITY ? 2 : 4);      // finite / non-finite: see conv3x3_halo_s1.hip
    // A new definition of every accumulator between the stage loop and the tile's tail: the register allocator may then place them
    // differently in the two regions (a few moves here) instead of spilling one whole accumulator -- and reloading / re-spilling it in
    // every stage -- because the tail's pressure peaks above the budget.
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) asm volatile("" : "+v"(acc[mi][ni]));

    // ---- split tiles (the protocol of the other stream-K kernels): 128 accumulator registers per thread = 256 KB per worker.
    // Addressed through a buffer descriptor over the partials area with the block offsets in SCALAR registers: with flat addresses
    // every one of the 32 rows needs its own 64-bit address pair (the offsets exceed the immediate field), which next to 128 live
    // accumulators costs spills.
    int tid_t = tid0;
    asm volatile("" : "+v"(tid_t));
    const __amdgpu_buffer_rsrc_t pres = __builtin_amdgcn_make_buffer_rsrc((void*)p.partials, 0, 0x7ffffff0, 0x00020000);
    constexpr unsigned WSLOT = NT * 32 * NI * 4;      // bytes per worker
    if (cc0 != 0) {
      const unsigned base = (unsigned)lw * WSLOT;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {
            const f32x4 v = {acc[mi][ni][e4 * 4 + 0], acc[mi][ni][e4 * 4 + 1], acc[mi][ni][e4 * 4 + 2], acc[mi][ni][e4 * 4 + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), pres, (unsigned)tid_t * 16u,
                                                   base + (unsigned)((mi * NI + ni) * 4 + e4) * (NT * 16u), 0);
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid_t == 0) {
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
        if (tid_t == 0) {
          int spins = 0;
          while (__hip_atomic_load(p.flags + pw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > SPIN_LIMIT) { atomicOr(p.flags + p.err_index, 1); break; }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        const unsigned base = (unsigned)pw * WSLOT;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
              const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pres, (unsigned)tid_t * 16u,
                                                                                               base + (unsigned)((mi * NI + ni) * 4 + e4) * (NT * 16u), 0));
              acc[mi][ni][e4 * 4 + 0] += v[0]; acc[mi][ni][e4 * 4 + 1] += v[1];
              acc[mi][ni][e4 * 4 + 2] += v[2]; acc[mi][ni][e4 * 4 + 3] += v[3];
            }
            __builtin_amdgcn_sched_barrier(0);      // one block's four rows in flight: hoisting all 32 loads takes 128 registers
          }
        __syncthreads();
        if (tid_t == 0) __hip_atomic_store(p.flags + pw, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }

    // ---- epilogue straight from the accumulators (no LDS, no barrier: a wave is done with its 64 x 128 block on its own).  Lane
    // (fi, fh) of an accumulator block holds channel fi of rows (e & 3) + 8 (e >> 2) + 4 fh: one dword store covers 32 consecutive
    // channels of two rows = two full 128-byte lines; the residual is read in the same pattern.  Rows past M fall outside the buffer
    // descriptors (stores dropped, loads zero).  Rows are visited in increasing order: the steps are +1 or +5, which keeps the
    // (image, y, x) decomposition the upsample-add needs to a compare and a subtract per row.
    {
      int tl = tid0;
      asm volatile("" : "+v"(tl));
      const int efi = tl & 31, efh = (tl >> 5) & 1;
      const int row0 = m0 + wm * 64 + 4 * efh;
      const int colb = n0 + wn * 128 + efi;              // + 32 ni
      const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (unsigned)p.M * (unsigned)p.ldy * 4u, 0x00020000);
      const unsigned rrec = p.res_mode == 0 ? 0u : p.res_mode == 1 ? (unsigned)p.M * (unsigned)p.ldr * 4u
                                                                   : (unsigned)(p.M >> 2) * (unsigned)p.ldr * 4u;
      const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res_mode ? p.res : p.y), 0, rrec, 0x00020000);
      float sc[NI], sh[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        sc[ni] = p.scale[colb + 32 * ni];
        sh[ni] = p.shift ? p.shift[colb + 32 * ni] : 0.f;
      }
      unsigned yo = ((unsigned)row0 * (unsigned)p.ldy + (unsigned)colb) * 4u;
      unsigned ro = 0;
      int r_n = 0, r_ho = 0, r_wo = 0;
      const bool row_in = row0 < p.M;
      if (p.res_mode == 1) ro = ((unsigned)row0 * (unsigned)p.ldr + (unsigned)colb) * 4u;
      if (p.res_mode == 2) {
        const int rm = row_in ? row0 : 0;
        r_n = rm / (p.Ho * p.Wo);
        const int rem = rm - r_n * (p.Ho * p.Wo);
        r_ho = rem / p.Wo;
        r_wo = rem - r_ho * p.Wo;
      }
      // residual rows are requested RD rows ahead of their use (a lane's 32 rows would otherwise be 32 memory round trips in series)
      constexpr int RD = 4;
      auto rows = [&](auto res_tag, auto act_tag) {
        constexpr int RESM = decltype(res_tag)::value;
        constexpr int ACT = decltype(act_tag)::value;
        float rv[RD][NI];
        auto res_row = [&](int slot) {      // request the residual of the lane's next row into rv[slot]; steps the row state
          if (RESM == 2) ro = ((unsigned)((r_n * (p.Ho >> 1) + (r_ho >> 1)) * (p.Wo >> 1) + (r_wo >> 1)) * (unsigned)p.ldr + (unsigned)colb) * 4u;
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) rv[slot][ni] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rres, ro + 128u * ni, 0, 0));
        };
        auto res_step = [&](int e) {
          const int step = (e & 3) == 3 ? 5 : 1;
          if (RESM == 1) ro += (unsigned)(step * p.ldr) * 4u;
          if (RESM == 2) {
            r_wo += step;
            if (r_wo >= p.Wo) {
              r_wo -= p.Wo;
              if (++r_ho == p.Ho) { r_ho = 0; ++r_n; }
            }
          }
        };
        if (RESM != 0) {
#pragma unroll
          for (int r = 0; r < RD; ++r) { res_row(r); res_step(r); }
        }
#pragma unroll
        for (int r = 0; r < 32; ++r) {
          const int mi = r >> 4, e = r & 15;
          float o[NI];
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            o[ni] = acc[mi][ni][e] * sc[ni] + sh[ni];
            if (RESM != 0) o[ni] += rv[r % RD][ni];
            if (ACT == 1) o[ni] = o[ni] > 0.f ? o[ni] : 0.f;
            else if (ACT == 2) o[ni] = o[ni] * 0.5f * (1.f + erff(o[ni] * 0.70710678118654752440f));
          }
          if (RESM != 0 && r + RD < 32) { res_row(r % RD); res_step(r + RD); }
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o[ni]), yres, yo + 128u * ni, 0, 0);
          yo += (unsigned)(((e & 3) == 3 ? 5 : 1) * p.ldy) * 4u;
        }
      };
      using R0 = std::integral_constant<int, 0>; using R1 = std::integral_constant<int, 1>; using R2 = std::integral_constant<int, 2>;
      using A0 = std::integral_constant<int, 0>; using A1 = std::integral_constant<int, 1>; using A2 = std::integral_constant<int, 2>;
      if (p.res_mode == 0) {
        if (p.relu == 1) rows(R0{}, A1{}); else if (p.relu == 2) rows(R0{}, A2{}); else rows(R0{}, A0{});
      } else if (p.res_mode == 1) {
        if (p.relu == 1) rows(R1{}, A1{}); else if (p.relu == 2) rows(R1{}, A2{}); else rows(R1{}, A0{});
      } else {
        if (p.relu == 1) rows(R2{}, A1{}); else if (p.relu == 2) rows(R2{}, A2{}); else rows(R2{}, A0{});
      }
    }
  }
}

#define LVC_MAX_WORKERS 1024
static int g_cus_pw_w = 0;

// Pointwise (R = S = 1, pad 0) layer y = act(conv(x, w) * scale + shift (+ residual)) on the 256 x 256 tile: x [N,H,W,C] fp32 NHWC with
// C % 16 == 0, K % 4 == 0 (meant for K >= 256; channel tiles past K compute on zero weight rows), w_split / scale from
// lvc_split_weights_rowscaled as for lvc_conv1x1_nhwc_f16s1 (single-accumulator form, |a| <= 4094).  relu: 0 none, 1 ReLU, 2 exact GELU.
// The packed weight planes must hold rows up to the next multiple of 256 when K % 256 > 128 (lvc_amd.kernels pads to 256 for this form).
extern "C" int lvc_conv1x1_nhwc_f16s1_w2(const float* x, const unsigned short* w_split, const float* scale, const float* shift,
                                          const float* residual, float* y, int N, int H, int W, int C, int K, int Kpad, int stride, int relu,
                                          int res_mode, int ldy, int ldr, void* workspace, void* stream) {
  LVC_CHECK_ARG(x && w_split && y && workspace && scale, "null pointer");
  LVC_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && K > 0 && stride >= 1, "non-positive dimension");
  LVC_CHECK_ARG(C % KD == 0, "needs C % 16 == 0");
  LVC_CHECK_ARG(K % PN == 0, "needs K % 256 == 0");
  LVC_CHECK_ARG(relu >= 0 && relu <= 2, "relu: 0 none, 1 ReLU, 2 GELU");
  LVC_CHECK_ARG(res_mode >= 0 && res_mode <= 2 && (res_mode == 0 || residual), "bad residual");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  if (res_mode == 2) LVC_CHECK_ARG(Ho % 2 == 0 && Wo % 2 == 0, "upsample-add needs even output size");
  PwArgsW a;
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
  a.tiles_n = lvc_cdiv(K, PN);
  LVC_CHECK_ARG(Kpad >= a.tiles_n * PN, "weight planes must be padded to a multiple of 256 rows");
  a.nk = C / KD;
  const int tiles_m = lvc_cdiv(a.M, PM);
  long long units = (long long)tiles_m * a.tiles_n * a.nk;
  LVC_CHECK_ARG(units < (1ll << 31), "iteration space too large");
  a.total_units = (int)units;
  a.w_plane_elems = (long long)Kpad * C;
  if (g_cus_pw_w == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    g_cus_pw_w = cus;
  }
  int cap = g_cus_pw_w;
  if (cap > LVC_MAX_WORKERS / 2) cap = LVC_MAX_WORKERS / 2;      // a worker's partial tile takes two 128 KB slots of the workspace
  a.ngroup = 1;
  if ((a.tiles_n == 2 || a.tiles_n == 4 || a.tiles_n == 8) && cap % a.tiles_n == 0 && units / a.tiles_n >= (long long)(cap / a.tiles_n) * 8) {
    a.ngroup = a.tiles_n;       // the workers of one row tile's channel tiles are neighbours on one XCD: the rows come from L2
    units /= a.tiles_n;
    cap /= a.tiles_n;
    a.total_units = (int)units;
  }
  const int min_units = 8;     // a worker's pipeline restarts per tile segment: keep segments >= 8 stages
  int workers = (int)((units + min_units - 1) / min_units);
  if (workers > cap) workers = cap;
  a.units_per_worker = (int)((units + workers - 1) / workers);
  a.nworkers = (int)((units + a.units_per_worker - 1) / a.units_per_worker) * a.ngroup;
  a.partials = (float*)workspace;
  a.flags = (int*)((char*)workspace + (size_t)LVC_MAX_WORKERS * 256 * 128 * 4);
  a.err_index = LVC_MAX_WORKERS + lvc_range_slot();   // the layer's own range word (common.cpp)
  hipLaunchKernelGGL(conv_pw_w2_kernel, dim3(a.nworkers), dim3(NT), 0, (hipStream_t)stream, a);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}
