// conv_pw_chain.hip -- two chained pointwise layers in ONE launch, the second fed from the first one's accumulators (round 4):
//
//     y1 = act1(conv1x1(x,  Wa) * sa + ta (+ residual))        stored     (a bottleneck's conv3 + FrozenBN + shortcut add + ReLU)
//     y2 = act2(conv1x1(y1, Wb) * sb + tb)                     stored     (the NEXT bottleneck's conv1 + FrozenBN + ReLU)
//
// Reference: detectron2/modeling/backbone/resnet.py:195-211 (the tail of block i) followed by :195-197 of block i+1.  As two
// launches the 4 Cm-channel tensor y1 is written, read back as the next residual AND read a third time by conv1; here conv1's
// input never comes back from HBM: res2 moves 1.37 GB per block boundary instead of 1.92 GB, res3 0.69 instead of 0.96 GB,
// and these layers run at the rate of a copy (scripts/micro/lane_pixel_access.hip: 5.1 TB/s coalesced, 4.6 TB/s in the access
// pattern below).
//
// Everything is computed TRANSPOSED: out^T [channels x pixels] = W [channels x k] . act^T [k x pixels], i.e. the MFMA's A operand
// is the weight block (from LDS) and its B operand the activations, so that an accumulator lane owns ONE pixel (lane % 32) and
// sixteen of a 32-channel block's channels -- four runs of four consecutive channels: c = 8 i + 4 (lane / 32) + {0..3}, i = 0..3.
// Consequences:
//   * a wave owns 32 pixels for the whole chain; its activations never pass through LDS.  x is read straight into the B-operand
//     layout (two dwordx4 per k16 step: 32 contiguous bytes per pixel and instruction), the residual is read and y1 / y2 are written
//     straight from / into the accumulator layout (the same pattern);
//   * y1's accumulator registers, after the epilogue and the fp16 split, ARE the B operand of the second layer (registers 8 s .. 8 s + 7
//     of a block = k16 step s) when Wb's contraction index is stored in the matching order: within every 16 channels the two middle
//     runs of four are swapped (position p holds channel perm16[p], perm16 = 0-3, 8-11, 4-7, 12-15).  x is read in the same order, so
//     Wa uses the same permutation (`lvc_amd.kernels.pack_chain`);
//   * LDS only holds weights: per 32-channel block j of y1 a ring stage carries Wa's rows 32 j .. 32 j + 31 (all K1 columns) and
//     Wb's columns 32 j .. 32 j + 31 (all N2 rows), both fp16 planes, by LDS-DMA with the granule XOR swizzle of conv_pw_dma.hip.
// Numerics: the single-accumulator two-way fp16 split of conv3x3_halo_s1.hip (row-scaled weight planes of
// lvc_split_weights_rowscaled, activations x 2^4, |a| <= 4094 for x AND for y1 or bit 1 of the workspace error word is raised).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define ACT_SCALE 16.f
#define ACT_MAX 4094.f
#define LVC_MAX_WORKERS 1024

struct ChainArgs {
  const float* x;            // [M][ldx]   first layer's input (K1 channels used)
  const unsigned short* wa;  // [2][wa_rows][K1] fp16 planes, k permuted
  const float* sa;           // [N1] (FrozenBN scale or 1) * row factor     -- never null
  const float* ta;           // [N1] shift or null
  const float* res;          // [M][ldr] or null
  float* y1;                 // [M][ldy1]
  const unsigned short* wb;  // [2][wb_rows][N1] fp16 planes, k permuted
  const float* sb;           // [N2]
  const float* tb;           // [N2] or null
  float* y2;                 // [M][ldy2]
  int* flags;
  int M, ldx, ldr, ldy1, ldy2, relu1, relu2, ngroups, err_index;
  long long wa_plane, wb_plane;   // elements per plane
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// buffer_load_dwordx4 the compiler does not track (it would wait for everything in flight, LDS-DMA and stores included, at the
// first use): rows past the end return zeros; completion through the counted waits below, which are TIED to the registers
template <int IMM> __device__ __forceinline__ f32x4 load_untracked(u32x4 rsrc, unsigned voff, unsigned soff) {
  f32x4 v;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff), "n"(IMM) : "memory");
  return v;
}
// Range tracking pinned in program order (volatile asm): written as plain fmaxf the compiler sank these maxima far below the
// split, kept the raw input rows alive for them and SPILLED those registers right after the untracked loads were issued --
// i.e. before their data had arrived.  (The build fails on any spill in this file: csrc/Makefile.)
__device__ __forceinline__ void track_abs(float& big, float a, float b) {
  asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(big) : "v"(a), "v"(b));
}
__device__ __forceinline__ void tie(f32x4& a, f32x4& b, f32x4& c, f32x4& d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
__device__ __forceinline__ u32x4 make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long b = (unsigned long long)base;
  return u32x4{(unsigned)__builtin_amdgcn_readfirstlane((unsigned)b), (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu),
               (unsigned)__builtin_amdgcn_readfirstlane(bytes), 0x00020000u};
}
// The scalar offset of a store is ALWAYS the literal 0.  The compiler inserts the wait state a > 64-bit VMEM store needs before its
// data registers are overwritten only when soffset is not an SGPR (GCNHazardRecognizer::createsVALUHazard); gfx950 needs it with an
// SGPR soffset as well: with `buffer_store_dwordx4 v[172:175], .., s93 offen` followed directly by a write of v172, the lanes
// 12-15 of every 16 stored 0 instead of the value (scripts/dbg_chain.py found exactly the registers that were rewritten next).
__device__ __forceinline__ void store_b128(f32x4 v, __amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, voff, 0, 0);
}

// K1: contraction of the first layer (64 / 128), N1: its outputs = the second layer's contraction, N2: the second layer's outputs.
// NW waves per workgroup (a wave = 32 pixels), NSLOT ring stages, RES: a residual operand exists.
//
// Vector-memory operations of a wave in program order, per stage = 32-channel block j of y1:
//     [wait: the weights of this stage] barrier | D: DMA of stage + NSLOT - 1 (DPW loads) | (all but the last block) R: residual rows
//     of block j + 1 (4 loads) | first-layer MFMAs | (last block) X: the NEXT pixel group's input rows (2 KS1 loads), R: its block 0
//     | [wait: residual rows of block j] epilogue, S: y1 stores (4) | second-layer MFMAs | (last block) S2: y2 stores (4 NB2)
// and at the top of a pixel group [wait: X].  LOADS retire in order among themselves, but stores are acknowledged out of order with
// them (measured: with the stores counted as "newer operations that may stay outstanding" a stage read weights that had not
// landed), so a wait allows exactly the number of LOADS issued behind its target: nothing is drained except where no load follows.
template <int K1, int N1, int N2, int NW, int NSLOT, bool RES>
__global__ __launch_bounds__(NW * 64, 2) void conv_pw_chain_kernel(ChainArgs p) {
  constexpr int KS1 = K1 / 16;             // k16 steps of the first layer
  constexpr int KC1 = K1 / 32;
  constexpr int NB1 = N1 / 32;             // 32-channel blocks of y1 = ring stages per pixel group
  constexpr int NB2 = N2 / 32;
  constexpr int WA_PLANE = KC1 * 32 * 64;  // bytes: [kc][32 rows][64 B]
  constexpr int WA_BYTES = 2 * WA_PLANE;
  constexpr int WB_PLANE = N2 * 64;        // bytes: [N2 rows][64 B]
  constexpr int WB_BYTES = 2 * WB_PLANE;
  constexpr int STAGE = WA_BYTES + WB_BYTES;
  constexpr int NDMA_A = KC1 * 2 * 2;      // 1 KB pieces (16 rows x 64 B) per stage
  constexpr int NDMA_B = (N2 / 16) * 2;
  constexpr int NDMA = NDMA_A + NDMA_B;
  static_assert(NDMA % NW == 0, "DMA pieces must divide over the waves");
  static_assert(NSLOT == 2 || NSLOT == 3, "ring of two or three stages");
  constexpr int DPW = NDMA / NW;
  constexpr int NR = RES ? 4 : 0;
  // operations issued behind a stage's DMA when the NEXT stage waits for it, and behind a block's residual rows at its epilogue
  constexpr int N_DMA = NSLOT == 2 ? NR : 2 * NR + DPW;
  constexpr int N_RES = DPW + NR;
  constexpr int N_X = NR;
  constexpr int TAB = (2 * N1 + 2 * N2) * 4;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NSLOT * STAGE + TAB];
  float* tab_sa = reinterpret_cast<float*>(smem + NSLOT * STAGE);
  float* tab_ta = tab_sa + N1;
  float* tab_sb = tab_ta + N1;
  float* tab_tb = tab_sb + N2;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 31, fh = lane >> 5;
  const int fx3 = (fi >> 2) & 3;
  const int fr_g[2] = {((0 + fh) ^ fx3) * 16, ((2 + fh) ^ fx3) * 16};
  const int fr_row = fi * 64;

  for (int i = tid; i < N1; i += NW * 64) {
    tab_sa[i] = p.sa[i];
    tab_ta[i] = p.ta ? p.ta[i] : 0.f;
  }
  for (int i = tid; i < N2; i += NW * 64) {
    tab_sb[i] = p.sb[i];
    tab_tb[i] = p.tb ? p.tb[i] : 0.f;
  }
  __syncthreads();

  // this wave's DMA pieces: source of stage 0, source step per stage (halves), destination inside a stage
  const unsigned short* dsrc[DPW];
  int dstep[DPW], ddst[DPW];
#pragma unroll
  for (int i = 0; i < DPW; ++i) {
    const int id = wave + NW * i;
    const int r16 = lane >> 2;
    if (id < NDMA_A) {
      const int pl = id / (KC1 * 2), rem = id - pl * (KC1 * 2), kc = rem >> 1, rb = rem & 1;
      const int row = rb * 16 + r16;
      const int G = (lane & 3) ^ ((row >> 2) & 3);
      dsrc[i] = p.wa + (size_t)pl * p.wa_plane + (size_t)row * K1 + kc * 32 + G * 8;
      dstep[i] = 32 * K1;
      ddst[i] = pl * WA_PLANE + kc * 2048 + rb * 1024;
    } else {
      const int idb = id - NDMA_A;
      const int pl = idb / (N2 / 16), rb = idb - pl * (N2 / 16);
      const int row = rb * 16 + r16;
      const int G = (lane & 3) ^ ((row >> 2) & 3);
      dsrc[i] = p.wb + (size_t)pl * p.wb_plane + (size_t)row * N1 + G * 8;
      dstep[i] = 32;
      ddst[i] = WA_BYTES + pl * WB_PLANE + rb * 1024;
    }
  }
  auto dma_stage = [&](int j, int slot) {
#pragma unroll
    for (int i = 0; i < DPW; ++i) glds16(dsrc[i] + (size_t)j * dstep[i], smem + slot * STAGE + ddst[i]);
  };

  const u32x4 xres = make_rsrc(p.x, (unsigned)p.M * (unsigned)p.ldx * 4u);
  const u32x4 rres = make_rsrc(RES ? p.res : p.x, RES ? (unsigned)p.M * (unsigned)p.ldr * 4u : 0u);
  const __amdgpu_buffer_rsrc_t y1res = __builtin_amdgcn_make_buffer_rsrc((void*)p.y1, 0, p.M * p.ldy1 * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t y2res = __builtin_amdgcn_make_buffer_rsrc((void*)p.y2, 0, p.M * p.ldy2 * 4, 0x00020000);

  const int ngl = p.ngroups > (int)blockIdx.x ? (p.ngroups - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;   // pixel groups of this workgroup
  if (ngl == 0) return;
  const int total_stages = ngl * NB1;
  float big = 0.f;
  // ReLU as ONE v_max against a uniform lower bound (0 or -inf) instead of compare / mask / select per element
  const float lo1 = p.relu1 ? 0.f : -INFINITY, lo2 = p.relu2 ? 0.f : -INFINITY;
#pragma unroll
  for (int s = 0; s < NSLOT - 1; ++s) dma_stage(s % NB1, s);

  // pixel of this lane in pixel group g; rows past M: loads return zeros, stores are dropped (buffer bounds)
  auto pixel = [&](int gl) { return (unsigned)(((int)blockIdx.x + gl * (int)gridDim.x) * NW + wave) * 32u + (unsigned)fi; };
  f32x4 xr[KS1][2];   // input rows of the current (then the next) pixel group, raw
  f32x4 rb[2][4];     // residual rows of block j in rb[j & 1]
  auto load_x = [&](unsigned px) {
    const unsigned vo = px * (unsigned)p.ldx * 4u + fh * 16u;
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
      xr[s][0] = load_untracked<0>(xres, vo, 64u * s);
      xr[s][1] = load_untracked<32>(xres, vo, 64u * s);
    }
  };
  auto load_r = [&](f32x4* dst, unsigned px, int j) {
    const unsigned vo = px * (unsigned)p.ldr * 4u + fh * 16u;
    dst[0] = load_untracked<0>(rres, vo, 128u * j);
    dst[1] = load_untracked<32>(rres, vo, 128u * j);
    dst[2] = load_untracked<64>(rres, vo, 128u * j);
    dst[3] = load_untracked<96>(rres, vo, 128u * j);
  };
  unsigned px = pixel(0);
  load_x(px);
  if (RES) load_r(rb[0], px, 0);
  wait_vm<0>();
#pragma unroll
  for (int s = 0; s < KS1; ++s) asm volatile("" : "+v"(xr[s][0]), "+v"(xr[s][1]));
  if (RES) tie(rb[0][0], rb[0][1], rb[0][2], rb[0][3]);

  int t = 0;   // stage counter of this workgroup
#pragma unroll 1
  for (int gl = 0; gl < ngl; ++gl) {
    const unsigned pxn = pixel(gl + 1 < ngl ? gl + 1 : gl);
    // ---- the first layer's B operand: x in the permuted channel order, split once
    f16x8 zh[KS1], zl[KS1];
    if (gl > 0) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_X) : "memory");
#pragma unroll
      for (int s = 0; s < KS1; ++s) asm volatile("" : "+v"(xr[s][0]), "+v"(xr[s][1]));
    }
#pragma unroll
    for (int s = 0; s < KS1; ++s)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float a = xr[s][h][c] * ACT_SCALE;
          const f16 hh = (f16)a;
          zh[s][4 * h + c] = hh;
          zl[s][4 * h + c] = (f16)(a - (float)hh);
          big = fmaxf(big, fabsf(xr[s][h][c]));
        }
    f32x16 acc2[NB2];
#pragma unroll
    for (int cb = 0; cb < NB2; ++cb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc2[cb][e] = 0.f;
    const unsigned y1off = px * (unsigned)p.ldy1 * 4u + fh * 16u;

    auto block = [&](int j, auto par_tag, auto last_tag) {
      constexpr int PAR = decltype(par_tag)::value;         // j & 1: the residual register set of this block
      constexpr bool LAST = decltype(last_tag)::value;
      f32x4* rr = rb[PAR];
      f32x4* rn = rb[PAR ^ 1];
      const int slot = t % NSLOT;
      // stage t has landed (this wave's pieces; behind the barrier everybody's), and everybody is done with stage t - 1
      wait_vm<N_DMA>();
      __builtin_amdgcn_s_barrier();
      {
        const int tn = t + NSLOT - 1;
        const int tc = tn < total_stages ? tn : total_stages - 1;
        dma_stage(tc % NB1, tn % NSLOT);
      }
      const unsigned char* S = smem + slot * STAGE;
      // residual rows of the NEXT block
      if (RES && !LAST) load_r(rn, px, j + 1);
      // ---- first layer, block j: acc = Wa[32 j .. +31][:] . x^T
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int s = 0; s < KS1; ++s) {
        const int off = (s >> 1) * 2048 + fr_row + fr_g[s & 1];
        const f16x8 wh = *reinterpret_cast<const f16x8*>(S + off);
        const f16x8 wl = *reinterpret_cast<const f16x8*>(S + WA_PLANE + off);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, zl[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, zh[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, zh[s], acc, 0, 0, 0);
      }
      if (LAST) {
        // the split planes of x are dead: the next pixel group's rows take their place while this block finishes
        __builtin_amdgcn_sched_barrier(0);
        load_x(pxn);
        if (RES) load_r(rn, pxn, 0);
      }
      if (RES) {
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(rr[0]), "+v"(rr[1]), "+v"(rr[2]), "+v"(rr[3]) : "n"(N_RES + (LAST ? 2 * KS1 : 0)) : "memory");
      }
      // ---- epilogue of block j in the accumulator layout; the result is split into the second layer's B operand
      f16x8 yh[2], yl[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(tab_sa + 32 * j + 8 * i + 4 * fh);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(tab_ta + 32 * j + 8 * i + 4 * fh);
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float o = acc[4 * i + c] * sc[c] + sh[c];
          if (RES) o += rr[i][c];
          o = fmaxf(o, lo1);
          v[c] = o;
          const float a = o * ACT_SCALE;
          const f16 hh = (f16)a;
          yh[i >> 1][4 * (i & 1) + c] = hh;
          yl[i >> 1][4 * (i & 1) + c] = (f16)(a - (float)hh);
        }
        track_abs(big, v[0], v[1]);
        track_abs(big, v[2], v[3]);
        store_b128(v, y1res, y1off + 128u * j + 32u * i);
      }
      // ---- second layer: acc2[cb] += Wb[32 cb .. +31][32 j .. +31] . y1_j^T
#pragma unroll
      for (int cb = 0; cb < NB2; ++cb)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int off = WA_BYTES + cb * 2048 + fr_row + fr_g[s];
          const f16x8 wh = *reinterpret_cast<const f16x8*>(S + off);
          const f16x8 wl = *reinterpret_cast<const f16x8*>(S + WB_PLANE + off);
          acc2[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, yl[s], acc2[cb], 0, 0, 0);
          acc2[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, yh[s], acc2[cb], 0, 0, 0);
          acc2[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, yh[s], acc2[cb], 0, 0, 0);
        }
      ++t;
    };
    using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
    block(0, P0{}, std::false_type{});
#pragma unroll 1
    for (int j = 1; j < NB1 - 1; j += 2) {
      block(j, P1{}, std::false_type{});
      block(j + 1, P0{}, std::false_type{});
    }
    block(NB1 - 1, P1{}, std::true_type{});
    // ---- the second layer's epilogue
    const unsigned y2off = px * (unsigned)p.ldy2 * 4u + fh * 16u;
#pragma unroll
    for (int cb = 0; cb < NB2; ++cb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(tab_sb + 32 * cb + 8 * i + 4 * fh);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(tab_tb + 32 * cb + 8 * i + 4 * fh);
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float o = acc2[cb][4 * i + c] * sc[c] + sh[c];
            o = fmaxf(o, lo2);
          v[c] = o;
        }
        store_b128(v, y2res, y2off + 128u * cb + 32u * i);
      }
    px = pxn;
  }
  wait_vm<0>();
  if (!(big <= ACT_MAX)) atomicOr(p.flags + p.err_index, big < INFINITY ? 2 : 4);      // finite / non-finite: see conv3x3_halo_s1.hip
}

static int g_cus_chain = 0;

template <int K1, int N1, int N2, int NW, int NSLOT>
static void chain_launch(const ChainArgs& a, int wgs_per_cu, hipStream_t st) {
  int grid = g_cus_chain * wgs_per_cu;
  if (grid > a.ngroups) grid = a.ngroups;
  if (a.res) hipLaunchKernelGGL((conv_pw_chain_kernel<K1, N1, N2, NW, NSLOT, true>), dim3(grid), dim3(NW * 64), 0, st, a);
  else hipLaunchKernelGGL((conv_pw_chain_kernel<K1, N1, N2, NW, NSLOT, false>), dim3(grid), dim3(NW * 64), 0, st, a);
}

// y1 = act1(x Wa^T * sa + ta (+ residual)), y2 = act2(y1 Wb^T * sb + tb), both stored; x [M][ldx] (K1 channels), y1 [M][ldy1]
// (N1), y2 [M][ldy2] (N2).  wa / wb: [2][rows][K1] / [2][rows][N1] fp16 planes of lvc_split_weights_rowscaled over weights whose
// contraction index is permuted within every 16 (0-3, 8-11, 4-7, 12-15); sa / sb = (per-channel scale or 1) x that call's row
// factors (never NULL), ta / tb shifts or NULL.  relu1 / relu2: 0 none, 1 ReLU.  (K1, N1, N2) must be one of the bottleneck shapes
// (64,256,64), (128,256,64), (128,512,128).  |x| or |y1| > 4094 (or NaN) raises bit 1 of the workspace error word.
extern "C" int lvc_conv1x1_chain_nhwc_f16s1(const float* x, int ldx, const unsigned short* wa, int wa_rows, const float* sa, const float* ta,
                                             const float* residual, int ldr, float* y1, int ldy1, int relu1,
                                             const unsigned short* wb, int wb_rows, const float* sb, const float* tb, float* y2, int ldy2,
                                             int relu2, int M, int K1, int N1, int N2, void* workspace, void* stream) {
  LVC_CHECK_ARG(x && wa && sa && y1 && wb && sb && y2 && workspace, "null pointer");
  LVC_CHECK_ARG((long long)(M + 128) * ldx < (1ll << 30) && (long long)(M + 128) * ldy1 < (1ll << 30) && (long long)(M + 128) * ldy2 < (1ll << 30) &&
                    (long long)(M + 128) * ldr < (1ll << 30), "tensors must stay below 4 GiB (32-bit buffer offsets)");
  LVC_CHECK_ARG(M > 0 && ldx >= K1 && ldy1 >= N1 && ldy2 >= N2 && (!residual || ldr >= N1), "bad dimension");
  LVC_CHECK_ARG((ldx & 3) == 0 && (ldy1 & 3) == 0 && (ldy2 & 3) == 0 && (ldr & 3) == 0, "row strides must be multiples of 4");
  LVC_CHECK_ARG(wa_rows >= N1 && wb_rows >= N2, "weight planes have too few rows");
  LVC_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)wa & 15) == 0 && ((uintptr_t)wb & 15) == 0 && ((uintptr_t)y1 & 15) == 0 &&
                    ((uintptr_t)y2 & 15) == 0 && ((uintptr_t)residual & 15) == 0 && ((uintptr_t)sa & 15) == 0 && ((uintptr_t)sb & 15) == 0,
                "pointers must be 16-byte aligned");
  ChainArgs a;
  a.x = x; a.wa = wa; a.sa = sa; a.ta = ta; a.res = residual; a.y1 = y1; a.wb = wb; a.sb = sb; a.tb = tb; a.y2 = y2;
  a.M = M; a.ldx = ldx; a.ldr = ldr; a.ldy1 = ldy1; a.ldy2 = ldy2; a.relu1 = relu1; a.relu2 = relu2;
  a.wa_plane = (long long)wa_rows * K1; a.wb_plane = (long long)wb_rows * N1;
  a.flags = (int*)((char*)workspace + (size_t)LVC_MAX_WORKERS * 256 * 128 * 4);
  a.err_index = LVC_MAX_WORKERS + lvc_range_slot();   // the layer's own range word (common.cpp)
  if (g_cus_chain == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    g_cus_chain = cus;
  }
  hipStream_t st = (hipStream_t)stream;
  constexpr int NW = 4;
  a.ngroups = lvc_cdiv(M, 32 * NW);
  if (K1 == 64 && N1 == 256 && N2 == 64) chain_launch<64, 256, 64, NW, 3>(a, 2, st);
  else if (K1 == 128 && N1 == 256 && N2 == 64) chain_launch<128, 256, 64, NW, 3>(a, 2, st);
  else if (K1 == 128 && N1 == 512 && N2 == 128) chain_launch<128, 512, 128, NW, 2>(a, 2, st);
  else LVC_CHECK_ARG(false, "unsupported (K1, N1, N2): one of (64,256,64), (128,256,64), (128,512,128)");
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}
