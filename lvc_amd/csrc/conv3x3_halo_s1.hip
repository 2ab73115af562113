// conv3x3_halo_s1.hip -- 3x3 / stride 1 / pad 1 convolution, two-way fp16 operand split with ONE fp32 accumulator and a
// software-pipelined tap loop (round 3).
//
// What changed against conv3x3_halo_h2.hip, and why (scripts/micro/mfma_ceiling.hip, profiles/r03_mfma_ceiling.txt): on this part a
// loop of nothing but v_mfma_f32_32x32x16_f16 on random data sustains 1.66 PFLOP/s (the power cap), 1.46 PFLOP/s with the
// fragment-read ratio of a 64 x 64 wave tile -- the h2 kernel delivers 1.03 PFLOP/s of raw MFMA work (345 TF/s fp32-equivalent),
// its matrix pipe busy 63 % of the cycles.  The rest are bubbles of its tap loop: every wave reads its fragments and THEN issues
// the twelve MFMAs that use them (single-buffered fragments: 128 of its 256 registers hold the main + cross accumulators), all
// eight waves leave the per-tap barrier at the same moment and read at the same moment, and the halo refill and the weight
// staging (global -> VGPR -> ds_write) sit between barriers.  Here
//   * ONE accumulator per 32 x 32 block: the residual planes are kept UNSCALED (a = a1 + a2 with a2 = fp16(a - a1)), so the three
//     products a1 b1, a1 b2, a2 b1 of a block chain into the same fp32 accumulator.  fp16's range is made to fit by operand
//     scaling with powers of two (exact): activations x 2^4 (residual plane normal for |a| >= 2^-7, absolute error <= 2^-29 below;
//     |a| <= 4094 or the range word is raised), every weight row x 2^e so that its largest entry lies in [2^13, 2^14) (residual
//     plane normal for every entry >= 2^-17 of the row's largest); the epilogue's per-channel scale carries 2^-(e+4)
//     (`lvc_split_weights_rowscaled`).  64 accumulator registers instead of 128;
//   * fragment reads run one group ahead of the MFMAs that use them, rotating through the registers earlier groups vacate (see
//     `step_body`): a wave never waits for LDS with an idle matrix pipe, and the barrier of a tap sits in the MIDDLE of its MFMA
//     work (the second k16 step needs nothing the barrier orders);
//   * weight planes arrive by LDS-DMA (`global_load_lds_dwordx4`, 64-byte rows, granule XOR swizzle on the source side as in
//     conv_pw_dma.hip) into a ring of three tap buffers, issued right behind a tap's barrier for the tap after next (a full tap
//     of latency budget): no VGPR staging, no ds_write;
//   * the halo window is double-buffered: the next 32-channel chunk is loaded at tap 1, split and written in six small pieces
//     under the MFMAs of taps 4..6 -- no refill phase between chunks, the tap pipeline runs through.
// Tiling (<= 256-pixel patches x 64 NI channels, 8 waves as 4 x 2), stream-K workers with grouped channel tiles, partial-tile
// hand-off and the epilogue through LDS are conv3x3_halo_h2.hip's.
#include "common.h"
#include <type_traits>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define HM 256            // output pixels per tile (patch area <= HM)
#define LROW 40           // fp16 elements per halo row in LDS (32 + 8 pad = 80 B: conflict-free ds_read_b128)
#define HALO_S1 340       // halo pixels per tile: two buffers x two planes x 340 x 80 B = 108,800 B
#define NJ 6              // halo pixels per thread (6 x 64 pixel slots x 8 float4 slots)
#define NT 512
#define SPIN_LIMIT (1 << 24)
#define ACT_SCALE 16.f    // activations x 2^4 before the split (see the header)
#define ACT_MAX 4094.f    // 65504 / 16
#define HALO_MAX_LEVELS 6

struct HaloArgsS {
  const float* x;
  const unsigned short* w;   // [2][Kpad][Kg] fp16 planes of the row-scaled weights (w1 = fp16(w 2^e), w2 = fp16(w 2^e - w1))
  const float* scale;        // per output channel: (FrozenBN scale or 1) * 2^-(e + 4)   -- never null
  const float* shift;
  const float* res;
  float* y;
  float* partials;
  int* flags;
  int N, H, W, C, K, relu, res_mode, ldy, ldr;
  int next_err_index;        // presplit: the consumer's range word (a value beyond its |a| <= 4094 is raised THERE, as its own split would)
  int presplit;              // 1: y is written as the fp16 planes its ONE consumer multiplies (see lvc_conv3x3_nhwc_f16s1_presplit)
  int PH, PW, HW, HP, MP;
  int inv_pw;                // ceil(2^16 / PW): (r * inv_pw) >> 16 == r / PW for the tile rows r < 256
  int tiles_x, tiles_y, tiles_n, nk, total_units, units_per_worker, nworkers, err_index;
  int ngroup;
  int x_bytes;
  long long w_plane_elems;
  // several maps in one launch (same weights, same channel counts, no residual: the RPN head over the pyramid levels): map l owns
  // the row tiles [lv_tile0[l], lv_tile0[l + 1]); nlev = 0: the one map above
  int nlev;
  const float* lv_x[HALO_MAX_LEVELS];
  float* lv_y[HALO_MAX_LEVELS];
  int lv_H[HALO_MAX_LEVELS], lv_W[HALO_MAX_LEVELS], lv_tx[HALO_MAX_LEVELS], lv_ty[HALO_MAX_LEVELS], lv_xbytes[HALO_MAX_LEVELS];
  int lv_tile0[HALO_MAX_LEVELS + 1];
  // ... or L layers of one shape, a map each (the FPN output convs): per-map weight planes, scale and shift (lv_w[0] == nullptr: p.w, p.scale, p.shift)
  const unsigned short* lv_w[HALO_MAX_LEVELS];
  const float* lv_scale[HALO_MAX_LEVELS];
  const float* lv_shift[HALO_MAX_LEVELS];
  // ... with a POINTWISE layer on top of act(conv) (the RPN predictor on the head's hidden map, pred_w != nullptr): y / lv_y are that
  // layer's outputs (rows of ldy floats, ZEROED by the caller), the hidden map is never written.  pred_w: its [2][pred_rows][K] fp16
  // planes (lvc_split_weights: w1 = fp16(w), w2 = fp16((w - w1) 2^11)), pred_K <= 32 outputs, pred_scale / pred_shift per output or null
  const unsigned short* pred_w;
  const float* pred_scale;
  const float* pred_shift;
  long long pred_plane_elems;
  int pred_K, pred_err_index;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ONEACC = true: the single-accumulator numerics of the header (row-scaled weight planes, activations x 2^4, residual planes unscaled).
// ONEACC = false: conv3x3_halo_h2.hip's numerics in this kernel's pipeline -- residual planes x 2^11, a main (a1 b1) and a cross
// (a1 b2 + a2 b1) accumulator folded once per tile, weights = lvc_split_weights planes, full fp16 range (|a| <= 65504).
// Measured with the diagnostics build (-DHALO_TIMELINE, scripts/probe_halo_timeline.py, profiles/r03_halo_timeline.txt) on the p2
// layer: a tap takes ~1950 cycles of which the SIMD's two waves need 1546 on the matrix pipe; the vmcnt wait in front of the barrier
// is 80 - 90 cycles (the weight DMA has landed: its latency is not the bound), the barrier 110 - 750 by wave: the arbiter serves the
// OLDER wave of a SIMD first (waves 0..3 run ahead, 1130 cycles per tap, and idle at the barrier while waves 4..7 finish alone).
// Alternating s_setprio between the two (HALO_PRIO_MODE 1) evens the arrival out (1630 / 1775) but not the tap period (-0.4 % launch
// time): left off.  Per tile, prologue 2.5 % and hand-off + epilogue 7 % of a workgroup's time run without MFMAs.
#ifndef HALO_PRIO_MODE
#define HALO_PRIO_MODE 0
#endif
#if HALO_PRIO_MODE == 1
#define HALO_PRIO(first) do { if ((wave < 4) == (first)) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0); } while (0)
#elif HALO_PRIO_MODE == 2
#define HALO_PRIO(first) do { if (first) { if (wave < 4) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(2); } } while (0)
#else
#define HALO_PRIO(first) do { } while (0)
#endif
// The two groups of a k16 step that read fragments, written out (NI = 2).  Left to the compiler, nearly every fragment wait of the tap
// loop is `s_waitcnt lgkmcnt(0)` placed in front of the group that uses the reads -- also the read issued one instruction earlier -- so a
// wave that runs alone on its SIMD (its partner waiting at the tap's barrier: the older wave of a SIMD is served first and runs ahead)
// stalls a full LDS round trip per group of four MFMAs.  Here the four reads of a group go out FIRST, in the order of their use, and
// every MFMA waits with a COUNT: LDS operations of a wave retire in order, so `lgkmcnt(n)` = "all but the youngest n have landed".
// The counts name only reads of these two blocks; anything the compiler adds to the queue in between (the halo stores) makes a wait
// stricter, never weaker.  The accumulation order per block (G1, G2, G3) is the compiler-scheduled loop's: bit-identical results.
//   halo_g1: late reads of THIS step (alo0, bhi0, alo1, bhi1) + G1 = a1 b2 into (c00, c01, c10, c11); the early fragments ahi / blo
//            were requested by the previous step's halo_g2 as (blo0, blo1, ahi0, ahi1), oldest first.
//   halo_g2: early reads of the NEXT step (blo_n0, blo_n1, ahi_n0, ahi_n1) + G2 = a2 b1 into (c00, c10, c01, c11).
// OA / OB: immediate byte offsets (fragment's k16 step and operand plane) on top of the per-lane LDS addresses.
template <int OA, int OB>
__device__ __forceinline__ void halo_g1(f32x16& c00, f32x16& c01, f32x16& c10, f32x16& c11, const f16x8& ah0, const f16x8& ah1,
                                        const f16x8& bl0, const f16x8& bl1, f16x8& al0, f16x8& al1, f16x8& bh0, f16x8& bh1,
                                        unsigned aA0, unsigned aA1, unsigned aB) {
  asm volatile(
      "ds_read_b128 %[al0], %[aA0] offset:%[oa]\n\t"
      "ds_read_b128 %[bh0], %[aB] offset:%[ob0]\n\t"
      "ds_read_b128 %[al1], %[aA1] offset:%[oa]\n\t"
      "ds_read_b128 %[bh1], %[aB] offset:%[ob1]\n\t"
      "s_waitcnt lgkmcnt(5)\n\t"
      "v_mfma_f32_32x32x16_f16 %[c00], %[ah0], %[bl0], %[c00]\n\t"
      "v_mfma_f32_32x32x16_f16 %[c01], %[ah0], %[bl1], %[c01]\n\t"
      "s_waitcnt lgkmcnt(4)\n\t"
      "v_mfma_f32_32x32x16_f16 %[c10], %[ah1], %[bl0], %[c10]\n\t"
      "v_mfma_f32_32x32x16_f16 %[c11], %[ah1], %[bl1], %[c11]"
      : [al0] "=&v"(al0), [al1] "=&v"(al1), [bh0] "=&v"(bh0), [bh1] "=&v"(bh1), [c00] "+v"(c00), [c01] "+v"(c01), [c10] "+v"(c10),
        [c11] "+v"(c11)
      : [ah0] "v"(ah0), [ah1] "v"(ah1), [bl0] "v"(bl0), [bl1] "v"(bl1), [aA0] "v"(aA0), [aA1] "v"(aA1), [aB] "v"(aB), [oa] "n"(OA),
        [ob0] "n"(OB), [ob1] "n"(OB + 2048));
}
// (bh0 / bh1 are in-out operands of halo_g2 although it only reads them: halo_g1 leaves while its reads are still in flight, and
// the G3 MFMAs the compiler schedules -- which read bh -- must not be placed before the waits in here.)
template <int OA, int OB>
__device__ __forceinline__ void halo_g2(f32x16& c00, f32x16& c01, f32x16& c10, f32x16& c11, const f16x8& al0, const f16x8& al1,
                                        f16x8& bh0, f16x8& bh1, f16x8& bln0, f16x8& bln1, f16x8& ahn0, f16x8& ahn1,
                                        unsigned aAn0, unsigned aAn1, unsigned aBn) {
  asm volatile(
      "ds_read_b128 %[bln0], %[aBn] offset:%[ob0]\n\t"
      "ds_read_b128 %[bln1], %[aBn] offset:%[ob1]\n\t"
      "ds_read_b128 %[ahn0], %[aAn0] offset:%[oa]\n\t"
      "ds_read_b128 %[ahn1], %[aAn1] offset:%[oa]\n\t"
      "s_waitcnt lgkmcnt(6)\n\t"
      "v_mfma_f32_32x32x16_f16 %[c00], %[al0], %[bh0], %[c00]\n\t"
      "s_waitcnt lgkmcnt(5)\n\t"
      "v_mfma_f32_32x32x16_f16 %[c10], %[al1], %[bh0], %[c10]\n\t"
      "s_waitcnt lgkmcnt(4)\n\t"
      "v_mfma_f32_32x32x16_f16 %[c01], %[al0], %[bh1], %[c01]\n\t"
      "v_mfma_f32_32x32x16_f16 %[c11], %[al1], %[bh1], %[c11]"
      : [bln0] "=&v"(bln0), [bln1] "=&v"(bln1), [ahn0] "=&v"(ahn0), [ahn1] "=&v"(ahn1), [c00] "+v"(c00), [c01] "+v"(c01),
        [c10] "+v"(c10), [c11] "+v"(c11), [bh0] "+v"(bh0), [bh1] "+v"(bh1)
      : [al0] "v"(al0), [al1] "v"(al1), [aAn0] "v"(aAn0), [aAn1] "v"(aAn1), [aBn] "v"(aBn), [oa] "n"(OA),
        [ob0] "n"(OB), [ob1] "n"(OB + 2048));
}

template <int NI, bool ONEACC, bool PRED = false>
__global__ __launch_bounds__(NT, 2) void conv3x3_halo_s1_kernel(HaloArgsS p) {
  constexpr int HN = 64 * NI;
  // the fragment groups written out with counted waits (halo_g1 / halo_g2): the one-accumulator instance with 128-channel tiles -- the
  // trunk and FPN layers.  The two-accumulator instance needs every register and spills a few (loop-invariant values); a spill of a
  // register with a read in flight would be silent corruption, so it keeps the compiler-scheduled groups (the Makefile fails the build
  // if the instance below ever spills a vector register).
  constexpr bool ASM_GROUPS = NI == 2 && ONEACC;
  constexpr int PLANE_A = HALO_S1 * LROW;          // halves
  constexpr int A_BUF = 2 * PLANE_A;               // halves per halo buffer (two planes)
  constexpr int A_BYTES = 2 * A_BUF * 2;           // two buffers
  constexpr int PLANE_B = HN * 64;                 // bytes: HN rows x 32 halves
  constexpr int B_BUF = 2 * PLANE_B;               // bytes per tap buffer
  constexpr int RING_BYTES = A_BYTES + 3 * B_BUF;
  constexpr int CS_STRIDE = HN + 4;
  constexpr int CS_BYTES = HM * CS_STRIDE * 4;
  constexpr int SMEM_BYTES = RING_BYTES > CS_BYTES ? RING_BYTES : CS_BYTES;
  constexpr int RBLK = HN / 16;                    // 16-row DMA pieces per plane
  static_assert(2 * RBLK == 8 * NI, "one DMA piece per wave and NI");
  __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[SMEM_BYTES];
  f16* sA = reinterpret_cast<f16*>(smem_raw);
  unsigned char* sB = smem_raw + A_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;   // wave tile: 64 pixels x 32 NI channels
  const int fi = lane & 31, fh = lane >> 5;
  // halo staging: thread = (pixel slot, float4 slot); the slot permutation keeps the 80-byte-pitch stores conflict-free
  const int q = tid & 7;
  const int arid = tid >> 3;
  const int hrow = (arid & 1) * 4 + ((arid >> 1) & 3) + (arid >> 3) * 8;    // halo pixels hrow + 64*j

  const int lw = lvc_xcd_remap(blockIdx.x, p.nworkers);
  const int wq = p.ngroup > 1 ? lw / p.ngroup : lw;
  const int wsel = p.ngroup > 1 ? lw - wq * p.ngroup : 0;
  int u = wq * p.units_per_worker;
  const int u_end = min(u + p.units_per_worker, p.total_units);


  // fragment offsets.  A (halves): halo pixel of output pixel m at tap (0,0); rows past the patch read pixel 0.
  // B (bytes within a plane): row * 64 + swizzled granule of k16 step s2: ((s2 * 2 + fh) ^ ((row >> 2) & 3)) * 16
  int a_frag[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int m = wm * 64 + mi * 32 + fi;
    const int mm = m < p.MP ? m : 0;
    const int py = mm / p.PW, px = mm - py * p.PW;
    a_frag[mi] = (py * p.HW + px) * LROW + fh * 8;
  }
  const int fx3 = (fi >> 2) & 3;
  const int b_row = (wn * 32 * NI + fi) * 64;
  const int b_g[2] = {((0 + fh) ^ fx3) * 16, ((2 + fh) ^ fx3) * 16};
  const int row_off = p.HW * LROW;   // halves per halo row

  int a_lds[NJ];    // LDS offset (halves) of this thread's halo slots
#pragma unroll
  for (int j = 0; j < NJ; ++j) a_lds[j] = min(hrow + 64 * j, p.HP - 1) * LROW + q * 4;
  float big = 0.f;   // largest |activation| staged: beyond the scaled fp16 range -> workspace error word
#ifdef HALO_TIMELINE
  // diagnostics build only (scripts/probe_halo_timeline.py): cycles of this wave between barriers, in the vmcnt wait, in the barrier
  unsigned long long tl_run = 0, tl_vm = 0, tl_bar = 0, tl_last = 0, tl_pro = 0, tl_epi = 0, tl_hand = 0, tl_cs = 0, tl_t0 = __builtin_readcyclecounter();
  unsigned tl_n = 0;
#endif
  while (u < u_end) {
    const int tile = u / p.nk;
    const int cc0 = u - tile * p.nk;
    const int cc1 = min(p.nk, cc0 + (u_end - u));
    const int tile_n = p.ngroup > 1 ? wsel : tile % p.tiles_n;
    const int tile_m = p.ngroup > 1 ? tile : tile / p.tiles_n;
    // the map this row tile belongs to (one map: the launch's own fields)
    int Hl = p.H, Wl = p.W, txl = p.tiles_x, tyl = p.tiles_y, xbl = p.x_bytes, tml = tile_m;
    const float* xl = p.x;
    float* yl = p.y;
    const unsigned short* wl = p.w;
    const float* scl = p.scale;
    const float* shl = p.shift;
    if (p.nlev > 0) {
      int lv = 0;
#pragma unroll
      for (int l = 1; l < HALO_MAX_LEVELS; ++l) lv += (l < p.nlev && tile_m >= p.lv_tile0[l]) ? 1 : 0;
      Hl = p.lv_H[lv]; Wl = p.lv_W[lv]; txl = p.lv_tx[lv]; tyl = p.lv_ty[lv]; xbl = p.lv_xbytes[lv];
      xl = p.lv_x[lv]; yl = p.lv_y[lv];
      if (p.lv_w[0]) { wl = p.lv_w[lv]; scl = p.lv_scale[lv]; shl = p.lv_shift[lv]; }
      tml = tile_m - p.lv_tile0[lv];
    }
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)xl, 0, xbl, 0x00020000);
    const int tx = tml % txl;
    const int t2 = tml / txl;
    const int ty = t2 % tyl;
    const int img = t2 / tyl;
    const int y0 = ty * p.PH, x0 = tx * p.PW;
    const int n0 = tile_n * HN;

    unsigned a_off[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      // slots past the halo re-stage its last pixel (same source, same destination, same data): every lane stays active and the
      // staging code has no branch to split the MFMA schedule around
      const int h = min(hrow + 64 * j, p.HP - 1);
      const int hy = h / p.HW, hx = h - hy * p.HW;
      const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
      const bool ok = yy >= 0 && yy < Hl && xx >= 0 && xx < Wl;
      a_off[j] = ok ? (unsigned)(((img * Hl + yy) * Wl + xx) * p.C + q * 4) * 4u : 0x80000000u;
    }
    // weight DMA: piece idx = wave * NI + j -> plane idx / RBLK, rows (idx % RBLK) * 16 + (lane >> 2), source granule swizzled
    const unsigned short* bsrc[NI];
    int bdst[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int idx = wave * NI + j;
      const int pl = idx / RBLK, rb = idx - pl * RBLK;
      const int row = rb * 16 + (lane >> 2);
      const int G = (lane & 3) ^ ((row >> 2) & 3);
      bsrc[j] = wl + (size_t)pl * p.w_plane_elems + (size_t)(n0 + row) * (9 * p.C) + G * 8;
      bdst[j] = pl * PLANE_B + rb * 16 * 64;
    }
    auto dma_B = [&](int step, int buf) {
#pragma unroll
      for (int j = 0; j < NI; ++j) glds16(bsrc[j] + step * 32, sB + buf * B_BUF + bdst[j]);
    };

    f32x4 areg[NJ];
    auto load_A = [&](int cc) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        areg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, a_off[j], cc * 128, 0));
    };
    auto store_A_piece = [&](f16* dstA, int j) {
      f16x4 h, m;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = ONEACC ? areg[j][e] * ACT_SCALE : areg[j][e];
        const f16 hh = (f16)a;
        h[e] = hh;
        m[e] = ONEACC ? (f16)(a - (float)hh) : (f16)((a - (float)hh) * 2048.f);
        big = fmaxf(big, fabsf(areg[j][e]));
      }
      const int o = a_lds[j];
      *reinterpret_cast<f16x4*>(dstA + o) = h;
      *reinterpret_cast<f16x4*>(dstA + PLANE_A + o) = m;
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

    // Fragments of ONE k16 step (32 registers at NI = 2), rotated in place.  A step issues its MFMAs in three groups -- G1: a1 b2,
    // G2: a2 b1, G3: a1 b1 -- so b2 is dead after G1 and a2 after G2.  Reads: the step's own a2 / b1 ("late", first needed by G2)
    // go out under G1; the NEXT step's a1 / b2 ("early", needed by its G1) go out under G2, into the registers b2 and a2 just
    // vacated.  One DS read per MFMA, every operand issued >= 3 MFMAs (~100 cycles) before its first use.
    f16x8 ahi[2], alo[2], bhi[NI], blo[NI];
    auto rdA = [&](const f16* A, int pl, int mi, int tap_off, int s2) {
      return *reinterpret_cast<const f16x8*>(A + pl * PLANE_A + a_frag[mi] + tap_off + s2 * 16);
    };
    auto rdB = [&](const unsigned char* B, int pl, int ni, int s2) {
      return *reinterpret_cast<const f16x8*>(B + pl * PLANE_B + b_row + ni * 32 * 64 + b_g[s2]);
    };
    // one k16 step: (A, toff, s2, B) = this step's operands; (An, toffn, s2n, Bn) = the next step's
    const unsigned ldsA = (unsigned)(size_t)(lds_ptr_t)(void*)sA, ldsB = (unsigned)(size_t)(lds_ptr_t)(void*)sB;
    auto step_body = [&](const f16* A, int toff, auto s2_tag, const unsigned char* B, const f16* An, int toffn, auto s2n_tag,
                         const unsigned char* Bn) {
      constexpr int s2 = decltype(s2_tag)::value, s2n = decltype(s2n_tag)::value;
      if constexpr (ASM_GROUPS) {
        // per-lane LDS byte addresses: A fragments of this step / of the next, B rows of this step / of the next
        const unsigned aoff = (unsigned)((A - sA) + toff) * 2u, aoffn = (unsigned)((An - sA) + toffn) * 2u;
        const unsigned aA0 = ldsA + aoff + (unsigned)a_frag[0] * 2u, aA1 = ldsA + aoff + (unsigned)a_frag[1] * 2u;
        const unsigned aAn0 = ldsA + aoffn + (unsigned)a_frag[0] * 2u, aAn1 = ldsA + aoffn + (unsigned)a_frag[1] * 2u;
        const unsigned aB = ldsB + (unsigned)(B - sB) + (unsigned)(b_row + b_g[s2]);
        const unsigned aBn = ldsB + (unsigned)(Bn - sB) + (unsigned)(b_row + b_g[s2n]);
        f32x16& g00 = ONEACC ? acc[0][0] : accx[0][0];
        f32x16& g01 = ONEACC ? acc[0][1] : accx[0][ONEACC ? 0 : 1];
        f32x16& g10 = ONEACC ? acc[1][0] : accx[ONEACC ? 0 : 1][0];
        f32x16& g11 = ONEACC ? acc[1][1] : accx[ONEACC ? 0 : 1][ONEACC ? 0 : 1];
        f16x8 blo_n[2], ahi_n[2];
        // late reads (a2 of this step: plane 1 of A; b1: plane 0 of B) + G1
        halo_g1<PLANE_A * 2 + s2 * 32, 0>(g00, g01, g10, g11, ahi[0], ahi[1], blo[0], blo[1], alo[0], alo[1], bhi[0], bhi[1], aA0, aA1, aB);
        // early reads of the next step (b2: plane 1 of B; a1: plane 0 of A) + G2
        halo_g2<s2n * 32, PLANE_B>(g00, g01, g10, g11, alo[0], alo[1], bhi[0], bhi[1], blo_n[0], blo_n[1], ahi_n[0], ahi_n[1], aAn0, aAn1, aBn);
        // G3: a1 b1
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[mi], bhi[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) ahi[mi] = ahi_n[mi];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) blo[ni] = blo_n[ni];
        return;
      }
      // late reads of this step, in the order G2 needs them
      alo[0] = rdA(A, 1, 0, toff, s2);
      bhi[0] = rdB(B, 0, 0, s2);
      alo[1] = rdA(A, 1, 1, toff, s2);
      if (NI == 2) bhi[NI - 1] = rdB(B, 0, NI - 1, s2);
      // G1: a1 b2
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          f32x16& c = ONEACC ? acc[mi][ni] : accx[ONEACC ? 0 : mi][ONEACC ? 0 : ni];
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[mi], blo[ni], c, 0, 0, 0);
        }
      // early reads of the next step: b2 first (its registers are free now), a1 after G2 has released a2
      f16x8 blo_n[NI], ahi_n[2];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) blo_n[ni] = rdB(Bn, 1, ni, s2n);
      // G2: a2 b1 (block order: the operands read first are used first)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          f32x16& c = ONEACC ? acc[mi][ni] : accx[ONEACC ? 0 : mi][ONEACC ? 0 : ni];
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[mi], bhi[ni], c, 0, 0, 0);
        }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) ahi_n[mi] = rdA(An, 0, mi, toffn, s2n);
      // G3: a1 b1
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
    // every fragment read of the written-out groups has landed (before anything may copy or reuse their registers: the chunk
    // loop's back edge, the end of the tile)
    auto drain_frags = [&]() {
      if constexpr (ASM_GROUPS)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ahi[0]), "+v"(ahi[1]), "+v"(blo[0]), "+v"(blo[1]));
    };
    // MFMA, DS read, MFMA, DS read ... for the 4 + 2 NI reads of a step, the remaining MFMAs behind
    auto interleave = [&]() {
      if constexpr (ASM_GROUPS) return;      // the written-out groups fix their own order
#pragma unroll
      for (int i = 0; i < 4 + 2 * NI; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 6 * NI - (4 + 2 * NI), 0);
    };

    const int step0 = cc0 * 9, last_step = cc1 * 9 - 1;
#ifdef HALO_TIMELINE
    const unsigned long long tl_p0 = __builtin_readcyclecounter();
#endif
    // ---- prologue: halo of the first chunk, weights of taps 0 and 1, the early fragments of (tap 0, k16 step 0)
    load_A(cc0);
    dma_B(step0, 0);
    dma_B(step0 + 1, 1);       // cc1 > cc0: a unit has nine taps, so step0 + 1 exists
#pragma unroll
    for (int j = 0; j < NJ; ++j) store_A_piece(sA, j);
    wait_vm<0>();
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) ahi[mi] = rdA(sA, 0, mi, 0, 0);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) blo[ni] = rdB(sB, 1, ni, 0);

#ifdef HALO_TIMELINE
    tl_last = __builtin_readcyclecounter();
    tl_pro += tl_last - tl_p0;
#endif
#pragma unroll 1
    for (int cc = cc0; cc < cc1; ++cc) {
      const int par = (cc - cc0) & 1;
      const f16* Acur = sA + par * A_BUF;
      f16* Anext = sA + (par ^ 1) * A_BUF;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int step = cc * 9 + tap;
        const int toff = (tap / 3) * row_off + (tap % 3) * LROW;
        const int ntap = (tap + 1) % 9;
        const int toff_n = (ntap / 3) * row_off + (ntap % 3) * LROW;
        const unsigned char* Bcur = sB + (tap % 3) * B_BUF;
        const unsigned char* Bnext = sB + ((tap + 1) % 3) * B_BUF;
        // ---- phase A: k16 step 0 of this tap; its early reads are step 1's (same buffers)
        HALO_PRIO(true);
        step_body(Acur, toff, std::integral_constant<int, 0>{}, Bcur, Acur, toff, std::integral_constant<int, 1>{}, Bcur);
        if (tap == 4) store_A_piece(Anext, 0);
        if (tap == 5) store_A_piece(Anext, 2);
        if (tap == 6) store_A_piece(Anext, 4);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
        // Everything this wave has in flight (the weights of tap + 1 issued one tap ago, at tap 2 the halo loads of tap 1) has
        // landed; behind the barrier that holds for every wave, and every wave is done with tap - 1's weight slot
#ifdef HALO_TIMELINE
        const unsigned long long tl_1 = __builtin_readcyclecounter();
        wait_vm<0>();
        const unsigned long long tl_2 = __builtin_readcyclecounter();
        __builtin_amdgcn_s_barrier();
        const unsigned long long tl_3 = __builtin_readcyclecounter();
        tl_run += tl_1 - tl_last; tl_vm += tl_2 - tl_1; tl_bar += tl_3 - tl_2; tl_last = tl_3; ++tl_n;
#else
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();
#endif
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase B: k16 step 1; the DMA of tap + 2 into tap - 1's slot; early reads of (tap + 1, step 0).  Past the end of the
        // unit the DMA re-fetches the last tap and the halo load the last chunk (into buffers nobody reads any more): the
        // instruction stream has no data-dependent branch
        dma_B(min(step + 2, last_step), (tap + 2) % 3);
        HALO_PRIO(false);
        step_body(Acur, toff, std::integral_constant<int, 1>{}, Bcur, tap == 8 ? Anext : Acur, toff_n, std::integral_constant<int, 0>{}, Bnext);
        if (tap == 1) load_A(min(cc + 1, cc1 - 1));
        if (tap == 4) store_A_piece(Anext, 1);
        if (tap == 5) store_A_piece(Anext, 3);
        if (tap == 6) store_A_piece(Anext, 5);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
      }
      drain_frags();
    }
    wait_vm<0>();
    __syncthreads();
#ifdef HALO_TIMELINE
    const unsigned long long tl_e0 = __builtin_readcyclecounter();
    tl_run += tl_e0 - tl_last;
#endif
    u += cc1 - cc0;
    if (!ONEACC) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[mi][ni][e] += accx[ONEACC ? 0 : mi][ONEACC ? 0 : ni][e] * (1.f / 2048.f);
    }

    // ---- split tiles: a worker that does not own the tile's first chunk hands its partial sums to the one that does
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

#ifdef HALO_TIMELINE
    const unsigned long long tl_e1 = __builtin_readcyclecounter();
    tl_hand += tl_e1 - tl_e0;
#endif
    // ---- epilogue through LDS: tile row r is patch pixel (r / PW, r % PW)
    float* Cs = reinterpret_cast<float*>(smem_raw);
    if constexpr (PRED) {
      // A pointwise layer on top (HaloArgsS::pred_w): this workgroup holds act(conv) for HN of the K hidden channels of its pixels --
      // a HN-deep slice of the layer's contraction.  The slice's products go to the output by atomic adds: with K <= 2 HN there are at
      // most two addends per element on a zeroed output, so the sum does not depend on their order.  Arithmetic of a slice: the
      // two-accumulator fp16 split (main a1 b1, cross a2 b1 + a1 b2 with the residual planes x 2^11), as the pointwise kernels'.
      float bigp = 0.f;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int col = wn * 32 * NI + ni * 32 + fi;
        const float sc1 = scl ? scl[n0 + col] : 1.f;
        const float sh1 = shl ? shl[n0 + col] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
            float v = acc[mi][ni][e] * sc1 + sh1;
            if (p.relu) v = v > 0.f ? v : 0.f;
            bigp = (fabsf(v) > bigp || v != v) ? fabsf(v) : bigp;
            Cs[row * CS_STRIDE + col] = v;
          }
      }
      __syncthreads();
      // B fragments from L1 / L2 (every workgroup reads the same 2 x 32 x HN halves), one k16 step ahead of their use
      const unsigned short* bsrcp = p.pred_w + (size_t)fi * p.K + n0 + fh * 8;
      f16x8 pbh = *reinterpret_cast<const f16x8*>(bsrcp), pbl = *reinterpret_cast<const f16x8*>(bsrcp + p.pred_plane_elems);
      f32x16 pm, px2;
#pragma unroll
      for (int e = 0; e < 16; ++e) { pm[e] = 0.f; px2[e] = 0.f; }
      const float* arow = Cs + (wave * 32 + fi) * CS_STRIDE + fh * 8;
#pragma unroll 1
      for (int ks = 0; ks < HN / 16; ++ks) {
        const int kn = ks + 1 < HN / 16 ? ks + 1 : ks;
        const f16x8 nbh = *reinterpret_cast<const f16x8*>(bsrcp + kn * 16);
        const f16x8 nbl = *reinterpret_cast<const f16x8*>(bsrcp + p.pred_plane_elems + kn * 16);
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(arow + ks * 16);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(arow + ks * 16 + 4);
        f16x8 ah, al;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float a = j < 4 ? a0[j & 3] : a1[j & 3];
          const f16 hh = (f16)a;
          ah[j] = hh;
          al[j] = (f16)((a - (float)hh) * 2048.f);
        }
        pm = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, pbh, pm, 0, 0, 0);
        px2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, pbh, px2, 0, 0, 0);
        px2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, pbl, px2, 0, 0, 0);
        pbh = nbh; pbl = nbl;
      }
      // The 32 x 32 result block goes back through the wave's OWN rows of the LDS tile (no other wave reads or writes them), then
      // out in a rolled loop, two pixels x 32 outputs per instruction: written out per accumulator element, the sixteen address
      // computations were spilled and every reload -- a vmcnt(0) -- waited for the atomic before it (+0.36 ms on the RPN head).
      float* rblk = Cs + wave * 32 * CS_STRIDE;
#pragma unroll
      for (int e = 0; e < 16; ++e) rblk[((e & 3) + 8 * (e >> 2) + 4 * fh) * 32 + fi] = pm[e] + px2[e] * (1.f / 2048.f);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      if (fi < p.pred_K) {
        const float psc = p.pred_scale ? p.pred_scale[fi] : 1.f;
        const float psh = (n0 == 0 && p.pred_shift) ? p.pred_shift[fi] : 0.f;     // the bias once: with the first slice
        const size_t origin = (size_t)(img * Hl + y0) * Wl + x0;
        float* const ybase = yl + origin * p.ldy + fi;
        const int ylim = Hl - y0, xlim = Wl - x0;
#pragma unroll 2
        for (int rr = 0; rr < 32; rr += 2) {
          const int r = wave * 32 + rr + fh;
          const int py = (r * p.inv_pw) >> 16, px = r - py * p.PW;
          const float v = rblk[(rr + fh) * 32 + fi];
          if (r < p.MP && py < ylim && px < xlim) unsafeAtomicAdd(ybase + (size_t)(unsigned)(py * Wl + px) * p.ldy, v * psc + psh);
        }
      }
      // the pointwise layer's own range word: a hidden value beyond fp16
      if (!(bigp <= 65504.f)) atomicOr(p.flags + p.pred_err_index, bigp < INFINITY ? 2 : 4);
      __syncthreads();
      continue;
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
          const int col = wn * 32 * NI + ni * 32 + fi;
          Cs[row * CS_STRIDE + col] = acc[mi][ni][e];
        }
    __syncthreads();
#ifdef HALO_TIMELINE
    tl_cs += __builtin_readcyclecounter() - tl_e1;
#endif
    // Output rows: the whole loop is address arithmetic around one LDS read and one store per row, and it ran on the vector ALU for
    // 5.6 % of the p2 layer's time (13.6 % on the 64-channel res2 layer; scripts/probe_halo_timeline.py): the pixel of tile row r comes
    // from a multiply-shift (no integer division), the row's address from the tile origin plus a 32-bit pixel offset.
    constexpr int C4 = HN / 4;
    constexpr int RPI = NT / C4;
    const int c4 = tid % C4, rsub = tid / C4;
    const int col = n0 + c4 * 4;
    if (col < p.K) {
      f32x4 sc = {1.f, 1.f, 1.f, 1.f};
      if (scl) sc = *reinterpret_cast<const f32x4*>(scl + col);
      f32x4 sh = {0.f, 0.f, 0.f, 0.f};
      if (shl) sh = *reinterpret_cast<const f32x4*>(shl + col);
      const size_t origin = (size_t)(img * Hl + y0) * Wl + x0;          // pixel index of the patch's corner
      float* const ybase = yl + origin * p.ldy + col;
      const int ylim = Hl - y0, xlim = Wl - x0;
      const float* cs = Cs + rsub * CS_STRIDE + c4 * 4;
      // one copy of the loop per (residual mode, ReLU): with the modes tested inside, every iteration ended in the compiler's
      // vmcnt(0) lgkmcnt(0) -- the LDS read of a row and the acknowledgement of the previous row's store, one after the other,
      // sixteen times per tile (the "output rows" share of scripts/probe_halo_timeline.py)
      float bigq = 0.f;      // presplit: the largest value handed to the consumer
      auto rows = [&](auto rm_tag, auto relu_tag, auto ps_tag) {
        constexpr int RM = decltype(rm_tag)::value;
        constexpr bool RELU = decltype(relu_tag)::value;
        constexpr bool PS = decltype(ps_tag)::value;
#pragma unroll 4
        for (int it = 0; it < HM / RPI; ++it) {
          const int r = it * RPI + rsub;
          const int py = (r * p.inv_pw) >> 16, px = r - py * p.PW;
          if (r < p.MP && py < ylim && px < xlim) {
            const unsigned pix = (unsigned)(py * Wl + px);
            f32x4 v = *reinterpret_cast<const f32x4*>(cs + it * RPI * CS_STRIDE);
            v = v * sc + sh;
            if (RM == 1) {
              v += *reinterpret_cast<const f32x4*>(p.res + (origin + pix) * p.ldr + col);
            } else if (RM == 2) {
              const int yy = y0 + py, xx = x0 + px;
              const size_t ro = ((size_t)(img * (Hl >> 1) + (yy >> 1)) * (Wl >> 1) + (xx >> 1));
              v += *reinterpret_cast<const f32x4*>(p.res + ro * p.ldr + col);
            }
            if (RELU) {
              v[0] = v[0] > 0.f ? v[0] : 0.f; v[1] = v[1] > 0.f ? v[1] : 0.f;
              v[2] = v[2] > 0.f ? v[2] : 0.f; v[3] = v[3] > 0.f ? v[3] : 0.f;
            }
            if (PS) {
#pragma unroll
              for (int e = 0; e < 4; ++e) bigq = (v[e] > bigq || v[e] != v[e]) ? v[e] : bigq;
              // the row's 32-channel chunk as [32 hi halves | 32 lo halves] (128 B, where the fp32 values would lie): exactly the
              // planes the consumer's own split would form (x 2^4, hi = fp16, lo = fp16 of the remainder) -- it skips the split
              typedef _Float16 f16x4s __attribute__((ext_vector_type(4)));
              f16x4s h, l;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float a = v[e] * ACT_SCALE;
                const f16 hh = (f16)a;
                h[e] = hh;
                l[e] = (f16)(a - (float)hh);
              }
              char* yc = reinterpret_cast<char*>(yl + (origin + pix) * p.ldy + (col & ~31)) + (col & 31) * 2;
              *reinterpret_cast<f16x4s*>(yc) = h;
              *reinterpret_cast<f16x4s*>(yc + 64) = l;
            } else {
              *reinterpret_cast<f32x4*>(ybase + (size_t)pix * p.ldy) = v;
            }
          }
        }
      };
      using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>; using T2 = std::integral_constant<int, 2>;
      if (p.presplit) {      // (the entry point admits it without residual, with ReLU)
        rows(T0{}, std::true_type{}, std::true_type{});
        if (!(bigq * ACT_SCALE <= ACT_MAX)) atomicOr(p.flags + p.next_err_index, bigq < INFINITY ? 2 : 4);
      }
      else if (p.res_mode == 0) { if (p.relu) rows(T0{}, std::true_type{}, std::false_type{}); else rows(T0{}, std::false_type{}, std::false_type{}); }
      else if (p.res_mode == 1) { if (p.relu) rows(T1{}, std::true_type{}, std::false_type{}); else rows(T1{}, std::false_type{}, std::false_type{}); }
      else { if (p.relu) rows(T2{}, std::true_type{}, std::false_type{}); else rows(T2{}, std::false_type{}, std::false_type{}); }
    }
    __syncthreads();
#ifdef HALO_TIMELINE
    tl_epi += __builtin_readcyclecounter() - tl_e0;
#endif
  }
#ifdef HALO_TIMELINE
  if (lane == 0 && (blockIdx.x & 15) == 0) {
    unsigned long long* d = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(p.partials) + (size_t)512 * 256 * 128 * 4) + ((blockIdx.x >> 4) * 8 + wave) * 16;
    d[0] = tl_run; d[1] = tl_vm; d[2] = tl_bar; d[3] = tl_n; d[4] = tl_pro; d[5] = tl_epi; d[6] = __builtin_readcyclecounter() - tl_t0; d[7] = 1;
    d[8] = tl_hand; d[9] = tl_cs;
  }
#endif
  // bit 1: a FINITE activation beyond the form's range; bit 2: a non-finite one (usually what an upstream layer that left ITS range
  // in this pass handed down: kernels.check_conv_error_word does not move this layer for it while another layer reports bit 1)
  if (!(big <= (ONEACC ? ACT_MAX : 65504.f))) atomicOr(p.flags + p.err_index, big < INFINITY ? 2 : 4);
}

#define LVC_MAX_WORKERS 1024
static int g_cus_halo_s = 0;

// Patch shape for an H x W output: PH * PW <= 256 pixels, (PH + 2) * (PW + 2) <= HALO_S1 halo pixels, fewest patches
// (every patch costs a full 256-row MFMA tile whatever its fill); ties go to the smaller halo.
static void pick_patch_s(int H, int W, int* PH, int* PW) {
  long long best_tiles = -1;
  int best_halo = 0, bh = 1, bw = 8;
  for (int pw = 4; pw <= 128; ++pw)
    for (int ph = 1; ph * pw <= HM; ++ph) {
      const int halo = (ph + 2) * (pw + 2);
      if (halo > HALO_S1) break;
      const long long tiles = (long long)lvc_cdiv(H, ph) * lvc_cdiv(W, pw);
      if (best_tiles < 0 || tiles < best_tiles || (tiles == best_tiles && halo < best_halo)) {
        best_tiles = tiles; best_halo = halo; bh = ph; bw = pw;
      }
    }
  *PH = bh; *PW = bw;
}

// tiles_m_total row tiles (all maps); fills the stream-K split and launches
static int halo_s1_finish(HaloArgsS& a, bool oneacc, long long tiles_m_total, int Kg, void* workspace, void* stream) {
  const int ni = a.K <= 64 ? 1 : 2;
  const int HN = 64 * ni;
  a.tiles_n = lvc_cdiv(a.K, HN);
  a.nk = a.C / 32;
  long long units = tiles_m_total * a.tiles_n * a.nk;
  LVC_CHECK_ARG(units < (1ll << 31), "iteration space too large");
  a.total_units = (int)units;
  a.w_plane_elems = (long long)(lvc_cdiv(a.K, 128) * 128) * Kg;   // planes are padded to 128 rows
  if (g_cus_halo_s == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    g_cus_halo_s = cus;
  }
  int cap = g_cus_halo_s;  // one worker per CU (154 KB of LDS per workgroup)
  if (cap > LVC_MAX_WORKERS) cap = LVC_MAX_WORKERS;
  a.ngroup = 1;
  if ((a.tiles_n == 2 || a.tiles_n == 4) && cap % a.tiles_n == 0 && units / a.tiles_n >= cap / a.tiles_n) {
    a.ngroup = a.tiles_n;
    units /= a.tiles_n;
    cap /= a.tiles_n;
    a.total_units = (int)units;
  }
  int workers = (int)(units < cap ? units : cap);
  a.units_per_worker = (int)((units + workers - 1) / workers);
  a.nworkers = (int)((units + a.units_per_worker - 1) / a.units_per_worker) * a.ngroup;
  a.partials = (float*)workspace;
  a.flags = (int*)((char*)workspace + (size_t)LVC_MAX_WORKERS * 256 * 128 * 4);
  a.err_index = LVC_MAX_WORKERS + lvc_range_slot();   // the layer's own range word (common.cpp)
  hipStream_t st = (hipStream_t)stream;
  if (a.pred_w) {
    LVC_CHECK_ARG(!oneacc && ni == 2, "pointwise layer on top: the two-accumulator form with >= 128 hidden channels");
    hipLaunchKernelGGL((conv3x3_halo_s1_kernel<2, false, true>), dim3(a.nworkers), dim3(NT), 0, st, a);
  } else if (oneacc) {
    if (ni == 1) hipLaunchKernelGGL((conv3x3_halo_s1_kernel<1, true>), dim3(a.nworkers), dim3(NT), 0, st, a);
    else hipLaunchKernelGGL((conv3x3_halo_s1_kernel<2, true>), dim3(a.nworkers), dim3(NT), 0, st, a);
  } else {
    if (ni == 1) hipLaunchKernelGGL((conv3x3_halo_s1_kernel<1, false>), dim3(a.nworkers), dim3(NT), 0, st, a);
    else hipLaunchKernelGGL((conv3x3_halo_s1_kernel<2, false>), dim3(a.nworkers), dim3(NT), 0, st, a);
  }
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// a pointwise layer on top of the 3x3 layer (lvc_conv3x3_nhwc_f16_levels_pred)
struct HaloPred {
  const unsigned short* w;
  const float* scale;
  const float* shift;
  int K, rows, ld, slot;
};

static void halo_s1_patch(HaloArgsS& a, int H, int W) {
  pick_patch_s(H, W, &a.PH, &a.PW);
#ifdef LVC_HALO_PATCH_HOOK   // experiment build only (scripts/sweep_halo_patch.sh: make HOOKS=-DLVC_HALO_PATCH_HOOK): no getenv on the launch path otherwise
  if (const char* e = getenv("LVC_HALO_PATCH")) {   // experiments: "PH,PW"
    int ph = 0, pw = 0;
    if (sscanf(e, "%d,%d", &ph, &pw) == 2 && ph > 0 && pw > 0 && ph * pw <= HM && (ph + 2) * (pw + 2) <= HALO_S1) { a.PH = ph; a.PW = pw; }
  }
#endif
  a.HW = a.PW + 2; a.HP = (a.PH + 2) * a.HW; a.MP = a.PH * a.PW;
  a.inv_pw = (65536 + a.PW - 1) / a.PW;
}

static int halo_s1_launch(bool oneacc, const float* x, const unsigned short* w_split, const float* scale, const float* shift,
                          const float* residual, float* y, int N, int H, int W, int C, int K, int Kg, int relu, int res_mode,
                          int ldy, int ldr, void* workspace, void* stream, int presplit = 0, int next_slot = 0) {
  LVC_CHECK_ARG(x && w_split && workspace && y, "null pointer");
  LVC_CHECK_ARG(!oneacc || scale, "the single-accumulator form needs the row factors in `scale`");
  LVC_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && K > 0, "non-positive dimension");
  LVC_CHECK_ARG(C % 32 == 0 && Kg == 9 * C, "needs C % 32 == 0 and Kg == 9*C");
  LVC_CHECK_ARG(res_mode >= 0 && res_mode <= 2 && (res_mode == 0 || residual), "bad residual");
  if (res_mode == 2) LVC_CHECK_ARG(H % 2 == 0 && W % 2 == 0, "upsample-add needs even output size");
  HaloArgsS a;
  memset(&a, 0, sizeof a);
  a.x = x; a.w = w_split; a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
  a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.relu = relu; a.res_mode = res_mode;
  a.ldy = ldy > 0 ? ldy : K; a.ldr = ldr > 0 ? ldr : K;
  LVC_CHECK_ARG((K & 3) == 0 && (a.ldy & 3) == 0 && (res_mode == 0 || (a.ldr & 3) == 0), "K, ldy, ldr must be multiples of 4");
  LVC_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)w_split & 15) == 0 && ((uintptr_t)y & 15) == 0 &&
                    ((uintptr_t)workspace & 15) == 0 && ((uintptr_t)scale & 15) == 0, "pointers must be 16-byte aligned");
  if (presplit) LVC_CHECK_ARG(oneacc && relu == 1 && res_mode == 0 && K % 32 == 0 && a.ldy == K, "pre-split output: single accumulator, ReLU, no residual, K % 32 == 0, dense rows");
  if (presplit) LVC_CHECK_ARG(next_slot > 0 && next_slot < lvc_range_slots(), "pre-split output: the consumer's range slot");
  a.presplit = presplit; a.next_err_index = LVC_MAX_WORKERS + next_slot;
  halo_s1_patch(a, H, W);
  a.tiles_x = lvc_cdiv(W, a.PW); a.tiles_y = lvc_cdiv(H, a.PH);
  const long long xb = (long long)N * H * W * C * 4;
  LVC_CHECK_ARG(xb < (1ll << 31), "input tensor must be smaller than 2 GiB");
  a.x_bytes = (int)xb;
  return halo_s1_finish(a, oneacc, (long long)N * a.tiles_x * a.tiles_y, Kg, workspace, stream);
}

// L maps [N, Hs[l], Ws[l], C] through the SAME 3x3 layer in one launch (outputs ys[l] [N, Hs[l], Ws[l], K], rows of K floats): the
// RPN head over the pyramid levels.  One stream-K split over the row tiles of all maps (in the order given), one patch shape
// (chosen for the largest map): the small maps no longer pay a launch each that cannot fill the chip.  Per output pixel the arithmetic is
// the single-map launch's.  oneacc as above; no residual.
static int halo_s1_levels(int oneacc, const float* const* xs, float* const* ys, const int* Hs, const int* Ws, int L,
                          const unsigned short* w_split, const float* scale, const float* shift, const unsigned short* const* ws,
                          const float* const* scales, const float* const* shifts, int N, int C, int K, int Kg, int relu, void* workspace,
                          void* stream, const HaloPred* pred = nullptr) {
  LVC_CHECK_ARG(xs && ys && Hs && Ws && L >= 1 && L <= HALO_MAX_LEVELS, "1..6 maps");
  if (ws) {
    LVC_CHECK_ARG(scales && shifts, "null pointer");
    w_split = ws[0]; scale = scales[0]; shift = shifts[0];
  }
  LVC_CHECK_ARG(w_split && workspace, "null pointer");
  LVC_CHECK_ARG(!oneacc || scale, "the single-accumulator form needs the row factors in `scale`");
  LVC_CHECK_ARG(N > 0 && C > 0 && K > 0 && C % 32 == 0 && Kg == 9 * C && (K & 3) == 0, "bad shape");
  HaloArgsS a;
  memset(&a, 0, sizeof a);
  a.w = w_split; a.scale = scale; a.shift = shift; a.res = nullptr;
  a.N = N; a.C = C; a.K = K; a.relu = relu; a.res_mode = 0; a.ldy = K; a.ldr = K;
  LVC_CHECK_ARG(((uintptr_t)w_split & 15) == 0 && ((uintptr_t)workspace & 15) == 0 && ((uintptr_t)scale & 15) == 0, "pointers must be 16-byte aligned");
  if (pred) {
    // at most two channel tiles per pixel tile: two addends per output element, whatever their order (see the kernel)
    LVC_CHECK_ARG(pred->w && ((uintptr_t)pred->w & 15) == 0 && K % 128 == 0 && K <= 256, "pointwise layer on top: needs K = 128 or 256");
    LVC_CHECK_ARG(pred->K >= 1 && pred->K <= 32 && pred->rows >= 32 && pred->ld >= pred->K, "pointwise layer on top: 1..32 outputs, planes of >= 32 rows");
    LVC_CHECK_ARG(pred->slot > 0 && pred->slot < lvc_range_slots(), "pointwise layer on top: its range slot");
    a.pred_w = pred->w; a.pred_scale = pred->scale; a.pred_shift = pred->shift; a.pred_K = pred->K;
    a.pred_plane_elems = (long long)pred->rows * K;
    a.pred_err_index = LVC_MAX_WORKERS + pred->slot;
    a.ldy = pred->ld;
  }
  int big = 0;      // the patch shape is chosen for the largest map, wherever it stands in the list
  for (int l = 1; l < L; ++l)
    if ((long long)Hs[l] * Ws[l] > (long long)Hs[big] * Ws[big]) big = l;
  halo_s1_patch(a, Hs[big], Ws[big]);
  long long tiles = 0;
  a.nlev = L;
  for (int l = 0; l < L; ++l) {
    LVC_CHECK_ARG(xs[l] && ys[l] && Hs[l] > 0 && Ws[l] > 0, "bad map");
    // outputs: float4 rows, or -- with a pointwise layer on top -- single floats added atomically
    LVC_CHECK_ARG(((uintptr_t)xs[l] & 15) == 0 && ((uintptr_t)ys[l] & (pred ? 3 : 15)) == 0, "pointers must be 16-byte aligned");
    const long long xb = (long long)N * Hs[l] * Ws[l] * C * 4;
    LVC_CHECK_ARG(xb < (1ll << 31), "input tensor must be smaller than 2 GiB");
    a.lv_x[l] = xs[l]; a.lv_y[l] = ys[l]; a.lv_H[l] = Hs[l]; a.lv_W[l] = Ws[l]; a.lv_xbytes[l] = (int)xb;
    if (ws) {
      LVC_CHECK_ARG(ws[l] && (!oneacc || scales[l]) && ((uintptr_t)ws[l] & 15) == 0 && ((uintptr_t)scales[l] & 15) == 0 && ((uintptr_t)shifts[l] & 15) == 0,
                    "per-map weights: null or misaligned pointer");
      a.lv_w[l] = ws[l]; a.lv_scale[l] = scales[l]; a.lv_shift[l] = shifts[l];
    }
    a.lv_tx[l] = lvc_cdiv(Ws[l], a.PW); a.lv_ty[l] = lvc_cdiv(Hs[l], a.PH);
    a.lv_tile0[l] = (int)tiles;
    tiles += (long long)N * a.lv_tx[l] * a.lv_ty[l];
    LVC_CHECK_ARG(tiles < (1ll << 30), "too many tiles");
  }
  a.lv_tile0[L] = (int)tiles;
  a.x = xs[0]; a.y = ys[0]; a.H = Hs[0]; a.W = Ws[0]; a.tiles_x = a.lv_tx[0]; a.tiles_y = a.lv_ty[0]; a.x_bytes = a.lv_xbytes[0];
  return halo_s1_finish(a, oneacc != 0, tiles, Kg, workspace, stream);
}

extern "C" int lvc_conv3x3_nhwc_f16_levels(int oneacc, const float* const* xs, float* const* ys, const int* Hs, const int* Ws, int L,
                                            const unsigned short* w_split, const float* scale, const float* shift, int N, int C, int K,
                                            int Kg, int relu, void* workspace, void* stream) {
  return halo_s1_levels(oneacc, xs, ys, Hs, Ws, L, w_split, scale, shift, nullptr, nullptr, nullptr, N, C, K, Kg, relu, workspace, stream);
}

// ... with a pointwise layer on top, act(conv) never written (the RPN head: 3x3 conv + ReLU, then objectness | anchor deltas): ys[l] are
// the POINTWISE layer's outputs [N, Hs[l], Ws[l], pred_ld floats per pixel], zeroed by the caller (the kernel adds the two 128-channel
// slices of the contraction atomically: two addends, order-free).  pred_w: the [2][pred_rows][K] fp16 planes of lvc_split_weights,
// pred_scale / pred_shift [pred_K] or null, pred_slot: that layer's range word (a hidden value beyond 65504 raises it).
extern "C" int lvc_conv3x3_nhwc_f16_levels_pred(int oneacc, const float* const* xs, float* const* ys, const int* Hs, const int* Ws, int L,
                                                 const unsigned short* w_split, const float* scale, const float* shift, int N, int C,
                                                 int K, int Kg, int relu, const unsigned short* pred_w, const float* pred_scale,
                                                 const float* pred_shift, int pred_K, int pred_rows, int pred_ld, int pred_slot,
                                                 void* workspace, void* stream) {
  HaloPred pr = {pred_w, pred_scale, pred_shift, pred_K, pred_rows, pred_ld, pred_slot};
  return halo_s1_levels(oneacc, xs, ys, Hs, Ws, L, w_split, scale, shift, nullptr, nullptr, nullptr, N, C, K, Kg, relu, workspace, stream, &pr);
}

// The same with a LAYER per map (L layers of one shape: the FPN output convs): ws / scales / shifts are [host] arrays of L device
// pointers (weight planes, per-channel scale, per-channel shift or NULL).
extern "C" int lvc_conv3x3_nhwc_f16_layers(int oneacc, const float* const* xs, float* const* ys, const int* Hs, const int* Ws, int L,
                                            const unsigned short* const* ws, const float* const* scales, const float* const* shifts, int N,
                                            int C, int K, int Kg, int relu, void* workspace, void* stream) {
  LVC_CHECK_ARG(ws && scales && shifts, "null pointer");
  return halo_s1_levels(oneacc, xs, ys, Hs, Ws, L, nullptr, nullptr, nullptr, ws, scales, shifts, N, C, K, Kg, relu, workspace, stream);
}

// Single-accumulator form.  Same arguments as lvc_conv3x3_nhwc_f16x2 except the weights: w_split = the [2][Kpad][Kg] fp16 planes
// written by lvc_split_weights_rowscaled, and `scale` = (the layer's per-channel scale or 1) x the row factors that call returned
// (never null).  An activation with |a| > 4094 (or NaN) raises bit 1 of the workspace error word.
extern "C" int lvc_conv3x3_nhwc_f16s1(const float* x, const unsigned short* w_split, const float* scale, const float* shift,
                                       const float* residual, float* y, int N, int H, int W, int C, int K, int Kg, int relu,
                                       int res_mode, int ldy, int ldr, void* workspace, void* stream) {
  return halo_s1_launch(true, x, w_split, scale, shift, residual, y, N, H, W, C, K, Kg, relu, res_mode, ldy, ldr, workspace, stream);
}

// lvc_conv3x3_nhwc_f16s1 whose output goes to ONE consumer, lvc_conv1x1_nhwc_f16s1_presplit (a bottleneck's conv2 -> conv3, reference
// resnet.py:200-205): y [N,H,W,K] holds, per pixel and 32-channel chunk, the 32 hi halves then the 32 lo halves of the two-way fp16
// split of relu(conv * scale + shift) x 2^4 (128 B, the chunk's place in a row of K floats) -- bit for bit the planes the consumer's own
// split forms, which it then skips (six vector instructions per MFMA in its chunk loop).  ReLU is applied; no residual; K % 32 == 0.
// A value whose x 2^4 leaves the consumer's range (|a| > 4094) raises range word `next_slot` (the CONSUMER's), as its split would have.
extern "C" int lvc_conv3x3_nhwc_f16s1_presplit(const float* x, const unsigned short* w_split, const float* scale, const float* shift, void* y,
                                                int N, int H, int W, int C, int K, int Kg, int next_slot, void* workspace, void* stream) {
  return halo_s1_launch(true, x, w_split, scale, shift, nullptr, (float*)y, N, H, W, C, K, Kg, 1, 0, K, 0, workspace, stream, 1, next_slot);
}

// lvc_conv3x3_nhwc_f16x2 (arguments, weight planes, numerics: main + cross accumulators, |a| <= 65504) on this file's pipeline.
extern "C" int lvc_conv3x3_nhwc_f16x2_pipe(const float* x, const unsigned short* w_split, const float* scale, const float* shift,
                                            const float* residual, float* y, int N, int H, int W, int C, int K, int Kg, int relu,
                                            int res_mode, int ldy, int ldr, void* workspace, void* stream) {
  return halo_s1_launch(false, x, w_split, scale, shift, residual, y, N, H, W, C, K, Kg, relu, res_mode, ldy, ldr, workspace, stream);
}

// Row-scaled two-plane split of packed weights wp [rows][Kg] fp32:  e = 13 - floor(log2(max |wp[row][:]|)) (0 for an all-zero
// row), w1 = fp16(wp 2^e), w2 = fp16(wp 2^e - w1) (the UNSCALED residual), row_factor[row] = 2^-e * act_unscale.
// planes_out [2][rows][Kg] fp16.  One workgroup per row.
__global__ __launch_bounds__(256) void split_rowscaled_kernel(const float* __restrict__ wp, int Kg, long long plane,
                                                              unsigned short* __restrict__ out, float* __restrict__ row_factor,
                                                              float act_unscale) {
  __shared__ float red[256];
  const int row = blockIdx.x;
  const float* src = wp + (size_t)row * Kg;
  float mx = 0.f;
  for (int i = threadIdx.x; i < Kg; i += 256) {
    const float v = fabsf(src[i]);
    mx = (v > mx || v != v) ? v : mx;    // NaN propagates: the row keeps e = 0 and its NaNs
  }
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const float o = red[threadIdx.x + s];
      if (o > red[threadIdx.x] || o != o) red[threadIdx.x] = o;
    }
    __syncthreads();
  }
  mx = red[0];
  int e = 0;
  if (mx > 0.f && mx < INFINITY) {
    int ex;
    frexpf(mx, &ex);          // mx = f * 2^ex, f in [0.5, 1)  ->  floor(log2 mx) = ex - 1
    e = 13 - (ex - 1);
    e = e > 100 ? 100 : e < -100 ? -100 : e;
  }
  const float s2 = ldexpf(1.f, e);
  for (int i = threadIdx.x; i < Kg; i += 256) {
    const float v = src[i] * s2;
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    out[(size_t)row * Kg + i] = __builtin_bit_cast(unsigned short, h);
    out[plane + (size_t)row * Kg + i] = __builtin_bit_cast(unsigned short, l);
  }
  if (threadIdx.x == 0) row_factor[row] = ldexpf(1.f, -e) * act_unscale;
}

extern "C" int lvc_split_weights_rowscaled(const float* wp, int rows, int Kg, void* planes_out, float* row_factor, void* stream) {
  LVC_CHECK_ARG(wp && planes_out && row_factor && rows > 0 && Kg > 0, "bad arguments");
  hipLaunchKernelGGL(split_rowscaled_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, wp, Kg, (long long)rows * Kg,
                     (unsigned short*)planes_out, row_factor, 1.f / ACT_SCALE);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}
