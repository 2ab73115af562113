// conv_bneck.hip -- a whole ResNet bottleneck block in ONE launch (round 6):
//
//     t1 = relu(bn1(conv1x1(x)))      never leaves the CU: fp16 hi / lo planes in LDS, on the output tile + its one-pixel halo
//     t2 = relu(bn2(conv3x3(t1)))     never leaves the wave: the accumulators, split, ARE the next layer's MFMA operand
//     y  = relu(bn3(conv1x1(t2)) + shortcut)      shortcut = x (identity blocks) or bn_s(conv1x1_s(x)) folded into the same GEMM
//
// Reference: detectron2/modeling/backbone/resnet.py:195-211 (BottleneckBlock.forward).  As separate launches (round 5: the 3x3 kernel
// and the conv3 -> next conv1 chain) a res2 block moves x, t1 (written, read with its halo), t2 (written, read) and y: 1.7 GB per
// block at batch 8 of 800x1333 on layers that already run at the rate of a copy.  Here a block reads x once (plus the halo ring of
// its tiles, served by L2 / the memory-side cache) and writes y once: 1.1 GB.
//
// Work decomposition.  A workgroup of 8 waves owns an output tile of 8 x 32 pixels of one image, ONE workgroup per CU (the first form
// -- 4 waves on 4 x 32 tiles, two workgroups per CU -- streamed twice the weight bytes per pixel through LDS-DMA, ~8 B/clk/CU, and its
// two workgroups walked the phases in lockstep: profiles/README.md).  Everything is computed TRANSPOSED as in conv_pw_chain.hip:
// out^T [channels x pixels] = W [channels x k] . act^T [k x pixels]; the MFMA's A operand is a weight fragment from LDS, its B operand
// the activations, an accumulator lane owns ONE pixel (lane % 32) and sixteen channels of a 32-channel block (c = 8 i + 4 (lane / 32)
// + {0..3}).
//   phase 1 (conv1): the 340 halo pixels are 11 groups of 32 over the 8 waves (three waves take two: the tile loop exists in two
//     copies, by group count, so that a stage is one scheduling region).  x is fetched in ROW order (8 lanes per 128 B line; lane =
//     pixel cost 64 tag look-ups per instruction), 32 channels a stage two stages ahead, goes through a per-wave LDS scratch into the
//     B-operand layout, is split in registers and contracted against W1 from the ring; the stage's MFMAs are issued one stage late,
//     under the next chunk's conversion.  Epilogue (FrozenBN, ReLU, zero outside the image = conv2's padding, x 2^4, fp16 split):
//     t1 into LDS as [k16 step][plane][k half][halo pixel][16 B] -- conflict-free for these writes and phase 2's tap-shifted reads.
//   phase 2 (conv2): a wave owns one output row for all 64 mid channels; a ring stage = two of the 36 (k16 step, tap) pairs; under it
//     the next tile's first two x chunks are fetched.
//   phase 3 (conv3): a ring stage = 32 output channels; B operand = t2 from the wave's own registers; the accumulators go through
//     the wave's scratch into row order, there FrozenBN, + residual (x rows again: L2 / memory-side cache) and ReLU, full-line stores;
//     a block's epilogue runs under the next block's MFMAs.
// The weights of all three layers come through ONE ring of 8 x 8 KB slots, six stages ahead; wait counts from a compile-time table.
// All weights come as ONE pre-swizzled image (lvc_amd.kernels.pack_bottleneck): a sequence of stages, each a sequence of 1 KB
// fragments in lane order (lane l: row l % 32, k half l / 32, 8 fp16), so a stage is a straight LDS-DMA copy and a fragment read is
// conflict-free; the contraction index is permuted within every 16 (0-3, 8-11, 4-7, 12-15) as in conv_pw_chain.hip.
// Numerics: the single-accumulator two-way fp16 split of conv3x3_halo_s1.hip / conv_pw_chain.hip (row-scaled weight planes,
// activations x 2^4, |a| <= 4094 for x, t1 and t2, else bit 1 of the layer's range word is raised and the host re-routes the block).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define ACT_SCALE 16.f
#define ACT_MAX 4094.f
#define LVC_MAX_WORKERS 1024
#define BN_MARK 0xFFFF0000u      // a byte offset beyond every tensor's buffer range: loads return zeros, stores are dropped

struct BneckArgs {
  const float* x;            // [N][H][W][ldx]
  float* y;                  // [N][H][W][ldy]
  const unsigned short* w;   // the stage images (pack_bottleneck)
  const float *s1, *t1, *s2, *t2, *s3, *t3;    // epilogue scale (x row factor) / shift of the three layers, never null
  int* flags;
  int N, H, W, ldx, ldy, tiles_x, tiles_y, ntiles, err_index;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// loads the compiler does not track (see conv_pw_chain.hip): completion through the waits below, which are tied to the registers
template <int IMM> __device__ __forceinline__ f32x4 load_untracked(u32x4 rsrc, unsigned voff, unsigned soff) {
  f32x4 v;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff), "n"(IMM) : "memory");
  return v;
}
__device__ __forceinline__ void track_abs(float& big, float a, float b) {
  asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(big) : "v"(a), "v"(b));
}
__device__ __forceinline__ void tie4(f32x4& a, f32x4& b, f32x4& c, f32x4& d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
__device__ __forceinline__ u32x4 make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long b = (unsigned long long)base;
  return u32x4{(unsigned)__builtin_amdgcn_readfirstlane((unsigned)b), (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu),
               (unsigned)__builtin_amdgcn_readfirstlane(bytes), 0x00020000u};
}
// soffset of a store is always the literal 0 (conv_pw_chain.hip: the >64-bit store data hazard with an SGPR soffset)
__device__ __forceinline__ void store_b128(f32x4 v, __amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, voff, 0, 0);
}
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, f16x8& h, f16x8& l) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float u = a[c] * ACT_SCALE, v = b[c] * ACT_SCALE;
    const f16 uh = (f16)u, vh = (f16)v;
    h[c] = uh; h[4 + c] = vh;
    l[c] = (f16)(u - (float)uh); l[4 + c] = (f16)(v - (float)vh);
  }
}
#define MFMA3(acc, wh, wl, zh, zl)                                       \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, zl, acc, 0, 0, 0);    \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, zh, acc, 0, 0, 0);    \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, zh, acc, 0, 0, 0)

// 64 mid channels, 256 output channels; CIN input channels (64: res2.0 with the projection shortcut folded into conv3's GEMM as 64
// more contraction channels, PROJ; 256: identity blocks, residual = x).
constexpr int BN_TH = 8, BN_TW = 32, BN_HW = BN_TW + 2, BN_HH = BN_TH + 2, BN_HPIX = BN_HW * BN_HH;      // 340 halo pixels
constexpr int BN_NG1 = (BN_HPIX + 31) / 32;                                                                // 11 groups of 32
constexpr int BN_T1_SUB = BN_HPIX * 16;                 // bytes of one (k16 step, plane, k half) sub-plane of t1
constexpr int BN_T1_BYTES = 16 * BN_T1_SUB;             // 4 steps x 2 planes x 2 halves = 87 040
constexpr int BN_NW = 8;                                // waves of a workgroup = rows of its tile
constexpr int BN_SLOT = 8192, BN_NSLOT = 8, BN_RING = BN_NSLOT * BN_SLOT;      // every stage image is 8 fragments = 8 KB; 7 stages ahead
constexpr int BN_TAB = (4 * 64 + 2 * 256) * 4;

// Vector-memory operations of a wave in program order, per ring stage t (all counts per wave):
//     [wait] barrier | D: LDS-DMA of stage t + 2 (2 pieces) | phase loads TWO stages ahead: phase 1 the x rows of chunk c + 2 (8 loads),
//     phase 3 the residual rows of block j + 2 (4) | fragment reads, MFMAs | (phase 3) 4 stores
// LOADS retire in order among themselves, stores are acknowledged out of order with them (conv_pw_chain.hip), so a wait names the
// number of LOADS issued behind its target -- what the previous stage issued -- and drains nothing else.
template <int CIN, bool PROJ>
__global__ __launch_bounds__(BN_NW * 64) void conv_bneck_kernel(BneckArgs p) {
  constexpr int NS1 = CIN / 32, NS2 = 18, NS3 = PROJ ? 16 : 8, NST = NS1 + NS2 + NS3;
  static_assert(!PROJ || CIN == 64, "projection blocks: 64 input channels");
  // pl(ts): loads stage ts of a tile issues BEHIND its own DMA piece; pl_sum(ts): over the six stages before ts (never reaching into
  // the previous tile: phase 2 starts at stage NS1 >= 2 ... see the static_assert) -- the wait counts of the weights-only stages
  struct Issue {
    static constexpr int pl(int ts) {
      if (ts < 0) ts += NST;                                                       // the previous tile
      if (ts < NS1) return ts + 2 < NS1 ? 8 : 0;                                   // x(c + 2)
      if (ts < NS1 + NS2) {
        const int q = ts - NS1;
        return (q == 2 || q == 10 ? 8 : 0) + (q == NS2 - 2 ? (PROJ ? 8 : 0) : q == NS2 - 1 ? (PROJ ? 0 : 4) : 0);
      }
      return PROJ ? 0 : (ts - NS1 - NS2 + 1 < 8 ? 4 : 0);                          // identity blocks: block j + 1's residual rows
    }
  };
  auto pl_sum = [](int ts) constexpr { int n = 0; for (int d = 1; d <= 6; ++d) n += Issue::pl(ts - d); return n; };
  static_assert(NS1 >= 2, "two chunks of x are in flight at a tile's start");
#ifdef BN_DIAG_SMALL_LDS      // timing only (wrong results): the tables alias t1
  __shared__ __attribute__((aligned(1024))) unsigned char smem[BN_RING + BN_T1_BYTES - BN_DIAG_SMALL_LDS];
  unsigned char* const t1s = smem + BN_RING - BN_DIAG_SMALL_LDS;
  float* tab_s1 = reinterpret_cast<float*>(smem + BN_RING);
#else
  __shared__ __attribute__((aligned(1024))) unsigned char smem[BN_RING + BN_T1_BYTES + BN_TAB];
  unsigned char* const t1s = smem + BN_RING;
  float* tab_s1 = reinterpret_cast<float*>(smem + BN_RING + BN_T1_BYTES);
#endif
  float* tab_t1 = tab_s1 + 64;
  float* tab_s2 = tab_t1 + 64;
  float* tab_t2 = tab_s2 + 64;
  float* tab_s3 = tab_t2 + 64;
  float* tab_t3 = tab_s3 + 256;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 31, fh = lane >> 5;

  // this workgroup's tiles: XCD k (= blockIdx % 8) owns a contiguous range of the row-major tile order, its workgroups walk it
  // together, so that the tiles in flight on one L2 are neighbours (shared halo rows)
  const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3, wgx = ((int)gridDim.x - xcd + 7) >> 3;
  const int tq = p.ntiles >> 3, tr = p.ntiles & 7;
  const int tbase = xcd * tq + (xcd < tr ? xcd : tr), tcnt = tq + (xcd < tr ? 1 : 0);
  if (within >= tcnt) return;

  for (int i = tid; i < 64; i += BN_NW * 64) {
    tab_s1[i] = p.s1[i]; tab_t1[i] = p.t1[i];
    tab_s2[i] = p.s2[i]; tab_t2[i] = p.t2[i];
  }
  for (int i = tid; i < 256; i += BN_NW * 64) {
    tab_s3[i] = p.s3[i]; tab_t3[i] = p.t3[i];
  }
  __syncthreads();

  // weights by LDS-DMA through a buffer resource: ONE vector register (lane * 16) addresses every piece, the stage offset is scalar
  const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, NST * 8192, 0x00020000);
  const int wlane = lane * 16;
  int rs = 0;      // ring slot of the current stage
  auto dma = [&](int ts, int slot) {      // stage ts of a tile (a constant at every call site): one 1 KB piece per wave
    unsigned char* dst = smem + slot * BN_SLOT + wave * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, (lds_ptr_t)dst, 16, wlane, ts * 8192 + wave * 1024, 0, 0);
  };
  // top of a stage: its weights (issued BN_NSLOT - 2 stages ago) and the phase's loads have landed, everybody is done with the slot
  // TWO stages back (phases 1 and 3 issue a stage's MFMAs one stage late, under the next one's vector work), which takes the stage
  // BN_NSLOT - 2 ahead; returns this stage's fragment base for the lane
#ifdef BN_DIAG_TIMELINE      // wave 0 of four workgroups stamps every stage of its first tiles into the head of the workspace
  unsigned long long* const tl_base = reinterpret_cast<unsigned long long*>(p.flags) - (size_t)LVC_MAX_WORKERS * 256 * 128 / 2;
  const int tl_slot = blockIdx.x == 0 ? 0 : blockIdx.x == (gridDim.x >> 1) ? 1 : blockIdx.x == 8 ? 2 : blockIdx.x == (gridDim.x >> 1) + 8 ? 3 : -1;
  int tl_iter = 0;
#endif
  auto stage_top = [&](auto n_, int ts) -> const unsigned char* {
#ifdef BN_DIAG_TIMELINE
    const unsigned long long tl0 = __builtin_amdgcn_s_memtime();
    wait_vm<decltype(n_)::value>();
    const unsigned long long tl1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_barrier();
    const unsigned long long tl2 = __builtin_amdgcn_s_memtime();
    if (tl_slot >= 0 && tl_iter < 6 && tid == 0) {
      unsigned long long* o = tl_base + ((tl_slot * 6 + tl_iter) * 48 + ts) * 4;
      o[0] = tl0; o[1] = tl1; o[2] = tl2; o[3] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    }
#else
    wait_vm<decltype(n_)::value>();
    __builtin_amdgcn_s_barrier();
#endif
    int tn = ts + BN_NSLOT - 2;
    if (tn >= NST) tn -= NST;
    dma(tn, (rs + BN_NSLOT - 2) & (BN_NSLOT - 1));
    const unsigned char* S = smem + rs * BN_SLOT + lane * 16;
    rs = (rs + 1) & (BN_NSLOT - 1);
    return S;
  };

  const unsigned xbytes = (unsigned)p.N * (unsigned)p.H * (unsigned)p.W * (unsigned)p.ldx * 4u;
  const unsigned ybytes = (unsigned)p.N * (unsigned)p.H * (unsigned)p.W * (unsigned)p.ldy * 4u;
  const u32x4 xres = make_rsrc(p.x, xbytes);
  const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, ybytes, 0x00020000);

  float big = 0.f;
  // this lane's base into the epilogue tables (beyond the 64 KB an LDS instruction's immediate reaches from a zero base): ONE opaque
  // register + immediates -- left to itself the compiler kept every table address in a register of its own and spilled them
  const float* tabl = tab_s1 + 4 * fh;
  asm volatile("" : "+v"(tabl));
  const float* tabr = tab_s1 + 4 * (lane & 7);      // the same for the row-order epilogue of phase 3 (channel run lane % 8 of a block)
  asm volatile("" : "+v"(tabr));
  constexpr int TS1 = 0, TT1 = 64, TS2 = 128, TT2 = 192, TS3 = 256, TT3 = 512;
  // phase 2's per-lane base inside t1: (row wave, column fi) of the halo tile, k half fh
  const unsigned char* const zb = t1s + fh * BN_T1_SUB + (wave * BN_HW + fi) * 16;
  const int tpi = p.tiles_x * p.tiles_y;
  // per-wave transposition scratch (8 KB inside t1's space: t1 is dead outside phase 2) and this lane's slots in it: sw(i) in row
  // order (instruction i: pixel 8 i + lane / 8, run lane % 8), sr(k) in MFMA order (pixel fi, run 2 k + fh)
  unsigned char* const scr = t1s + wave * 8192;
  // sw(i) = 16 (64 i + 8 (lane / 8) + (lane % 8 ^ (lane / 16 % 4 + 4 (i % 2)))) = (sw0 ^ 64 (i % 2)) + 1024 i;  sr(k) = 16 (8 fi + (2 k + fh ^ fi / 2 % 8)) =
  // sr0 ^ 32 k: two registers instead of eight (the tile loop has none to spare)
  const int sw0 = (8 * (lane >> 3) + ((lane & 7) ^ ((lane >> 4) & 3))) * 16;
  const int sr0 = (8 * fi + (fh ^ ((fi >> 1) & 7))) * 16;
  auto sw = [&](int i) { return (sw0 ^ ((i & 1) << 6)) + 1024 * i; };
  auto sr = [&](int k) { return sr0 ^ (k << 5); };
  static_assert(NST >= BN_NSLOT - 2, "prologue");
#pragma unroll
  for (int i = 0; i < BN_NSLOT - 2; ++i) dma(i, i);
#ifdef BN_STAGGER
  // de-phase the CUs: every workgroup walks load-bound, matrix-bound and store-bound phases of equal length, and all of them start together
  {
    const int ph = (blockIdx.x >> 3) & 3;
#pragma unroll 1
    for (int i = 0; i < BN_STAGGER * ph; ++i) __builtin_amdgcn_s_sleep(127);
  }
#endif
  // two copies of the tile loop: waves with two groups of halo pixels in phase 1 (11 groups over 8 waves) and waves with one -- no
  // branches inside a stage, so that its MFMAs and its vector work are ONE scheduling region
  auto tiles = [&](auto g2_) {
  constexpr bool G2 = decltype(g2_)::value;      // this wave has a second group of halo pixels in phase 1
  constexpr bool LATE = false;
  // x offsets (row order) of a tile's halo pixels for this lane; tile < 0: none (loads return zeros)
  unsigned xo[2][4];
  f32x4 xr[2][2][4];                      // [chunk % 2][group][instruction]: row order
  auto set_xo = [&](int tile) {
    int lane_o = lane;      // opaque: the halo coordinates are recomputed per tile instead of living in registers across the loop
    asm volatile("" : "+v"(lane_o));
    const int n = tile / tpi, trem = tile - n * tpi;
    const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    const int y0 = ty * BN_TH, x0 = tx * BN_TW;
#pragma unroll
    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int h = 32 * (wave + BN_NW * gi) + 8 * i + (lane_o >> 3);
        const int hy = h / BN_HW, hx = h - hy * BN_HW;
        const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
        const bool ok = tile >= 0 && h < BN_HPIX && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
        xo[gi][i] = ok ? (unsigned)((n * p.H + yy) * p.W + xx) * (unsigned)p.ldx * 4u + (lane_o & 7) * 16u : BN_MARK;
#ifdef BN_DIAG_NOX
        xo[gi][i] = BN_MARK;
#endif
      }
  };
  auto load_x = [&](int slot, int c) {
#pragma unroll
    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
      for (int i = 0; i < 4; ++i) xr[slot][gi][i] = load_untracked<0>(xres, xo[gi][i], 128u * c);
  };
  // the first tile's first two chunks; later tiles' are issued under the previous tile's phase 2
  set_xo(tbase + within);
  load_x(0, 0);
  load_x(1, 1);
  wait_vm<0>();
#pragma unroll 1
  for (int tl = within; tl < tcnt; tl += wgx) {
    const int tile = tbase + tl;
    const int n = tile / tpi, trem = tile - n * tpi;
    const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    const int y0 = ty * BN_TH, x0 = tx * BN_TW;

    // ---------------------------------------------------------------- phase 1: t1 = relu(bn1(W1 x)) on the tile + halo
    // x is fetched in ROW order -- lane l of the group's instruction i reads 16 B of pixel 8 i + l / 8, channel run l % 8 of the stage's
    // 32 channels: 8 lanes per 128 B line, 8 lines an instruction (as lane = pixel it would be 64 lookups of 32 B lines halves: the
    // texture unit's tag rate, not HBM, bounded the phase) -- and goes through a per-wave LDS scratch into the MFMA layout (below)
    set_xo(tile);      // again: not kept across phase 3 (registers)
    float msk[2];
    int hl[2];
    {
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
#pragma unroll
      for (int gi = 0; gi < 2; ++gi) {
        const int h = 32 * (wave + BN_NW * gi) + (lane_o & 31);
        const int hy = h / BN_HW, hx = h - hy * BN_HW;
        const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
        const bool ok = h < BN_HPIX && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
        msk[gi] = ok ? 1.f : 0.f;
        hl[gi] = h;
      }
    }
    f32x16 acc1[2][2];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1[gi][cb][e] = 0.f;

    // software pipeline: stage c converts chunk c (scratch round trip, fp16 split: vector work) while the matrix pipe runs chunk c - 1
    f16x8 zh[2][2][2], zl[2][2][2];      // [chunk % 2][group][k16 step]
    const unsigned char* Sp = nullptr;   // the previous stage's fragments
    auto mfma1 = [&](int par, const unsigned char* S) {
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) {
        f16x8 wf[2][2];      // [channel block][plane]
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) wf[cb][pl] = *reinterpret_cast<const f16x8*>(S + (((sp * 2 + cb) * 2 + pl) << 10));
#pragma unroll
        for (int gi = 0; gi < 2; ++gi)
          if (gi == 0 || G2) {
#pragma unroll
#ifdef BN_DIAG_NOMFMA1
            for (int cb = 0; cb < 2; ++cb) { asm volatile("" : "+v"(acc1[gi][cb]) : "v"(wf[cb][0]), "v"(wf[cb][1]), "v"(zh[par][gi][sp]), "v"(zl[par][gi][sp])); }
#else
            for (int cb = 0; cb < 2; ++cb) { MFMA3(acc1[gi][cb], wf[cb][0], wf[cb][1], zh[par][gi][sp], zl[par][gi][sp]); }
#endif
          }
      }
    };
    static_for<0, NS1>([&](auto c_) {
      constexpr int c = decltype(c_)::value;
      constexpr int XS = c & 1;
      // loads behind x(c): at c = 0 chunk 1's (8); else the previous stage's DMA (1) and, if it issued them, chunk c + 1's rows (8)
      // chunks 0 and 1 were fetched under the previous tile's phase 2 and are older than every load its phase 3 waited for: stages 0
      // and 1 wait for their weights (issued six stages back) -- behind those: five DMA pieces, the residual rows of the last phase-3
      // stages (identity blocks: 4 each up to block 5's stage) and chunk 2's rows (stage 0); from stage 2 on: x(c), issued two stages back
      constexpr int NW = c < 2 ? BN_NSLOT - 3 + pl_sum(c) : 1 + (c + 1 < NS1 ? 8 : 0);
      const unsigned char* S = stage_top(std::integral_constant<int, NW>{}, c);
#ifdef BN_DIAG_NOSCR
      f32x4 xbk[2][4];
#endif
#pragma unroll
      for (int gi = 0; gi < 2; ++gi) tie4(xr[XS][gi][0], xr[XS][gi][1], xr[XS][gi][2], xr[XS][gi][3]);
      // row order -> MFMA order through the wave's scratch: 16 B slot of (pixel q, run r) = 8 q + (r ^ (q / 2 % 8)), conflict-free both
      // ways; the registers are free once the writes are issued and take the chunk two stages ahead
#pragma unroll
      for (int gi = 0; gi < 2; ++gi)
        if (gi == 0 || G2) {
#pragma unroll
#ifdef BN_DIAG_NOSCR
          for (int i = 0; i < 4; ++i) xbk[gi][i] = xr[XS][gi][i];
#else
          for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(scr + gi * 4096 + sw(i)) = xr[XS][gi][i];
#endif
        }
      asm volatile("" ::: "memory");
      if (c + 2 < NS1) load_x(XS, c + 2);
#ifdef BN_DIAG_PF0      // experiment: expose the load latency of every stage (how long is it under the kernel's own traffic?)
      wait_vm<0>();
#endif
      // the two waves of a SIMD (w, w + 4) take the halves in opposite order: one's MFMAs run under the other's vector and LDS work
      auto prep = [&]() {
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
          if (gi == 0 || G2) {
            f32x4 xb[4];      // lane (pixel fi, half fh): runs 2 k + fh, k = 0..3 = k16 step k / 2, its first / second four channels
#pragma unroll
#ifdef BN_DIAG_NOSCR
            for (int k = 0; k < 4; ++k) xb[k] = xbk[gi][k];
#else
            for (int k = 0; k < 4; ++k) xb[k] = *reinterpret_cast<const f32x4*>(scr + gi * 4096 + sr(k));
#endif
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
#ifdef BN_DIAG_NOSPLIT
              zh[XS][gi][sp] = __builtin_bit_cast(f16x8, xb[2 * sp]);
              zl[XS][gi][sp] = __builtin_bit_cast(f16x8, xb[2 * sp + 1]);
#else
              split8(xb[2 * sp], xb[2 * sp + 1], zh[XS][gi][sp], zl[XS][gi][sp]);
              track_abs(big, xb[2 * sp][0], xb[2 * sp][1]);
              track_abs(big, xb[2 * sp][2], xb[2 * sp][3]);
              track_abs(big, xb[2 * sp + 1][0], xb[2 * sp + 1][1]);
              track_abs(big, xb[2 * sp + 1][2], xb[2 * sp + 1][3]);
#endif
            }
          }
        }
      };
#ifdef BN_HANDMIX
      // ONE instruction stream, interleaved by hand and pinned: an MFMA, then a two-value slice of the next chunk's conversion in its shadow
      // (left to the scheduler the stage is a block of MFMAs followed by a block of vector work: the wave stalls at every MFMA issue
      // for the pipe and then runs the conversion with the pipe idle -- removing either block saved its full time).  Consecutive MFMAs
      // alternate between the two channel blocks' accumulators.
      {
        constexpr int NGR = G2 ? 2 : 1;
        constexpr int NM = c > 0 ? 12 * NGR : 0, NH = 8 * NGR;      // MFMAs / two-value slices of this stage
        f32x4 xb[4];            // the scratch rows of the group being converted (group 1's replace group 0's at the stage's midpoint)
        f16x8 wf[2][2];         // [channel block][plane] of the k16 step being multiplied (step 1's replace step 0's at the midpoint)
        auto load_xb = [&](int gi) {
#pragma unroll
          for (int k = 0; k < 4; ++k) xb[k] = *reinterpret_cast<const f32x4*>(scr + gi * 4096 + sr(k));
        };
        auto load_wf = [&](int sp) {
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) wf[cb][pl] = *reinterpret_cast<const f16x8*>(Sp + (((sp * 2 + cb) * 2 + pl) << 10));
        };
        load_xb(0);
        if (c > 0) load_wf(0);
        auto slice = [&](auto h_) {      // slice h: group h / 8, register (h / 2) % 4, values 2 (h % 2) + {0, 1}
          constexpr int h = decltype(h_)::value;
          constexpr int gi = h / 8, k = (h / 2) % 4, e0 = 2 * (h % 2);
          if (NGR == 2 && h == 8 && NM == 0) load_xb(1);
          const float a = xb[k][e0], b2 = xb[k][e0 + 1];
          const float u = a * ACT_SCALE, v = b2 * ACT_SCALE;
          const f16 uh = (f16)u, vh = (f16)v;
          constexpr int d0 = 4 * (k & 1) + e0;
          zh[XS][gi][k >> 1][d0] = uh; zh[XS][gi][k >> 1][d0 + 1] = vh;
          zl[XS][gi][k >> 1][d0] = (f16)(u - (float)uh); zl[XS][gi][k >> 1][d0 + 1] = (f16)(v - (float)vh);
          track_abs(big, a, b2);
        };
        if constexpr (NM == 0) {
          static_for<0, NH>([&](auto h_) { slice(h_); });
        } else {
          static_for<0, NM>([&](auto m_) {
            constexpr int m = decltype(m_)::value;
            // order: k16 step, group, then the triple's three products with the channel block alternating
            constexpr int sp = m / (6 * NGR), r = m % (6 * NGR), gi = r / 6, t = (r % 6) / 2, cb = r % 2;
            if (m == NM / 2) {      // the midpoint: every step-0 MFMA is issued (it read its operands), group 0 is converted
              load_wf(1);
              if (NGR == 2) load_xb(1);
            }
            const f16x8 wa = t == 1 ? wf[cb][1] : wf[cb][0];
            const f16x8 zb2 = t == 0 ? zl[XS ^ 1][gi][sp] : zh[XS ^ 1][gi][sp];
            acc1[gi][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa, zb2, acc1[gi][cb], 0, 0, 0);
            constexpr int h0 = m * NH / NM, h1 = (m + 1) * NH / NM;
            static_for<h0, h1>([&](auto h_) { slice(h_); });
            __builtin_amdgcn_sched_barrier(0);
          });
        }
      }
#else
#ifdef BN_INTERLEAVE      // experiment: one instruction stream, the conversion's vector work between the MFMAs (scripts/probe_bneck_variants.sh)
      if (c > 0) mfma1(XS ^ 1, Sp);
      prep();
      if (c > 0) {
#pragma unroll
        for (int i = 0; i < (G2 ? 24 : 12); ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, BN_INTERLEAVE, 0);
        }
      }
#else
      if (LATE) {
        prep();
        __builtin_amdgcn_sched_barrier(0);
        if (c > 0) mfma1(XS ^ 1, Sp);
      } else {
        if (c > 0) mfma1(XS ^ 1, Sp);
        __builtin_amdgcn_sched_barrier(0);
        prep();
      }
#endif
#endif
      Sp = S;
    });
    __builtin_amdgcn_s_barrier();      // the scratch of every wave lies inside t1: nobody writes t1 before everybody has read its last chunk
    mfma1((NS1 - 1) & 1, Sp);          // (behind the barrier: the last chunk's MFMAs and the epilogue's vector work are one region)
    // epilogue of phase 1: FrozenBN + ReLU, zero outside the image (conv2's padding), split, into LDS
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      if (gi == 0 || G2) {
        const bool live = hl[gi] < BN_HPIX;      // only the stores are predicated (the group's tail lanes have no halo pixel)
        unsigned char* dst = t1s + fh * BN_T1_SUB + hl[gi] * 16;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          f16x8 yh[2], yl[2];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(tabl + TS1 + 32 * cb + 8 * i);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(tabl + TT1 + 32 * cb + 8 * i);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float o = fmaxf(acc1[gi][cb][4 * i + e] * sc[e] + sh[e], 0.f) * msk[gi];
              v[e] = o;
              const float a = o * ACT_SCALE;
              const f16 hh = (f16)a;
              yh[i >> 1][4 * (i & 1) + e] = hh;
              yl[i >> 1][4 * (i & 1) + e] = (f16)(a - (float)hh);
            }
            track_abs(big, v[0], v[1]);
            track_abs(big, v[2], v[3]);
          }
#pragma unroll
          for (int sp = 0; sp < 2; ++sp) {
            const int s = 2 * cb + sp;
            if (live) {
              *reinterpret_cast<f16x8*>(dst + ((s * 2 + 0) * 2) * BN_T1_SUB) = yh[sp];
              *reinterpret_cast<f16x8*>(dst + ((s * 2 + 1) * 2) * BN_T1_SUB) = yl[sp];
            }
          }
        }
      }
    }

    // ---------------------------------------------------------------- phase 2: t2 = relu(bn2(conv3x3(t1))), one output row per wave
    const int oy = y0 + wave;
    unsigned yo[4], ro[4];      // row order: instruction i covers pixels 8 i .. 8 i + 7 of the wave's row, 128 B each
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ox = x0 + 8 * i + (lane >> 3);
      const bool ook = oy < p.H && ox < p.W;
      const unsigned opix = (unsigned)((n * p.H + oy) * p.W + ox);
      yo[i] = ook ? opix * (unsigned)p.ldy * 4u + (lane & 7) * 16u : BN_MARK;
      ro[i] = ook ? opix * (unsigned)p.ldx * 4u + (lane & 7) * 16u : BN_MARK;
#ifdef BN_DIAG_NOSTORE
      yo[i] = BN_MARK;
#endif
#ifdef BN_DIAG_NORES
      ro[i] = BN_MARK;
#endif
    }
    // projection blocks: x of the lane's own pixel (MFMA order), 64 channels
    const unsigned rop = (oy < p.H && x0 + fi < p.W) ? (unsigned)((n * p.H + oy) * p.W + x0 + fi) * (unsigned)p.ldx * 4u + fh * 16u : BN_MARK;
    f32x4 rb[2][4];      // identity blocks: the residual rows of output block j in rb[j % 2]
    f32x4 xi[8];         // projection blocks: x of the own pixel
    auto load_r = [&](int slot, int j) {
#pragma unroll
      for (int i = 0; i < 4; ++i) rb[slot][i] = load_untracked<0>(xres, ro[i], 128u * j);
    };
    f32x16 acc2[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc2[cb][e] = 0.f;

    // a stage = two of the 36 (k16 step, tap) pairs: 8 fragments [pair][channel block][plane]
    static_for<0, NS2>([&](auto q_) {
      constexpr int q = decltype(q_)::value;
      constexpr int ts = NS1 + q;
      // loads behind this stage's weights: the previous stage's DMA (2) and what it issued of phase 3's first operands
      // loads behind this stage's weights (issued six stages back): five DMA pieces and whatever stages ts - 6 .. ts - 1 issued after
      // their own piece -- phase 1's rows two chunks ahead, the next tile's chunks 0 / 1 (stages 2 and 10 of this phase), phase 3's first operands
      constexpr int NW = BN_NSLOT - 3 + pl_sum(ts);
      if (q == 0) wait_lgkm0();      // t1: this wave's writes are done before the barrier lets anybody read
      const unsigned char* S = stage_top(std::integral_constant<int, NW>{}, ts);
      if (q == 2) {
        const int tn = tl + wgx;
        set_xo(tn < tcnt ? tbase + tn : -1);
        load_x(0, 0);
      }
      if (q == 10) load_x(1, 1);
      if (PROJ) {
        if (q == NS2 - 2) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            xi[2 * u] = load_untracked<0>(xres, rop, 64u * u);
            xi[2 * u + 1] = load_untracked<32>(xres, rop, 64u * u);
          }
        }
      } else {
        if (q == NS2 - 1) load_r(0, 0);
      }
#ifndef BN_DIAG_NOP2
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int u = 2 * q + e;
        const int s = u / 9, tp = u % 9, dy = tp / 3, dx = tp % 3;
        const f16x8 zh = *reinterpret_cast<const f16x8*>(zb + ((s * 2 + 0) * 2) * BN_T1_SUB + (dy * BN_HW + dx) * 16);
#ifdef BN_DIAG_HALFZ      // timing only: half of phase 2's activation reads
        const f16x8 zl = zh;
#else
        const f16x8 zl = *reinterpret_cast<const f16x8*>(zb + ((s * 2 + 1) * 2) * BN_T1_SUB + (dy * BN_HW + dx) * 16);
#endif
        f16x8 wh[2], wl[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          wh[cb] = *reinterpret_cast<const f16x8*>(S + (((e * 2 + cb) * 2 + 0) << 10));
#ifdef BN_DIAG_HALFW      // timing only (wrong results): half of phase 2's weight-fragment reads
          wl[cb] = wh[cb];
#else
          wl[cb] = *reinterpret_cast<const f16x8*>(S + (((e * 2 + cb) * 2 + 1) << 10));
#endif
        }
        // the two channel blocks' accumulators alternate: no MFMA waits for the result of the one before it (each block's own
        // order -- wh zl, wl zh, wh zh -- is unchanged: same sums)
        acc2[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[0], zl, acc2[0], 0, 0, 0);
        acc2[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[1], zl, acc2[1], 0, 0, 0);
        acc2[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[0], zh, acc2[0], 0, 0, 0);
        acc2[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[1], zh, acc2[1], 0, 0, 0);
        acc2[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[0], zh, acc2[0], 0, 0, 0);
        acc2[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[1], zh, acc2[1], 0, 0, 0);
      }
#endif
    });
    // epilogue of phase 2: the split accumulators are phase 3's B operand (run behind phase 3's first stage top: one region with its MFMAs)
    f16x8 t2h[4], t2l[4];
    auto epi2 = [&]() {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(tabl + TS2 + 32 * cb + 8 * i);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(tabl + TT2 + 32 * cb + 8 * i);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float o = fmaxf(acc2[cb][4 * i + e] * sc[e] + sh[e], 0.f);
          v[e] = o;
          const float a = o * ACT_SCALE;
          const f16 hh = (f16)a;
          t2h[2 * cb + (i >> 1)][4 * (i & 1) + e] = hh;
          t2l[2 * cb + (i >> 1)][4 * (i & 1) + e] = (f16)(a - (float)hh);
        }
        track_abs(big, v[0], v[1]);
        track_abs(big, v[2], v[3]);
      }
    };

    // ---------------------------------------------------------------- phase 3: y = relu(bn3(W3 t2) + shortcut), 32 channels a stage
    // software pipeline: a stage issues its block's MFMAs and, under them, the epilogue of the block the previous stage completed
    f16x8 xh[4], xl[4];
    f32x16 acc3[2];
    auto epi3 = [&](auto j_) {
      constexpr int j = decltype(j_)::value;
      constexpr int RS = j % 2;
#ifdef BN_DIAG_NOEPI3
      asm volatile("" :: "v"(acc3[j & 1]), "v"(rb[RS][0]), "v"(rb[RS][1]), "v"(rb[RS][2]), "v"(rb[RS][3]));
      return;
#endif
      // the accumulators through the wave's scratch into row order; there FrozenBN (a lane's four channels are the same in all four
      // instructions: two table reads a stage instead of eight in the accumulator layout), + residual, ReLU, full-line stores
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc3[j & 1][4 * i + e];
        *reinterpret_cast<f32x4*>(scr + sr(i)) = v;
      }
      const f32x4 sc = *reinterpret_cast<const f32x4*>(tabr + TS3 + 32 * j);
      const f32x4 sh = *reinterpret_cast<const f32x4*>(tabr + TT3 + 32 * j);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 v = *reinterpret_cast<const f32x4*>(scr + sw(i));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float o = v[e] * sc[e] + sh[e];
          if (!PROJ) o += rb[RS][i][e];
          v[e] = fmaxf(o, 0.f);
        }
        store_b128(v, yres, yo[i] + 128u * j);
      }
    };
    static_for<0, NS3>([&](auto st_) {
      constexpr int st = decltype(st_)::value;
      constexpr int j = PROJ ? st / 2 : st, half = PROJ ? st % 2 : 0, ts = NS1 + NS2 + st;
      // identity blocks: the stage runs the epilogue of block j - 1, whose residual rows were issued two stages back; behind them the
      // previous stage's DMA piece and block j's rows.  Otherwise the weights only (six stages back).
      constexpr int NW = PROJ ? (st == 0 ? 1 : BN_NSLOT - 3) : j == 0 ? BN_NSLOT - 3 + pl_sum(ts) : 5;
      const unsigned char* S = stage_top(std::integral_constant<int, NW>{}, ts);
      if (st == 0) epi2();
      if (PROJ) {
        if (st == 0) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            asm volatile("" : "+v"(xi[2 * u]), "+v"(xi[2 * u + 1]));
            split8(xi[2 * u], xi[2 * u + 1], xh[u], xl[u]);
          }
        }
      } else if (j > 0) {
        tie4(rb[(j - 1) & 1][0], rb[(j - 1) & 1][1], rb[(j - 1) & 1][2], rb[(j - 1) & 1][3]);
      }
      if (half == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc3[j & 1][e] = 0.f;
      }
      auto mfma3 = [&]() {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const f16x8 wh = *reinterpret_cast<const f16x8*>(S + ((s * 2 + 0) << 10));
          const f16x8 wl = *reinterpret_cast<const f16x8*>(S + ((s * 2 + 1) << 10));
#ifdef BN_DIAG_NOMFMA3
          asm volatile("" : "+v"(acc3[j & 1]) : "v"(wh), "v"(wl));
#else
          if (half == 0) { MFMA3(acc3[j & 1], wh, wl, t2h[s], t2l[s]); }
          else { MFMA3(acc3[j & 1], wh, wl, xh[s], xl[s]); }
#endif
        }
      };
      if (LATE) {
        if (j > 0 && half == 0) epi3(std::integral_constant<int, (j > 0 ? j - 1 : 0)>{});
        __builtin_amdgcn_sched_barrier(0);
        mfma3();
      } else {
        mfma3();
        __builtin_amdgcn_sched_barrier(0);
        if (j > 0 && half == 0) epi3(std::integral_constant<int, (j > 0 ? j - 1 : 0)>{});
      }
      // the next block's residual rows go where block j - 1's just were read
      if (!PROJ && j + 1 < 8) load_r((j + 1) & 1, j + 1);
    });
    if (!PROJ) {
      wait_vm<1>();      // block 7's rows (issued in stage 6; behind them stage 7's DMA piece)
      tie4(rb[1][0], rb[1][1], rb[1][2], rb[1][3]);
    }
    epi3(std::integral_constant<int, 7>{});
#ifdef BN_DIAG_TIMELINE
    ++tl_iter;
#endif
  }
  };
  if (wave < BN_NG1 - BN_NW) tiles(std::true_type{}); else tiles(std::false_type{});
  wait_vm<0>();
  if (!(big <= ACT_MAX)) atomicOr(p.flags + p.err_index, big < INFINITY ? 2 : 4);
}

static int g_cus_bneck = 0;

// One bottleneck block, 64 mid and 256 output channels, stride 1 (detectron2/modeling/backbone/resnet.py:195-211):
// y = relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(x)))))))) + shortcut(x)).  x [N][H][W][ldx] (cin channels used), y [N][H][W][ldy].
// proj = 0: cin = 256, shortcut = x (ldx >= 256).  proj = 1: cin = 64, the projection shortcut's weights are the last 64 contraction
// columns of the third layer (kernels.pack_bottleneck over BottleneckBlock._fused_projection).  w: the stage images of
// kernels.pack_bottleneck ((cin/32) x 8 KB, 18 x 8 KB, 8 or 16 x 8 KB); s1..t3: epilogue scales (x row factors) and shifts, 64 / 64 /
// 256 entries, never NULL.  |x|, |t1| or |t2| > 4094 (or non-finite) raises bit 1 / 2 of the launch's range word.
extern "C" int lvc_bottleneck_nhwc_f16s1(const float* x, int ldx, float* y, int ldy, int N, int H, int W, int cin, int proj,
                                          const unsigned short* w, const float* s1, const float* t1, const float* s2, const float* t2,
                                          const float* s3, const float* t3, void* workspace, void* stream) {
  LVC_CHECK_ARG(x && y && w && s1 && t1 && s2 && t2 && s3 && t3 && workspace, "null pointer");
  LVC_CHECK_ARG(N > 0 && H > 0 && W > 0, "bad dimension");
  LVC_CHECK_ARG((cin == 256 && !proj) || (cin == 64 && proj), "cin must be 256 (identity block) or 64 (projection block)");
  LVC_CHECK_ARG(ldx >= cin && ldy >= 256 && (ldx & 3) == 0 && (ldy & 3) == 0, "row strides");
  LVC_CHECK_ARG((long long)N * H * W * ldx * 4 < 0xFFF00000ll && (long long)N * H * W * ldy * 4 < 0xFFF00000ll,
                "tensors must stay below 4 GiB (32-bit buffer offsets)");
  LVC_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)s1 & 15) == 0 &&
                    ((uintptr_t)t1 & 15) == 0 && ((uintptr_t)s2 & 15) == 0 && ((uintptr_t)t2 & 15) == 0 && ((uintptr_t)s3 & 15) == 0 &&
                    ((uintptr_t)t3 & 15) == 0, "pointers must be 16-byte aligned");
  BneckArgs a;
  a.x = x; a.y = y; a.w = w; a.s1 = s1; a.t1 = t1; a.s2 = s2; a.t2 = t2; a.s3 = s3; a.t3 = t3;
  a.N = N; a.H = H; a.W = W; a.ldx = ldx; a.ldy = ldy;
  a.tiles_x = lvc_cdiv(W, BN_TW); a.tiles_y = lvc_cdiv(H, BN_TH);
  a.ntiles = N * a.tiles_x * a.tiles_y;
  a.flags = (int*)((char*)workspace + (size_t)LVC_MAX_WORKERS * 256 * 128 * 4);
  a.err_index = LVC_MAX_WORKERS + lvc_range_slot();
  if (g_cus_bneck == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    g_cus_bneck = cus;
  }
#ifdef BN_DIAG_WGS1
  int grid = g_cus_bneck;
#else
  int grid = g_cus_bneck;
#endif
#ifdef BN_DIAG_GRID
  grid = BN_DIAG_GRID;
#endif
  if (grid > a.ntiles) grid = a.ntiles;
  hipStream_t st = (hipStream_t)stream;
#ifdef BN_DIAG_OCC
  {
    int nb0 = -1, nb1 = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb0, conv_bneck_kernel<256, false>, BN_NW * 64, 0);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb1, conv_bneck_kernel<64, true>, BN_NW * 64, 0);
    fprintf(stderr, "conv_bneck occupancy (blocks/CU): identity %d projection %d\n", nb0, nb1);
  }
#endif
  if (proj) hipLaunchKernelGGL((conv_bneck_kernel<64, true>), dim3(grid), dim3(BN_NW * 64), 0, st, a);
  else hipLaunchKernelGGL((conv_bneck_kernel<256, false>), dim3(grid), dim3(BN_NW * 64), 0, st, a);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}
