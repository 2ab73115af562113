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
// Work decomposition.  A workgroup (4 waves) owns an output tile of TH x 32 pixels of one image (TH = 4 at 64 mid channels, 2 at
// 128: the tile's t1 planes must leave room for TWO workgroups per CU, so that one tile's 3x3 phase, which touches no HBM, runs
// under the other's loads and stores).  Everything is computed TRANSPOSED as in conv_pw_chain.hip: out^T [channels x pixels] =
// W [channels x k] . act^T [k x pixels]; the MFMA's A operand is a weight fragment from LDS, its B operand the activations, an
// accumulator lane owns ONE pixel (lane % 32) and sixteen channels of a 32-channel block (c = 8 i + 4 (lane / 32) + {0..3}).
//   phase 1 (conv1): a wave takes groups of 32 halo pixels; x goes from HBM straight into the B-operand layout (two dwordx4 per k16
//     step and lane), is split in registers, contracted against W1 streamed through the LDS ring in 32-channel chunks; the epilogue
//     (FrozenBN, ReLU, zero outside the image = conv2's padding, x 2^4, fp16 split) writes t1 into LDS as
//     [k16 step][plane][k half][halo pixel][16 B]: conflict-free for these writes and for phase 2's tap-shifted reads.
//   phase 2 (conv2): a wave owns one output row (two at TH = 2 ... see below) for ALL mid channels; per ring stage = (k16 step, dx)
//     it reads the three t1 rows at that column shift and the 3 dy x {channel blocks} x 2 planes weight fragments.
//   phase 3 (conv3): per ring stage = 32 output channels; B operand = t2 from the wave's own registers; the residual is read in the
//     accumulator layout (the rows were fetched by phase 1 moments ago: L2 / memory-side cache hits) and y is stored from it.
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
constexpr int BN_TH = 4, BN_TW = 32, BN_HW = BN_TW + 2, BN_HH = BN_TH + 2, BN_HPIX = BN_HW * BN_HH;      // 204 halo pixels
constexpr int BN_NG1 = (BN_HPIX + 31) / 32;                                                                // 7 groups of 32
constexpr int BN_T1_SUB = BN_HPIX * 16;                 // bytes of one (k16 step, plane, k half) sub-plane of t1
constexpr int BN_T1_BYTES = 16 * BN_T1_SUB;             // 4 steps x 2 planes x 2 halves = 52 224
constexpr int BN_SLOT = 12288, BN_RING = 2 * BN_SLOT;
constexpr int BN_TAB = (4 * 64 + 2 * 256) * 4;

template <int CIN, bool PROJ>
__global__ __launch_bounds__(256, 2) void conv_bneck_kernel(BneckArgs p) {
  constexpr int NS1 = CIN / 32, NS2 = 12, NS3 = PROJ ? 16 : 8, NST = NS1 + NS2 + NS3;
  static_assert(NST % 2 == 0, "the ring slot of a stage is its parity inside a tile");
  static_assert(!PROJ || CIN == 64, "projection blocks: 64 input channels");
  constexpr int OFF2 = NS1 * 8192, OFF3 = OFF2 + NS2 * 12288;      // byte offsets of the phases' stage images
  __shared__ __attribute__((aligned(1024))) unsigned char smem[BN_RING + BN_T1_BYTES + BN_TAB];
  unsigned char* const t1s = smem + BN_RING;
  float* tab_s1 = reinterpret_cast<float*>(smem + BN_RING + BN_T1_BYTES);
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

  for (int i = tid; i < 64; i += 256) {
    tab_s1[i] = p.s1[i]; tab_t1[i] = p.t1[i];
    tab_s2[i] = p.s2[i]; tab_t2[i] = p.t2[i];
  }
  for (int i = tid; i < 256; i += 256) {
    tab_s3[i] = p.s3[i]; tab_t3[i] = p.t3[i];
  }
  __syncthreads();

  // weights by LDS-DMA through a buffer resource: ONE vector register (lane * 16) addresses every piece, the stage offset is scalar
  const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, OFF3 + NS3 * 8192, 0x00020000);
  const int wlane = lane * 16;
  auto dma = [&](int ts) {      // stage ts of a tile (a constant at every call site) into its ring slot
    const int off = ts < NS1 ? ts * 8192 : ts < NS1 + NS2 ? OFF2 + (ts - NS1) * 12288 : OFF3 + (ts - NS1 - NS2) * 8192;
    const bool wide = ts >= NS1 && ts < NS1 + NS2;
    const int so = off + wave * 1024;
    unsigned char* dst = smem + (ts & 1) * BN_SLOT + wave * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, (lds_ptr_t)dst, 16, wlane, so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, (lds_ptr_t)(dst + 4096), 16, wlane, so + 4096, 0, 0);
    if (wide) __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, (lds_ptr_t)(dst + 8192), 16, wlane, so + 8192, 0, 0);
  };

  const unsigned xbytes = (unsigned)p.N * (unsigned)p.H * (unsigned)p.W * (unsigned)p.ldx * 4u;
  const unsigned ybytes = (unsigned)p.N * (unsigned)p.H * (unsigned)p.W * (unsigned)p.ldy * 4u;
  const u32x4 xres = make_rsrc(p.x, xbytes);
  const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, ybytes, 0x00020000);

  float big = 0.f;
  const int ftab = 4 * fh;
  // phase 2's per-lane base inside t1: (row wave, column fi) of the halo tile, k half fh
  const unsigned char* const zb = t1s + fh * BN_T1_SUB + (wave * BN_HW + fi) * 16;
  const int tpi = p.tiles_x * p.tiles_y;

  dma(0);
#pragma unroll 1
  for (int tl = within; tl < tcnt; tl += wgx) {
    const int tile = tbase + tl;
    const int n = tile / tpi, trem = tile - n * tpi;
    const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    const int y0 = ty * BN_TH, x0 = tx * BN_TW;

    // ---------------------------------------------------------------- phase 1: t1 = relu(bn1(W1 x)) on the tile + halo
    unsigned xo[2];
    float msk[2];
    int hl[2];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      const int h = 32 * (wave + 4 * gi) + fi;
      const int hy = h / BN_HW, hx = h - hy * BN_HW;
      const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
      const bool ok = h < BN_HPIX && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
      xo[gi] = ok ? (unsigned)((n * p.H + yy) * p.W + xx) * (unsigned)p.ldx * 4u + fh * 16u : BN_MARK;
      msk[gi] = ok ? 1.f : 0.f;
      hl[gi] = h;
    }
    const bool g1 = wave < BN_NG1 - 4;      // this wave's second group exists
    f32x4 xr[2][2][4];                      // [stage parity][group][2 k16 steps x 2 runs of four channels]
    auto load_x = [&](int par, int c) {
#pragma unroll
      for (int gi = 0; gi < 2; ++gi) {
        xr[par][gi][0] = load_untracked<0>(xres, xo[gi], 128u * c);
        xr[par][gi][1] = load_untracked<32>(xres, xo[gi], 128u * c);
        xr[par][gi][2] = load_untracked<64>(xres, xo[gi], 128u * c);
        xr[par][gi][3] = load_untracked<96>(xres, xo[gi], 128u * c);
      }
    };
    load_x(0, 0);
    f32x16 acc1[2][2];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1[gi][cb][e] = 0.f;

    static_for<0, NS1>([&](auto c_) {
      constexpr int c = decltype(c_)::value;
      constexpr int PAR = c & 1;
      wait_vm<0>();
#pragma unroll
      for (int gi = 0; gi < 2; ++gi) tie4(xr[PAR][gi][0], xr[PAR][gi][1], xr[PAR][gi][2], xr[PAR][gi][3]);
      __builtin_amdgcn_s_barrier();
      dma(c + 1);
      if (c + 1 < NS1) load_x(PAR ^ 1, c + 1);
      const unsigned char* S = smem + (c & 1) * BN_SLOT + lane * 16;
      f16x8 wf[2][2][2];      // [k16 step][channel block][plane]
#pragma unroll
      for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) wf[sp][cb][pl] = *reinterpret_cast<const f16x8*>(S + (((sp * 2 + cb) * 2 + pl) << 10));
#pragma unroll
      for (int gi = 0; gi < 2; ++gi) {
        if (gi == 0 || g1) {
          f16x8 zh[2], zl[2];
#pragma unroll
          for (int sp = 0; sp < 2; ++sp) {
            split8(xr[PAR][gi][2 * sp], xr[PAR][gi][2 * sp + 1], zh[sp], zl[sp]);
            track_abs(big, xr[PAR][gi][2 * sp][0], xr[PAR][gi][2 * sp][1]);
            track_abs(big, xr[PAR][gi][2 * sp][2], xr[PAR][gi][2 * sp][3]);
            track_abs(big, xr[PAR][gi][2 * sp + 1][0], xr[PAR][gi][2 * sp + 1][1]);
            track_abs(big, xr[PAR][gi][2 * sp + 1][2], xr[PAR][gi][2 * sp + 1][3]);
          }
#pragma unroll
          for (int sp = 0; sp < 2; ++sp)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) { MFMA3(acc1[gi][cb], wf[sp][cb][0], wf[sp][cb][1], zh[sp], zl[sp]); }
        }
      }
    });
    // epilogue of phase 1: FrozenBN + ReLU, zero outside the image (conv2's padding), split, into LDS
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      if ((gi == 0 || g1) && hl[gi] < BN_HPIX) {
        unsigned char* dst = t1s + fh * BN_T1_SUB + hl[gi] * 16;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          f16x8 yh[2], yl[2];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(tab_s1 + 32 * cb + 8 * i + ftab);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(tab_t1 + 32 * cb + 8 * i + ftab);
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
            *reinterpret_cast<f16x8*>(dst + ((s * 2 + 0) * 2) * BN_T1_SUB) = yh[sp];
            *reinterpret_cast<f16x8*>(dst + ((s * 2 + 1) * 2) * BN_T1_SUB) = yl[sp];
          }
        }
      }
    }

    // ---------------------------------------------------------------- phase 2: t2 = relu(bn2(conv3x3(t1))), one output row per wave
    const int oy = y0 + wave, ox = x0 + fi;
    const bool ook = oy < p.H && ox < p.W;
    const unsigned opix = (unsigned)((n * p.H + oy) * p.W + ox);
    const unsigned yo = ook ? opix * (unsigned)p.ldy * 4u + fh * 16u : BN_MARK;
    const unsigned ro = ook ? opix * (unsigned)p.ldx * 4u + fh * 16u : BN_MARK;
    f32x4 rb[2][4];      // identity blocks: the residual rows of output block j in rb[j & 1]; projection blocks: x of the own pixel
    f32x4 xi[8];
    f32x16 acc2[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc2[cb][e] = 0.f;

    static_for<0, NS2>([&](auto st_) {
      constexpr int st = decltype(st_)::value;
      constexpr int s = st / 3, dx = st % 3, ts = NS1 + st;
      wait_vm<0>();
      if (st == 0) wait_lgkm0();
      __builtin_amdgcn_s_barrier();
      dma(ts + 1);
      if (st == NS2 - 1) {
        if (PROJ) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            xi[2 * q] = load_untracked<0>(xres, ro, 64u * q);
            xi[2 * q + 1] = load_untracked<32>(xres, ro, 64u * q);
          }
        } else {
          rb[0][0] = load_untracked<0>(xres, ro, 0u);
          rb[0][1] = load_untracked<32>(xres, ro, 0u);
          rb[0][2] = load_untracked<64>(xres, ro, 0u);
          rb[0][3] = load_untracked<96>(xres, ro, 0u);
        }
      }
      const unsigned char* S = smem + (ts & 1) * BN_SLOT + lane * 16;
      f16x8 zh[3], zl[3];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        zh[dy] = *reinterpret_cast<const f16x8*>(zb + ((s * 2 + 0) * 2) * BN_T1_SUB + (dy * BN_HW + dx) * 16);
        zl[dy] = *reinterpret_cast<const f16x8*>(zb + ((s * 2 + 1) * 2) * BN_T1_SUB + (dy * BN_HW + dx) * 16);
      }
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const f16x8 wh = *reinterpret_cast<const f16x8*>(S + (((dy * 2 + cb) * 2 + 0) << 10));
          const f16x8 wl = *reinterpret_cast<const f16x8*>(S + (((dy * 2 + cb) * 2 + 1) << 10));
          MFMA3(acc2[cb], wh, wl, zh[dy], zl[dy]);
        }
    });
    // epilogue of phase 2: the split accumulators are phase 3's B operand
    f16x8 t2h[4], t2l[4];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(tab_s2 + 32 * cb + 8 * i + ftab);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(tab_t2 + 32 * cb + 8 * i + ftab);
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

    // ---------------------------------------------------------------- phase 3: y = relu(bn3(W3 t2) + shortcut), 32 channels a stage
    f16x8 xh[4], xl[4];
    f32x16 acc3;
    static_for<0, NS3>([&](auto st_) {
      constexpr int st = decltype(st_)::value;
      constexpr int j = PROJ ? st / 2 : st, half = PROJ ? st % 2 : 0, ts = NS1 + NS2 + st;
      constexpr int PAR = j & 1;
      wait_vm<0>();
      if (PROJ) {
        if (st == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            asm volatile("" : "+v"(xi[2 * q]), "+v"(xi[2 * q + 1]));
            split8(xi[2 * q], xi[2 * q + 1], xh[q], xl[q]);
          }
        }
      } else {
        tie4(rb[PAR][0], rb[PAR][1], rb[PAR][2], rb[PAR][3]);
      }
      __builtin_amdgcn_s_barrier();
      dma(ts + 1 < NST ? ts + 1 : 0);
      if (!PROJ && j + 1 < 8) {
        rb[PAR ^ 1][0] = load_untracked<0>(xres, ro, 128u * (j + 1));
        rb[PAR ^ 1][1] = load_untracked<32>(xres, ro, 128u * (j + 1));
        rb[PAR ^ 1][2] = load_untracked<64>(xres, ro, 128u * (j + 1));
        rb[PAR ^ 1][3] = load_untracked<96>(xres, ro, 128u * (j + 1));
      }
      const unsigned char* S = smem + (ts & 1) * BN_SLOT + lane * 16;
      if (half == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc3[e] = 0.f;
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const f16x8 wh = *reinterpret_cast<const f16x8*>(S + ((s * 2 + 0) << 10));
        const f16x8 wl = *reinterpret_cast<const f16x8*>(S + ((s * 2 + 1) << 10));
        if (half == 0) { MFMA3(acc3, wh, wl, t2h[s], t2l[s]); }
        else { MFMA3(acc3, wh, wl, xh[s], xl[s]); }
      }
      if (!PROJ || half == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(tab_s3 + 32 * j + 8 * i + ftab);
          const f32x4 sh = *reinterpret_cast<const f32x4*>(tab_t3 + 32 * j + 8 * i + ftab);
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float o = acc3[4 * i + e] * sc[e] + sh[e];
            if (!PROJ) o += rb[PAR][i][e];
            v[e] = fmaxf(o, 0.f);
          }
          store_b128(v, yres, yo + 128u * j + 32u * i);
        }
      }
    });
  }
  wait_vm<0>();
  if (!(big <= ACT_MAX)) atomicOr(p.flags + p.err_index, big < INFINITY ? 2 : 4);
}

static int g_cus_bneck = 0;

// One bottleneck block, 64 mid and 256 output channels, stride 1 (detectron2/modeling/backbone/resnet.py:195-211):
// y = relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(x)))))))) + shortcut(x)).  x [N][H][W][ldx] (cin channels used), y [N][H][W][ldy].
// proj = 0: cin = 256, shortcut = x (ldx >= 256).  proj = 1: cin = 64, the projection shortcut's weights are the last 64 contraction
// columns of the third layer (kernels.pack_bottleneck over BottleneckBlock._fused_projection).  w: the stage images of
// kernels.pack_bottleneck ((cin/32) x 8 KB, 12 x 12 KB, 8 or 16 x 8 KB); s1..t3: epilogue scales (x row factors) and shifts, 64 / 64 /
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
  int grid = g_cus_bneck * 2;
  if (grid > a.ntiles) grid = a.ntiles;
  hipStream_t st = (hipStream_t)stream;
  if (proj) hipLaunchKernelGGL((conv_bneck_kernel<64, true>), dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((conv_bneck_kernel<256, false>), dim3(grid), dim3(256), 0, st, a);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}
