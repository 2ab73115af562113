// stem_pool_h2.hip -- the fused BasicStem kernel of stem_pool.hip with the two-way fp16 operand split of
// conv3x3_halo_h2.hip (two planes, three MFMAs per block, main + cross accumulators; operands beyond fp16's range --
// here the mean-subtracted pixels, |x| < 256 -- would raise bit 1 of the caller's error word).
// Original header: BasicStem in one kernel: conv 7x7 / stride 2 / pad 3 (3 -> 64 channels) -> FrozenBN -> ReLU ->
// max-pool 3x3 / stride 2 / pad 1 (reference detectron2/modeling/backbone/resnet.py:588-592), on the bf16 matrix cores
// with the 3-way operand split of conv_bf16x3.hip.
//
// Why fused: unfused, the stem writes its 8 x 400 x 672 x 64 fp32 output (550 MB) only for the pooling kernel to read
// it back and keep a quarter; and on the generic fp32 kernel the 224-deep, 64-wide product ran at 59 TF/s (0.67 ms +
// 0.13 ms per batch).  Here one workgroup owns a patch of 3 x 17 POOLED pixels = 7 x 35 = 245 conv pixels (the 3x3/s2
// pooling windows of neighbouring patches overlap by one conv row / column, so ~20 % of the conv pixels are computed
// twice -- cheaper than the round trip through HBM):
//   * the 19 x 76 input window (NHWC4: 16 B per pixel) is loaded once, split into three bf16 planes in LDS; the k index
//     of the stem GEMM is (filter row r, 8 input pixels x 4 channel slots): for conv pixel (cy, cx) and filter row r the
//     32 k values are CONTIGUOUS in the window row 2cy + r starting at column 2cx, so the A fragments are plain
//     ds_read_b128 at a per-lane base + r * row pitch (same idea as conv3x3_halo.hip, stride 2, 7 row taps);
//   * the weight planes of filter row r stream through a double buffer (12 KB per tap);
//   * 8 waves as 4 (M) x 2 (N), wave tile 64 x 32, 12 MFMAs per k16 step, one barrier per filter row;
//   * epilogue: scale/shift/ReLU into an LDS tile [245][64], then each thread reduces 3 x 3 windows of float4 and writes
//     the pooled pixels.  Conv pixels outside the conv map (the pad ring of the pooling) are skipped, as max_pool2d
//     pads with -inf.
// Packed weights: the mode-1 stem layout of lvc_conv2d_nhwc_f32 ([Kpad][7 * 32], k = r*32 + s*4 + c, s = 7 and c = 3
// zero), split into [3][Kpad][224] bf16 planes.
#include "common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define SP_PH 3                      // pooled rows per patch
#define SP_PW 17                     // pooled cols per patch
#define SP_CH (2 * SP_PH + 1)        // 7 conv rows
#define SP_CW (2 * SP_PW + 1)        // 35 conv cols
#define SP_M (SP_CH * SP_CW)         // 245 conv pixels (<= 256)
#define SP_IH (2 * SP_CH + 5)        // 19 input rows
#define SP_IW (2 * SP_CW + 6)        // 76 input cols
#define SP_IROW (SP_IW * 4)          // bf16 elements per window row (304 = 608 B, a multiple of 16 B)
#define SP_PLANE_IN (SP_IH * SP_IROW)
#define SP_LROW 40
#define SP_PLANE_B (64 * SP_LROW)
#define SP_CS 68                     // floats per row of the conv tile in LDS
#define SP_NT 512
#define SP_NLD 3                     // float4 window loads per thread (19 * 76 = 1444 <= 3 * 512)

struct StemArgsH {
  const float* x;            // [N, H, W, 4]
  const unsigned short* w;   // [2][Kpad][224] fp16 planes
  int* err;                  // error word (bit 1: operand beyond fp16's range) or NULL
  const float* scale;
  const float* shift;
  float* y;                  // [N, Hp, Wp, 64]
  float* y2;                 // optional second copy of the output: rows of ldy2 floats (NULL: none)
  int ldy2;
  int N, H, W, Ho, Wo, Hp, Wp, relu;
  int tiles_x, tiles_y, ntiles, nworkers;
  int x_bytes, w_plane_bytes;
};

__device__ __forceinline__ void split2s(float a, f16& h, f16& m) {
  h = (f16)a;
  m = (f16)((a - (float)h) * 2048.f);
}

__global__ __launch_bounds__(SP_NT, 4) void stem_pool_h2_kernel(StemArgsH p) {     // <= 128 registers: two workgroups (16 waves) per CU
  constexpr int IN_ELEMS = 2 * SP_PLANE_IN;        // 11,552 fp16 = 23,104 B
  constexpr int B_ELEMS = 2 * SP_PLANE_B;          // 5,120 fp16 = 10,240 B per buffer
  // The conv tile of the epilogue (66,640 B) lies OVER the input window and the weight buffers (43,584 B): they are dead while it is
  // pooled, and 66.6 KB per workgroup lets TWO workgroups share a CU -- one's barriers (a filter row has only 12 MFMAs per wave
  // between them) and epilogue run under the other's MFMAs.  The next tile's window and first weight row wait in registers until
  // the pooled rows are out.
  constexpr int CS_OFF = 0;
  constexpr int RING_BYTES = (IN_ELEMS + 2 * B_ELEMS) * 2;
  constexpr int CS_BYTES = SP_M * SP_CS * 4;
  constexpr int SMEM_BYTES = RING_BYTES > CS_BYTES ? RING_BYTES : CS_BYTES;   // 66,640
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[SMEM_BYTES];
  f16* sIn = reinterpret_cast<f16*>(smem_raw);
  f16* sB = sIn + IN_ELEMS;
  float* Cs = reinterpret_cast<float*>(smem_raw + CS_OFF);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;   // wave tile: 64 conv pixels x 32 channels
  const int fi = lane & 31, fh = lane >> 5;
  // weight staging: 256 pieces of 16 B per plane (64 rows x 4), two planes: thread t stages piece t % 256 of plane t / 256
  const int brid = (tid & 255) >> 2;
  const int b_row = (brid & 1) * 4 + ((brid >> 1) & 3) + (brid >> 3) * 8;
  const int b_q4 = tid & 3;
  const int b_pl0 = tid >> 8;
  const unsigned b_off = (unsigned)(b_row * 224 + b_q4 * 8) * 2u;

  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 2 * p.w_plane_bytes, 0x00020000);

  // A fragment base (bf16 elements inside a plane): conv pixel m = (cyl, cxl) -> window row 2*cyl (+ r), column 2*cxl
  int a_frag[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int m = wm * 64 + mi * 32 + fi;
    const int mm = m < SP_M ? m : 0;
    const int cyl = mm / SP_CW, cxl = mm - cyl * SP_CW;
    a_frag[mi] = 2 * cyl * SP_IROW + 8 * cxl + fh * 8;
  }
  const int b_frag = (wn * 32 + fi) * SP_LROW + fh * 8;

  // per-channel affine of this lane's accumulator column
  const int ocol = wn * 32 + fi;
  const float sc = p.scale ? p.scale[ocol] : 1.f;
  const float sh = p.shift ? p.shift[ocol] : 0.f;

  int range_err = 0;
  f32x4 wreg[SP_NLD];
  u32x4 breg[1], breg2;
  auto window_offsets = [&](int tile, unsigned* off) {
    const int tx = tile % p.tiles_x;
    const int t2 = tile / p.tiles_x;
    const int ty = t2 % p.tiles_y;
    const int img = t2 / p.tiles_y;
    const int iy0 = 2 * (2 * ty * SP_PH - 1) - 3, ix0 = 2 * (2 * tx * SP_PW - 1) - 3;
#pragma unroll
    for (int j = 0; j < SP_NLD; ++j) {
      const int idx = tid + SP_NT * j;
      const int iy = idx / SP_IW, ix = idx - iy * SP_IW;
      const int yy = iy0 + iy, xx = ix0 + ix;
      const bool ok = idx < SP_IH * SP_IW && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
      off[j] = ok ? (unsigned)(((img * p.H + yy) * p.W + xx) * 4) * 4u : 0x80000000u;
    }
  };
  auto load_window = [&](int tile) {
    unsigned off[SP_NLD];
    window_offsets(tile, off);
#pragma unroll
    for (int j = 0; j < SP_NLD; ++j)
      wreg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, off[j], 0, 0));
  };
  auto store_window = [&]() {
#pragma unroll
    for (int j = 0; j < SP_NLD; ++j) {
      const int idx = tid + SP_NT * j;
      if (idx < SP_IH * SP_IW) {
        f16x4 h, m;
        float big = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f16 hh, mm;
          split2s(wreg[j][e], hh, mm);
          h[e] = hh; m[e] = mm;
          big = fmaxf(big, fabsf(wreg[j][e]));
        }
        if (!(big <= 65504.f)) range_err = 1;
        const int o = idx * 4;   // (iy * SP_IW + ix) * 4
        *reinterpret_cast<f16x4*>(sIn + o) = h;
        *reinterpret_cast<f16x4*>(sIn + SP_PLANE_IN + o) = m;
      }
    }
  };
  int ld_r = 0;   // filter row of the next weight request (cycles 0..6 across tiles)
  auto load_B = [&]() {
    breg[0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                            wres, b_off + (unsigned)(b_pl0 * p.w_plane_bytes), ld_r * 64, 0));
    ld_r = ld_r == 6 ? 0 : ld_r + 1;
  };
  auto load_B2 = [&]() {     // the same request into the second register: the row behind the one still waiting in breg
    breg2 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                          wres, b_off + (unsigned)(b_pl0 * p.w_plane_bytes), ld_r * 64, 0));
    ld_r = ld_r == 6 ? 0 : ld_r + 1;
  };
  auto store_B = [&](int buf) {
    f16* sb = sB + buf * B_ELEMS;
    *reinterpret_cast<u32x4*>(sb + b_pl0 * SP_PLANE_B + b_row * SP_LROW + b_q4 * 8) = breg[0];
  };

  int tile = lvc_xcd_remap(blockIdx.x, p.nworkers);
  if (tile >= p.ntiles) return;
  load_window(tile);
  load_B();
  store_window();
  store_B(0);
  load_B();
  if (tile + p.nworkers < p.ntiles) load_window(tile + p.nworkers);
  __syncthreads();
  int cur = 0;

  for (; tile < p.ntiles; tile += p.nworkers) {
    f32x16 acc[2], accx[2];   // main (a1 b1) and cross (a1 b2 + a2 b1, weight 2^-11) accumulators
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[a][e] = 0.f; accx[a][e] = 0.f; }

    f16x8 fa[2][2], fb[2];
    auto read_frags = [&](int row_off, const f16* sb, int s2) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          fa[mi][pl] = *reinterpret_cast<const f16x8*>(sIn + pl * SP_PLANE_IN + a_frag[mi] + row_off + s2 * 16);
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        fb[pl] = *reinterpret_cast<const f16x8*>(sb + pl * SP_PLANE_B + b_frag + s2 * 16);
    };
    auto mfma_group = [&]() {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) accx[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi][0], fb[1], accx[mi], 0, 0, 0);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi][0], fb[0], acc[mi], 0, 0, 0);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) accx[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi][1], fb[0], accx[mi], 0, 0, 0);
    };
    for (int r = 0; r < 7; ++r) {
      const f16* sb = sB + cur * B_ELEMS;
      read_frags(r * SP_IROW, sb, 0);
      if (r < 6) store_B(cur ^ 1);      // the next TILE's first row stays in breg: its buffer is under the conv tile until the pooling is done
      mfma_group();
      __builtin_amdgcn_sched_barrier(0);
      read_frags(r * SP_IROW, sb, 1);
      if (r < 6) load_B(); else load_B2();
      mfma_group();
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      cur ^= 1;
    }

    // ---- conv tile -> LDS (affine + ReLU), next window -> LDS, then pooled output
    const int tx = tile % p.tiles_x;
    const int t2 = tile / p.tiles_x;
    const int ty = t2 % p.tiles_y;
    const int img = t2 / p.tiles_y;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        if (row < SP_M) {
          float v = (acc[mi][e] + accx[mi][e] * (1.f / 2048.f)) * sc + sh;
          if (p.relu) v = v > 0.f ? v : 0.f;
          Cs[row * SP_CS + ocol] = v;
        }
      }
    const bool more = tile + p.nworkers < p.ntiles;
    __syncthreads();
    const int cy0 = 2 * ty * SP_PH - 1, cx0 = 2 * tx * SP_PW - 1;   // conv coordinates of the patch origin
    for (int o = tid; o < SP_PH * SP_PW * 16; o += SP_NT) {
      const int c4 = o & 15, pp = o >> 4;
      const int pi = pp / SP_PW, pj = pp - pi * SP_PW;
      const int py = ty * SP_PH + pi, px = tx * SP_PW + pj;
      if (py < p.Hp && px < p.Wp) {
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int cyl = 2 * pi + dy;
          const int cy = cy0 + cyl;
          if (cy < 0 || cy >= p.Ho) continue;
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const int cxl = 2 * pj + dx;
            const int cx = cx0 + cxl;
            if (cx < 0 || cx >= p.Wo) continue;
            const f32x4 v = *reinterpret_cast<const f32x4*>(Cs + (cyl * SP_CW + cxl) * SP_CS + c4 * 4);
            best[0] = v[0] > best[0] ? v[0] : best[0];
            best[1] = v[1] > best[1] ? v[1] : best[1];
            best[2] = v[2] > best[2] ? v[2] : best[2];
            best[3] = v[3] > best[3] ? v[3] : best[3];
          }
        }
        *reinterpret_cast<f32x4*>(p.y + ((size_t)(img * p.Hp + py) * p.Wp + px) * 64 + c4 * 4) = best;
        if (p.y2) *reinterpret_cast<f32x4*>(p.y2 + ((size_t)(img * p.Hp + py) * p.Wp + px) * p.ldy2 + c4 * 4) = best;
      }
    }
    __syncthreads();             // the conv tile has been read: its LDS is the window and the weight buffers again
    if (more) store_window();
    store_B(cur);                // filter row 0 of the next tile, then row 1 becomes the pending one
    breg[0] = breg2;
    if (tile + 2 * p.nworkers < p.ntiles) load_window(tile + 2 * p.nworkers);
    __syncthreads();
  }
  if (range_err && p.err) atomicOr(p.err, 2);
}

static int g_cus_stem_h = 0;

// x [N,H,W,4] fp32 (NHWC4), w_split [2][Kpad][224] fp16 planes (split2h) of the mode-1 packed stem weights (Kpad >= 64 rows,
// plane stride = Kpad * 224), scale/shift [64] or NULL, y [N,Hp,Wp,64] with Ho = (H - 1) / 2 + 1, Hp = (Ho - 1) / 2 + 1.
extern "C" int lvc_stem_conv_pool_nhwc4_f16x2(const float* x, const unsigned short* w_split, const float* scale,
                                              const float* shift, float* y, int N, int H, int W, int Kpad, int relu,
                                              int* d_error_word, float* y2, int ldy2, void* stream) {
  LVC_CHECK_ARG(x && w_split && y, "null pointer");
  LVC_CHECK_ARG(!y2 || (ldy2 >= 64 && ldy2 % 4 == 0 && ((uintptr_t)y2 & 15) == 0), "bad second output");
  LVC_CHECK_ARG(N > 0 && H > 0 && W > 0 && Kpad >= 64, "bad shape");
  LVC_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)w_split & 15) == 0 && ((uintptr_t)y & 15) == 0,
                "pointers must be 16-byte aligned");
  StemArgsH a;
  a.x = x; a.w = w_split; a.scale = scale; a.shift = shift; a.y = y; a.y2 = y2; a.ldy2 = ldy2;
  a.N = N; a.H = H; a.W = W; a.relu = relu; a.err = d_error_word;
  a.Ho = (H + 6 - 7) / 2 + 1; a.Wo = (W + 6 - 7) / 2 + 1;
  a.Hp = (a.Ho + 2 - 3) / 2 + 1; a.Wp = (a.Wo + 2 - 3) / 2 + 1;
  a.tiles_x = lvc_cdiv(a.Wp, SP_PW); a.tiles_y = lvc_cdiv(a.Hp, SP_PH);
  const long long nt = (long long)N * a.tiles_x * a.tiles_y;
  const long long xb = (long long)N * H * W * 16, wb = (long long)Kpad * 224 * 2;
  LVC_CHECK_ARG(nt < (1ll << 31) && xb < (1ll << 31), "input too large");
  a.ntiles = (int)nt; a.x_bytes = (int)xb; a.w_plane_bytes = (int)wb;
  if (g_cus_stem_h == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    g_cus_stem_h = cus;
  }
  a.nworkers = a.ntiles < 2 * g_cus_stem_h ? a.ntiles : 2 * g_cus_stem_h;   // two 66.6 KB workgroups per CU, tiles dealt round-robin
  hipLaunchKernelGGL(stem_pool_h2_kernel, dim3(a.nworkers), dim3(SP_NT), 0, (hipStream_t)stream, a);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}
