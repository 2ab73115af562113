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
//   k = (r, s, c) with c fastest      -- a BK=32 chunk of k is 128 contiguous bytes of one input pixel
// Arithmetic: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain; 64 FLOP/clk/SIMD = 157 TF chip peak).
// The 1e-3 box/score parity bar of the path needs true fp32; bf16 MFMA would not hold it.
//
// Tile: 128(M) x 128(N) x 32(K) per 256-thread workgroup; 4 waves as 2x2, each wave a 64x64 sub-tile
// = 2x2 MFMA 32x32 accumulators (64 acc VGPRs).  LDS: A and B tiles, k-contiguous rows padded to 36
// floats (144 B) so that the 16-lane groups of ds_read_b128 hit 16 distinct 16-B slots; double buffered
// (73.7 KB -> 2 workgroups per CU).  Global->LDS staging goes through registers (prefetch chunk k+1
// before the MFMA block of chunk k, ds_write after it), one barrier per chunk.
//
// Lane/fragment map for 32x32x2 (A: lane l holds A[i=l&31][k=l>>5]; B: B[k=l>>5][j=l&31]):
// lane (i,h) ds_read_b128's 4 consecutive k (= 8*kk + 4*h + t, t=0..3) from row i; MFMA step t then
// contracts k in {8kk+t, 8kk+4+t}; A and B use the same permutation so every k is used exactly once.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define BM 128
#define BN 128
#define BK 32
#define LDS_STRIDE 36  // floats per tile row (BK + 4 pad)

struct ConvArgs {
  const float* x;      // input  [N,H,W,C]   (C = physical channel count, multiple of 4)
  const float* w;      // packed weights [Kpad][Kg]  (Kpad multiple of BN, Kg multiple of BK)
  const float* scale;  // per out-channel multiplier or nullptr (=1)
  const float* shift;  // per out-channel addend or nullptr (=0)
  const float* res;    // residual tensor or nullptr
  float* y;            // output [M][ldy]
  int N, H, W, C;
  int K;               // real out channels
  int R, S, stride, pad;
  int Ho, Wo, M;
  int Kg;              // padded gemm-K
  int relu;
  int res_mode;        // 0 none | 1 same shape [M][ldr] | 2 nearest-x2-upsampled: res is [N,Ho/2,Wo/2,ldr]
  int ldy, ldr;
  int tiles_n;
  int mode;            // 0: chunk -> (r,s,c0) ; 1: "row mode" (stem): chunk -> r, 32 floats = 8 pixels x 4 ch
};

template <int MODE>
__global__ __launch_bounds__(256, 2) void conv_igemm_f32_kernel(ConvArgs p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDS_STRIDE];
  float* As = smem;                          // [2][BM][LDS_STRIDE]
  float* Bs = smem + 2 * BM * LDS_STRIDE;    // [2][BN][LDS_STRIDE]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int nwg = gridDim.x;
  const int wg = lvc_xcd_remap(blockIdx.x, nwg);
  const int tile_n = wg % p.tiles_n;
  const int tile_m = wg / p.tiles_n;
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;

  // ---- staging assignment: thread loads float4 q of rows (tid>>3)+32*j, j=0..3 (A and B alike)
  const int q = tid & 7;
  const int row0 = tid >> 3;

  int a_base[4];   // element offset of pixel (n, base_h, base_w) relative to x, or <0 when row >= M
  int a_bh[4], a_bw[4];
  bool a_ok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int m = m0 + row0 + 32 * j;
    a_ok[j] = m < p.M;
    int mm = a_ok[j] ? m : 0;
    int n = mm / (p.Ho * p.Wo);
    int rem = mm - n * (p.Ho * p.Wo);
    int ho = rem / p.Wo;
    int wo = rem - ho * p.Wo;
    a_bh[j] = ho * p.stride - p.pad;
    a_bw[j] = wo * p.stride - p.pad;
    a_base[j] = n * p.H * p.W;  // pixel index base of image n
  }
  const float* wrow[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wrow[j] = p.w + (size_t)(n0 + row0 + 32 * j) * p.Kg + q * 4;

  const int nk = p.Kg / BK;
  const int cpc = (MODE == 0) ? (p.C / BK) : 1;  // chunks per (r,s)

  f32x4 areg[4], breg[4];

  auto load_chunk = [&](int kc) {
    int r, s, c0;
    if (MODE == 0) {
      int rs = kc / cpc;
      c0 = (kc - rs * cpc) * BK;
      r = rs / p.S;
      s = rs - r * p.S;
    } else {
      r = kc; s = 0; c0 = 0;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int hi = a_bh[j] + r;
      int wi = a_bw[j] + s + (MODE == 1 ? q : 0);
      bool ok = a_ok[j] && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) {
        size_t off = (size_t)(a_base[j] + hi * p.W + wi) * p.C + c0 + (MODE == 0 ? q * 4 : 0);
        v = *reinterpret_cast<const f32x4*>(p.x + off);
      }
      areg[j] = v;
      breg[j] = *reinterpret_cast<const f32x4*>(wrow[j] + (size_t)kc * BK);
    }
  };
  auto store_chunk = [&](int buf) {
    float* a = As + buf * BM * LDS_STRIDE;
    float* b = Bs + buf * BN * LDS_STRIDE;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *reinterpret_cast<f32x4*>(a + (row0 + 32 * j) * LDS_STRIDE + q * 4) = areg[j];
      *reinterpret_cast<f32x4*>(b + (row0 + 32 * j) * LDS_STRIDE + q * 4) = breg[j];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  const int fi = lane & 31, fh = lane >> 5;
  const int a_frag_off = (wm * 64 + fi) * LDS_STRIDE + fh * 4;
  const int b_frag_off = (wn * 64 + fi) * LDS_STRIDE + fh * 4;

  load_chunk(0);
  store_chunk(0);
  __syncthreads();

  int cur = 0;
  for (int kc = 0; kc < nk; ++kc) {
    if (kc + 1 < nk) load_chunk(kc + 1);
    const float* a = As + cur * BM * LDS_STRIDE + a_frag_off;
    const float* b = Bs + cur * BN * LDS_STRIDE + b_frag_off;
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      f32x4 af0 = *reinterpret_cast<const f32x4*>(a + kk * 8);
      f32x4 af1 = *reinterpret_cast<const f32x4*>(a + 32 * LDS_STRIDE + kk * 8);
      f32x4 bf0 = *reinterpret_cast<const f32x4*>(b + kk * 8);
      f32x4 bf1 = *reinterpret_cast<const f32x4*>(b + 32 * LDS_STRIDE + kk * 8);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af0[t], bf0[t], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af0[t], bf1[t], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af1[t], bf0[t], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af1[t], bf1[t], acc[1][1], 0, 0, 0);
      }
    }
    if (kc + 1 < nk) store_chunk(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: y = act(acc*scale + shift + residual)
  // C/D map of 32x32x2: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    int col = n0 + wn * 64 + ni * 32 + fi;
    bool col_ok = col < p.K;
    float sc = (p.scale && col_ok) ? p.scale[col] : 1.f;
    float sh = (p.shift && col_ok) ? p.shift[col] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        int row = m0 + wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        if (row < p.M && col_ok) {
          float v = acc[mi][ni][e] * sc + sh;
          if (p.res_mode == 1) {
            v += p.res[(size_t)row * p.ldr + col];
          } else if (p.res_mode == 2) {
            int n = row / (p.Ho * p.Wo);
            int rem = row - n * (p.Ho * p.Wo);
            int ho = rem / p.Wo;
            int wo = rem - ho * p.Wo;
            size_t ro = ((size_t)(n * (p.Ho >> 1) + (ho >> 1)) * (p.Wo >> 1) + (wo >> 1));
            v += p.res[ro * p.ldr + col];
          }
          if (p.relu) v = v > 0.f ? v : 0.f;
          p.y[(size_t)row * p.ldy + col] = v;
        }
      }
    }
  }
}

// C ABI -- see include/lvc_amd.h for the contract of each argument.
extern "C" int lvc_conv2d_nhwc_f32(const float* x, const float* w_packed, const float* scale,
                                   const float* shift, const float* residual, float* y, int N, int H,
                                   int W, int C, int K, int R, int S, int stride, int pad, int Kg,
                                   int relu, int res_mode, int ldy, int ldr, int mode, void* stream) {
  LVC_CHECK_ARG(x && w_packed && y, "null pointer");
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
  a.tiles_n = lvc_cdiv(K, BN);
  a.mode = mode;
  int tiles_m = lvc_cdiv(a.M, BM);
  dim3 grid(tiles_m * a.tiles_n), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0)
    hipLaunchKernelGGL(conv_igemm_f32_kernel<0>, grid, block, 0, st, a);
  else
    hipLaunchKernelGGL(conv_igemm_f32_kernel<1>, grid, block, 0, st, a);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}
