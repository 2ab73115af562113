// conv_wgrad.hip -- backward of the NHWC convolution w.r.t. its weights, plus the small data-movement kernels the
// backward of the trunk needs (stride-2 scatter, 2x2 down-sum, column sums).
//
// Reference: the reference trains through ATen's conv2d backward (cudnn / mkldnn wgrad): BottleneckBlock
// (detectron2/modeling/backbone/resnet.py:195-211), FPN (backbone/fpn.py:109-144), StandardRPNHead
// (proposal_generator/rpn.py:120-139) when their parameters require grad -- `cascade_ubbr_R_50_FPN_base.yaml`
// (`FREEZE_AT 2`), `faster_rcnn_R_50_FPN_base.yaml`, and the RPN / box head of the `ft_all` fine-tune yaml.
//
// GEMM view:   dW[k][tap][c] = scale[k] * sum_m dY[m][k] * X[pix(m, tap)][c]
//   m   = output pixel (n, oy, ox): the CONTRACTION runs over pixels, so both operands are read exactly as they lie
//         in HBM (a pixel's channels are contiguous in NHWC) and arrive in LDS already "k-major" for the MFMA:
//         v_mfma_f32_32x32x2_f32 takes A[i][kk] / B[kk][j] with kk = lane/32, i.e. lanes 0-31 read 32 consecutive
//         channels of pixel 2*ks, lanes 32-63 of pixel 2*ks+1 -- no transpose anywhere.
//   tap = (r, s); X's pixel for tap is (oy*stride + r - pad, ox*stride + s - pad), zero outside the map.
//   scale[k] = the FrozenBatchNorm2d affine scale that follows the conv (batch_norm.py:45-65): y = conv(x)*scale+shift,
//         so dL/dW = scale[k] * (dL/dy (*) x).  NULL = 1.
// Arithmetic: exact fp32 FMA chain (157 TF chip peak); gradients span too many binades for the fp16 split used by
// the forward kernels.
//
// Work decomposition: one 256-thread workgroup = one 128(k) x 128(c) tile of one tap over a contiguous range of
// 32-pixel chunks; grid.y splits the pixel range so that the launch has >= ~4 workgroups per CU whatever the
// layer's tile count is (a 1x1 128->512 layer has 4 tiles and 33 600 pixels).  Partial sums are combined with fp32
// atomic adds into the zero-initialised gradient (order not deterministic, as in the reference's GPU path).
// 4 waves as 2x2, wave tile 64x64 = 2x2 MFMA tiles (64 accumulator VGPRs).  LDS rows are 128 floats with the two
// 32-float halves of every 64 swapped on odd rows, so lanes 32-63 (odd pixel) hit the other 32 banks.
#include "common.h"
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgradParams {
  const float* x;
  const float* dy;
  const float* scale;
  float* dw;
  int N, H, W, C, Ho, Wo, K, R, S, stride, pad, lddy;
  int M;                 // N*Ho*Wo
  int chunks_per_split;  // 32-pixel chunks per grid.y slice
  int k_tiles, c_tiles, tiles;
};

__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
  __shared__ float sA[2][32][128];  // dY [pixel][k]
  __shared__ float sB[2][32][128];  // X  [pixel][c]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // workgroups of one pixel range (all tiles, all taps) are consecutive logical ids -> same XCD, same L2: they walk
  // the same dY / X rows at the same pace, so the slab is fetched from HBM about once per XCD
  const int lid = lvc_xcd_remap(blockIdx.x, gridDim.x);
  int t = lid % p.tiles;
  const int split = lid / p.tiles;
  const int ct = t % p.c_tiles; t /= p.c_tiles;
  const int kt = t % p.k_tiles; t /= p.k_tiles;
  const int tap = t, r = tap / p.S, s = tap % p.S;
  const int k0 = kt * 128, c0 = ct * 128;
  const int nchunks = (p.M + 31) >> 5;
  const int chunk0 = split * p.chunks_per_split;
  int chunk1 = chunk0 + p.chunks_per_split;
  if (chunk1 > nchunks) chunk1 = nchunks;
  if (chunk0 >= chunk1) return;

  const int lrow = tid >> 5;        // rows lrow + 8*i
  const int lcol = (tid & 31) * 4;  // first of 4 consecutive channels
  const bool k_ok = k0 + lcol < p.K, c_ok = c0 + lcol < p.C;
  f32x4 ra[4], rb[4];
  auto load = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = chunk * 32 + lrow + 8 * i;
      f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
      if (m < p.M) {
        if (k_ok) a = *reinterpret_cast<const f32x4*>(p.dy + (size_t)m * p.lddy + k0 + lcol);
        const int ox = m % p.Wo, q = m / p.Wo, oy = q % p.Ho, n = q / p.Ho;
        const int iy = oy * p.stride + r - p.pad, ix = ox * p.stride + s - p.pad;
        if (c_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
          b = *reinterpret_cast<const f32x4*>(p.x + (((size_t)n * p.H + iy) * p.W + ix) * p.C + c0 + lcol);
      }
      ra[i] = a;
      rb[i] = b;
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = lrow + 8 * i;
      const int col = lcol ^ ((row & 1) << 5);
      *reinterpret_cast<f32x4*>(&sA[buf][row][col]) = ra[i];
      *reinterpret_cast<f32x4*>(&sB[buf][row][col]) = rb[i];
    }
  };

  const int wk = (wave >> 1) * 64, wc = (wave & 1) * 64;
  const int half = lane >> 5, l31 = lane & 31, sw = half << 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[mi][ni][j] = 0.f;

  load(chunk0);
  store(0);
  __syncthreads();
  int cur = 0;
  for (int chunk = chunk0; chunk < chunk1; ++chunk) {
    const bool more = chunk + 1 < chunk1;
    if (more) load(chunk + 1);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int row = 2 * ks + half;
      const float a0 = sA[cur][row][(wk + l31) ^ sw], a1 = sA[cur][row][(wk + 32 + l31) ^ sw];
      const float b0 = sB[cur][row][(wc + l31) ^ sw], b1 = sB[cur][row][(wc + 32 + l31) ^ sw];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (more) store(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // D[i][j]: j = lane & 31, i = 8*(e/4) + 4*(lane/32) + e%4
  const int RS = p.R * p.S;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int k = k0 + wk + mi * 32 + 8 * (e >> 2) + 4 * half + (e & 3);
      if (k >= p.K) continue;
      const float sc = p.scale ? p.scale[k] : 1.f;
      float* row = p.dw + ((size_t)k * RS + tap) * p.C;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int c = c0 + wc + ni * 32 + l31;
        if (c < p.C) unsafeAtomicAdd(row + c, acc[mi][ni][e] * sc);
      }
    }
}

// dw [K][R][S][C] (the packed "KRSC" order of the forward kernels' weights); zeroed here.
// x [N,H,W,C], dy [N,Ho,Wo,K] with row pitch lddy >= K floats.  K, C, lddy multiples of 4.
extern "C" int lvc_conv_wgrad_nhwc(const float* x, const float* dy, const float* scale, float* dw, int N, int H, int W,
                                   int C, int K, int R, int S, int stride, int pad, int lddy, void* stream) {
  LVC_CHECK_ARG(x && dy && dw, "null pointer");
  LVC_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && K > 0 && R > 0 && S > 0 && stride > 0 && pad >= 0, "bad shape");
  LVC_CHECK_ARG(C % 4 == 0 && K % 4 == 0 && lddy % 4 == 0 && lddy >= K, "C, K and lddy must be multiples of 4");
  LVC_CHECK_ARG((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dw) & 15) == 0, "pointers must be 16-byte aligned");
  const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
  LVC_CHECK_ARG(Ho > 0 && Wo > 0, "empty output map");
  const long long M64 = (long long)N * Ho * Wo;
  LVC_CHECK_ARG(M64 < (1ll << 31) - 64, "too many output pixels");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(dw, 0, (size_t)K * R * S * C * sizeof(float), st) != hipSuccess) {
    lvc_set_error("%s: hipMemsetAsync failed", __func__);
    return LVC_ERR_HIP;
  }
  WgradParams p;
  p.x = x; p.dy = dy; p.scale = scale; p.dw = dw;
  p.N = N; p.H = H; p.W = W; p.C = C; p.Ho = Ho; p.Wo = Wo; p.K = K; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
  p.lddy = lddy; p.M = (int)M64;
  p.k_tiles = lvc_cdiv(K, 128); p.c_tiles = lvc_cdiv(C, 128);
  const int tiles = p.k_tiles * p.c_tiles * R * S;
  p.tiles = tiles;
  const int nchunks = lvc_cdiv(p.M, 32);
  // every workgroup ends with 128x128 atomic adds (64 KB), the traffic of two 32-pixel chunks: a pixel range must be
  // long enough to amortise it, even if that leaves fewer workgroups than the chip has slots
  constexpr int min_chunks = 16;
  int splits = lvc_cdiv(1024, tiles);                 // ~4 workgroups per CU
  const int max_splits = lvc_cdiv(nchunks, min_chunks);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  p.chunks_per_split = lvc_cdiv(nchunks, splits);
  splits = lvc_cdiv(nchunks, p.chunks_per_split);
  hipLaunchKernelGGL(conv_wgrad_kernel, dim3(tiles * splits), dim3(256), 0, st, p);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// ------------------------------------------------------------------------------------------------------------
// y[n, 2i, 2j, :] = x[n, i, j, :], every other pixel of y = 0.  Backward of a stride-2 1x1 convolution's input
// sampling (conv1 / shortcut of the first block of res3..res5, resnet.py:117-160 with STRIDE_IN_1X1) and of
// LastLevelMaxPool (fpn.py:165-177: max_pool2d(k=1, s=2)).  x [N,Hs,Ws,C] -> y [N,H,W,C], Hs = (H-1)/2+1.
__global__ __launch_bounds__(256) void scatter_stride2_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y, int H,
                                                              int W, int Hs, int Ws, int C4, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C4);
  long long q = i / C4;
  const int ix = (int)(q % W); q /= W;
  const int iy = (int)(q % H);
  const int n = (int)(q / H);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (!((iy | ix) & 1)) v = x[(((long long)n * Hs + (iy >> 1)) * Ws + (ix >> 1)) * C4 + c];
  y[i] = v;
}

extern "C" int lvc_scatter_stride2_nhwc(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  LVC_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "bad arguments");
  const int Hs = (H - 1) / 2 + 1, Ws = (W - 1) / 2 + 1;
  const long long total = (long long)N * H * W * (C / 4);
  hipLaunchKernelGGL(scatter_stride2_kernel, dim3((unsigned)lvc_cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const f32x4*>(x), reinterpret_cast<f32x4*>(y), H, W, Hs, Ws, C / 4, total);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// y[n, i, j, 0..C) = x[n, 2i, 2j, :] with rows of ldy floats in y (ldy >= C: y may be the tail channels of a wider buffer).  The
// sampling of a stride-2 1x1 convolution as a copy: the block input of res3.0 / res4.0 / res5.0 next to conv2's output, so that
// conv3 and the projection shortcut run as ONE pointwise GEMM over [conv2 output | sampled input] (BottleneckBlock.can_fuse_projection).
__global__ __launch_bounds__(256) void subsample2_kernel(const f32x4* __restrict__ x, float* __restrict__ y, int H, int W, int Hs,
                                                         int Ws, int C4, int ldy, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C4);
  long long q = i / C4;
  const int ix = (int)(q % Ws); q /= Ws;
  const int iy = (int)(q % Hs);
  const int n = (int)(q / Hs);
  const f32x4 v = x[(((long long)n * H + 2 * iy) * W + 2 * ix) * C4 + c];
  *reinterpret_cast<f32x4*>(y + (((long long)n * Hs + iy) * Ws + ix) * ldy + c * 4) = v;
}

extern "C" int lvc_subsample2_nhwc(const float* x, float* y, int N, int H, int W, int C, int ldy, void* stream) {
  LVC_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "bad arguments");
  const int ld = ldy > 0 ? ldy : C;
  LVC_CHECK_ARG(ld >= C && ld % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0, "rows must be 16-byte aligned");
  const int Hs = (H - 1) / 2 + 1, Ws = (W - 1) / 2 + 1;
  const long long total = (long long)N * Hs * Ws * (C / 4);
  hipLaunchKernelGGL(subsample2_kernel, dim3((unsigned)lvc_cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const f32x4*>(x), y, H, W, Hs, Ws, C / 4, ld, total);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// y[n, i, j, :] = sum of the 2x2 block x[n, 2i..2i+1, 2j..2j+1, :].  Backward of the nearest x2 upsample of the FPN
// top-down path (fpn.py:131-133).  x [N,2Hs,2Ws,C] -> y [N,Hs,Ws,C].
__global__ __launch_bounds__(256) void downsum2x2_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y, int Hs, int Ws,
                                                         int C4, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C4);
  long long q = i / C4;
  const int jx = (int)(q % Ws); q /= Ws;
  const int jy = (int)(q % Hs);
  const int n = (int)(q / Hs);
  const long long W = 2ll * Ws;
  const f32x4* r0 = x + (((long long)n * 2 * Hs + 2 * jy) * W + 2 * jx) * C4 + c;
  const f32x4* r1 = r0 + W * C4;
  y[i] = (r0[0] + r0[C4]) + (r1[0] + r1[C4]);
}

extern "C" int lvc_downsum2x2_nhwc(const float* x, float* y, int N, int Hs, int Ws, int C, void* stream) {
  LVC_CHECK_ARG(x && y && N > 0 && Hs > 0 && Ws > 0 && C > 0 && C % 4 == 0, "bad arguments");
  const long long total = (long long)N * Hs * Ws * (C / 4);
  hipLaunchKernelGGL(downsum2x2_kernel, dim3((unsigned)lvc_cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const f32x4*>(x), reinterpret_cast<f32x4*>(y), Hs, Ws, C / 4, total);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// out[c] = sum over rows of x[r][c] (bias gradients of the FPN / RPN convolutions: 10^5 rows x 256 columns).
// One workgroup sums a 256-row slab for 64 columns (4 row phases x 64 lanes), then one atomic per column.
__global__ __launch_bounds__(256) void colsum_slab_kernel(const float* __restrict__ x, int M, int Ncol, int ldx,
                                                          float* __restrict__ out) {
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
  const int r0 = blockIdx.y * 256;
  int r1 = r0 + 256;
  if (r1 > M) r1 = M;
  float sacc = 0.f;
  if (c < Ncol)
    for (int r = r0 + ph; r < r1; r += 4) sacc += x[(size_t)r * ldx + c];
  part[ph][threadIdx.x & 63] = sacc;
  __syncthreads();
  if (ph == 0 && c < Ncol) unsafeAtomicAdd(out + c, (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]));
}

// float4 form for 16-byte aligned rows: a wave reads 1 KiB of one row per request (64 lanes x 4 columns), the four waves
// take rows r0 + w + 4 i of a 512-row slab; the 64-column form above moved 256-byte pieces (2.3 TB/s on the p2 tensors).
__global__ __launch_bounds__(256) void colsum_slab4_kernel(const float* __restrict__ x, int M, int Ncol, int ldx,
                                                           float* __restrict__ out) {
  __shared__ f32x4 part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  const int r0 = blockIdx.y * 512;
  int r1 = r0 + 512;
  if (r1 > M) r1 = M;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
  if (c < Ncol) {
    int r = r0 + w;
    for (; r + 4 < r1; r += 8) {
      a0 += *reinterpret_cast<const f32x4*>(x + (size_t)r * ldx + c);
      a1 += *reinterpret_cast<const f32x4*>(x + (size_t)(r + 4) * ldx + c);
    }
    if (r < r1) a0 += *reinterpret_cast<const f32x4*>(x + (size_t)r * ldx + c);
  }
  part[w][lane] = a0 + a1;
  __syncthreads();
  if (w == 0 && c < Ncol) {
    const f32x4 t = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
#pragma unroll
    for (int e = 0; e < 4; ++e) unsafeAtomicAdd(out + c + e, t[e]);
  }
}

extern "C" int lvc_colsum_atomic(const float* x, int M, int Ncol, int ldx, float* out, void* stream) {
  LVC_CHECK_ARG(M >= 0 && Ncol > 0 && ldx >= Ncol && x && out, "bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(out, 0, (size_t)Ncol * sizeof(float), st) != hipSuccess) {
    lvc_set_error("%s: hipMemsetAsync failed", __func__);
    return LVC_ERR_HIP;
  }
  if (M == 0) return LVC_OK;
  if (Ncol % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)x & 15) == 0)
    hipLaunchKernelGGL(colsum_slab4_kernel, dim3(lvc_cdiv(Ncol, 256), lvc_cdiv(M, 512)), dim3(256), 0, st, x, M, Ncol, ldx, out);
  else
    hipLaunchKernelGGL(colsum_slab_kernel, dim3(lvc_cdiv(Ncol, 64), lvc_cdiv(M, 256)), dim3(256), 0, st, x, M, Ncol, ldx, out);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Two-way fp16 split form of the weight gradient (used when lvc_amd.solver.LossScaler has scaled the upstream gradients
// into fp16's range): the same pixel-contraction GEMM on v_mfma_f32_32x32x16_f16, three MFMAs per 32x32x16 block into a
// main (a1 b1) and a cross (a1 b2 + a2 b1, weight 2^-11) accumulator as in the forward kernels -- 833 TF/s effective peak
// instead of the 157 TF/s of the fp32 MFMA form.
// The obstacle is the operand layout: a 16-bit MFMA wants 8 consecutive k values (= PIXELS here) per lane, but NHWC keeps a
// pixel's channels contiguous and pixels 2*C bytes apart.  gfx950's LDS transpose read solves it without a transposing
// store: ds_read_b64_tr_b16 hands lane i of a 16-lane group column i of a 4 x 16 tile whose rows the group's lanes address
// (lane t supplies row t/4, columns 4(t%4)..+3; scripts/micro/tr_read.hip prints the mapping; the kernel uses the compiler
// builtin __builtin_amdgcn_ds_read_tr16_b64_v4f16, so waits and scheduling are the compiler's).  With the LDS image kept
// [pixel][channel] exactly as it arrives from HBM, two such reads per lane yield pixels {0..3, 4..7} (lanes 0-31) and
// {8..11, 12..15} (lanes 32-63) of channels lane % 32 -- the 32x32x16 operand.  Both operands (dY^T and X) are read this way.
typedef _Float16 wg_f16;
typedef _Float16 wg_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 wg_f16x8 __attribute__((ext_vector_type(8)));
#define WG_PITCH 128                    // fp16 elements per LDS row, unpadded: the 32-byte channel groups of row r are XOR-
                                        // swizzled by 2 * (r & 3), so the 4 rows x 2 channel groups a 32-lane half of a transpose
                                        // read touches fall on 8 different 8-bank groups (the padded 288-byte pitch had the two
                                        // groups of a half collide: SQ_LDS_BANK_CONFLICT = 79 % of the LDS cycles)
#define WG_PLANE (32 * WG_PITCH)        // one operand plane of a 32-pixel chunk

// the compiler's own builtin for the transpose read: it tracks the read like any LDS load (waits, scheduling under MFMAs)
typedef __fp16 wg_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) wg_fp16x4 wg_lds_fp16x4;
typedef __attribute__((address_space(3))) char wg_lds_char;
__device__ __forceinline__ wg_f16x4 wg_tr_read(const wg_lds_char* base, unsigned byte_off) {
  return __builtin_bit_cast(wg_f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((wg_lds_fp16x4*)(base + byte_off)));
}

// BUF: both tensors are smaller than 2 GiB, so the loader uses buffer loads whose out-of-range offset returns zeros: no
// divergent branch around each of the eight requests of a chunk and 32-bit address arithmetic.
template <bool BUF>
__global__ __launch_bounds__(256, 2) void conv_wgrad_f16x2_kernel(const WgradParams p, int* __restrict__ err_word,
                                                                  unsigned x_bytes, unsigned dy_bytes) {
  __shared__ __attribute__((aligned(16))) wg_f16 lds[2 * 4 * WG_PLANE];   // [buffer][dY1, dY2, X1, X2][32][WG_PITCH]
  const wg_lds_char* lds3 = (const wg_lds_char*)lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lid = lvc_xcd_remap(blockIdx.x, gridDim.x);
  int t = lid % p.tiles;
  const int split = lid / p.tiles;
  const int ct = t % p.c_tiles; t /= p.c_tiles;
  const int kt = t % p.k_tiles; t /= p.k_tiles;
  const int tap = t, r = tap / p.S, s = tap % p.S;
  const int k0 = kt * 128, c0 = ct * 128;
  const int nchunks = (p.M + 31) >> 5;
  const int chunk0 = split * p.chunks_per_split;
  int chunk1 = chunk0 + p.chunks_per_split;
  if (chunk1 > nchunks) chunk1 = nchunks;
  if (chunk0 >= chunk1) return;

  // loader: thread = (pixel row of the chunk, 16-byte slot): the eight lanes of a row fetch 128 contiguous bytes per
  // request and four requests cover the row's 128 channels, so a thread follows ONE pixel (one set of coordinates and
  // bounds per chunk; with four pixels per thread the address arithmetic was half of the kernel's VALU instructions)
  const int lrow = tid >> 3;
  const int lq = (tid & 7) * 4;            // channel of slot j: lq + 32 j
  f32x4 ra[4], rb[4];
  int range_err = 0;
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, BUF ? x_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t dyres = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, BUF ? dy_bytes : 0, 0x00020000);
  // byte offsets of the four slots inside a pixel; a slot or a pixel out of range is 2^31, and the saturating sum of the
  // two stays beyond the buffer (both tensors are < 2 GiB), so the load returns zeros
  unsigned k_off[4], c_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    k_off[j] = k0 + lq + 32 * j < p.K ? (unsigned)(k0 + lq + 32 * j) * 4u : 0x80000000u;
    c_off[j] = c0 + lq + 32 * j < p.C ? (unsigned)(c0 + lq + 32 * j) * 4u : 0x80000000u;
  }
  int pm = chunk0 * 32 + lrow;             // this thread's pixel, advanced by 32 per chunk
  int px = pm % p.Wo, py = (pm / p.Wo) % p.Ho, pn = pm / (p.Wo * p.Ho);
  auto load = [&](int) {
    const int iy = py * p.stride + r - p.pad, ix = px * p.stride + s - p.pad;
    const bool okm = pm < p.M;
    const bool okx = okm && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    if constexpr (BUF) {
      const unsigned ab = okm ? (unsigned)(pm * p.lddy) * 4u : 0x80000000u;
      const unsigned bb = okx ? (unsigned)(((pn * p.H + iy) * p.W + ix) * p.C) * 4u : 0x80000000u;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dyres, __builtin_elementwise_add_sat(ab, k_off[j]), 0, 0));
        rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, __builtin_elementwise_add_sat(bb, c_off[j]), 0, 0));
      }
    } else {
      const float* ap = p.dy + (size_t)pm * p.lddy;
      const float* bp = p.x + (((size_t)pn * p.H + iy) * p.W + ix) * p.C;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        ra[j] = (okm && !(k_off[j] >> 31)) ? *reinterpret_cast<const f32x4*>(ap + (k_off[j] >> 2)) : z;
        rb[j] = (okx && !(c_off[j] >> 31)) ? *reinterpret_cast<const f32x4*>(bp + (c_off[j] >> 2)) : z;
      }
    }
    pm += 32;
    if (p.Wo >= 32) {                      // the next chunk's pixel: at most one wrap
      px += 32;
      if (px >= p.Wo) {
        px -= p.Wo;
        if (++py == p.Ho) { py = 0; ++pn; }
      }
    } else {                               // narrow maps (and the Linear case, Wo = 1): divide
      px = pm % p.Wo;
      const int q2 = pm / p.Wo;
      py = q2 % p.Ho;
      pn = q2 / p.Ho;
    }
  };
  auto split4 = [&](const f32x4& v, wg_f16x4& h, wg_f16x4& m) {
    float big = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const wg_f16 hh = (wg_f16)v[e];
      h[e] = hh;
      m[e] = (wg_f16)((v[e] - (float)hh) * 2048.f);
      big = fmaxf(big, fabsf(v[e]));
    }
    if (!(big <= 65504.f)) range_err = 1;
  };
  auto store = [&](int buf) {
    wg_f16* base = lds + buf * 4 * WG_PLANE + lrow * WG_PITCH;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = (lq + 32 * j) ^ ((lrow & 3) << 5);
      wg_f16x4 h, m;
      split4(ra[j], h, m);
      *reinterpret_cast<wg_f16x4*>(base + o) = h;
      *reinterpret_cast<wg_f16x4*>(base + WG_PLANE + o) = m;
      split4(rb[j], h, m);
      *reinterpret_cast<wg_f16x4*>(base + 2 * WG_PLANE + o) = h;
      *reinterpret_cast<wg_f16x4*>(base + 3 * WG_PLANE + o) = m;
    }
  };

  const int wk = (wave >> 1) * 64, wc = (wave & 1) * 64;
  // transpose-read address of this lane inside a 32-channel block at pixel row 0: group g = lane / 16, t = lane % 16
  const int g = lane >> 4, tt = lane & 15;
  // row of the tile this lane addresses: 8 * (g / 2) + tt / 4 (+4, +16 for the other reads: row & 3 == tt / 4 throughout)
  const unsigned tr_row = (unsigned)((8 * (g >> 1) + (tt >> 2)) * WG_PITCH * 2);
  unsigned tr_a[2], tr_b[2];   // byte offsets of the lane's swizzled column inside a row, per 32-channel block
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    tr_a[blk] = (unsigned)(((wk + blk * 32 + 16 * (g & 1) + 4 * (tt & 3)) ^ ((tt >> 2) << 5)) * 2);
    tr_b[blk] = (unsigned)(((wc + blk * 32 + 16 * (g & 1) + 4 * (tt & 3)) ^ ((tt >> 2) << 5)) * 2);
  }
  f32x16 acc[2][2], accx[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int j = 0; j < 16; ++j) { acc[mi][ni][j] = 0.f; accx[mi][ni][j] = 0.f; }

  load(chunk0);
  store(0);
  __syncthreads();
  int cur = 0;
  for (int chunk = chunk0; chunk < chunk1; ++chunk) {
    const bool more = chunk + 1 < chunk1;
    if (more) load(chunk + 1);
    const unsigned bufb = (unsigned)(cur * 4 * WG_PLANE * 2);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const unsigned rowb = bufb + (unsigned)(ks * 16 * WG_PITCH * 2) + tr_row;
      wg_f16x8 fa[2][2], fb[2][2];   // [32-channel block][plane]
      auto cat = [](const wg_f16x4& lo, const wg_f16x4& hi) { return wg_f16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; };
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          const unsigned aa = rowb + (unsigned)(pl * WG_PLANE * 2) + tr_a[blk];
          const unsigned bb = rowb + (unsigned)((2 + pl) * WG_PLANE * 2) + tr_b[blk];
          fa[blk][pl] = cat(wg_tr_read(lds3, aa), wg_tr_read(lds3, aa + 4 * WG_PITCH * 2));
          fb[blk][pl] = cat(wg_tr_read(lds3, bb), wg_tr_read(lds3, bb + 4 * WG_PITCH * 2));
        }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          accx[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi][0], fb[ni][1], accx[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi][0], fb[ni][0], acc[mi][ni], 0, 0, 0);
          accx[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi][1], fb[ni][0], accx[mi][ni], 0, 0, 0);
        }
    }
    if (more) store(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  const int RS = p.R * p.S;
  const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int k = k0 + wk + mi * 32 + 8 * (e >> 2) + 4 * half + (e & 3);
      if (k >= p.K) continue;
      const float sc = p.scale ? p.scale[k] : 1.f;
      float* row = p.dw + ((size_t)k * RS + tap) * p.C;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int c = c0 + wc + ni * 32 + l31;
        if (c < p.C) unsafeAtomicAdd(row + c, (acc[mi][ni][e] + accx[mi][ni][e] * (1.f / 2048.f)) * sc);
      }
    }
  if (range_err && err_word) atomicOr(err_word, 2);
}

// lvc_conv_wgrad_nhwc on the two-way fp16 split kernel.  dy (and x) must lie inside fp16's range (|v| <= 65504; gradients
// scaled by lvc_amd.solver.LossScaler); a value outside raises bit 1 (value 2) of *err_word (the conv error word).
extern "C" int lvc_conv_wgrad_nhwc_f16x2(const float* x, const float* dy, const float* scale, float* dw, int N, int H, int W,
                                         int C, int K, int R, int S, int stride, int pad, int lddy, int* err_word,
                                         void* stream) {
  LVC_CHECK_ARG(x && dy && dw, "null pointer");
  LVC_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && K > 0 && R > 0 && S > 0 && stride > 0 && pad >= 0, "bad shape");
  LVC_CHECK_ARG(C % 4 == 0 && K % 4 == 0 && lddy % 4 == 0 && lddy >= K, "C, K and lddy must be multiples of 4");
  LVC_CHECK_ARG((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dw) & 15) == 0, "pointers must be 16-byte aligned");
  const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
  LVC_CHECK_ARG(Ho > 0 && Wo > 0, "empty output map");
  const long long M64 = (long long)N * Ho * Wo;
  LVC_CHECK_ARG(M64 < (1ll << 31) - 64, "too many output pixels");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(dw, 0, (size_t)K * R * S * C * sizeof(float), st) != hipSuccess) {
    lvc_set_error("%s: hipMemsetAsync failed", __func__);
    return LVC_ERR_HIP;
  }
  WgradParams p;
  p.x = x; p.dy = dy; p.scale = scale; p.dw = dw;
  p.N = N; p.H = H; p.W = W; p.C = C; p.Ho = Ho; p.Wo = Wo; p.K = K; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
  p.lddy = lddy; p.M = (int)M64;
  p.k_tiles = lvc_cdiv(K, 128); p.c_tiles = lvc_cdiv(C, 128);
  const int tiles = p.k_tiles * p.c_tiles * R * S;
  p.tiles = tiles;
  const int nchunks = lvc_cdiv(p.M, 32);
  // two workgroups fit a CU (250 VGPRs): four rounds of them (measured over the layer set on one box: 1024 -> 8.88 ms,
  // 1536 -> 8.65, 2048 -> 8.54, 2560 -> 8.47, 3072 -> 8.40; the whole training step does not resolve 2048 from 3072)
  constexpr int target_h = 2048;
  int splits = lvc_cdiv(target_h, tiles);
  const int max_splits = lvc_cdiv(nchunks, 16);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  p.chunks_per_split = lvc_cdiv(nchunks, splits);
  splits = lvc_cdiv(nchunks, p.chunks_per_split);
  const long long xb = (long long)N * H * W * C * 4, dyb = M64 * lddy * 4;
  if (xb < (1ll << 31) && dyb < (1ll << 31))
    hipLaunchKernelGGL(conv_wgrad_f16x2_kernel<true>, dim3(tiles * splits), dim3(256), 0, st, p, err_word, (unsigned)xb,
                       (unsigned)dyb);
  else
    hipLaunchKernelGGL(conv_wgrad_f16x2_kernel<false>, dim3(tiles * splits), dim3(256), 0, st, p, err_word, 0u, 0u);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Three-way bf16 split form of the weight gradient: the default (unscaled) training path.  bf16 keeps fp32's exponent,
// so raw gradients (1e-8 .. 1e-2) need no loss scale; a = hi + mid + lo exactly, and six of the nine plane products
// (everything down to 2^-24 relative) go into ONE fp32 accumulator on v_mfma_f32_32x32x16_bf16 -- 417 TF/s effective peak
// against the 157 TF/s of the fp32 MFMA form, with the same transpose-read operand path as the fp16 kernel above.
// Chunks are 16 pixels (one k16 step): six planes of a 32-pixel chunk double-buffered would be 96 KB, one workgroup per
// CU; with 16 pixels a workgroup holds 48 KB and several stay resident.
typedef __bf16 wg_bf16;
typedef __bf16 wg_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
#define WB_PLANE (16 * WG_PITCH)        // one operand plane of a 16-pixel chunk (bf16 elements)

__device__ __forceinline__ void wg_split3(float a, wg_bf16& h, wg_bf16& m, wg_bf16& l) {
  h = (wg_bf16)a;
  const float r1 = a - (float)h;
  m = (wg_bf16)r1;
  const float r2 = r1 - (float)m;
  l = (wg_bf16)r2;
}

template <bool BUF>
__device__ __forceinline__ void wgrad_bf16x3_body(const WgradParams& p, const int lid, unsigned x_bytes, unsigned dy_bytes, wg_bf16* lds) {
  const wg_lds_char* lds3 = (const wg_lds_char*)lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int t = lid % p.tiles;
  const int split = lid / p.tiles;
  const int ct = t % p.c_tiles; t /= p.c_tiles;
  const int kt = t % p.k_tiles; t /= p.k_tiles;
  const int tap = t, r = tap / p.S, s = tap % p.S;
  const int k0 = kt * 128, c0 = ct * 128;
  const int nchunks = (p.M + 15) >> 4;                     // chunks_per_split counts 16-pixel chunks here
  const int chunk0 = split * p.chunks_per_split;
  int chunk1 = chunk0 + p.chunks_per_split;
  if (chunk1 > nchunks) chunk1 = nchunks;
  if (chunk0 >= chunk1) return;

  // loader: a 16-lane group follows one pixel of the chunk (256 contiguous bytes per request, two requests per tensor);
  // the four groups of a wave take rows 4w + {0, 2, 1, 3}: the two rows of a 32-lane store pass then differ in bit 1 of
  // the row, so the XOR swizzle below puts their 128-byte pieces on different halves of the 64 banks
  const int lg = lane >> 4;
  const int prow = 4 * wave + (((lg & 1) << 1) | (lg >> 1));
  const int lq = (lane & 15) * 4;          // channel of slot j: lq + 64 j
  f32x4 ra[2], rb[2];
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, BUF ? x_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t dyres = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, BUF ? dy_bytes : 0, 0x00020000);
  unsigned k_off[2], c_off[2];             // 2^31 = out of range (see the fp16 kernel's loader)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    k_off[j] = k0 + lq + 64 * j < p.K ? (unsigned)(k0 + lq + 64 * j) * 4u : 0x80000000u;
    c_off[j] = c0 + lq + 64 * j < p.C ? (unsigned)(c0 + lq + 64 * j) * 4u : 0x80000000u;
  }
  int pm = chunk0 * 16 + prow;
  int px = pm % p.Wo, py = (pm / p.Wo) % p.Ho, pn = pm / (p.Wo * p.Ho);
  auto load = [&]() {
    const int iy = py * p.stride + r - p.pad, ix = px * p.stride + s - p.pad;
    const bool okm = pm < p.M;
    const bool okx = okm && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    if constexpr (BUF) {
      const unsigned ab = okm ? (unsigned)(pm * p.lddy) * 4u : 0x80000000u;
      const unsigned bb = okx ? (unsigned)(((pn * p.H + iy) * p.W + ix) * p.C) * 4u : 0x80000000u;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dyres, __builtin_elementwise_add_sat(ab, k_off[j]), 0, 0));
        rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, __builtin_elementwise_add_sat(bb, c_off[j]), 0, 0));
      }
    } else {
      const float* ap = p.dy + (size_t)pm * p.lddy;
      const float* bp = p.x + (((size_t)pn * p.H + iy) * p.W + ix) * p.C;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        ra[j] = (okm && !(k_off[j] >> 31)) ? *reinterpret_cast<const f32x4*>(ap + (k_off[j] >> 2)) : z;
        rb[j] = (okx && !(c_off[j] >> 31)) ? *reinterpret_cast<const f32x4*>(bp + (c_off[j] >> 2)) : z;
      }
    }
    pm += 16;
    if (p.Wo >= 16) {
      px += 16;
      if (px >= p.Wo) {
        px -= p.Wo;
        if (++py == p.Ho) { py = 0; ++pn; }
      }
    } else {
      px = pm % p.Wo;
      const int q2 = pm / p.Wo;
      py = q2 % p.Ho;
      pn = q2 / p.Ho;
    }
  };
  auto store = [&](int buf) {
    wg_bf16* base = lds + buf * 6 * WB_PLANE + prow * WG_PITCH;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int o = (lq + 64 * j) ^ ((prow & 3) << 5);
      wg_bf16x4 h, m, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) { wg_bf16 a, b, c; wg_split3(ra[j][e], a, b, c); h[e] = a; m[e] = b; l[e] = c; }
      *reinterpret_cast<wg_bf16x4*>(base + o) = h;
      *reinterpret_cast<wg_bf16x4*>(base + WB_PLANE + o) = m;
      *reinterpret_cast<wg_bf16x4*>(base + 2 * WB_PLANE + o) = l;
#pragma unroll
      for (int e = 0; e < 4; ++e) { wg_bf16 a, b, c; wg_split3(rb[j][e], a, b, c); h[e] = a; m[e] = b; l[e] = c; }
      *reinterpret_cast<wg_bf16x4*>(base + 3 * WB_PLANE + o) = h;
      *reinterpret_cast<wg_bf16x4*>(base + 4 * WB_PLANE + o) = m;
      *reinterpret_cast<wg_bf16x4*>(base + 5 * WB_PLANE + o) = l;
    }
  };

  const int wk = (wave >> 1) * 64, wc = (wave & 1) * 64;
  const int g = lane >> 4, tt = lane & 15;
  const unsigned tr_row = (unsigned)((8 * (g >> 1) + (tt >> 2)) * WG_PITCH * 2);
  unsigned tr_a[2], tr_b[2];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    tr_a[blk] = (unsigned)(((wk + blk * 32 + 16 * (g & 1) + 4 * (tt & 3)) ^ ((tt >> 2) << 5)) * 2);
    tr_b[blk] = (unsigned)(((wc + blk * 32 + 16 * (g & 1) + 4 * (tt & 3)) ^ ((tt >> 2) << 5)) * 2);
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[mi][ni][j] = 0.f;

  load();
  store(0);
  __syncthreads();
  int cur = 0;
  for (int chunk = chunk0; chunk < chunk1; ++chunk) {
    const bool more = chunk + 1 < chunk1;
    if (more) load();
    const unsigned rowb = (unsigned)(cur * 6 * WB_PLANE * 2) + tr_row;
    wg_bf16x8 fa[2][3], fb[2][3];   // [32-channel block][plane]
    auto rd = [&](unsigned off) {
      const wg_bf16x4 lo = __builtin_bit_cast(wg_bf16x4, wg_tr_read(lds3, off));
      const wg_bf16x4 hi = __builtin_bit_cast(wg_bf16x4, wg_tr_read(lds3, off + 4 * WG_PITCH * 2));
      return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        fa[blk][pl] = rd(rowb + (unsigned)(pl * WB_PLANE * 2) + tr_a[blk]);
        fb[blk][pl] = rd(rowb + (unsigned)((3 + pl) * WB_PLANE * 2) + tr_b[blk]);
      }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        f32x16 c = acc[mi][ni];     // smallest terms first
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][0], fb[ni][2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][2], fb[ni][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][1], fb[ni][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][0], fb[ni][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][1], fb[ni][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][0], fb[ni][0], c, 0, 0, 0);
        acc[mi][ni] = c;
      }
    if (more) store(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  const int RS = p.R * p.S;
  const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int k = k0 + wk + mi * 32 + 8 * (e >> 2) + 4 * half + (e & 3);
      if (k >= p.K) continue;
      const float sc = p.scale ? p.scale[k] : 1.f;
      float* row = p.dw + ((size_t)k * RS + tap) * p.C;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int c = c0 + wc + ni * 32 + l31;
        if (c < p.C) unsafeAtomicAdd(row + c, acc[mi][ni][e] * sc);
      }
    }
}

template <bool BUF>
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16x3_kernel(const WgradParams p, unsigned x_bytes, unsigned dy_bytes) {
  __shared__ __attribute__((aligned(16))) wg_bf16 lds[2 * 6 * WB_PLANE];   // [buffer][dY hi,mid,lo, X hi,mid,lo][16][WG_PITCH]
  wgrad_bf16x3_body<BUF>(p, lvc_xcd_remap(blockIdx.x, gridDim.x), x_bytes, dy_bytes, lds);
}

// ---- the weight gradients of SEVERAL layers in one launch (round 5).  At the training batch of the shipped box-corrector configs
// (2 images per GPU) a res4 layer has 8 400 output pixels: its 16-36 tiles cannot fill 256 CUs, the pixel range had to be cut
// into 512-pixel slices whose workgroups spend their time in prologue, atomics and a zeroing launch (65-95 us per layer for
// 10-25 us of MFMA work).  The weight gradient is not on the backward's critical path -- only the data gradients chain -- so
// the host queues the (x, dy) pairs and launches them together (lvc_amd.kernels.defer_wgrad): the tiles of ~24 layers fill the
// chip with long pixel loops (a few slices per tile instead of 17).  Jobs travel as kernel arguments (no table in memory).
#define WG_GROUP_MAX 24
struct WgradJob {
  WgradParams p;
  int wg_begin;                  // first logical workgroup of this job
  unsigned x_bytes, dy_bytes;
  int pad_;
};
struct WgradGroup {
  WgradJob job[WG_GROUP_MAX];
  int njobs, total_wgs;
};

__global__ __launch_bounds__(256, 2) void conv_wgrad_group_bf16x3_kernel(const WgradGroup g) {
  __shared__ __attribute__((aligned(16))) wg_bf16 lds[2 * 6 * WB_PLANE];
  const int lid = lvc_xcd_remap(blockIdx.x, gridDim.x);
  int j = 0;
#pragma unroll 1
  for (int i = 1; i < g.njobs; ++i)
    if (lid >= g.job[i].wg_begin) j = i;
  j = __builtin_amdgcn_readfirstlane(j);
  const WgradJob& job = g.job[j];
  wgrad_bf16x3_body<true>(job.p, lid - job.wg_begin, job.x_bytes, job.dy_bytes, lds);
}

// dst (OIHW, the parameter's layout) = (beta ? dst : 0) + src ([K][R*S][C], what the weight-gradient kernels write): the
// transposition ATen's conv backward never needs because it writes the parameter's layout directly.  One launch per group.
struct WgradFinJob { const float* src; float* dst; int K, C, RS, beta; long long begin; };   // begin: first (k, c) pair of this job in the launch
struct WgradFinGroup { WgradFinJob job[WG_GROUP_MAX]; int njobs; long long total; };
__global__ __launch_bounds__(256) void wgrad_finalize_group_kernel(const WgradFinGroup g) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= g.total) return;
  int j = 0;
  for (int q = 1; q < g.njobs; ++q)
    if (i >= g.job[q].begin) j = q;
  const WgradFinJob& f = g.job[j];
  const long long e = i - f.begin;
  const int c = (int)(e % f.C);
  const long long k = e / f.C;
  const float* s = f.src + k * f.RS * f.C + c;
  float* d = f.dst + (k * f.C + c) * f.RS;
  for (int t = 0; t < f.RS; ++t) {
    const float v = s[(long long)t * f.C];
    d[t] = f.beta ? d[t] + v : v;
  }
}

static int wgrad_bf16x3_params(WgradParams& p, const float* x, const float* dy, const float* scale, float* dw, int N, int H, int W,
                               int C, int K, int R, int S, int stride, int pad, int lddy) {
  const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
  p.x = x; p.dy = dy; p.scale = scale; p.dw = dw;
  p.N = N; p.H = H; p.W = W; p.C = C; p.Ho = Ho; p.Wo = Wo; p.K = K; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
  p.lddy = lddy; p.M = N * Ho * Wo;
  p.k_tiles = lvc_cdiv(K, 128); p.c_tiles = lvc_cdiv(C, 128);
  p.tiles = p.k_tiles * p.c_tiles * R * S;
  return lvc_cdiv(p.M, 16);
}

extern "C" int lvc_conv_wgrad_group_bf16x3(int njobs, const float* const* x, const float* const* dy, const float* const* scale,
                                           float* const* dw, const int* shapes, void* stream) {
  LVC_CHECK_ARG(njobs > 0 && x && dy && scale && dw && shapes, "null pointer / no jobs");
  hipStream_t st = (hipStream_t)stream;
  for (int j0 = 0; j0 < njobs; j0 += WG_GROUP_MAX) {
    const int nj = njobs - j0 < WG_GROUP_MAX ? njobs - j0 : WG_GROUP_MAX;
    WgradGroup g;
    long long total_chunks = 0;
    int nchunks[WG_GROUP_MAX];
    for (int j = 0; j < nj; ++j) {
      const int* sh = shapes + (size_t)(j0 + j) * 10;
      const int N = sh[0], H = sh[1], W = sh[2], C = sh[3], K = sh[4], R = sh[5], S = sh[6], stride = sh[7], pad = sh[8], lddy = sh[9];
      LVC_CHECK_ARG(x[j0 + j] && dy[j0 + j] && dw[j0 + j], "null pointer");
      LVC_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && K > 0 && R > 0 && S > 0 && stride > 0 && pad >= 0, "bad shape");
      LVC_CHECK_ARG(C % 4 == 0 && K % 4 == 0 && lddy % 4 == 0 && lddy >= K, "C, K and lddy must be multiples of 4");
      LVC_CHECK_ARG((((uintptr_t)x[j0 + j] | (uintptr_t)dy[j0 + j] | (uintptr_t)dw[j0 + j]) & 15) == 0, "pointers must be 16-byte aligned");
      const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
      LVC_CHECK_ARG(Ho > 0 && Wo > 0, "empty output map");
      const long long xb = (long long)N * H * W * C * 4, dyb = (long long)N * Ho * Wo * lddy * 4;
      LVC_CHECK_ARG(xb < (1ll << 31) && dyb < (1ll << 31), "a grouped job's tensors must be smaller than 2 GiB (launch it alone)");
      nchunks[j] = wgrad_bf16x3_params(g.job[j].p, x[j0 + j], dy[j0 + j], scale[j0 + j], dw[j0 + j], N, H, W, C, K, R, S, stride, pad, lddy);
      g.job[j].x_bytes = (unsigned)xb; g.job[j].dy_bytes = (unsigned)dyb; g.job[j].pad_ = 0;
      total_chunks += (long long)g.job[j].p.tiles * nchunks[j];
    }
    // equal slices of the pixel range over the whole group: about three rounds of the 768 resident workgroups, never shorter than 32 chunks
    constexpr int target_wgs = 2304, min_chunks = 32;
    long long per_wg = (total_chunks + target_wgs - 1) / target_wgs;
    if (per_wg < min_chunks) per_wg = min_chunks;
    int wgs = 0;
    for (int j = 0; j < nj; ++j) {
      WgradParams& p = g.job[j].p;
      int splits = (int)((nchunks[j] + per_wg / 2) / per_wg);
      if (splits < 1) splits = 1;
      p.chunks_per_split = lvc_cdiv(nchunks[j], splits);
      splits = lvc_cdiv(nchunks[j], p.chunks_per_split);
      g.job[j].wg_begin = wgs;
      wgs += p.tiles * splits;
    }
    g.njobs = nj; g.total_wgs = wgs;
    hipLaunchKernelGGL(conv_wgrad_group_bf16x3_kernel, dim3(wgs), dim3(256), 0, st, g);
    LVC_CHECK_LAUNCH();
  }
  return LVC_OK;
}

extern "C" int lvc_wgrad_finalize_group(int njobs, const float* const* src, float* const* dst, const int* shapes, void* stream) {
  LVC_CHECK_ARG(njobs > 0 && src && dst && shapes, "null pointer / no jobs");
  hipStream_t st = (hipStream_t)stream;
  for (int j0 = 0; j0 < njobs; j0 += WG_GROUP_MAX) {
    const int nj = njobs - j0 < WG_GROUP_MAX ? njobs - j0 : WG_GROUP_MAX;
    WgradFinGroup g;
    long long total = 0;
    for (int j = 0; j < nj; ++j) {
      const int* sh = shapes + (size_t)(j0 + j) * 4;
      LVC_CHECK_ARG(src[j0 + j] && dst[j0 + j] && sh[0] > 0 && sh[1] > 0 && sh[2] > 0, "bad job");
      g.job[j].src = src[j0 + j]; g.job[j].dst = dst[j0 + j];
      g.job[j].K = sh[0]; g.job[j].C = sh[1]; g.job[j].RS = sh[2]; g.job[j].beta = sh[3];
      g.job[j].begin = total;
      total += (long long)sh[0] * sh[1];
    }
    g.njobs = nj; g.total = total;
    hipLaunchKernelGGL(wgrad_finalize_group_kernel, dim3((unsigned)lvc_cdiv64(total, 256)), dim3(256), 0, st, g);
    LVC_CHECK_LAUNCH();
  }
  return LVC_OK;
}

// lvc_conv_wgrad_nhwc on the three-way bf16 split kernel: same arguments, same semantics, no range restriction.
extern "C" int lvc_conv_wgrad_nhwc_bf16x3(const float* x, const float* dy, const float* scale, float* dw, int N, int H, int W,
                                          int C, int K, int R, int S, int stride, int pad, int lddy, void* stream) {
  LVC_CHECK_ARG(x && dy && dw, "null pointer");
  LVC_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && K > 0 && R > 0 && S > 0 && stride > 0 && pad >= 0, "bad shape");
  LVC_CHECK_ARG(C % 4 == 0 && K % 4 == 0 && lddy % 4 == 0 && lddy >= K, "C, K and lddy must be multiples of 4");
  LVC_CHECK_ARG((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dw) & 15) == 0, "pointers must be 16-byte aligned");
  const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
  LVC_CHECK_ARG(Ho > 0 && Wo > 0, "empty output map");
  const long long M64 = (long long)N * Ho * Wo;
  LVC_CHECK_ARG(M64 < (1ll << 31) - 64, "too many output pixels");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(dw, 0, (size_t)K * R * S * C * sizeof(float), st) != hipSuccess) {
    lvc_set_error("%s: hipMemsetAsync failed", __func__);
    return LVC_ERR_HIP;
  }
  WgradParams p;
  p.x = x; p.dy = dy; p.scale = scale; p.dw = dw;
  p.N = N; p.H = H; p.W = W; p.C = C; p.Ho = Ho; p.Wo = Wo; p.K = K; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
  p.lddy = lddy; p.M = (int)M64;
  p.k_tiles = lvc_cdiv(K, 128); p.c_tiles = lvc_cdiv(C, 128);
  const int tiles = p.k_tiles * p.c_tiles * R * S;
  p.tiles = tiles;
  const int nchunks = lvc_cdiv(p.M, 16);
  // three workgroups fit a CU (146 VGPRs, 48 KB of LDS): aim at three full rounds of them (2304 on 256 CUs; measured over
  // the layer set: 1024 -> 12.4 ms, 1536 -> 12.1, 2304 -> 10.5, 3072 -> 11.0, 4608 -> 10.6)
  constexpr int target_wgs = 2304, min_chunks = 32;
  int splits = lvc_cdiv(target_wgs, tiles);
  const int max_splits = lvc_cdiv(nchunks, min_chunks);   // at least 32 chunks (512 pixels) per slice
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  p.chunks_per_split = lvc_cdiv(nchunks, splits);
  splits = lvc_cdiv(nchunks, p.chunks_per_split);
  const long long xb = (long long)N * H * W * C * 4, dyb = M64 * lddy * 4;
  if (xb < (1ll << 31) && dyb < (1ll << 31))
    hipLaunchKernelGGL(conv_wgrad_bf16x3_kernel<true>, dim3(tiles * splits), dim3(256), 0, st, p, (unsigned)xb, (unsigned)dyb);
  else
    hipLaunchKernelGGL(conv_wgrad_bf16x3_kernel<false>, dim3(tiles * splits), dim3(256), 0, st, p, 0u, 0u);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}
