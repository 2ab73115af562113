// vit.hip -- the non-GEMM pieces of the DINO ViT-S/8 descriptor network that feeds the label-verification kNN
// (reference tools/run_nearest_neighbours.py:102-128 get_descriptors, :292-293 torch.hub 'facebookresearch/dino' dino_vits8;
// the network itself is third-party: published architecture, vision_transformer.py of that repository, restated in
// oracle/vit.py).  The linear layers run on the conv/GEMM kernels; here: patch gathering, token assembly (class token +
// position embedding), LayerNorm, exact (erf) GELU and multi-head self-attention with an online softmax.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// img [B,C,H,W] -> patches [B * (H/ps) * (W/ps), C*ps*ps], column = c*ps*ps + r*ps + s: the row order of
// patch_embed.proj.weight.reshape(D, C*ps*ps), so that the 8x8 stride-8 convolution is one GEMM.  NORM: the crops' per-channel
// normalisation (x - mean[c]) / std[c] of get_descriptors (tools/run_nearest_neighbours.py:95-99) on the way -- the same fp32
// subtraction and IEEE division, so the same bits as the separate elementwise pass.
struct PatchNorm { float mean[8], std[8]; };
template <bool NORM>
__global__ void vit_patchify_kernel(const float* __restrict__ img, float* __restrict__ out, int B, int C, int H, int W, int ps, PatchNorm nm) {
  const int Ph = H / ps, Pw = W / ps, KC = C * ps * ps;
  const long long total = (long long)B * Ph * Pw * KC;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int col = (int)(i % KC);
    const long long row = i / KC;
    const int px = (int)(row % Pw), py = (int)((row / Pw) % Ph), b = (int)(row / ((long long)Pw * Ph));
    const int s = col % ps, r = (col / ps) % ps, c = col / (ps * ps);
    float v = img[(((size_t)b * C + c) * H + py * ps + r) * W + px * ps + s];
    if (NORM) v = (v - nm.mean[c]) / nm.std[c];
    out[i] = v;
  }
}

static int patchify_launch(const float* img, const float* mean, const float* std, float* out, int B, int C, int H, int W, int ps,
                           void* stream) {
  LVC_CHECK_ARG(img && out && B > 0 && C > 0 && ps > 0 && H % ps == 0 && W % ps == 0, "bad arguments");
  LVC_CHECK_ARG((mean == nullptr) == (std == nullptr) && (!mean || C <= 8), "mean and std come together, for at most 8 channels");
  const long long total = (long long)B * C * H * W;
  const int blocks = (int)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535);
  PatchNorm nm = {};
  if (mean) {
    for (int c = 0; c < C; ++c) { nm.mean[c] = mean[c]; nm.std[c] = std[c]; }
    hipLaunchKernelGGL(vit_patchify_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, out, B, C, H, W, ps, nm);
  } else {
    hipLaunchKernelGGL(vit_patchify_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, out, B, C, H, W, ps, nm);
  }
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

extern "C" int lvc_vit_patchify(const float* img, float* out, int B, int C, int H, int W, int ps, void* stream) {
  return patchify_launch(img, nullptr, nullptr, out, B, C, H, W, ps, stream);
}

// lvc_vit_patchify of (img - mean[c]) / std[c]: mean / std are HOST arrays of C <= 8 floats
extern "C" int lvc_vit_patchify_norm(const float* img, const float* mean, const float* std, float* out, int B, int C, int H, int W,
                                     int ps, void* stream) {
  LVC_CHECK_ARG(mean && std, "null pointer");
  return patchify_launch(img, mean, std, out, B, C, H, W, ps, stream);
}

// tokens[b][0] = cls + pos[0];  tokens[b][1 + p] = emb[b * P + p] + pos[1 + p]   (D % 4 == 0)
__global__ void vit_tokens_kernel(const float* __restrict__ emb, const float* __restrict__ cls, const float* __restrict__ pos,
                                  float* __restrict__ out, int B, int P, int D) {
  const int D4 = D / 4, N = P + 1;
  const long long total = (long long)B * N * D4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int d4 = (int)(i % D4);
    const long long row = i / D4;
    const int n = (int)(row % N), b = (int)(row / N);
    const f32x4 pe = *reinterpret_cast<const f32x4*>(pos + (size_t)n * D + d4 * 4);
    const f32x4 v = n == 0 ? *reinterpret_cast<const f32x4*>(cls + d4 * 4)
                           : *reinterpret_cast<const f32x4*>(emb + ((size_t)b * P + n - 1) * D + d4 * 4);
    *reinterpret_cast<f32x4*>(out + (size_t)row * D + d4 * 4) = v + pe;
  }
}

extern "C" int lvc_vit_tokens(const float* emb, const float* cls, const float* pos, float* out, int B, int P, int D, void* stream) {
  LVC_CHECK_ARG(emb && cls && pos && out && B > 0 && P > 0 && D > 0 && D % 4 == 0, "bad arguments");
  const long long total = (long long)B * (P + 1) * (D / 4);
  const int blocks = (int)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535);
  hipLaunchKernelGGL(vit_tokens_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, emb, cls, pos, out, B, P, D);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// LayerNorm over the last dimension (torch.nn.LayerNorm: biased variance, eps inside the sqrt).  One wave per row; the row
// lives in registers (D <= 64 * LN_MAX_PER_LANE), mean and variance are two exact passes over it.
#define LN_MAX_PER_LANE 32
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ y, int ldy, int M, int D,
                                                        float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* xr = x + (size_t)row * ldx;
  float v[LN_MAX_PER_LANE];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAX_PER_LANE; ++j) {
    const int d = j * 64 + lane;
    v[j] = d < D ? xr[d] : 0.f;
    s += v[j];
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)D;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAX_PER_LANE; ++j) {
    const int d = j * 64 + lane;
    const float t = d < D ? v[j] - mean : 0.f;
    q += t * t;
  }
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = 1.f / sqrtf(q / (float)D + eps);
  float* yr = y + (size_t)row * ldy;
#pragma unroll
  for (int j = 0; j < LN_MAX_PER_LANE; ++j) {
    const int d = j * 64 + lane;
    if (d < D) yr[d] = (v[j] - mean) * rstd * (w ? w[d] : 1.f) + (b ? b[d] : 0.f);
  }
}

extern "C" int lvc_layernorm(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int M, int D, float eps,
                             void* stream) {
  LVC_CHECK_ARG(M >= 0 && D > 0 && D <= 64 * LN_MAX_PER_LANE, "row length must be in 1..2048");
  if (M == 0) return LVC_OK;
  LVC_CHECK_ARG(x && y && ldx >= D && ldy >= D, "bad arguments");
  hipLaunchKernelGGL(layernorm_kernel, dim3(lvc_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, w, b, y, ldy, M, D, eps);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// torch.nn.GELU() (approximate='none'): x * 0.5 * (1 + erf(x / sqrt(2)))
__global__ void gelu_kernel(const float* __restrict__ x, float* __restrict__ y, long long n4) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] * 0.5f * (1.f + erff(v[e] * 0.70710678118654752440f));
    *reinterpret_cast<f32x4*>(y + i * 4) = v;
  }
}

extern "C" int lvc_gelu(const float* x, float* y, long long n, void* stream) {
  LVC_CHECK_ARG(n >= 0 && n % 4 == 0, "element count must be a multiple of 4");
  if (n == 0) return LVC_OK;
  LVC_CHECK_ARG(x && y, "null pointer");
  const long long n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 65535 ? (n4 + 255) / 256 : 65535);
  hipLaunchKernelGGL(gelu_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, n4);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// Multi-head self-attention of the ViT block: qkv [B*N, 3*H*64] (column = which * H*64 + h * 64 + d, the layout of
// nn.Linear(dim, 3 dim) followed by reshape(B, N, 3, H, 64)), out [B*N, H*64] = softmax(q k^T * scale) v per (b, h).
// One thread per query: q (pre-scaled) and the running output in registers; keys / values of 64 tokens at a time in LDS, read
// by every lane at the same address (broadcast); softmax online in blocks of 8 keys (one rescale per block).
#define MHA_DH 64
#define MHA_TQ 128
#define MHA_TK 64
__global__ __launch_bounds__(MHA_TQ) void mha_kernel(const float* __restrict__ qkv, float* __restrict__ out, int N, int H, float scale) {
  __shared__ __attribute__((aligned(16))) float sk[MHA_TK * MHA_DH];
  __shared__ __attribute__((aligned(16))) float sv[MHA_TK * MHA_DH];
  const int b = blockIdx.z, h = blockIdx.y, tid = threadIdx.x;
  const int nq = blockIdx.x * MHA_TQ + tid;
  const int ld = 3 * H * MHA_DH;
  const bool live = nq < N;
  const float* qp = qkv + ((size_t)b * N + (live ? nq : N - 1)) * ld + h * MHA_DH;
  float q[MHA_DH], o[MHA_DH];
#pragma unroll
  for (int d4 = 0; d4 < MHA_DH / 4; ++d4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(qp + d4 * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { q[d4 * 4 + e] = t[e] * scale; o[d4 * 4 + e] = 0.f; }
  }
  float m = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < N; k0 += MHA_TK) {
    __syncthreads();
    // stage keys k0 .. k0+63 and their values: 64 x 64 floats each = 1024 float4 per matrix, 8 per thread
    for (int i = tid; i < MHA_TK * MHA_DH / 4; i += MHA_TQ) {
      const int r = i / (MHA_DH / 4), c4 = i % (MHA_DH / 4);
      const int kn = k0 + r < N ? k0 + r : N - 1;
      const float* base = qkv + ((size_t)b * N + kn) * ld + h * MHA_DH + c4 * 4;
      *reinterpret_cast<f32x4*>(sk + r * MHA_DH + c4 * 4) = *reinterpret_cast<const f32x4*>(base + H * MHA_DH);
      *reinterpret_cast<f32x4*>(sv + r * MHA_DH + c4 * 4) = *reinterpret_cast<const f32x4*>(base + 2 * H * MHA_DH);
    }
    __syncthreads();
    const int kend = min(MHA_TK, N - k0);
    for (int j0 = 0; j0 < kend; j0 += 8) {
      float s[8];
      float bm = -INFINITY;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float acc = 0.f;
        const float* kr = sk + (j0 + j) * MHA_DH;
#pragma unroll
        for (int d4 = 0; d4 < MHA_DH / 4; ++d4) {
          const f32x4 kv = *reinterpret_cast<const f32x4*>(kr + d4 * 4);
          acc += q[d4 * 4] * kv[0] + q[d4 * 4 + 1] * kv[1] + q[d4 * 4 + 2] * kv[2] + q[d4 * 4 + 3] * kv[3];
        }
        s[j] = j0 + j < kend ? acc : -INFINITY;
        bm = fmaxf(bm, s[j]);
      }
      const float mn = fmaxf(m, bm);
      const float corr = expf(m - mn);     // first block: exp(-inf) = 0
      l *= corr;
#pragma unroll
      for (int d = 0; d < MHA_DH; ++d) o[d] *= corr;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float pj = expf(s[j] - mn);  // masked keys: exp(-inf) = 0
        l += pj;
        const float* vr = sv + (j0 + j) * MHA_DH;
#pragma unroll
        for (int d4 = 0; d4 < MHA_DH / 4; ++d4) {
          const f32x4 vv = *reinterpret_cast<const f32x4*>(vr + d4 * 4);
          o[d4 * 4] += pj * vv[0]; o[d4 * 4 + 1] += pj * vv[1]; o[d4 * 4 + 2] += pj * vv[2]; o[d4 * 4 + 3] += pj * vv[3];
        }
      }
      m = mn;
    }
  }
  if (live) {
    const float inv = 1.f / l;
    float* op = out + ((size_t)b * N + nq) * (H * MHA_DH) + h * MHA_DH;
#pragma unroll
    for (int d4 = 0; d4 < MHA_DH / 4; ++d4) {
      f32x4 t = {o[d4 * 4] * inv, o[d4 * 4 + 1] * inv, o[d4 * 4 + 2] * inv, o[d4 * 4 + 3] * inv};
      *reinterpret_cast<f32x4*>(op + d4 * 4) = t;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same attention on the matrix cores (round 3).  fp32-accurate through the two-way fp16 operand split of the conv kernels
// (x = x1 + x2 with x1 = fp16(x), x2 = fp16(x - x1); three v_mfma_f32_32x32x16_f16 per block: x1 y2, x2 y1, x1 y1 into one fp32
// accumulator) for BOTH products, so the result stays at the fp32 reference's error level (tests/test_gpu_vit.py).
//   1. mha_split_kernel: qkv -> six fp16 planes [B*H][Npad][64]: q * scale * log2(e) (the softmax runs in base 2), k, v, each as
//      (x1, x2); rows N..Npad-1 are zero.  One pass over qkv instead of a split per query block.
//   2. mha_mfma_kernel: a workgroup = 128 queries of one (image, head), a wave = 32 of them.  Per tile of 64 keys the K and V
//      planes arrive by LDS-DMA (double-buffered, a wave fetches one plane).  Everything is computed TRANSPOSED so that a lane
//      owns one query: S^T = K Q^T (A operand = K rows from LDS, B operand = the wave's Q fragments, kept in registers) leaves
//      lane (query, g) with 16 keys of each 32-key block -- row max and row sum are in-lane loops plus ONE exchange with lane
//      ^ 32 -- and P^T is then already the B operand of O^T = V^T P^T: the probabilities never leave the registers.  Two
//      details make the register hand-off exact: K's rows enter the A operand through the permutation that swaps bits 2 and 3
//      of the row index, which makes accumulator registers 8s .. 8s+7 of lane group g the keys 16s + 8g .. 16s + 8g + 7 --
//      the k order of a 32x32x16 operand; and V^T (k = key, the ROW index of V) is read from the [key][d] LDS image with
//      gfx950's transpose read ds_read_b64_tr_b16 (two per fragment), so V is never transposed in memory.
//      LDS rows are 128 B, DMA-written lane-linear; conflicts are avoided by swizzling on the SOURCE side: K's 16-byte chunks by
//      (row >> 1) & 7, V's 32-byte groups by f(row) = bit1(row) | bit3(row) << 1 (the four even rows a transpose read touches
//      land on the two halves of the banks).
typedef _Float16 h16;
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* at_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* at_glb_ptr_t;
typedef __fp16 at_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) at_fp16x4 at_lds_fp16x4;
#define AT_TQ 128
#define AT_TK 64

__global__ __launch_bounds__(256) void mha_split_kernel(const float* __restrict__ qkv, h16* __restrict__ ws, int B, int N, int H,
                                                        int Npad, float qscale, int* __restrict__ err) {
  float big = 0.f;      // largest |operand| rounded to fp16: beyond 65504 (or NaN) the planes would carry inf / NaN silently
  const long long total = (long long)B * Npad * 3 * H * 16;
  const size_t PS = (size_t)B * H * Npad * 64;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int d4 = (int)(i & 15);
    long long r = i >> 4;
    const int h = (int)(r % H); r /= H;
    const int which = (int)(r % 3); r /= 3;
    const int n = (int)(r % Npad);
    const int b = (int)(r / Npad);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
      v = *reinterpret_cast<const f32x4*>(qkv + ((size_t)b * N + n) * (3 * H * 64) + which * H * 64 + h * 64 + d4 * 4);
      if (which == 0) v *= qscale;
      big = fmaxf(big, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
      if (v[0] != v[0] || v[1] != v[1] || v[2] != v[2] || v[3] != v[3]) big = INFINITY;
    }
    h16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = v[e];
      asm volatile("" : "+v"(a));      // a VALUE (the rounded q * qscale), not a product the compiler may fuse into the subtraction below:
      hi[e] = (h16)a;                  // the qkv GEMM's planes epilogue (conv_pw_s1.hip) forms the same two numbers
      lo[e] = (h16)(a - (float)hi[e]);
    }
    const size_t o = ((size_t)(b * H + h) * Npad + n) * 64 + d4 * 4;
    *reinterpret_cast<h16x4*>(ws + (size_t)(2 * which) * PS + o) = hi;
    *reinterpret_cast<h16x4*>(ws + (size_t)(2 * which + 1) * PS + o) = lo;
  }
  if (err && !(big <= 65504.f)) atomicOr(err, 2);
}

__global__ __launch_bounds__(256, 2) void mha_mfma_kernel(const h16* __restrict__ ws, float* __restrict__ out, int B, int N, int H,
                                                          int Npad) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * 4 * AT_TK * 128];   // [buffer][Kh, Kl, Vh, Vl][64 rows][128 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 31, fh = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const size_t PS = (size_t)B * H * Npad * 64;
  const size_t bh = ((size_t)b * H + h) * Npad * 64;
  const int q0 = blockIdx.x * AT_TQ + wave * 32;
  // Q fragments (B operand): lane (query q0 + fi, k group fh) holds d = 16 ks + 8 fh .. + 7
  h16x8 qh[4], ql[4];
  {
    const h16* qp = ws + bh + (size_t)(q0 + fi) * 64 + 8 * fh;     // q0 + fi < Npad (a multiple of 128)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qh[ks] = *reinterpret_cast<const h16x8*>(qp + 16 * ks);
      ql[ks] = *reinterpret_cast<const h16x8*>(qp + PS + 16 * ks);
    }
  }
  // DMA: wave w fetches plane w (0 Kh, 1 Kl, 2 Vh, 3 Vl) of a tile: 8 pieces of 8 rows x 128 B; lane = (row lane>>3, chunk lane&7)
  const h16* plane = ws + (size_t)(2 + wave) * PS + bh;
  int src_off[8];     // halves, relative to the tile's first row
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = 8 * j + (lane >> 3), c = lane & 7;
    int lc;
    if (wave < 2) lc = c ^ ((row >> 1) & 7);
    else lc = ((((c >> 1) ^ (((row >> 1) & 1) | (((row >> 3) & 1) << 1))) << 1) | (c & 1));
    src_off[j] = row * 64 + lc * 8;
  }
  auto issue = [&](int t, int buf) {
    unsigned char* dst = lds + (buf * 4 + wave) * (AT_TK * 128);
    const h16* base = plane + (size_t)t * AT_TK * 64;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      __builtin_amdgcn_global_load_lds((at_glb_ptr_t)(base + src_off[j]), (at_lds_ptr_t)(dst + j * 1024), 16, 0, 0);
  };
  const int nt = (N + AT_TK - 1) / AT_TK;
  f32x16 O[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) O[d][e] = 0.f;
  float m = -INFINITY, l = 0.f;
  const int pfi = (fi & ~12) | ((fi & 4) << 1) | ((fi & 8) >> 1);     // K row permutation: bits 2 and 3 swapped
  const int t16 = lane & 15, q1 = (lane >> 4) & 1;
  issue(0, 0);
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    if (t + 1 < nt) {
      issue(t + 1, buf ^ 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const unsigned char* Kh = lds + (buf * 4 + 0) * (AT_TK * 128);
    const unsigned char* Kl = lds + (buf * 4 + 1) * (AT_TK * 128);
    const at_lds_fp16x4* Vh = (const at_lds_fp16x4*)(lds + (buf * 4 + 2) * (AT_TK * 128));
    const at_lds_fp16x4* Vl = (const at_lds_fp16x4*)(lds + (buf * 4 + 3) * (AT_TK * 128));
    // ---- S^T = K Q^T for the tile's two 32-key blocks
    f32x16 S[2];
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
#pragma unroll
      for (int e = 0; e < 16; ++e) S[jb][e] = 0.f;
      const int row = 32 * jb + pfi;
      const int sw = (row >> 1) & 7;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int off = row * 128 + (((ks * 2 + fh) ^ sw) << 4);
        const h16x8 kh = *reinterpret_cast<const h16x8*>(Kh + off);
        const h16x8 kl = *reinterpret_cast<const h16x8*>(Kl + off);
        S[jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], S[jb], 0, 0, 0);
        S[jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], S[jb], 0, 0, 0);
        S[jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], S[jb], 0, 0, 0);
      }
    }
    // register e of block jb in lane group fh is key k0 + 32 jb + 16 (e >> 3) + 8 fh + (e & 7)
    const int k0 = t * AT_TK;
    if (k0 + AT_TK > N) {
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (k0 + 32 * jb + 16 * (e >> 3) + 8 * fh + (e & 7) >= N) S[jb][e] = -INFINITY;
    }
    // ---- online softmax (base 2), one query per lane pair (lane, lane ^ 32)
    float mloc = -INFINITY;
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int e = 0; e < 16; ++e) mloc = fmaxf(mloc, S[jb][e]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
    const float mn = fmaxf(m, mloc);
    const float corr = __builtin_amdgcn_exp2f(m - mn);
    m = mn;
    l *= corr;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) O[d][e] *= corr;
    h16x8 ph[2][2], pl[2][2];
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pv = __builtin_amdgcn_exp2f(S[jb][e] - mn);
        l += pv;
        const h16 hi = (h16)pv;
        ph[jb][e >> 3][e & 7] = hi;
        pl[jb][e >> 3][e & 7] = (h16)(pv - (float)hi);
      }
    // ---- O^T += V^T P^T: four k16 steps of 16 keys, two 32-wide blocks of d
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int r1 = 32 * jb + 16 * s + 8 * fh + (t16 >> 2), r2 = r1 + 4;
        const int f1 = ((r1 >> 1) & 1) | (((r1 >> 3) & 1) << 1), f2 = ((r2 >> 1) & 1) | (((r2 >> 3) & 1) << 1);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const int lg = 2 * db + q1;
          const int a1 = (r1 * 128 + ((lg ^ f1) << 5) + 8 * (t16 & 3)) >> 3, a2 = (r2 * 128 + ((lg ^ f2) << 5) + 8 * (t16 & 3)) >> 3;
          const h16x4 vh1 = __builtin_bit_cast(h16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((at_lds_fp16x4*)(Vh + a1)));
          const h16x4 vh2 = __builtin_bit_cast(h16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((at_lds_fp16x4*)(Vh + a2)));
          const h16x4 vl1 = __builtin_bit_cast(h16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((at_lds_fp16x4*)(Vl + a1)));
          const h16x4 vl2 = __builtin_bit_cast(h16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((at_lds_fp16x4*)(Vl + a2)));
          const h16x8 vh = {vh1[0], vh1[1], vh1[2], vh1[3], vh2[0], vh2[1], vh2[2], vh2[3]};
          const h16x8 vl = {vl1[0], vl1[1], vl1[2], vl1[3], vl2[0], vl2[1], vl2[2], vl2[3]};
          O[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[jb][s], O[db], 0, 0, 0);
          O[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[jb][s], O[db], 0, 0, 0);
          O[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[jb][s], O[db], 0, 0, 0);
        }
      }
    __syncthreads();     // every wave is done with this buffer before the tile after next is fetched into it
  }
  l += __shfl_xor(l, 32);
  const float inv = 1.f / l;
  const int q = q0 + fi;
  if (q < N) {
    float* op = out + ((size_t)b * N + q) * (H * 64) + h * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        const f32x4 v = {O[db][4 * e4] * inv, O[db][4 * e4 + 1] * inv, O[db][4 * e4 + 2] * inv, O[db][4 * e4 + 3] * inv};
        *reinterpret_cast<f32x4*>(op + 32 * db + 8 * e4 + 4 * fh) = v;
      }
  }
}

// bytes of the fp16 plane workspace lvc_mha_mfma needs
extern "C" long long lvc_mha_workspace_bytes(int B, int N, int H) {
  const long long Npad = (long long)((N + AT_TQ - 1) / AT_TQ) * AT_TQ;
  return 6ll * B * H * Npad * 64 * 2;
}

// lvc_mha on the matrix cores: same arguments and result (to fp32 rounding), plus `workspace` (lvc_mha_workspace_bytes, 16-byte
// aligned) for the fp16 operand planes.  head_dim is 64.
extern "C" int lvc_mha_mfma(const float* qkv, float* out, void* workspace, int B, int N, int H, float scale, int* d_error_word,
                            void* stream) {
  LVC_CHECK_ARG(B >= 0 && N > 0 && H > 0, "bad sizes");
  if (B == 0) return LVC_OK;
  LVC_CHECK_ARG(qkv && out && workspace && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)workspace & 15) == 0,
                "null or unaligned pointer");
  LVC_CHECK_ARG(H <= 65535 && B <= 65535, "too many heads / images for one launch");
  const int Npad = (N + AT_TQ - 1) / AT_TQ * AT_TQ;
  const long long total = (long long)B * Npad * 3 * H * 16;
  const int blocks = (int)((total + 255) / 256 < 1048576 ? (total + 255) / 256 : 1048576);
  hipLaunchKernelGGL(mha_split_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, qkv, (h16*)workspace, B, N, H, Npad,
                     scale * 1.44269504088896340736f, d_error_word);
  LVC_CHECK_LAUNCH();
  hipLaunchKernelGGL(mha_mfma_kernel, dim3(Npad / AT_TQ, H, B), dim3(256), 0, (hipStream_t)stream, (const h16*)workspace, out, B, N, H,
                     Npad);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// lvc_mha_mfma's second pass alone, on operand planes that already exist (lvc_conv1x1_qkv_planes_f16s1 wrote them from the qkv GEMM's epilogue):
// planes [6][B*H][Npad][64] fp16 with q pre-multiplied by scale * log2(e) and rows N..Npad-1 zero.
extern "C" int lvc_mha_mfma_planes(const void* planes, float* out, int B, int N, int H, void* stream) {
  LVC_CHECK_ARG(B >= 0 && N > 0 && H > 0, "bad sizes");
  if (B == 0) return LVC_OK;
  LVC_CHECK_ARG(planes && out && ((uintptr_t)planes & 15) == 0 && ((uintptr_t)out & 15) == 0, "null or unaligned pointer");
  LVC_CHECK_ARG(H <= 65535 && B <= 65535, "too many heads / images for one launch");
  const int Npad = (N + AT_TQ - 1) / AT_TQ * AT_TQ;
  hipLaunchKernelGGL(mha_mfma_kernel, dim3(Npad / AT_TQ, H, B), dim3(256), 0, (hipStream_t)stream, (const h16*)planes, out, B, N, H, Npad);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

// Attention of ONE query per image -- token 0, the class token -- against all N keys / values: what the LAST block of the descriptor network
// needs (its output is read at the class rows only, run_nearest_neighbours.py:102-128 -> `x[:, 0]` of the DINO forward).  One wave
// per (image, head): the lanes split the keys (scores, two-pass softmax with wave-wide max / sum), then the 64 output dimensions.
__global__ __launch_bounds__(64) void mha_cls_kernel(const float* __restrict__ qkv, float* __restrict__ out, int N, int H, float scale) {
  __shared__ float sp[1024];           // probabilities of up to 1024 keys
  const int b = blockIdx.y, h = blockIdx.x, lane = threadIdx.x;
  const int ld = 3 * H * MHA_DH;
  const float* base = qkv + (size_t)b * N * ld + h * MHA_DH;
  float q[MHA_DH];
#pragma unroll
  for (int d4 = 0; d4 < MHA_DH / 4; ++d4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(base + d4 * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) q[d4 * 4 + e] = t[e] * scale;
  }
  float mx = -INFINITY;
  for (int k = lane; k < N; k += 64) {
    const float* kr = base + (size_t)k * ld + H * MHA_DH;
    float acc = 0.f;
#pragma unroll
    for (int d4 = 0; d4 < MHA_DH / 4; ++d4) {
      const f32x4 kv = *reinterpret_cast<const f32x4*>(kr + d4 * 4);
      acc += q[d4 * 4] * kv[0] + q[d4 * 4 + 1] * kv[1] + q[d4 * 4 + 2] * kv[2] + q[d4 * 4 + 3] * kv[3];
    }
    sp[k] = acc;
    mx = fmaxf(mx, acc);
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
  for (int k = lane; k < N; k += 64) {
    const float pk = expf(sp[k] - mx);
    sp[k] = pk;
    sum += pk;
  }
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  __syncthreads();
  // lane = output dimension: sum over the keys of p_k v[k][lane] (a row of V is 64 consecutive floats: coalesced)
  float acc = 0.f;
  const float* vb = base + 2 * H * MHA_DH + lane;
  for (int k = 0; k < N; ++k) acc += sp[k] * vb[(size_t)k * ld];
  out[(size_t)b * H * MHA_DH + h * MHA_DH + lane] = acc / sum;
}

// qkv [B*N, 3*H*64] -> out [B, H*64]: the attention output of token 0 of every image (fp32 arithmetic throughout).  N <= 1024.
extern "C" int lvc_mha_cls(const float* qkv, float* out, int B, int N, int H, float scale, void* stream) {
  LVC_CHECK_ARG(B >= 0 && N > 0 && N <= 1024 && H > 0 && H <= 65535 && B <= 65535, "bad sizes");
  if (B == 0) return LVC_OK;
  LVC_CHECK_ARG(qkv && out && ((uintptr_t)qkv & 15) == 0, "null or unaligned pointer");
  hipLaunchKernelGGL(mha_cls_kernel, dim3(H, B), dim3(64), 0, (hipStream_t)stream, qkv, out, N, H, scale);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

extern "C" int lvc_mha(const float* qkv, float* out, int B, int N, int H, int head_dim, float scale, void* stream) {
  LVC_CHECK_ARG(B >= 0 && N > 0 && H > 0, "bad sizes");
  LVC_CHECK_ARG(head_dim == MHA_DH, "head dimension must be 64");
  if (B == 0) return LVC_OK;
  LVC_CHECK_ARG(qkv && out && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0, "null or unaligned pointer");
  hipLaunchKernelGGL(mha_kernel, dim3(lvc_cdiv(N, MHA_TQ), H, B), dim3(MHA_TQ), 0, (hipStream_t)stream, qkv, out, N, H, scale);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}
