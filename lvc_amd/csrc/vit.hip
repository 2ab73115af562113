// vit.hip -- the non-GEMM pieces of the DINO ViT-S/8 descriptor network that feeds the label-verification kNN
// (reference tools/run_nearest_neighbours.py:102-128 get_descriptors, :292-293 torch.hub 'facebookresearch/dino' dino_vits8;
// the network itself is third-party: published architecture, vision_transformer.py of that repository, restated in
// oracle/vit.py).  The linear layers run on the conv/GEMM kernels; here: patch gathering, token assembly (class token +
// position embedding), LayerNorm, exact (erf) GELU and multi-head self-attention with an online softmax.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// img [B,C,H,W] -> patches [B * (H/ps) * (W/ps), C*ps*ps], column = c*ps*ps + r*ps + s: the row order of
// patch_embed.proj.weight.reshape(D, C*ps*ps), so that the 8x8 stride-8 convolution is one GEMM.
__global__ void vit_patchify_kernel(const float* __restrict__ img, float* __restrict__ out, int B, int C, int H, int W, int ps) {
  const int Ph = H / ps, Pw = W / ps, KC = C * ps * ps;
  const long long total = (long long)B * Ph * Pw * KC;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int col = (int)(i % KC);
    const long long row = i / KC;
    const int px = (int)(row % Pw), py = (int)((row / Pw) % Ph), b = (int)(row / ((long long)Pw * Ph));
    const int s = col % ps, r = (col / ps) % ps, c = col / (ps * ps);
    out[i] = img[(((size_t)b * C + c) * H + py * ps + r) * W + px * ps + s];
  }
}

extern "C" int lvc_vit_patchify(const float* img, float* out, int B, int C, int H, int W, int ps, void* stream) {
  LVC_CHECK_ARG(img && out && B > 0 && C > 0 && ps > 0 && H % ps == 0 && W % ps == 0, "bad arguments");
  const long long total = (long long)B * C * H * W;
  const int blocks = (int)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535);
  hipLaunchKernelGGL(vit_patchify_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, out, B, C, H, W, ps);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
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

extern "C" int lvc_mha(const float* qkv, float* out, int B, int N, int H, int head_dim, float scale, void* stream) {
  LVC_CHECK_ARG(B >= 0 && N > 0 && H > 0, "bad sizes");
  LVC_CHECK_ARG(head_dim == MHA_DH, "head dimension must be 64");
  if (B == 0) return LVC_OK;
  LVC_CHECK_ARG(qkv && out && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0, "null or unaligned pointer");
  hipLaunchKernelGGL(mha_kernel, dim3(lvc_cdiv(N, MHA_TQ), H, B), dim3(MHA_TQ), 0, (hipStream_t)stream, qkv, out, N, H, scale);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}
