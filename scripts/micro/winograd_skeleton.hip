// Microbenchmark: what the matrix pipe sustains on THIS part under its power cap, as a function of
//   * operand type (fp16 / bf16) and operand data (zeros, N(0,1) "hi-plane-like" values, full-entropy "lo-plane-like" values),
//   * LDS fragment reads per MFMA (0 = operands stay in registers; 2/3 = the 64x64 wave tile of conv3x3_halo_h2_kernel;
//     1/2 = a 64x128 wave tile; 1/3 = a 128x128 wave tile),
//   * waves per SIMD (2 = 8-wave workgroups, 1 = 4-wave workgroups).
// Every launch runs ~30-60 ms so that the clock has settled to the power budget; reported: TFLOP/s of
// v_mfma_f32_32x32x16_{f16,bf16} (2 * 32 * 32 * 16 flop each) over all 256 CUs.  This is the ceiling any split-precision
// kernel built on these instructions can be priced against (three MFMAs per fp32-accurate product: divide by 3).
// This copy (mfma_ingest_ceiling.hip, round 4) adds what a GEMM chunk loop does besides MFMAs and fragment reads: DMA wave instructions
// (global_load_lds_dwordx4, 1 KB each, from an L2-resident region) per k16 step and a workgroup barrier every second step -- what do the
// pointwise kernel's 48 KB per 32-deep chunk cost under the power cap, and what would a 64 x 128 wave tile (64 KB per chunk of twice the MFMAs) give?
// This copy (winograd_skeleton.hip, round 5): only the direct-3x3 and Winograd F(2,3) chunk-loop skeletons.
// Build + run:  hipcc --offload-arch=gfx950 -O3 winograd_skeleton.hip -o /tmp/wsk && /tmp/wsk
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Split form (what the conv kernels do): per k16 step a wave reads 2 * (MI + NI) fragments (hi and lo plane of each operand,
// 16 B per lane each) and issues 3 * MI * NI MFMAs (hi*lo', hi*hi, lo*hi') into ONE accumulator per 32x32 block (TWOACC = 0)
// or into a main and a cross accumulator (TWOACC = 1: the current kernels).  READS = 0 keeps the fragments in registers.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
// DMA = 1 KB DMA instructions per wave and k16 step; BAR = 1: s_barrier every second step
template <int MI, int NI, bool BF, int READS, int NW, int TWOACC, int DMA, int BAR>
__global__ __launch_bounds__(NW * 64) void k(const u32x4* __restrict__ src, float* __restrict__ out, int iters, const char* __restrict__ stream, unsigned stream_mask) {
  __shared__ u32x4 lds[NW * 64 * 8];   // 8 x 16 B per lane, lane-linear: conflict-free ds_read_b128
  __shared__ __attribute__((aligned(1024))) unsigned char ring[DMA ? 3 * NW * DMA * 1024 : 16];
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned spos = (blockIdx.x * 7919u + wv * 104729u) * 1024u;
  const int tid = threadIdx.x;
  for (int i = 0; i < 8; ++i) lds[i * NW * 64 + tid] = src[(blockIdx.x * 8 + i) * NW * 64 + tid];
  __syncthreads();
  f32x16 acc[MI][NI], accx[TWOACC ? MI : 1][TWOACC ? NI : 1];
  for (int a = 0; a < MI; ++a)
    for (int b = 0; b < NI; ++b)
      for (int e = 0; e < 16; ++e) { acc[a][b][e] = 0.f; if (TWOACC) accx[TWOACC ? a : 0][TWOACC ? b : 0][e] = 0.f; }
  u32x4 fa[MI][2], fb[NI][2];
  for (int a = 0; a < MI; ++a) for (int pl = 0; pl < 2; ++pl) fa[a][pl] = lds[((2 * a + pl) % 8) * NW * 64 + tid];
  for (int b = 0; b < NI; ++b) for (int pl = 0; pl < 2; ++pl) fb[b][pl] = lds[((2 * b + pl + 3) % 8) * NW * 64 + tid];
  auto mm = [&](u32x4 x, u32x4 y, f32x16 c) {
    if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0);
  };
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    if (DMA) {
#pragma unroll
      for (int q = 0; q < DMA; ++q) {
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(stream + ((spos + (threadIdx.x & 63) * 16u) & stream_mask)),
                                         (lds_ptr_t)(ring + ((it % 3) * NW * DMA + wv * DMA + q) * 1024), 16, 0, 0);
        spos += 1024u * 61u;
      }
#ifdef STRICT_WAIT      // everything issued ONE step ago has landed (the kernels' rule: the barrier publishes the buffer to all waves)
      if (DMA == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if (DMA == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
#else
      if (DMA == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (DMA == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#endif
    }
    if (BAR && (it & 1)) __builtin_amdgcn_s_barrier();
    if (READS) {
      const int rot = it & 7;
#pragma unroll
      for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) fa[a][pl] = lds[((2 * a + pl + rot) & 7) * NW * 64 + tid];
#pragma unroll
      for (int b = 0; b < NI; ++b)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) fb[b][pl] = lds[((2 * b + pl + 3 + rot) & 7) * NW * 64 + tid];
    }
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
      for (int b = 0; b < NI; ++b) {
        if (TWOACC) accx[TWOACC ? a : 0][TWOACC ? b : 0] = mm(fa[a][0], fb[b][1], accx[TWOACC ? a : 0][TWOACC ? b : 0]);
        else acc[a][b] = mm(fa[a][0], fb[b][1], acc[a][b]);
      }
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
      for (int b = 0; b < NI; ++b) acc[a][b] = mm(fa[a][0], fb[b][0], acc[a][b]);
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
      for (int b = 0; b < NI; ++b) {
        if (TWOACC) accx[TWOACC ? a : 0][TWOACC ? b : 0] = mm(fa[a][1], fb[b][0], accx[TWOACC ? a : 0][TWOACC ? b : 0]);
        else acc[a][b] = mm(fa[a][1], fb[b][0], acc[a][b]);
      }
  }
  float s = 0.f;
  for (int a = 0; a < MI; ++a)
    for (int b = 0; b < NI; ++b)
      for (int e = 0; e < 16; ++e) s += acc[a][b][e] + (TWOACC ? accx[TWOACC ? a : 0][TWOACC ? b : 0][e] : 0.f);
  if (s == 12345.678f) out[tid] = s;
}

static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; __builtin_memcpy(&u, &h, 2); return u; }
static unsigned short f2bf(float f) { unsigned u; __builtin_memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
static float gauss() {
  float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
  return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
}

int main() {
  const size_t n16 = (size_t)256 * 8 * 512 * 8;   // halves
  std::vector<unsigned short> h(n16);
  unsigned short* d; float* out; char* stream;
  hipMalloc(&d, n16 * 2); hipMalloc(&out, 4096);
  unsigned stream_bytes = 2u << 20;           // 2 MB: resident in every XCD's L2 (the last lines: 128 MB = Infinity Cache, 1 GB = HBM)
  hipMalloc(&stream, (1u << 30) + 4096); hipMemset(stream, 0x3c, (1u << 30) + 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  srand(1);
  for (size_t i = 0; i < n16; ++i) { const float a = gauss(); h[i] = f2h((i & 1) ? (a - (float)(_Float16)a) * 2048.f : a); }
  hipMemcpy(d, h.data(), n16 * 2, hipMemcpyHostToDevice);
  auto run = [&](const char* name, auto kern, int nw, int mi, int ni, int iters, int dma) {
    hipLaunchKernelGGL(kern, dim3(256), dim3(nw * 64), 0, 0, (const u32x4*)d, out, iters / 8, (const char*)stream, stream_bytes - 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(nw * 64), 0, 0, (const u32x4*)d, out, iters, (const char*)stream, stream_bytes - 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * 32 * 32 * 16 * 3 * mi * ni * (double)iters * nw * 256;
    printf("%-78s %7.2f ms  %7.1f TF/s  (/3 = %5.1f)  ingest %5.1f GB/s per CU\n", name, ms, fl / ms / 1e9, fl / ms / 3e9, (double)dma * nw * 1024.0 * iters / ms / 1e6);
    fflush(stdout);
  };
  const int IT = 100000;
  // round 5: a 3x3 layer as Winograd F(2,3) along x.  Per k16 step and filter ROW a wave multiplies four transformed
  // positions, each with ITS OWN weight fragments: 32 pixel pairs x 64 channels per position = the skeleton's 1 x 2 block grid run four
  // times (24 MFMAs for 24 fragment reads, 32 KB of weight planes per workgroup); the direct kernel does a 2 x 2 grid per TAP (12 MFMAs,
  // 8 reads, 8 KB).  Work per 64 pixels x 64 channels x 16 input channels: direct 9 taps x 12 = 108 MFMAs, Winograd 3 rows x 4 x 6 = 72.
  run("direct 3x3: 64x64 wave tile, ONE acc, 8 KB DMA / tap step, barrier        ", k<2, 2, false, 1, 8, 0, 1, 1>, 8, 2, 2, IT, 1);
  run("Winograd F(2,3) 1-D: 32x64 per position, ONE acc, 8 KB DMA / position step, barrier", k<1, 2, false, 1, 8, 0, 1, 1>, 8, 1, 2, 2 * IT, 1);
  run("Winograd F(2,3) 1-D, 64 pairs x 32 channels per position (2 x 1 grid)              ", k<2, 1, false, 1, 8, 0, 1, 1>, 8, 2, 1, 2 * IT, 1);
  printf("# time per 64 px x 64 ch x k16 x 3x3: direct = 9 steps of line 1, Winograd = 12 steps of line 2 (or 3)\n");
  return 0;
}
