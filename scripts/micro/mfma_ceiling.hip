// Microbenchmark: what the matrix pipe sustains on THIS part under its power cap, as a function of
//   * operand type (fp16 / bf16) and operand data (zeros, N(0,1) "hi-plane-like" values, full-entropy "lo-plane-like" values),
//   * LDS fragment reads per MFMA (0 = operands stay in registers; 2/3 = the 64x64 wave tile of conv3x3_halo_h2_kernel;
//     1/2 = a 64x128 wave tile; 1/3 = a 128x128 wave tile),
//   * waves per SIMD (2 = 8-wave workgroups, 1 = 4-wave workgroups).
// Every launch runs ~30-60 ms so that the clock has settled to the power budget; reported: TFLOP/s of
// v_mfma_f32_32x32x16_{f16,bf16} (2 * 32 * 32 * 16 flop each) over all 256 CUs.  This is the ceiling any split-precision
// kernel built on these instructions can be priced against (three MFMAs per fp32-accurate product: divide by 3).
// Build + run:  hipcc --offload-arch=gfx950 -O3 mfma_ceiling.hip -o /tmp/mfma_ceiling && /tmp/mfma_ceiling
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
template <int MI, int NI, bool BF, int READS, int NW, int TWOACC>
__global__ __launch_bounds__(NW * 64) void k(const u32x4* __restrict__ src, float* __restrict__ out, int iters) {
  __shared__ u32x4 lds[NW * 64 * 8];   // 8 x 16 B per lane, lane-linear: conflict-free ds_read_b128
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
  unsigned short* d; float* out;
  hipMalloc(&d, n16 * 2); hipMalloc(&out, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* data_names[] = {"zeros", "N(0,1) (hi plane)", "residual plane (a - fp16(a)) * 2^11"};
  for (int bf = 0; bf < 2; ++bf)
    for (int data = 0; data < 3; ++data) {
      srand(1);
      for (size_t i = 0; i < n16; ++i) {
        float v = 0.f;
        if (data == 1) v = gauss();
        if (data == 2) {
          const float a = gauss();
          if (bf) { unsigned short t = f2bf(a); unsigned u = (unsigned)t << 16; float hi; __builtin_memcpy(&hi, &u, 4); v = (a - hi) * 256.f; }
          else { v = (a - (float)(_Float16)a) * 2048.f; }
        }
        h[i] = bf ? f2bf(v) : f2h(v);
      }
      hipMemcpy(d, h.data(), n16 * 2, hipMemcpyHostToDevice);
      auto run = [&](const char* name, auto kern, int nw, int mi, int ni, int iters) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(nw * 64), 0, 0, (const u32x4*)d, out, iters / 8);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(nw * 64), 0, 0, (const u32x4*)d, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = 2.0 * 32 * 32 * 16 * 3 * mi * ni * (double)iters * nw * 256;
        printf("%-5s %-38s %-52s %7.2f ms  %7.1f TF/s  (/3 = %5.1f)\n", bf ? "bf16" : "fp16", data_names[data], name, ms, fl / ms / 1e9, fl / ms / 3e9);
        fflush(stdout);
      };
      const int IT = 100000;   // 2x2 blocks, 8 waves: 24 MFMAs per SIMD and iteration = 0.5 us at 1.5 GHz -> ~50 ms
      if (bf) {
        run("regs only, 2x2, 2 waves/SIMD, two acc", k<2, 2, true, 0, 8, 1>, 8, 2, 2, IT);
        run("LDS 2/3 reads/MFMA (64x64 wave tile), 2/SIMD, two acc", k<2, 2, true, 1, 8, 1>, 8, 2, 2, IT);
        run("LDS 1/2 reads/MFMA (64x128 wave tile), 2/SIMD, one acc", k<2, 4, true, 1, 8, 0>, 8, 2, 4, IT / 2);
      } else {
        run("regs only, 2x2, 2 waves/SIMD, two acc", k<2, 2, false, 0, 8, 1>, 8, 2, 2, IT);
        run("regs only, 2x2, 1 wave/SIMD, two acc", k<2, 2, false, 0, 4, 1>, 4, 2, 2, 2 * IT);
        run("LDS 2/3 reads/MFMA (64x64 wave tile), 2/SIMD, two acc", k<2, 2, false, 1, 8, 1>, 8, 2, 2, IT);
        run("LDS 2/3 reads/MFMA (64x64 wave tile), 2/SIMD, one acc", k<2, 2, false, 1, 8, 0>, 8, 2, 2, IT);
        run("LDS 1/2 reads/MFMA (64x128 wave tile), 2/SIMD, one acc", k<2, 4, false, 1, 8, 0>, 8, 2, 4, IT / 2);
        run("LDS 1/3 reads/MFMA (128x128 wave tile), 1/SIMD, one acc", k<4, 4, false, 1, 4, 0>, 4, 4, 4, IT / 2);
      }
    }
  return 0;
}
