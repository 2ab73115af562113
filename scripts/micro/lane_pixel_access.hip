// Microbenchmark: copy rate (read + write) of an [M, C] fp32 matrix when a wave's lanes are laid out the way a transposed-MFMA
// accumulator is -- lane % 32 = pixel (row), lane / 32 = half -- against the fully coalesced form (8 lanes x 16 B per 128-byte
// row piece).  The fused conv3 -> conv1 kernel wants to read its residual and write its output straight from / into the
// accumulator layout (no LDS transposition); this says what the vector memory path makes of 32 rows x 32 B per instruction.
//   mode 0: coalesced: instruction = 8 rows x 128 B
//   mode 1: accumulator layout: instruction i of a 32-channel block = 32 rows x 32 B (lane reads 16 B at i*32 + half*16)
//   mode 2: lane owns 64 contiguous bytes (permuted weight rows): instruction i = 32 rows x 2 x 16 B (64 B apart)
//   mode 3: mode 1 reads, no writes;  mode 4: mode 0 reads, no writes
// hipcc --offload-arch=gfx950 -O3 lane_pixel_access.hip -o /tmp/lpa && /tmp/lpa
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(const float* __restrict__ x, float* __restrict__ y, int M, int C, int nblk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fi = lane & 31, fh = lane >> 5;
  const int gw = blockIdx.x * 8 + wave, nw = gridDim.x * 8;
  f32x4 sink = {0, 0, 0, 0};
  // a wave owns 32 rows at a time and walks their C / 32 channel blocks
  for (int pb = gw; pb < nblk; pb += nw) {
    const size_t row0 = (size_t)pb * 32;
    for (int cb = 0; cb < C / 32; cb += 2) {
      f32x4 v[2][4];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          size_t off;
          if (MODE == 0 || MODE == 4) off = (row0 + i * 8 + (lane >> 3)) * C + (cb + b) * 32 + (lane & 7) * 4;
          else if (MODE == 1 || MODE == 3) off = (row0 + fi) * C + (cb + b) * 32 + i * 8 + fh * 4;
          else off = (row0 + fi) * C + (cb + b) * 32 + fh * 16 + i * 4;
          v[b][i] = *reinterpret_cast<const f32x4*>(x + off);
        }
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          size_t off;
          if (MODE == 0 || MODE == 4) off = (row0 + i * 8 + (lane >> 3)) * C + (cb + b) * 32 + (lane & 7) * 4;
          else if (MODE == 1 || MODE == 3) off = (row0 + fi) * C + (cb + b) * 32 + i * 8 + fh * 4;
          else off = (row0 + fi) * C + (cb + b) * 32 + fh * 16 + i * 4;
          f32x4 o = v[b][i];
          o[0] = o[0] > 0.f ? o[0] : 0.f; o[1] = o[1] > 0.f ? o[1] : 0.f; o[2] = o[2] > 0.f ? o[2] : 0.f; o[3] = o[3] > 0.f ? o[3] : 0.f;
          if (MODE >= 3) sink += o;
          else *reinterpret_cast<f32x4*>(y + off) = o;
        }
    }
  }
  if (MODE >= 3 && sink[0] + sink[1] + sink[2] + sink[3] == 12345.678f) y[0] = sink[0];
}

template <int MODE> void run(const float* x, float* y, int M, int C, int wgs) {
  const int nblk = M / 32;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(512), 0, 0, x, y, M, C, nblk);
  hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(512), 0, 0, x, y, M, C, nblk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double bytes = (double)M * C * 4 * (MODE >= 3 ? 1 : 2);
  printf("mode %d  C %4d  wgs %4d : %.3f ms  %.2f TB/s (%s)\n", MODE, C, wgs, ms, bytes / ms * 1e-9, MODE >= 3 ? "read only" : "read + write");
}

int main() {
  const int M = 537600;
  for (int C : {256, 512}) {
    const int m = C == 256 ? M : M / 4;
    float *x, *y;
    hipMalloc(&x, (size_t)m * C * 4); hipMalloc(&y, (size_t)m * C * 4);
    hipMemset(x, 0, (size_t)m * C * 4);
    for (int wgs : {256, 512, 1024}) {
      run<0>(x, y, m, C, wgs); run<1>(x, y, m, C, wgs); run<2>(x, y, m, C, wgs); run<3>(x, y, m, C, wgs); run<4>(x, y, m, C, wgs);
    }
    hipFree(x); hipFree(y);
  }
  return 0;
}
