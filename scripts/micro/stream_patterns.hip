// Microbenchmark: HBM read rate of the access patterns a pointwise-conv A loader can use on an [M, C] fp32 matrix
// (C = 256): 128-byte pieces per row chunk by chunk (what the GEMM kernels do), the same with all chunks of a tile in
// flight, and whole rows.  hipcc --offload-arch=gfx950 -O3 stream_patterns.hip -o /tmp/stream_patterns && /tmp/stream_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int ROWS>
__global__ __launch_bounds__(512) void k(const float* __restrict__ x, float* __restrict__ out, int M, int C, int ntiles) {
  const int tid = threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const size_t base = (size_t)tile * ROWS * C;
    if (MODE == 0) {  // chunk by chunk: 8 lanes x 16 B per row, rows ROWS/..; wait between chunks
      const int q = tid & 7, r0 = tid >> 3;
      for (int kc = 0; kc < C / 32; ++kc) {
        f32x4 v[ROWS / 64];
#pragma unroll
        for (int j = 0; j < ROWS / 64; ++j) v[j] = *reinterpret_cast<const f32x4*>(x + base + (size_t)(r0 + 64 * j) * C + kc * 32 + q * 4);
#pragma unroll
        for (int j = 0; j < ROWS / 64; ++j) acc += v[j];
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (MODE == 1) {  // all chunks of the tile in flight (same addresses per instruction as MODE 0)
      const int q = tid & 7, r0 = tid >> 3;
      f32x4 v[ROWS / 64][8];
#pragma unroll
      for (int kc = 0; kc < 8; ++kc)
#pragma unroll
        for (int j = 0; j < ROWS / 64; ++j) v[j][kc] = *reinterpret_cast<const f32x4*>(x + base + (size_t)(r0 + 64 * j) * C + kc * 32 + q * 4);
#pragma unroll
      for (int kc = 0; kc < 8; ++kc)
#pragma unroll
        for (int j = 0; j < ROWS / 64; ++j) acc += v[j][kc];
    } else {  // whole rows: a wave instruction reads one full 1 KB row
      const int n = ROWS * C / 4 / 512;
#pragma unroll 8
      for (int i = 0; i < n; ++i) acc += *reinterpret_cast<const f32x4*>(x + base + (size_t)(tid + 512 * i) * 4);
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[tid] = acc[0];
}

int main() {
  const int M = 8 * 200 * 336, C = 256;
  float *x, *out;
  hipMalloc(&x, (size_t)M * C * 4); hipMalloc(&out, 4096);
  hipMemset(x, 0, (size_t)M * C * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-46s %.3f ms  %.2f TB/s\n", name, ms, (double)M * C * 4 / ms / 1e9);
  };
  for (int wg : {256, 512, 1024}) {
    printf("workgroups %d\n", wg);
    run("chunked 128 B pieces, 128-row tiles", [&] { hipLaunchKernelGGL((k<0, 128>), dim3(wg), dim3(512), 0, 0, x, out, M, C, M / 128); });
    run("chunked 128 B pieces, 256-row tiles", [&] { hipLaunchKernelGGL((k<0, 256>), dim3(wg), dim3(512), 0, 0, x, out, M, C, M / 256); });
    run("all 8 chunks in flight, 128-row tiles", [&] { hipLaunchKernelGGL((k<1, 128>), dim3(wg), dim3(512), 0, 0, x, out, M, C, M / 128); });
    run("all 8 chunks in flight, 256-row tiles", [&] { hipLaunchKernelGGL((k<1, 256>), dim3(wg), dim3(512), 0, 0, x, out, M, C, M / 256); });
    run("whole rows, 128-row tiles", [&] { hipLaunchKernelGGL((k<2, 128>), dim3(wg), dim3(512), 0, 0, x, out, M, C, M / 128); });
  }
  return 0;
}
