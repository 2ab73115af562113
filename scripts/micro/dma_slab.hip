// Microbenchmark: rate at which a CU fills LDS with [256 rows x 128 B] slabs of an [M, C] fp32 matrix by
// global_load_lds_dwordx4 (the activation stream of csrc/conv_pw_dma.hip), as a function of the row pitch C (power-of-two
// pitches put a slab's rows on few memory channels), of the re-read factor (tiles re-read `rep` times: L2 hits) and of the
// chunks in flight.  Build + run:  hipcc --offload-arch=gfx950 -O3 dma_slab.hip -o /tmp/dma_slab && /tmp/dma_slab
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int DEPTH>
__global__ __launch_bounds__(512) void k(const float* __restrict__ x, float* __restrict__ out, int M, int C, int ntiles, int rep) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * 32768];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = C / 32;
  float acc = 0.f;
  int issued = 0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const float* src[4];
    for (int i = 0; i < 4; ++i) {
      const int r = wave * 32 + i * 8 + (lane >> 3);
      src[i] = x + (size_t)(tile * 256 + r) * C + ((lane & 7) ^ ((r >> 1) & 7)) * 4;
    }
    for (int rp = 0; rp < rep; ++rp)
      for (int kc = 0; kc < nk; ++kc) {
        unsigned char* st = smem + (issued % 3) * 32768;
        for (int i = 0; i < 4; ++i)
          __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[i] + kc * 32), (lds_ptr_t)(st + (wave * 32 + i * 8) * 128), 16, 0, 0);
        ++issued;
        if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        acc += *reinterpret_cast<const float*>(smem + ((issued + 1) % 3) * 32768 + tid * 4);
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 12345.678f) out[tid] = acc;
}

int main() {
  float *x, *out;
  const size_t cap = (size_t)1 << 30;
  hipMalloc(&x, cap + (1 << 20)); hipMalloc(&out, 4096);
  hipMemset(x, 0, cap);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](int C, size_t bytes, int rep, int depth) {
    const int M = (int)(bytes / ((size_t)C * 4)) / 256 * 256;
    const int ntiles = M / 256;
    auto launch = [&]() {
      if (depth == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, x, out, M, C, ntiles, rep);
      else if (depth == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, x, out, M, C, ntiles, rep);
      else hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, x, out, M, C, ntiles, rep);
    };
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double moved = (double)M * C * 4 * rep;
    printf("C %5d  footprint %5.0f MB  rep %d  depth %d : %.3f ms  %6.2f TB/s into LDS  (%5.1f GB/s per CU)\n", C, (double)M * C * 4 / 1e6, rep,
           depth, ms, moved / ms / 1e9, moved / ms / 1e6 / 256);
  };
  for (int depth : {1, 2, 3})
    for (int C : {128, 256, 512, 1024, 1056, 2048, 2080}) run(C, (size_t)1 << 30, 1, depth);
  for (int C : {256, 1024, 1056}) run(C, (size_t)32 << 20, 8, 3);     // 32 MB: L2 / MALL resident
  for (int C : {256, 1024, 1056}) run(C, (size_t)1 << 30, 8, 3);      // each tile re-read 8 times back to back
  return 0;
}
