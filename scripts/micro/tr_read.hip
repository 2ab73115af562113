// Semantics probe of gfx950's ds_read_b64_tr_b16 (LDS transpose read of 16-bit elements).
// LDS holds img[row][col] = 256 * row + col as 16-bit integers, pitch P elements.  Every lane of a 16-lane group t supplies
// the address of img[r0 + (t >> 2)][c0 + 4 * (t & 3)]; the probe prints the four 16-bit values each lane receives.
// hipcc --offload-arch=gfx950 -O2 tr_read.hip -o /tmp/tr_read && /tmp/tr_read
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define P 144   // elements per LDS row (128 + 16: the pitch the wgrad kernel would use)

__global__ void probe(unsigned long long* out) {
  __shared__ unsigned short img[32 * P];
  for (int i = threadIdx.x; i < 32 * P; i += 64) img[i] = (unsigned short)(256 * (i / P) + (i % P));
  __syncthreads();
  typedef __attribute__((address_space(3))) unsigned short lds_u16;
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_u16*)img;   // makes the array escape (the asm below reads it)
  const int lane = threadIdx.x & 63;
  const int grp = lane >> 4, t = lane & 15;
  // group g: channels 16*(g&1).., pixels 8*(g>>1)..
  const int r0 = 8 * (grp >> 1), c0 = 16 * (grp & 1);
  const unsigned addr = lds_base + (unsigned)(((r0 + (t >> 2)) * P + c0 + 4 * (t & 3)) * 2);   // bytes
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[lane] = v;
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, 64 * 8);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  unsigned long long h[64];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) {
      unsigned v = (unsigned)((h[l] >> (16 * j)) & 0xffff);
      printf("  (r%2u,c%3u)", v >> 8, v & 255);
    }
    printf("\n");
  }
  return 0;
}
