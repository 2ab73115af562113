// Microbenchmark (companion of dma_slab.hip): the same LDS-DMA fill of 32 KB activation slabs [256 rows x 128 B] and 8 KB weight slabs
// [128 rows x 64 B], row-major with a pitch (what the pointwise kernels fetch today: rows of an [M, C] fp32 matrix / of [K, C] fp16 planes)
// against K-BLOCKED storage ([C/32][M][32]: a slab is one contiguous block).  Question: is the ~30 GB/s per CU the pipelined pointwise kernel
// takes in from L2 (profiles/r04_pw_fabric_pmc.txt) a property of the strided slabs?
// Build + run:  hipcc --offload-arch=gfx950 -O3 dma_slab_kblocked.hip -o /tmp/dsk && /tmp/dsk
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// ROWB = bytes per slab row (128: activations, 64: weight plane rows), ROWS rows per slab; blocked = 1: slab (tile, kc) contiguous
template <int ROWB, int ROWS>
__global__ __launch_bounds__(512) void k(const char* __restrict__ x, float* __restrict__ out, long long pitch, int nk, int ntiles, int rep, int blocked,
                                         int shared_tiles) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * 32768];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int SLAB = ROWB * ROWS;            // bytes
  constexpr int PIECES = SLAB / 1024;          // wave instructions per slab
  constexpr int LPR = ROWB / 16;               // lanes per row
  float acc = 0.f;
  int issued = 0;
  // shared_tiles: every workgroup of a group of `shared_tiles` walks the SAME tiles (the weight slabs all row tiles re-read; the row
  // tile the channel-tile siblings share), 1 = every workgroup its own
  // workgroup i runs on XCD i % 8: the logical id puts a group of `shared_tiles` neighbours on ONE XCD (lvc_xcd_remap in the kernels)
  const int lw = (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;
  for (int tile = lw / shared_tiles; tile < ntiles; tile += gridDim.x / shared_tiles) {
    for (int rp = 0; rp < rep; ++rp)
      for (int kc = 0; kc < nk; ++kc) {
        unsigned char* st = smem + (issued % 3) * 32768;
        for (int pc = wave; pc < PIECES; pc += 8) {
          const int r = pc * (64 / LPR) + lane / LPR;
          const char* src = blocked ? x + ((long long)tile * nk + kc) * SLAB + (long long)r * ROWB + (lane % LPR) * 16
                                    : x + ((long long)tile * ROWS + r) * pitch + (long long)kc * ROWB + (lane % LPR) * 16;
          __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(st + pc * 1024), 16, 0, 0);
        }
        ++issued;
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        acc += *reinterpret_cast<const float*>(smem + ((issued + 1) % 3) * 32768 + tid * 4);
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 12345.678f) out[tid] = acc;
}

int main() {
  char* x; float* out;
  const size_t cap = (size_t)1 << 30;
  hipMalloc(&x, cap + (1 << 20)); hipMalloc(&out, 4096);
  hipMemset(x, 0, cap);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* what, int rowb, int rows, long long pitch, int nk, int ntiles, int rep, int blocked, int shared) {
    auto launch = [&]() {
      if (rowb == 128) hipLaunchKernelGGL((k<128, 256>), dim3(256), dim3(512), 0, 0, x, out, pitch, nk, ntiles, rep, blocked, shared);
      else hipLaunchKernelGGL((k<64, 256>), dim3(256), dim3(512), 0, 0, x, out, pitch, nk, ntiles, rep, blocked, shared);
    };
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double per_wg_tiles = (double)ntiles / (256 / shared);
    const double moved = per_wg_tiles * 256.0 * nk * rep * rowb * rows;
    printf("%-58s %s: %.3f ms  %6.2f TB/s into LDS (%5.1f GB/s per CU)\n", what, blocked ? "K-blocked " : "row-major ", ms, moved / ms / 1e9, moved / ms / 1e6 / 256);
  };
  for (int blocked = 0; blocked < 2; ++blocked) {
    // activation slabs of fc1: M = 8000 rows of 12544 fp32 (pitch 50 176 B), 392 chunks, the 8 channel-tile siblings share a row tile
    run("fc1 activations (pitch 50 KB), 8 siblings per row tile", 128, 256, 50176, 392, 31, 1, blocked, 8);
    // activation slabs of res4 conv1: 33 600 rows of 1024 fp32 (pitch 4 KB), 32 chunks, 2 siblings
    run("res4 conv1 activations (pitch 4 KB), 2 siblings", 128, 256, 4096, 32, 131, 1, blocked, 2);
    // weight-plane slabs (64-byte rows; two planes = 256 rows of 64 B here): fc1 pitch 25 088 B, every workgroup the same 8 tiles
    run("fc1 weights (pitch 25 KB), all row tiles re-read them", 64, 256, 25088, 392, 8, 4, blocked, 32);
    run("res4 conv1 weights (pitch 2 KB)", 64, 256, 2048, 32, 2, 66, blocked, 128);
    run("res4 conv3 weights (pitch 512 B)", 64, 256, 512, 8, 8, 66, blocked, 32);
  }
  return 0;
}
