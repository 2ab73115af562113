// conv_pw_dma.hip -- pointwise (1x1 / FC) layers with the two-way fp16 operand split, operands streamed by LDS-DMA.
//
// Why another pointwise kernel.  conv_pw256_f16x2_kernel (conv_f16x2.hip) stages operands through registers: a thread
// loads a chunk, splits it (VALU), writes both planes to LDS (ds_write) -- one chunk ahead, with the epilogue's LDS
// transpose aliasing the stage buffers, so every tile boundary drains the pipeline.  Measured on the R50-FPN layer set
// (profiles/README.md, round 2): the 1x1 layers run at 1.3 - 3.3 TB/s and 90 - 250 TF/s, i.e. neither at the HBM nor
// at the matrix-pipe roof: they wait on latency.  Here
//   * activations arrive as RAW fp32 rows by `global_load_lds_dwordx4` (no VGPR round trip, no ds_write) into a ring of
//     three 48 KB stages that runs two chunks ahead and does not stop at tile boundaries;
//   * a wave owns 32 pixel rows x all output channels of the tile (wave grid 8 x 1), so every activation element is split
//     into its fp16 pair exactly once chip-wide, in registers, right before the MFMAs that consume it (a wave also DMAs
//     its own 32 rows: activation data needs no workgroup barrier at all, only the wave's own vmcnt);
//   * the weight planes (pre-split fp16, L2 resident) take the same DMA path; one s_barrier per chunk orders them;
//   * the epilogue goes from the accumulators straight to HBM (lane = output channel: 128-byte row segments per store),
//     touches no LDS, so the ring keeps prefetching the next tile underneath it; residual rows travel through the ring too
//     (one more chunk per 32-channel block, added to the accumulator as a product with the identity: see the epilogue).
// LDS images are lane-linear (the DMA writes base + lane * 16), bank conflicts are avoided by permuting the 16-byte
// granules of a row on the SOURCE side and on the fragment-read side with the same XOR (cdna_hip_programming.md rule 21):
//   A rows (128 B, 8 granules):  slot = G ^ ((row >> 1) & 7)      B rows (64 B, 4 granules):  slot = G ^ ((row >> 2) & 3)
// so that every 16-lane group of a ds_read_b128 covers the 64 banks exactly once.
// Arithmetic is that of conv_f16x2.hip (a = a1 + 2^-11 a2, three fp16 MFMAs per block into a main and a cross fp32
// accumulator); the stream-K split of the chunk sequence and its release/acquire hand-off are the same as well.
#include "common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// <NI, 8, 3, 1>: one workgroup of 8 waves per CU on 256-row tiles with a three-stage ring (other shapes were measured and lost:
// profiles/README.md round 2).
#define SPIN_LIMIT (1 << 24)
#define LVC_MAX_WORKERS 1024

struct ConvArgsD {
  const float* x;
  const unsigned short* w;   // [2][Kpad][Kg] fp16 planes
  const float* scale;
  const float* shift;
  const float* res;
  float* y;
  float* partials;
  int* flags;
  int H, W, C, K, stride, Ho, Wo, M, Kg, relu, res_mode, ldy, ldr;
  int tiles_n, nk, total_units, units_per_worker, nworkers, err_index, ngroup, y_bytes;
  unsigned long long* dbg;   // -DPW_DMA_TIMELINE build (scripts/probe_pw_timeline.py): cycle stamps of every 64th workgroup, else null
  long long w_plane_elems;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
}

template <int NI, int NW, int D_NS, int MI>
__global__ __launch_bounds__(NW * 64, MI == 2 ? 1 : 2) void conv_pw_dma_kernel(ConvArgsD p) {
  constexpr int WR = 32 * MI;                  // rows of a wave
  constexpr int D_BM = NW * WR, D_NT = NW * 64, D_A_BYTES = D_BM * 128;
  constexpr int GBN = 32 * NI;                 // output channels per tile
  constexpr int B_PLANE = GBN * 64;            // bytes of one weight plane chunk (GBN rows x 32 halves)
  constexpr int STAGE = D_A_BYTES + 2 * B_PLANE;
  constexpr int NBW = (4 * NI + NW - 1) / NW;  // weight DMA instructions per wave and chunk (4 * NI pieces over NW waves)
  constexpr int NPER = 4 * MI + NBW;           // DMA instructions per wave and chunk
  __shared__ __attribute__((aligned(1024))) unsigned char smem[D_NS * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 31, fh = lane >> 5;
  const int fx7 = (fi >> 1) & 7, fx3 = (fi >> 2) & 3;

  unsigned long long* const dbg = (p.dbg && tid == 0 && (blockIdx.x & 63) == 0) ? p.dbg + (blockIdx.x >> 6) * 512 : nullptr;
  int dbg_i = 1;
#define STAMP(tag) do { if (dbg && dbg_i < 255) { dbg[2 * dbg_i] = __builtin_readcyclecounter(); dbg[2 * dbg_i + 1] = (tag); ++dbg_i; dbg[0] = dbg_i; } } while (0)
  STAMP(1);
  const int lw = lvc_xcd_remap(blockIdx.x, p.nworkers);
  // ngroup > 1: the workers lw .. lw + ngroup - 1 (neighbours on one XCD) walk the same row tiles, one output-channel tile
  // each, so an activation tile comes from the fabric once and from that XCD's L2 for the others
  const int wq = p.ngroup > 1 ? lw / p.ngroup : lw;
  const int wsel = p.ngroup > 1 ? lw - wq * p.ngroup : 0;
  auto tile_n_of = [&](int tile) { return p.ngroup > 1 ? wsel : tile % p.tiles_n; };
  auto tile_m_of = [&](int tile) { return p.ngroup > 1 ? tile : tile / p.tiles_n; };
  int u = wq * p.units_per_worker;
  const int u_end = min(u + p.units_per_worker, p.total_units);
  if (u >= u_end) return;
  const bool res_layer = p.res_mode != 0;

  // ---- loader.  The chunk stream of a worker: for every tile segment its K chunks (activation rows + weight planes), then --
  // when the layer has a residual and the worker owns the tile's epilogue (its segment starts at chunk 0) -- one chunk per
  // 32-channel block of the residual rows: they take the activation slot of a stage and reach the accumulators as a product
  // with the identity (see the epilogue), so they are prefetched like any operand instead of being waited for per tile.
  const float* asrc[4 * MI];
  const float* rsrc[4 * MI];
  const unsigned short* bsrc[NBW];
  int lu = u, l_tile = u / p.nk, l_kc = u - l_tile * p.nk;
  int l_seg0 = l_kc, l_phase = 0, l_j = 0, l_nj = 0;
  bool l_done = false;
  int l_tm = -1, l_tn = -1;
  auto loader_enter = [&](int tile) {
    l_tile = tile;
    const int tm = tile_m_of(tile), tn = tile_n_of(tile);
    if (tm != l_tm) {
      l_tm = tm;
#pragma unroll
      for (int i = 0; i < 4 * MI; ++i) {
        const int r = wave * WR + i * 8 + (lane >> 3);
        int m = tm * D_BM + r;
        m = m < p.M ? m : p.M - 1;          // rows past the end read a valid row; their outputs are never stored
        const int n = m / (p.Ho * p.Wo);
        const int rem = m - n * (p.Ho * p.Wo);
        const int ho = rem / p.Wo;
        const int wo = rem - ho * p.Wo;
        const int G = (lane & 7) ^ ((r >> 1) & 7);
        asrc[i] = p.x + ((size_t)(n * p.H + ho * p.stride) * p.W + wo * p.stride) * p.C + G * 4;
        if (res_layer) {
          const int rr = p.res_mode == 2 ? (n * (p.Ho >> 1) + (ho >> 1)) * (p.Wo >> 1) + (wo >> 1) : m;
          rsrc[i] = p.res + (size_t)rr * p.ldr + G * 4;
        }
      }
    }
    if (tn != l_tn) {
      l_tn = tn;
#pragma unroll
      for (int j = 0; j < NBW; ++j) {
        const int idx = (wave * NBW + j) % (4 * NI);
        const int pl = idx / (2 * NI), rb = idx % (2 * NI);
        const int row = rb * 16 + (lane >> 2);
        const int G = (lane & 3) ^ ((row >> 2) & 3);
        bsrc[j] = p.w + (size_t)pl * p.w_plane_elems + (size_t)(tn * GBN + row) * p.Kg + G * 8;
      }
    }
    l_nj = min(NI, (p.K - tn * GBN + 31) >> 5);   // 32-channel blocks of this tile that exist
  };
  unsigned res_mask = 0;          // bit (i & 31): chunk i is a residual chunk (4 MI DMA pieces instead of NPER)
  int issued = 0, consumed = 0;   // chunks of this worker, both count from 0; ring slot = counter % D_NS
  int landed = 0;                 // chunks [0, landed) are known to have landed (a full drain happened after their issue)
  auto next_segment = [&]() {
    if (lu < u_end) { loader_enter(l_tile + 1); l_kc = 0; l_seg0 = 0; l_phase = 0; }
    else l_done = true;
  };
  auto issue_chunk = [&]() {
    if (l_done) return;
    unsigned char* st = smem + (issued % D_NS) * STAGE;
    const bool dma = true;
    if (l_phase == 0) {
      if (dma) {
#pragma unroll
        for (int i = 0; i < 4 * MI; ++i) glds16(asrc[i] + l_kc * 32, st + (wave * WR + i * 8) * 128);
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
          const int idx = (wave * NBW + j) % (4 * NI);
          const int pl = idx / (2 * NI), rb = idx % (2 * NI);
          glds16(bsrc[j] + l_kc * 32, st + D_A_BYTES + pl * B_PLANE + rb * 16 * 64);
        }
      }
      res_mask &= ~(1u << (issued & 31));
      ++l_kc; ++lu; ++issued;
      if (l_kc == p.nk || lu == u_end) {
        if (res_layer && l_seg0 == 0 && l_nj > 0) { l_phase = 1; l_j = 0; }
        else next_segment();
      }
    } else {
      if (dma) {
        const int c0 = tile_n_of(l_tile) * GBN + l_j * 32;
#pragma unroll
        for (int i = 0; i < 4 * MI; ++i) glds16(rsrc[i] + c0, st + (wave * WR + i * 8) * 128);
      }
      // a residual chunk carries only the 4 MI activation-slot pieces (no weight planes): remembered per ring position so that
      // the counted wait of the chunk BEFORE it leaves exactly these outstanding
      res_mask |= 1u << (issued & 31);
      ++issued;
      if (++l_j == l_nj) next_segment();
    }
  };
  // top of a chunk: chunk `consumed` has landed once this wave's own DMA pieces are done (loads retire in order: at most the
  // NPER pieces of the following chunk may still be outstanding) and every other wave has said the same at the barrier; past
  // the barrier every wave is done with chunk consumed - 1, whose slot takes chunk consumed + D_NS - 1
  auto chunk_top = [&]() {
    if (consumed >= landed) {
      if (D_NS > 2 && issued - consumed >= 2) {
        if ((res_mask >> ((consumed + 1) & 31)) & 1u) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * MI) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPER) : "memory");
      } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    issue_chunk();
  };
  loader_enter(l_tile);
  issue_chunk();
  if (D_NS > 2) issue_chunk();
  STAMP(2);

  float big = 0.f;   // largest |operand| met: beyond fp16 -> workspace error word

#pragma unroll 1
  while (u < u_end) {
    const int tile = u / p.nk;
    const int kc0 = u - tile * p.nk;
    const int kc1 = min(p.nk, kc0 + (u_end - u));
    const int tile_n = tile_n_of(tile);
    const int tile_m = tile_m_of(tile);
    const int m0 = tile_m * D_BM + wave * WR;   // first row of this wave
    const int n0 = tile_n * GBN;

    // main (a1 b1) and cross (a1 b2 + a2 b1, weight 2^-11) accumulators
    f32x16 acc[MI][NI], accx[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
      for (int b = 0; b < NI; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    {
#pragma unroll
      for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
          for (int e = 0; e < 16; ++e) accx[a][b][e] = 0.f;
    }

#pragma unroll 1
    for (int kc = kc0; kc < kc1; ++kc) {
      chunk_top();
      STAMP(3);
      const unsigned char* st = smem + (consumed % D_NS) * STAGE;
      const unsigned char* sa = st + (wave * WR + fi) * 128;
      const unsigned char* sb = st + D_A_BYTES + fi * 64;
      {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int G0 = s * 4 + fh * 2;
        const int bo = ((s * 2 + fh) ^ fx3) * 16;
        f16x8 ha[MI], la[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(sa + mi * 32 * 128 + ((G0 ^ fx7) * 16));
          const f32x4 a1 = *reinterpret_cast<const f32x4*>(sa + mi * 32 * 128 + (((G0 + 1) ^ fx7) * 16));
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f16 h0 = (f16)a0[e], h1 = (f16)a1[e];
            ha[mi][e] = h0; ha[mi][4 + e] = h1;
            la[mi][e] = (f16)((a0[e] - (float)h0) * 2048.f);
            la[mi][4 + e] = (f16)((a1[e] - (float)h1) * 2048.f);
            big = fmaxf(big, fmaxf(fabsf(a0[e]), fabsf(a1[e])));
          }
        }
        // weight fragments one 32-channel block at a time; the two updates of accx are an accumulate chain (D -> C of the
        // next MFMA on that accumulator: no wait states)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const f16x8 hb = *reinterpret_cast<const f16x8*>(sb + ni * 32 * 64 + bo);
          const f16x8 lb = *reinterpret_cast<const f16x8*>(sb + B_PLANE + ni * 32 * 64 + bo);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) accx[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[mi], lb, accx[mi][ni], 0, 0, 0);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[mi], hb, acc[mi][ni], 0, 0, 0);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) accx[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(la[mi], hb, accx[mi][ni], 0, 0, 0);
        }
      }
      }
      ++consumed;
    }
    u += kc1 - kc0;
    STAMP(4);
    {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[mi][ni][e] += accx[mi][ni][e] * (1.f / 2048.f);
    }

    // ---- split tiles (same protocol as conv_f16x2.hip / conv_igemm.hip): a worker that starts inside a tile hands its
    // partial sums to the worker that owns the tile's first chunk
    if (kc0 != 0) {
      float* dst = p.partials + (size_t)lw * (D_NT * 16 * MI * NI);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {
            f32x4 v = {acc[mi][ni][e4 * 4 + 0], acc[mi][ni][e4 * 4 + 1], acc[mi][ni][e4 * 4 + 2], acc[mi][ni][e4 * 4 + 3]};
            *reinterpret_cast<f32x4*>(dst + ((size_t)((mi * NI + ni) * 4 + e4) * D_NT + tid) * 4) = v;
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      landed = issued;
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(p.flags + lw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      STAMP(5);
      continue;
    }
    if (kc1 < p.nk) {
      const int last_unit = tile * p.nk + p.nk - 1;
      const int wstep = p.ngroup > 1 ? p.ngroup : 1;
      const int last_worker = (last_unit / p.units_per_worker) * wstep + wsel;
      for (int pw = lw + wstep; pw <= last_worker; pw += wstep) {
        if (tid == 0) {
          int spins = 0;
          while (__hip_atomic_load(p.flags + pw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > SPIN_LIMIT) { atomicOr(p.flags + p.err_index, 1); break; }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        const float* src = p.partials + (size_t)pw * (D_NT * 16 * MI * NI);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(src + ((size_t)((mi * NI + ni) * 4 + e4) * D_NT + tid) * 4);
              acc[mi][ni][e4 * 4 + 0] += v[0]; acc[mi][ni][e4 * 4 + 1] += v[1];
              acc[mi][ni][e4 * 4 + 2] += v[2]; acc[mi][ni][e4 * 4 + 3] += v[3];
            }
        __syncthreads();
        if (tid == 0) __hip_atomic_store(p.flags + pw, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }

    STAMP(6);
    // ---- epilogue: per-channel affine, residual, ReLU, accumulators -> HBM.
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int col = n0 + ni * 32 + fi;
      const int colc = col < p.K ? col : p.K - 1;
      const float sc = p.scale ? p.scale[colc] : 1.f;
      const float sh = p.shift ? p.shift[colc] : 0.f;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][ni][e] = acc[mi][ni][e] * sc + sh;
    }
    // Residual rows of block j sit in the activation slot of the next ring stage as raw fp32 [32 rows x 32 channels] per wave.
    // They join the accumulator as the product with the 32 x 32 identity on the same split operands (res = r1 + 2^-11 r2 to
    // 2^-23, r1 / r2 times 1.0 are exact in the fp32 accumulate): the identity's fragment is built in registers -- lane (fi, fh)
    // of k16 step s holds B[fi][s*16 + fh*8 + t], t = 0..7 -- and only block j's accumulator has a non-zero product.
    if (res_layer) {
      const int nj = min(NI, (p.K - n0 + 31) >> 5);
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        if (j < nj) {
          chunk_top();
          const unsigned char* sa = smem + (consumed % D_NS) * STAGE + (wave * WR + fi) * 128;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            f32x16 rx;
#pragma unroll
            for (int e = 0; e < 16; ++e) rx[e] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
              const int G0 = s * 4 + fh * 2;
              const f32x4 a0 = *reinterpret_cast<const f32x4*>(sa + mi * 32 * 128 + ((G0 ^ fx7) * 16));
              const f32x4 a1 = *reinterpret_cast<const f32x4*>(sa + mi * 32 * 128 + (((G0 + 1) ^ fx7) * 16));
              f16x8 ha, la, ib;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const f16 h0 = (f16)a0[e], h1 = (f16)a1[e];
                ha[e] = h0; ha[4 + e] = h1;
                la[e] = (f16)((a0[e] - (float)h0) * 2048.f);
                la[4 + e] = (f16)((a1[e] - (float)h1) * 2048.f);
                big = fmaxf(big, fmaxf(fabsf(a0[e]), fabsf(a1[e])));
              }
              const bool mine = (fi >> 3) == s * 2 + fh;
#pragma unroll
              for (int t = 0; t < 8; ++t) ib[t] = (mine && (fi & 7) == t) ? (f16)1.f : (f16)0.f;
              acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, ib, acc[mi][j], 0, 0, 0);
              rx = __builtin_amdgcn_mfma_f32_32x32x16_f16(la, ib, rx, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][j][e] += rx[e] * (1.f / 2048.f);
          }
          ++consumed;
        }
      }
    }
    // Stores go through a buffer descriptor that ends after row M - 1: rows past the end (and, through an out-of-range
    // offset, channels past K) are dropped by the bounds check; a store instruction writes 2 rows x 32 consecutive channels
    // (128-byte segments).  The chunks prefetched so far are drained FIRST and remembered as landed: the counted waits of the
    // next chunks would otherwise also wait for these 16 NI stores (one vmcnt queue).
    STAMP(7);
    {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      landed = issued;
    }
    STAMP(8);
    const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, p.y_bytes, 0x00020000);
    const float lo = p.relu == 1 ? 0.f : -INFINITY;
    const unsigned ldy4 = (unsigned)p.ldy * 4u;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const unsigned rbase = (unsigned)(m0 + mi * 32 + 4 * fh) * ldy4;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int col = n0 + ni * 32 + fi;
        const unsigned cbase = col < p.K ? rbase + (unsigned)col * 4u : 0x80000000u;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float v = acc[mi][ni][e];
          // relu == 2: torch.nn.GELU() (exact erf form), the same expression as lvc_gelu (vit.hip) -> the same bits
          v = p.relu == 2 ? v * 0.5f * (1.f + erff(v * 0.70710678118654752440f)) : fmaxf(v, lo);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yres, cbase + (unsigned)((e & 3) + 8 * (e >> 2)) * ldy4, 0, 0);
        }
      }
    }
    STAMP(10);
  }
  STAMP(9);
#undef STAMP
  if (!(big <= 65504.f)) atomicOr(p.flags + p.err_index, big < INFINITY ? 2 : 4);      // finite / non-finite: see conv3x3_halo_s1.hip
}

static int g_cus_d = 0;

// Same arguments and results as lvc_conv2d_nhwc_f16x2 (conv_f16x2.hip) for the pointwise layers it routes to its 256-row
// shape (R = S = 1, pad 0, C % 32 == 0, at least 2048 output rows); returns LVC_ERR_INVALID for anything else.
extern "C" int lvc_conv2d_nhwc_f16x2_dma(const float* x, const unsigned short* w_split, const float* scale,
                                          const float* shift, const float* residual, float* y, int N, int H, int W, int C,
                                          int K, int R, int S, int stride, int pad, int Kg, int relu, int res_mode, int ldy,
                                          int ldr, void* workspace, void* stream) {
  LVC_CHECK_ARG(x && w_split && y && workspace, "null pointer");
  LVC_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && K > 0, "non-positive dimension");
  LVC_CHECK_ARG(R == 1 && S == 1 && pad == 0, "pointwise layers only");
  LVC_CHECK_ARG(C % 32 == 0 && Kg == C, "needs C % 32 == 0 and Kg == C");
  LVC_CHECK_ARG(res_mode >= 0 && res_mode <= 2 && (res_mode == 0 || residual), "bad residual");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  if (res_mode == 2) LVC_CHECK_ARG(Ho % 2 == 0 && Wo % 2 == 0, "upsample-add needs even output size");
  ConvArgsD a;
  a.x = x; a.w = w_split; a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
  a.H = H; a.W = W; a.C = C; a.K = K; a.stride = stride; a.Ho = Ho; a.Wo = Wo;
  const long long Mll = (long long)N * Ho * Wo;
  LVC_CHECK_ARG(Mll < (1ll << 31), "too many output pixels");
  a.M = (int)Mll; a.Kg = Kg; a.relu = relu; a.res_mode = res_mode;
  a.ldy = ldy > 0 ? ldy : K; a.ldr = ldr > 0 ? ldr : K;
  LVC_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)w_split & 15) == 0 && ((uintptr_t)workspace & 15) == 0,
                "x, w_split and workspace must be 16-byte aligned");
  if (res_mode) LVC_CHECK_ARG(((uintptr_t)residual & 15) == 0 && (a.ldr & 3) == 0 && (K & 31) == 0,
                              "residual layers need a 16-byte aligned residual, ldr % 4 == 0 and K % 32 == 0");
  const long long yb = (long long)a.M * a.ldy * 4, rb = res_mode ? (long long)a.M * a.ldr * 4 : 0;
  LVC_CHECK_ARG(yb < (1ll << 31) && rb < (1ll << 31), "output / residual tensor must be smaller than 2 GiB");
  a.y_bytes = (int)yb;
  a.nk = Kg / 32;
  const int gbn = K <= 32 ? 32 : K <= 64 ? 64 : 128;
  a.tiles_n = lvc_cdiv(K, gbn);
  const int nw = 8, bm = 256;     // 8 waves x 32 rows, one workgroup per CU
  const int tiles_m = lvc_cdiv(a.M, bm);
  long long units = (long long)tiles_m * a.tiles_n * a.nk;
  LVC_CHECK_ARG(units < (1ll << 31), "iteration space too large");
  a.total_units = (int)units;
  a.w_plane_elems = (long long)(lvc_cdiv(K, 128) * 128) * Kg;   // planes are padded to 128 rows
  if (g_cus_d == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    g_cus_d = cus;
  }
  int cap = g_cus_d;   // one resident workgroup per CU
  if (cap > LVC_MAX_WORKERS) cap = LVC_MAX_WORKERS;
  const int min_units = 4;
  int workers = (int)((units + min_units - 1) / min_units);
  if (workers > cap) workers = cap;
  a.ngroup = 1;
  constexpr int ngroup_on = 1;
  const int tn = a.tiles_n;
  if (ngroup_on && workers == cap && (tn == 2 || (ngroup_on > 1 && tn <= 16 && (tn & (tn - 1)) == 0)) && cap % tn == 0 &&
      units / tn >= (long long)(cap / tn) * min_units) {
    a.ngroup = tn;
    units /= tn;
    workers = cap / tn;
    a.total_units = (int)units;
  }
  a.units_per_worker = (int)((units + workers - 1) / workers);
  a.nworkers = (int)((units + a.units_per_worker - 1) / a.units_per_worker) * a.ngroup;
  a.partials = (float*)workspace;
  a.flags = (int*)((char*)workspace + (size_t)LVC_MAX_WORKERS * 256 * 128 * 4);
  a.err_index = LVC_MAX_WORKERS + lvc_range_slot();   // the layer's own range word (common.cpp)
#ifdef PW_DMA_TIMELINE     // diagnostics build only: stamps into the upper half of the partial-tile area (never used by <= 256 workers)
  a.dbg = (unsigned long long*)((char*)workspace + (size_t)512 * 256 * 128 * 4);
#else
  a.dbg = nullptr;
#endif
  hipStream_t st = (hipStream_t)stream;
#define PW_LAUNCH(NI_, NW_, NS_) hipLaunchKernelGGL((conv_pw_dma_kernel<NI_, NW_, NS_, 1>), dim3(a.nworkers), dim3(NW_ * 64), 0, st, a)
  (void)nw;
  if (gbn == 32) PW_LAUNCH(1, 8, 3); else if (gbn == 64) PW_LAUNCH(2, 8, 3); else PW_LAUNCH(4, 8, 3);
#undef PW_LAUNCH
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}
