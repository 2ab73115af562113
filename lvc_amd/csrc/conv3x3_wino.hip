// conv3x3_wino.hip -- 3x3 / stride 1 / pad 1 layers as Winograd F(2,3) ALONG X (round 5): two output pixels of a row from four
// transformed input positions, the three filter rows accumulated directly -- 6 products per output instead of 9, i.e. two thirds of the
// MFMAs of csrc/conv3x3_halo_s1.hip at the same operand precision (single-accumulator two-way fp16 split: activations x 2^4, row-scaled
// weight planes, fp32 accumulation).
//
//     d = x[.., 2t-1 .. 2t+2] (one filter row, one input channel)        g = that row's three taps
//     V0 = d0 - d2   V1 = d1 + d2   V2 = d2 - d1   V3 = d1 - d3           U0 = g0   U1 = (g0+g1+g2)/2   U2 = (g0-g1+g2)/2   U3 = g2
//     M_p = sum over input channels and filter rows of V_p U_p            y[2t] = M0 + M1 + M2      y[2t+1] = M1 - M2 - M3
//
// Priced before it was built: scripts/winograd_error.py -- its fp32 error equals the direct evaluation's (rms x 0.99 on
// a 256 -> 256 layer; the two-dimensional F(2x2,3x3) form: x 2.5, not built); scripts/micro/winograd_skeleton.hip -- a chunk loop with
// this form's MFMA / fragment-read / weight-DMA mix runs the 3x3 work in 0.70 of the direct loop's time under the power cap
// (profiles/r05_winograd_skeleton.txt).
//
// Tile: 128 pixel PAIRS (a PH x 2 PWP pixel patch, PH PWP = 128) x 128 output channels per workgroup, 8 waves = 2 position halves x 2
// row blocks x 2 channel blocks: a wave owns 64 pairs x 64 channels for TWO of the four positions (128 accumulator registers) and reads
// 16 fragments per 24 MFMAs -- the direct kernel's ratio; the other two positions are its partner wave's, and the output transform joins
// them through LDS once per tile.  Per 16 input channels: the raw fp32 window is read to registers, transformed (V, fp32), split and
// written as two fp16 planes x four positions [(PH+2) PWP rows x 32 B] into one of two V buffers (40 KB each) while the previous chunk
// is multiplied; the transformed weight planes of a (filter row, chunk) -- 4 positions x 2 planes x 128 channels x 32 B = 32 KB, stored
// in exactly that order by lvc_amd.kernels.pack_wino -- arrive by LDS-DMA one stage ahead into a ring of two.  A stage = (chunk, filter
// row): 24 MFMAs per wave, one barrier.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define WN_NT 512
#define WN_PAIRS 128
#define WN_CH 128
#define WN_VROWS 160                       // most (PH + 2) * PWP rows of a V buffer: PH x PWP = 8 x 16 (160) or 16 x 8 (144)
#define WN_PP (WN_VROWS * 32)              // bytes of one (plane, position) image
#define WN_VBUF (8 * WN_PP)                // 40,960 B: 2 planes x 4 positions
#define WN_USLOT (8 * WN_CH * 32)          // 32,768 B: 4 positions x 2 planes x 128 channel rows x 32 B
#define WN_SMEM (2 * WN_VBUF + 2 * WN_USLOT)
#define ACT_SCALE 16.f
#define ACT_MAX 4094.f
#define LVC_MAX_WORKERS 1024

struct WinoArgs {
  const float* x;              // [N,H,W,C]
  const unsigned short* u;     // [3][C/16][4][2][Kpad][16] fp16: transformed, row-scaled weight planes (kernels.pack_wino)
  const float* scale;          // [K] row factor (x FrozenBN scale)
  const float* shift;          // [K] or null
  float* y;                    // [N,H,W,ldy]
  int* flags;
  float* partials;             // stream-K: 256 KB per worker (the conv workspace of the other kernels)
  int N, H, W, C, K, Kpad, ldy, relu;
  int PH, PWP, lg_pwp, tiles_x, tiles_y, tiles_n, ntiles, nk, err_index;
  int units_per_worker, nworkers;   // stream-K: a unit = one 16-channel chunk of one tile
  unsigned x_bytes, y_bytes;
  // PRED: a pointwise layer on top of act(conv) (the RPN predictor): y = ITS output rows (ldy floats, zeroed by the caller)
  const unsigned short* pred_w;     // [2][pred_rows][K] fp16 planes of lvc_split_weights
  const float* pred_scale;
  const float* pred_shift;
  long long pred_plane;
  int pred_K, pred_err_index;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
__device__ __forceinline__ void wn_glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)l, 16, 0, 0); }
// behind it: at most N vector-memory operations outstanding and every LDS operation of this wave complete
template <int N> __device__ __forceinline__ void wn_wait_vm_lds() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }
// One tile segment: chunks [cc0, cc1) of tile t.  Whole tiles (cc0 = 0, cc1 = nk) go straight to the epilogue.  Stream-K (SK): a worker
// whose segment does not hold the tile's first chunk hands its partial sums (still in the transformed domain: the output transform is
// linear) to the worker that does -- the protocol of conv3x3_halo_s1.hip: partials [worker][128 values][512 threads], a flag per
// worker, the owner adds the later workers' parts in worker order (deterministic) and runs the epilogue.  lw = this worker.
template <bool PRED, bool SK>
__device__ __forceinline__ void wino_tile(const WinoArgs& p, const int t, const int cc0, const int cc1, const int lw, unsigned char* const smem) {
  unsigned char* const sV = smem;
  unsigned char* const sU = smem + 2 * WN_VBUF;

  int tid_ = threadIdx.x;
  if constexpr (SK) asm volatile("" : "+v"(tid_));      // opaque per segment: what derives from it is not hoisted out of the worker's loop (and spilled)
  const int tid = tid_;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ph = wave & 1, wn = (wave >> 1) & 1, wm = wave >> 2;
  const int fi = lane & 31, fh = lane >> 5;

  // ---- tile: (image, patch row, patch column, channel tile); consecutive ids share the patch (channel tile fastest) and land on one XCD
  const int tn = t % p.tiles_n;
  int r0 = t / p.tiles_n;
  const int tx = r0 % p.tiles_x;
  r0 /= p.tiles_x;
  const int ty = r0 % p.tiles_y;
  const int img = r0 / p.tiles_y;
  const int y0 = ty * p.PH, x0 = tx * 2 * p.PWP, n0 = tn * WN_CH;
  const int VR = (p.PH + 2) * p.PWP;                  // V rows of this patch shape

  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);

  // ---- transform work of a thread per 16-channel chunk.  Round 0: ONE item = (V row tid / 4 < 128, 4-channel group tid % 4): four
  // pixels in, four positions out.  Round 1 (the V rows >= 128: the last two window rows): ONE (item, position) = (V row 128 + tid / 16,
  // position (tid / 4) % 4, group tid % 4): two pixels in, one position out -- every wave carries the same load (with whole items the
  // two waves that owned the extra rows kept the other six waiting at every barrier).  Per item the byte offset of its first pixel
  // (x0 + 2 vx - 1) and which of its pixels lie inside the image.
  const int q = tid & 3;
  const unsigned pix_b = (unsigned)p.C * 4u;
  unsigned off0, mask0, off1[2];
  int lds0, lds1;
  bool on1;
  float sgn1;
  {
    const int vrow = tid >> 2;
    const int vy = vrow >> p.lg_pwp, vx = vrow & (p.PWP - 1);
    const int iy = y0 - 1 + vy, ix = x0 + 2 * vx - 1;
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) m |= (iy >= 0 && iy < p.H && ix + j >= 0 && ix + j < p.W) ? (1u << j) : 0u;
    mask0 = m;
    off0 = (unsigned)((((long long)img * p.H + iy) * p.W + ix) * p.C + q * 4) * 4u;      // wraps for masked pixels: never used
    lds0 = vrow * 32 + (((q >> 1) ^ ((vrow >> 3) & 1)) << 4) + (q & 1) * 8;
  }
  {
    const int vrow = 128 + (tid >> 4), pz = (tid >> 2) & 3;
    on1 = vrow < VR;
    const int vy = vrow >> p.lg_pwp, vx = vrow & (p.PWP - 1);
    const int iy = y0 - 1 + vy, ix = x0 + 2 * vx - 1;
    // position -> (first pixel, second pixel, sign):  V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3
    const int ja = pz == 0 ? 0 : pz == 2 ? 2 : 1, jb = pz == 3 ? 3 : pz == 2 ? 1 : 2;
    sgn1 = pz == 1 ? 1.f : -1.f;
    const unsigned base = (unsigned)((((long long)img * p.H + iy) * p.W + ix) * p.C + q * 4) * 4u;
    const bool rowok = on1 && iy >= 0 && iy < p.H;
    off1[0] = (rowok && ix + ja >= 0 && ix + ja < p.W) ? base + (unsigned)ja * pix_b : 0x80000000u;
    off1[1] = (rowok && ix + jb >= 0 && ix + jb < p.W) ? base + (unsigned)jb * pix_b : 0x80000000u;
    lds1 = pz * WN_PP + vrow * 32 + (((q >> 1) ^ ((vrow >> 3) & 1)) << 4) + (q & 1) * 8;
  }
  f32x4 raw0[4], raw1[2];
  float big = 0.f;
  auto load0 = [&](int kc) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned off = ((mask0 >> j) & 1u) ? off0 + (unsigned)j * pix_b + (unsigned)kc * 64u : 0x80000000u;
      raw0[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, off, 0, 0));
    }
  };
  auto load1 = [&](int kc) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned off = off1[j] == 0x80000000u ? 0x80000000u : off1[j] + (unsigned)kc * 64u;
      raw1[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, off, 0, 0));
    }
  };
  // Transform = conversion (VALU) + LDS writes, both under the MFMAs of a stage.  Diagnostics builds (scripts/build_variant.sh; p2 layer
  // 1.42 ms): without the LDS writes 1.25, without conversion and writes 1.15 (the window loads themselves cost nothing), with half of
  // the fragment reads 1.39; writes issued a stage later at its top instead (conversion results parked in registers): slower (1.40 vs
  // 1.37 at equal direct-kernel time) -- kept as below.
  f16x4 pk0[4][2], pk1[2];           // round 0: [position][hi, lo]; round 1: hi, lo
  auto split = [&](f32x4 v, f16x4& h, f16x4& l) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = v[e] * ACT_SCALE;
      const f16 hh = (f16)a;
      h[e] = hh;
      l[e] = (f16)(a - (float)hh);
      big = fmaxf(big, fabsf(v[e]));
    }
  };
  auto prep0 = [&]() {
    split(raw0[0] - raw0[2], pk0[0][0], pk0[0][1]);
    split(raw0[1] + raw0[2], pk0[1][0], pk0[1][1]);
    split(raw0[2] - raw0[1], pk0[2][0], pk0[2][1]);
    split(raw0[1] - raw0[3], pk0[3][0], pk0[3][1]);
  };
  auto write0 = [&](unsigned char* vb) {
#ifndef WN_DIAG_NO_LDSWRITE
#pragma unroll
    for (int pz = 0; pz < 4; ++pz) {
      *reinterpret_cast<f16x4*>(vb + pz * WN_PP + lds0) = pk0[pz][0];
      *reinterpret_cast<f16x4*>(vb + (4 + pz) * WN_PP + lds0) = pk0[pz][1];
    }
#endif
  };
  auto prep1 = [&]() { split(raw1[0] + sgn1 * raw1[1], pk1[0], pk1[1]); };
  auto write1 = [&](unsigned char* vb) {      // no branch: V rows VR .. 159 of the smaller patch shape exist in the buffer and are never read
#ifndef WN_DIAG_NO_LDSWRITE
    *reinterpret_cast<f16x4*>(vb + lds1) = pk1[0];
    *reinterpret_cast<f16x4*>(vb + 4 * WN_PP + lds1) = pk1[1];
#endif
  };

  // ---- weight DMA: wave w brings (position, plane) image w of a stage (4 KB = four 1 KB pieces of 32 channel rows)
  const int nk = p.nk;
  const size_t u_img = (size_t)p.Kpad * 16;                        // halves per (row, chunk, position, plane) image
  const int u_row = lane >> 1, u_g = (lane & 1) ^ ((u_row >> 3) & 1);
  const unsigned short* const u_src = p.u + (size_t)wave * u_img + (size_t)(n0 + u_row) * 16 + u_g * 8;
  auto dma_u = [&](int kc, int r, int slot) {
    const unsigned short* s = u_src + (size_t)((r * nk + kc) * 8) * u_img;
#pragma unroll
    for (int j = 0; j < 4; ++j) wn_glds16(s + j * 32 * 16, sU + slot * WN_USLOT + wave * 4096 + j * 1024);
  };

  // ---- fragments.  A: V row m + r PWP of (plane, position); B: channel row of (position, plane)
  const int m_lo = wm * 64 + fi;                                  // + 32 mi
  const int b_off = (wn * 64 + fi) * 32 + ((fh ^ ((fi >> 3) & 1)) << 4);          // + 32 * 32 ni
  f32x16 acc[2][2][2];                                           // [position of the half][mi][ni]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][b][c][e] = 0.f;

  auto stage_mma = [&](const unsigned char* vb, const unsigned char* ub, int r) {
    // V row of this lane's fragment rows for filter row r, and its swizzled granule
    int arow[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int vrow = m_lo + 32 * mi + r * p.PWP;
      arow[mi] = vrow * 32 + ((fh ^ ((vrow >> 3) & 1)) << 4);
    }
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      const int pz = 2 * ph + pp;
      f16x8 ahi[2], alo[2], bhi[2], blo[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        ahi[mi] = *reinterpret_cast<const f16x8*>(vb + pz * WN_PP + arow[mi]);
#ifdef WN_DIAG_HALF_READS      // diagnostics build only (wrong results): is the stage bound by the LDS port?
        alo[mi] = ahi[mi];
#else
        alo[mi] = *reinterpret_cast<const f16x8*>(vb + (4 + pz) * WN_PP + arow[mi]);
#endif
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        bhi[ni] = *reinterpret_cast<const f16x8*>(ub + (pz * 2 + 0) * 4096 + b_off + ni * 1024);
#ifdef WN_DIAG_HALF_READS
        blo[ni] = bhi[ni];
#else
        blo[ni] = *reinterpret_cast<const f16x8*>(ub + (pz * 2 + 1) * 4096 + b_off + ni * 1024);
#endif
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[pp][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[mi], blo[ni], acc[pp][mi][ni], 0, 0, 0);
#ifndef WN_DIAG_DROP_CROSS      // gate check (scripts/perturbed_build_check.sh): without this term the layer is a 2^-11 product, and smoke / bench / e2e must fail
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[pp][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[mi], bhi[ni], acc[pp][mi][ni], 0, 0, 0);
#endif
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[pp][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[mi], bhi[ni], acc[pp][mi][ni], 0, 0, 0);
    }
  };

  // ---- prologue: the segment's first chunk transformed into V buffer 0, the weights of its stage 0 in slot 0
  dma_u(cc0, 0, 0);
  load0(cc0);
  load1(cc0);
  prep0();
  prep1();
  write0(sV);
  write1(sV);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- stages.  No branch inside a stage (a branch ends the scheduling region: the transform's VALU work must share one with the MFMAs
  // to issue in their shadow): the last chunk transforms ITSELF again into the idle buffer.
  // Transform schedule (chunk kc multiplies V(kc)):  round 0 of chunk kc+1 (V rows 0..127): loads at the top of r = 0, transform + store
  // under the MFMAs of r = 2;  round 1 (V rows 128..): loads at the top of r = 1, transform + store under the MFMAs of the NEXT chunk's
  // r = 0 -- into the buffer that stage reads, but it reads rows < 128 only (pair row + filter row 0); rows >= 128 are first needed by
  // r = 1, behind r = 0's barrier.  In-order queue of a wave (D = 4 DMA ops, L0 = 4 loads, L1 = 2):  r0: D L0 | r1: D L1 | r2: D: the
  // explicit waits before the barriers cover the DMA (other waves read that LDS); the loads' uses carry the compiler's own waits.
  // (Issuing DMA and loads as untracked inline asm -- no compiler wait before LDS reads behind a DMA -- was measured: no difference.)
  __builtin_assume(cc1 > cc0);       // at least one chunk: the zero accumulators never reach the code behind the loop
#pragma unroll 1
  for (int kc = cc0; kc < cc1; ++kc) {
    const int kr = kc - cc0;
    unsigned char* vb = sV + (kr & 1) * WN_VBUF;
    unsigned char* vnext = sV + ((kr + 1) & 1) * WN_VBUF;
    const int kn = min(kc + 1, cc1 - 1);
    const int s0 = (kr * 3) & 1;
    // r = 0: round 0 of the next chunk is requested; round 1 of THIS chunk (rows >= 128: not read before r = 1) is converted and written
    dma_u(kc, 1, 1 - s0);
    load0(kn);
    stage_mma(vb, sU + s0 * WN_USLOT, 0);
    prep1();                              // (chunk 0: the prologue's values once more)
    write1(vb);
    __builtin_amdgcn_sched_barrier(0);
    wn_wait_vm_lds<4>();                  // the DMA; round 0's loads stay in flight
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // r = 1: round 1 of the next chunk is requested
    dma_u(kc, 2, s0);
    load1(kn);
    stage_mma(vb, sU + (1 - s0) * WN_USLOT, 1);
    __builtin_amdgcn_sched_barrier(0);
    wn_wait_vm_lds<2>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // r = 2: round 0 of the next chunk is converted and written
    dma_u(kn, 0, 1 - s0);
    stage_mma(vb, sU + s0 * WN_USLOT, 2);
    prep0();
    write0(vnext);
    __builtin_amdgcn_sched_barrier(0);
    wn_wait_vm_lds<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (!(big <= ACT_MAX)) atomicOr(p.flags + p.err_index, big < INFINITY ? 2 : 4);      // finite / non-finite: see conv3x3_halo_s1.hip

  if constexpr (SK) {
    if (cc0 != 0) {          // not the tile's owner: publish the partial sums, done
      // one descriptor + scalar offsets: 32 flat stores 8 KB apart would each carry their own 64-bit address (64 registers: spills)
      const __amdgpu_buffer_rsrc_t pres = __builtin_amdgcn_make_buffer_rsrc((void*)(p.partials + (size_t)lw * (WN_NT * 128)), 0, WN_NT * 128 * 4, 0x00020000);
      const unsigned toff = (unsigned)tid * 16u;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
              const f32x4 v = {acc[a][b][c][e4 * 4 + 0], acc[a][b][c][e4 * 4 + 1], acc[a][b][c][e4 * 4 + 2], acc[a][b][c][e4 * 4 + 3]};
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), pres, toff, ((((a * 2 + b) * 2 + c) * 4 + e4)) * WN_NT * 16, 0);
            }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(p.flags + lw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return;
    }
    if (cc1 < nk) {          // the owner of a split tile: add the later workers' parts in worker order
      const int last_worker = (t * nk + nk - 1) / p.units_per_worker;
#pragma unroll 1
      for (int pw = lw + 1; pw <= last_worker; ++pw) {
        if (tid == 0) {
          int spins = 0;
          while (__hip_atomic_load(p.flags + pw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1 << 24)) { atomicOr(p.flags + p.err_index, 1); break; }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        const __amdgpu_buffer_rsrc_t pres = __builtin_amdgcn_make_buffer_rsrc((void*)(p.partials + (size_t)pw * (WN_NT * 128)), 0, WN_NT * 128 * 4, 0x00020000);
        const unsigned toff = (unsigned)tid * 16u;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              f32x4 v[4];      // one accumulator block at a time: 16 registers in flight, not 128
#pragma unroll
              for (int e4 = 0; e4 < 4; ++e4)
                v[e4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pres, toff, ((((a * 2 + b) * 2 + c) * 4 + e4)) * WN_NT * 16, 0));
#pragma unroll
              for (int e4 = 0; e4 < 4; ++e4) {
                acc[a][b][c][e4 * 4 + 0] += v[e4][0]; acc[a][b][c][e4 * 4 + 1] += v[e4][1];
                acc[a][b][c][e4 * 4 + 2] += v[e4][2]; acc[a][b][c][e4 * 4 + 3] += v[e4][3];
              }
              __builtin_amdgcn_sched_barrier(0);
            }
        __syncthreads();
        if (tid == 0) __hip_atomic_store(p.flags + pw, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }

  // ---- output transform.  This wave holds M_{2 ph}, M_{2 ph + 1}; its partner (same rows and channels, other half) the other two.
  // Half 0 finishes the EVEN pixel y[2t] = (M0 + M1) + M2 and needs M2; half 1 the odd one y[2t+1] = (M1 - M2) - M3 and needs M1:
  // each writes the block the other needs (lane-linear, 16 KB per wave) into the ring the stage loop has left.
  float* xch = reinterpret_cast<float*>(smem);
  const int pair_slot = wm * 2 + wn;
  {
    float* mine = xch + ((size_t)(ph * 4 + pair_slot) * 4) * 1024;
    const int give = ph == 0 ? 1 : 0;                            // half 0 gives M1 (its second position), half 1 gives M2 (its first)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) mine[((mi * 2 + ni) * 16 + e) * 64 + lane] = ph == 0 ? acc[1][mi][ni][e] : acc[0][mi][ni][e];
    (void)give;
  }
  __syncthreads();
  const float* theirs = xch + ((size_t)((1 - ph) * 4 + pair_slot) * 4) * 1024;
  const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, p.y_bytes, 0x00020000);
  float sc[2], sh[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int ch = n0 + wn * 64 + ni * 32 + fi;
    sc[ni] = ch < p.K ? p.scale[ch] : 0.f;
    sh[ni] = (p.shift && ch < p.K) ? p.shift[ch] : 0.f;
  }
  if constexpr (!PRED) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;       // pair of the tile
        const int py = m >> p.lg_pwp, px = m & (p.PWP - 1);
        const int oy = y0 + py, ox = x0 + 2 * px + ph;
        const bool ok = oy < p.H && ox < p.W;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const float o = theirs[((mi * 2 + ni) * 16 + e) * 64 + lane];
          float v = ph == 0 ? (acc[0][mi][ni][e] + acc[1][mi][ni][e]) + o : (o - acc[0][mi][ni][e]) - acc[1][mi][ni][e];
          v = v * sc[ni] + sh[ni];
          if (p.relu) v = v > 0.f ? v : 0.f;
          const int ch = n0 + wn * 64 + ni * 32 + fi;
          const unsigned off = (ok && ch < p.K) ? (unsigned)((((long long)img * p.H + oy) * p.W + ox) * p.ldy + ch) * 4u : 0xfffffff0u;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yres, off, 0, 0);
        }
      }
  } else {
    // ---- a pointwise layer on top (the RPN predictor; csrc/conv3x3_halo_s1.hip has the same epilogue): the workgroup holds act(conv)
    // for 128 of the K hidden channels of its 256 pixels = a 128-deep slice of that layer's contraction.  The hidden tile goes to LDS
    // ([pixel][channel] fp32; pixel row 2 m + parity), every wave multiplies 32 pixels x 128 channels by the layer's fp16 planes (read
    // from L1 / L2; two-accumulator split: main a1 b1, cross (a2 b1 + a1 b2) x 2^-11) and adds its 32 x pred_K block to the ZEROED
    // output atomically: K <= 256 -> at most two addends per element, the sum does not depend on their order.
    constexpr int CSS = WN_CH + 4;
    float bigp = 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int e = 0; e < 16; ++e)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const float o = theirs[((mi * 2 + ni) * 16 + e) * 64 + lane];
          float v = ph == 0 ? (acc[0][mi][ni][e] + acc[1][mi][ni][e]) + o : (o - acc[0][mi][ni][e]) - acc[1][mi][ni][e];
          v = v * sc[ni] + sh[ni];
          if (p.relu) v = v > 0.f ? v : 0.f;
          bigp = (fabsf(v) > bigp || v != v) ? fabsf(v) : bigp;
          acc[0][mi][ni][e] = v;
        }
    __syncthreads();                      // every wave has read its partner's block: the ring may be overwritten
    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) Cs[(2 * m + ph) * CSS + wn * 64 + ni * 32 + fi] = acc[0][mi][ni][e];
      }
    __syncthreads();
    const unsigned short* bsrcp = p.pred_w + (size_t)fi * p.K + n0 + fh * 8;
    f16x8 pbh = *reinterpret_cast<const f16x8*>(bsrcp), pbl = *reinterpret_cast<const f16x8*>(bsrcp + p.pred_plane);
    f32x16 pm, px2;
#pragma unroll
    for (int e = 0; e < 16; ++e) { pm[e] = 0.f; px2[e] = 0.f; }
    const float* arow = Cs + (wave * 32 + fi) * CSS + fh * 8;
#pragma unroll 1
    for (int ks = 0; ks < WN_CH / 16; ++ks) {
      const int kn = ks + 1 < WN_CH / 16 ? ks + 1 : ks;
      const f16x8 nbh = *reinterpret_cast<const f16x8*>(bsrcp + kn * 16);
      const f16x8 nbl = *reinterpret_cast<const f16x8*>(bsrcp + p.pred_plane + kn * 16);
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(arow + ks * 16);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(arow + ks * 16 + 4);
      f16x8 ah, al;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float a = j < 4 ? a0[j & 3] : a1[j & 3];
        const f16 hh = (f16)a;
        ah[j] = hh;
        al[j] = (f16)((a - (float)hh) * 2048.f);
      }
      pm = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, pbh, pm, 0, 0, 0);
      px2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, pbh, px2, 0, 0, 0);
      px2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, pbl, px2, 0, 0, 0);
      pbh = nbh; pbl = nbl;
    }
    // the 32 x 32 result block back through the wave's OWN rows of the tile (no other wave touches them), then out two pixels x pred_K
    // outputs per instruction
    float* rblk = Cs + wave * 32 * CSS;
#pragma unroll
    for (int e = 0; e < 16; ++e) rblk[((e & 3) + 8 * (e >> 2) + 4 * fh) * 32 + fi] = pm[e] + px2[e] * (1.f / 2048.f);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (fi < p.pred_K) {
      const float psc = p.pred_scale ? p.pred_scale[fi] : 1.f;
      const float psh = (n0 == 0 && p.pred_shift) ? p.pred_shift[fi] : 0.f;     // the bias once: with the first slice
#pragma unroll 2
      for (int rr = 0; rr < 32; rr += 2) {
        const int r = wave * 32 + rr + fh;                                      // tile pixel row: pair r / 2, parity r & 1
        const int m = r >> 1;
        const int oy = y0 + (m >> p.lg_pwp), ox = x0 + 2 * (m & (p.PWP - 1)) + (r & 1);
        const float v = rblk[(rr + fh) * 32 + fi];
        if (oy < p.H && ox < p.W) unsafeAtomicAdd(p.y + ((size_t)((long long)img * p.H + oy) * p.W + ox) * p.ldy + fi, v * psc + psh);
      }
    }
    if (!(bigp <= 65504.f)) atomicOr(p.flags + p.pred_err_index, bigp < INFINITY ? 2 : 4);      // the pointwise layer's own range word
  }
}

template <bool PRED>
__global__ __launch_bounds__(WN_NT, 2) void conv3x3_wino_kernel(WinoArgs p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[WN_SMEM];
  wino_tile<PRED, false>(p, lvc_xcd_remap(blockIdx.x, p.ntiles), 0, p.nk, 0, smem);
}

// Stream-K form: one resident workgroup per CU, worker w takes the units [w upw, (w + 1) upw) of the (tile, chunk) list -- consecutive
// workers sit on one XCD (lvc_xcd_remap) and walk neighbouring tiles.  For the maps whose tile count does not fill whole rounds of the
// 256 CUs (p3 / res3: 2.4 - 4.8 rounds, res4 / p4: 1.03): every CU gets the same number of stages.
template <bool PRED, bool SPLIT>     // SPLIT = false: units_per_worker is a multiple of nk (whole tiles per worker: no hand-off code)
__global__ __launch_bounds__(WN_NT, 2) void conv3x3_wino_sk_kernel(WinoArgs p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[WN_SMEM];
  const int lw = lvc_xcd_remap(blockIdx.x, p.nworkers);
  int u = lw * p.units_per_worker;
  const int u_end = min(u + p.units_per_worker, p.ntiles * p.nk);
#pragma unroll 1
  while (u < u_end) {
    const int t = u / p.nk, cc0 = u - t * p.nk;
    const int cc1 = min(p.nk, cc0 + (u_end - u));
    if constexpr (SPLIT) wino_tile<PRED, true>(p, t, cc0, cc1, lw, smem);
    else wino_tile<PRED, false>(p, t, 0, p.nk, lw, smem);
    __syncthreads();
    u += cc1 - cc0;
  }
}

// y = act(conv3x3(x, w) * scale + shift), stride 1, pad 1, as Winograd F(2,3) along x.  x [N,H,W,C] fp32 NHWC (C % 16 == 0), u = the
// transformed weight planes [3][C/16][4][2][Kpad][16] fp16 with Kpad % 128 == 0 and scale [K] = their row factors (x the layer's
// per-channel scale) as lvc_amd.kernels.pack_wino makes them; y [N,H,W,ldy].  An activation window value |V| > 4094 (or NaN) raises the
// layer's range word in `workspace` (the conv workspace of the other kernels; only its error words are used).
// Work distribution of lvc_conv3x3_nhwc_wino*: 0 (default) one workgroup per tile; 1 stream-K (the (tile, chunk) list split evenly over one
// resident workgroup per CU, split tiles completed through the workspace); 2 persistent workgroups on whole tiles.  Measured on the layers
// the detector routes here (scripts/probe_wino_modes.py, alternated): p2 1.33-1.44 / 1.38-1.39 / 1.34-1.36 ms, p3 0.361-0.374 / 0.397-0.406 /
// 0.370-0.380, res3 0.116-0.119 / 0.118-0.120 / 0.115-0.118, res4 0.131-0.133 / 0.117-0.120 / 0.135-0.143 (direct kernel there: 0.109-0.115):
// a hand-off moves 128 accumulator values per thread (256 KB per worker each way) and costs more than the partial last round it removes.
static int g_wino_streamk = 0;
extern "C" void lvc_set_wino_streamk(int mode) { g_wino_streamk = mode; }

static int wn_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus = n / 8 * 8;
    if (cus < 8) cus = 8;
  }
  return cus;
}

static int wino_launch(const float* x, const unsigned short* u, const float* scale, const float* shift, float* y, int N, int H, int W, int C,
                       int K, int Kpad, int relu, int ldy, const unsigned short* pred_w, const float* pred_scale, const float* pred_shift,
                       int pred_K, int pred_rows, int pred_slot, void* workspace, void* stream) {
  LVC_CHECK_ARG(x && u && scale && y && workspace, "null pointer");
  LVC_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && K > 0, "non-positive dimension");
  LVC_CHECK_ARG(C % 16 == 0 && Kpad % WN_CH == 0 && Kpad >= K, "needs C % 16 == 0 and weight planes padded to 128 rows");
  WinoArgs a;
  a.x = x; a.u = u; a.scale = scale; a.shift = shift; a.y = y;
  a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.Kpad = Kpad; a.ldy = ldy > 0 ? ldy : K; a.relu = relu;
  const long long xb = (long long)N * H * W * C * 4, yb = (long long)N * H * W * a.ldy * 4;
  LVC_CHECK_ARG(xb < (1ll << 31) && yb < (1ll << 31), "tensors must be smaller than 2 GiB");
  a.x_bytes = (unsigned)xb; a.y_bytes = (unsigned)yb;
  // patch shape: 8 rows x 16 pairs (32 pixels), or 16 rows x 8 pairs for maps that waste less on it
  auto waste = [&](int phh, int pwp) { return (long long)lvc_cdiv(H, phh) * phh * lvc_cdiv(W, 2 * pwp) * 2 * pwp; };
  if (waste(16, 8) < waste(8, 16)) { a.PH = 16; a.PWP = 8; a.lg_pwp = 3; } else { a.PH = 8; a.PWP = 16; a.lg_pwp = 4; }
  a.tiles_x = lvc_cdiv(W, 2 * a.PWP); a.tiles_y = lvc_cdiv(H, a.PH); a.tiles_n = lvc_cdiv(K, WN_CH);
  const long long nt = (long long)N * a.tiles_x * a.tiles_y * a.tiles_n;
  LVC_CHECK_ARG(nt < (1ll << 31), "too many tiles");
  a.ntiles = (int)nt;
  a.nk = C / 16;
  a.flags = (int*)((char*)workspace + (size_t)LVC_MAX_WORKERS * 256 * 128 * 4);
  a.err_index = LVC_MAX_WORKERS + lvc_range_slot();
  a.pred_w = pred_w; a.pred_scale = pred_scale; a.pred_shift = pred_shift; a.pred_K = pred_K;
  a.pred_plane = (long long)pred_rows * K;
  a.pred_err_index = LVC_MAX_WORKERS + pred_slot;
  if (pred_w) {
    LVC_CHECK_ARG(K % WN_CH == 0 && K <= 2 * WN_CH, "the pointwise layer on top needs 128 or 256 hidden channels (at most two slices per output)");
    LVC_CHECK_ARG(pred_K >= 1 && pred_K <= 32 && pred_rows >= 32 && pred_slot >= 0 && pred_slot < lvc_range_slots(), "bad pointwise layer");
  }
  // stream-K / persistent forms only on request (see g_wino_streamk)
  a.partials = (float*)workspace;
  const long long nunits = (long long)a.ntiles * a.nk;
  const int cus = wn_cus() < LVC_MAX_WORKERS / 2 ? wn_cus() : LVC_MAX_WORKERS / 2;      // 256 KB of partial sums per worker in a 128 MB region
  bool sk = g_wino_streamk != 0 && nunits < (1ll << 31) && a.ntiles % cus != 0;
  if (sk) {
    constexpr int min_units = 4;        // a segment restarts the pipeline: at least four chunks (12 stages) per worker
    long long workers = (nunits + min_units - 1) / min_units;
    if (workers > cus) workers = cus;
    a.units_per_worker = (int)((nunits + workers - 1) / workers);
    const bool whole = g_wino_streamk == 2;
    if (whole) a.units_per_worker = (a.units_per_worker + a.nk - 1) / a.nk * a.nk;      // whole tiles per worker: persistent workgroups, no hand-off
    a.nworkers = (int)((nunits + a.units_per_worker - 1) / a.units_per_worker);
    if (pred_w) {
      if (whole) hipLaunchKernelGGL((conv3x3_wino_sk_kernel<true, false>), dim3(a.nworkers), dim3(WN_NT), 0, (hipStream_t)stream, a);
      else hipLaunchKernelGGL((conv3x3_wino_sk_kernel<true, true>), dim3(a.nworkers), dim3(WN_NT), 0, (hipStream_t)stream, a);
    } else {
      if (whole) hipLaunchKernelGGL((conv3x3_wino_sk_kernel<false, false>), dim3(a.nworkers), dim3(WN_NT), 0, (hipStream_t)stream, a);
      else hipLaunchKernelGGL((conv3x3_wino_sk_kernel<false, true>), dim3(a.nworkers), dim3(WN_NT), 0, (hipStream_t)stream, a);
    }
  } else {
    a.units_per_worker = a.nk; a.nworkers = a.ntiles;
    if (pred_w) hipLaunchKernelGGL(conv3x3_wino_kernel<true>, dim3(a.ntiles), dim3(WN_NT), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(conv3x3_wino_kernel<false>, dim3(a.ntiles), dim3(WN_NT), 0, (hipStream_t)stream, a);
  }
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}

extern "C" int lvc_conv3x3_nhwc_wino(const float* x, const unsigned short* u, const float* scale, const float* shift, float* y, int N, int H,
                                     int W, int C, int K, int Kpad, int relu, int ldy, void* workspace, void* stream) {
  return wino_launch(x, u, scale, shift, y, N, H, W, C, K, Kpad, relu, ldy, nullptr, nullptr, nullptr, 0, 0, 0, workspace, stream);
}

// ... with a pointwise layer `pred` (<= 32 outputs) on top of act(conv): y [N,H,W,ldy] = ITS outputs, ZEROED by the caller; the hidden map is
// never written (each workgroup adds its 128-channel slice of the contraction atomically: K in {128, 256}, at most two addends per
// element).  pred_w: [2][pred_rows][K] fp16 planes of lvc_split_weights (pred_rows >= 32), pred_scale / pred_shift [pred_K] or NULL;
// pred_slot: the pointwise layer's range word (a hidden value beyond fp16's range).  The arguments of lvc_conv3x3_nhwc_f16_levels_pred.
extern "C" int lvc_conv3x3_nhwc_wino_pred(const float* x, const unsigned short* u, const float* scale, const float* shift, float* y, int N,
                                          int H, int W, int C, int K, int Kpad, int relu, int ldy, const unsigned short* pred_w,
                                          const float* pred_scale, const float* pred_shift, int pred_K, int pred_rows, int pred_slot,
                                          void* workspace, void* stream) {
  LVC_CHECK_ARG(pred_w, "null pointer");
  return wino_launch(x, u, scale, shift, y, N, H, W, C, K, Kpad, relu, ldy, pred_w, pred_scale, pred_shift, pred_K, pred_rows, pred_slot, workspace,
                     stream);
}
