// conv3x3_halo.hip -- 3x3 / stride 1 / pad 1 NHWC convolution on the bf16 matrix cores (3-way operand split), with the
// activation window staged ONCE per 32-channel chunk and reused by all nine taps.
//
// Why a second kernel: profiles/README.md ("power-bound") -- on gfx950 the split-precision convolution sits at the
// socket power cap, and what the generic implicit-GEMM kernel (conv_bf16x3.hip) spends its joules on besides the
// MFMAs is moving operands: per 128x128x32 chunk it pulls 16 KB of fp32 activations + 24 KB of bf16 weight planes from
// L2, splits 4096 activations on the VALU and writes 48 KB to LDS -- and for a 3x3 filter the nine taps of one channel
// chunk re-stage the same activations nine times, shifted by one pixel.  72 % of the flops of the Faster R-CNN
// R50-FPN forward are 3x3 stride-1 convolutions (res2-5 conv2, FPN output convs, RPN conv), so this kernel:
//   * tiles the OUTPUT as 2-D patches of PH x PW <= 256 pixels (host picks PH, PW per layer to minimise patch count)
//     by 128 output channels; 8 waves as 4 (M) x 2 (N), wave tile 64 x 64 (24 MFMAs per 12 fragment reads);
//   * per 32-channel chunk loads the (PH+2) x (PW+2) halo window once (zero outside the image), splits it once into
//     three bf16 planes in LDS, and runs the nine taps as nine k-steps whose A fragments are read from the same halo
//     at a tap-dependent pixel offset: activation traffic, split work and LDS writes drop ~7x per flop;
//   * streams only the weight planes per tap (24 KB, double-buffered), amortised over 256 output pixels instead of 128.
// Everything else -- split-precision product (6 MFMAs, fp32 accumulate, smallest terms first), packed weight layout
// [3][Kpad][Kg] with k order (c/32, r, s, c%32), stream-K workers with partial-tile hand-off, fused
// scale/shift/residual/ReLU epilogue through LDS -- is the contract of conv_bf16x3.hip, so both kernels are
// interchangeable on a 3x3 layer (tests/test_gpu_kernels.py runs both against the same oracle).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define HM 256           // output pixels per tile (patch area <= HM)
#define LROW 40          // bf16 elements per LDS row (32 + 8 pad = 80 B: conflict-free ds_read_b128)
#define HALO_MAX 384     // halo pixels per tile: 6 x 512 threads x one float4
#define NJ 6
#define PLANE_A (HALO_MAX * LROW)
#define NT 512
#define SPIN_LIMIT (1 << 24)

struct HaloArgs {
  const float* x;
  const unsigned short* w;   // [3][Kpad][Kg] bf16 planes
  const float* scale;
  const float* shift;
  const float* res;
  float* y;
  float* partials;
  int* flags;
  int N, H, W, C, K, relu, res_mode, ldy, ldr;
  int PH, PW, HW, HP, MP;    // patch rows / cols, halo row pitch (PW + 2), halo pixels, patch pixels
  int tiles_x, tiles_y, tiles_n, nk, total_units, units_per_worker, nworkers, err_index;
  int x_bytes, w_plane_bytes;
};

__device__ __forceinline__ void split3h(float a, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)a;
  const float r1 = a - (float)h;
  m = (__bf16)r1;
  const float r2 = r1 - (float)m;
  l = (__bf16)r2;
}

// NI = 32-column MFMA blocks per wave: NI = 2 -> 128 output channels per tile (wave tile 64 x 64), NI = 1 -> 64 output
// channels per tile (wave tile 64 x 32; the 64-channel res2 layers, which would waste half of a 128-wide tile).
template <int NI>
__global__ __launch_bounds__(NT, 2) void conv3x3_halo_kernel(HaloArgs p) {
  constexpr int HN = 64 * NI;
  constexpr int PLANE_B = HN * LROW;
  constexpr int A_ELEMS = 3 * PLANE_A;
  constexpr int B_ELEMS = 3 * PLANE_B;
  constexpr int STAGE_BYTES = (A_ELEMS + 2 * B_ELEMS) * 2;   // 92,160 + 61,440
  constexpr int CS_STRIDE = HN + 4;
  constexpr int CS_BYTES = HM * CS_STRIDE * 4;                // 135,168
  constexpr int SMEM_BYTES = STAGE_BYTES > CS_BYTES ? STAGE_BYTES : CS_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[SMEM_BYTES];
  __bf16* sA = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* sB = sA + A_ELEMS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;   // wave tile: 64 (M) x 64 (N)
  const int fi = lane & 31, fh = lane >> 5;
  // halo staging: thread = (pixel slot, float4 slot); the slot permutation keeps the 80-byte-pitch stores conflict-free
  const int q = tid & 7;
  const int arid = tid >> 3;
  const int hrow = (arid & 1) * 4 + ((arid >> 1) & 3) + (arid >> 3) * 8;    // halo pixels hrow + 64*j
  // weight staging: 16-byte pieces, HN rows x 4 pieces per plane.  NI = 2: 512 pieces per plane, thread t stages piece t
  // of every plane.  NI = 1: 256 pieces per plane, thread t stages piece t % 256 of plane t / 256 and (t < 256) of plane 2.
  const int brid = (NI == 2 ? tid : (tid & 255)) >> 2;
  const int b_row = (brid & 1) * 4 + ((brid >> 1) & 3) + (brid >> 3) * 8;
  const int b_q4 = tid & 3;
  constexpr int NB = NI == 2 ? 3 : 2;                 // staged pieces per thread
  const int b_pl0 = NI == 2 ? 0 : (tid >> 8);         // plane of piece i: b_pl0 + (NI == 2 ? i : 2 * i)

  const int lw = lvc_xcd_remap(blockIdx.x, p.nworkers);
  int u = lw * p.units_per_worker;
  const int u_end = min(u + p.units_per_worker, p.total_units);

  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 3 * p.w_plane_bytes, 0x00020000);

  // fragment offsets (bf16 elements).  A: halo pixel of output pixel m at tap (0,0); rows past the patch read pixel 0
  int a_frag[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int m = wm * 64 + mi * 32 + fi;
    const int mm = m < p.MP ? m : 0;
    const int py = mm / p.PW, px = mm - py * p.PW;
    a_frag[mi] = (py * p.HW + px) * LROW + fh * 8;
  }
  const int b_frag = (wn * 32 * NI + fi) * LROW + fh * 8;

  while (u < u_end) {
    const int tile = u / p.nk;
    const int cc0 = u - tile * p.nk;
    const int cc1 = min(p.nk, cc0 + (u_end - u));
    const int tile_n = tile % p.tiles_n;
    const int tile_m = tile / p.tiles_n;
    const int tx = tile_m % p.tiles_x;
    const int t2 = tile_m / p.tiles_x;
    const int ty = t2 % p.tiles_y;
    const int img = t2 / p.tiles_y;
    const int y0 = ty * p.PH, x0 = tx * p.PW;
    const int n0 = tile_n * HN;

    unsigned a_off[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int h = hrow + 64 * j;
      const int hy = h / p.HW, hx = h - hy * p.HW;
      const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
      const bool ok = h < p.HP && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
      a_off[j] = ok ? (unsigned)(((img * p.H + yy) * p.W + xx) * p.C + q * 4) * 4u : 0x80000000u;
    }
    const unsigned b_off = (unsigned)((n0 + b_row) * (9 * p.C) + b_q4 * 8) * 2u;

    f32x4 areg[NJ];
    u32x4 breg[NB];
    auto load_A = [&](int cc) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        areg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, a_off[j], cc * 128, 0));
    };
    auto store_A = [&]() {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (hrow + 64 * j < p.HP) {
          bf16x4 h, m, l;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            __bf16 hh, mm, ll;
            split3h(areg[j][e], hh, mm, ll);
            h[e] = hh; m[e] = mm; l[e] = ll;
          }
          const int o = (hrow + 64 * j) * LROW + q * 4;
          *reinterpret_cast<bf16x4*>(sA + o) = h;
          *reinterpret_cast<bf16x4*>(sA + PLANE_A + o) = m;
          *reinterpret_cast<bf16x4*>(sA + 2 * PLANE_A + o) = l;
        }
      }
    };
    int ld_step = cc0 * 9;
    const int step_end = cc1 * 9;
    auto load_B = [&]() {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int pl = b_pl0 + (NI == 2 ? i : 2 * i);
        if (pl < 3)
          breg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  wres, b_off + (unsigned)(pl * p.w_plane_bytes), ld_step * 64, 0));
      }
      if (ld_step + 1 < step_end) ++ld_step;
    };
    auto store_B = [&](int buf) {
      __bf16* sb = sB + buf * B_ELEMS;
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int pl = b_pl0 + (NI == 2 ? i : 2 * i);
        if (pl < 3) *reinterpret_cast<u32x4*>(sb + pl * PLANE_B + b_row * LROW + b_q4 * 8) = breg[i];
      }
    };

    f32x16 acc[2][NI];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < NI; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    // Fragments are single-buffered (48 VGPRs): with two waves per SIMD the sibling wave's MFMAs cover this wave's
    // fragment reads, and 96 VGPRs of double-buffered fragments next to 64 accumulators + 36 prefetch registers
    // spill (the first version of this kernel reloaded its LDS addresses from scratch every tap).
    bf16x8 fa[2][3], fb[NI][3];   // [mi|ni][plane]
    auto read_frags = [&](int tap_off, const __bf16* sb, int s2) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          fa[mi][pl] = *reinterpret_cast<const bf16x8*>(sA + pl * PLANE_A + a_frag[mi] + tap_off + s2 * 16);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          fb[ni][pl] = *reinterpret_cast<const bf16x8*>(sb + pl * PLANE_B + b_frag + ni * 32 * LROW + s2 * 16);
    };
    // six split-precision terms, smallest first; the four accumulators are independent MFMA chains
    auto mfma_group = [&]() {
      constexpr int TA[6] = {0, 2, 1, 0, 1, 0};
      constexpr int TB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][TA[t]], fb[ni][TB[t]], acc[mi][ni], 0, 0, 0);
    };

    // Per tap: two k16 groups out of the current weight buffer, the next tap's weight planes written to the idle
    // buffer and the tap after that requested from L2, then ONE barrier (all reads of this buffer done, writes of the
    // other visible).  Per 32-channel chunk one extra barrier pair around the halo refill (~1 % of the chunk).
    int cur = 0;
    auto tap_step = [&](int tap_off) {
      const __bf16* sb = sB + cur * B_ELEMS;
      read_frags(tap_off, sb, 0);
      store_B(cur ^ 1);
      mfma_group();
      __builtin_amdgcn_sched_barrier(0);
      read_frags(tap_off, sb, 1);
      load_B();
      mfma_group();
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      cur ^= 1;
    };

    load_A(cc0);
    load_B();
    store_A();
    store_B(0);
    load_B();
    if (cc0 + 1 < cc1) load_A(cc0 + 1);
    __syncthreads();
    for (int cc = cc0; cc < cc1; ++cc) {
      int tap_off = 0;
      for (int r = 0; r < 3; ++r) {
        tap_step(tap_off);
        tap_step(tap_off + LROW);
        tap_step(tap_off + 2 * LROW);
        tap_off += p.HW * LROW;
      }
      if (cc + 1 < cc1) {
        store_A();                  // every wave passed the last tap's barrier: nobody reads the old halo
        if (cc + 2 < cc1) load_A(cc + 2);
        __syncthreads();
      }
    }
    __syncthreads();
    u += cc1 - cc0;

    // ---- split tiles: a worker that does not own the tile's first chunk hands its partial sums to the one that does
    if (cc0 != 0) {
      float* dst = p.partials + (size_t)lw * (NT * 32 * NI);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {
            f32x4 v = {acc[mi][ni][e4 * 4 + 0], acc[mi][ni][e4 * 4 + 1], acc[mi][ni][e4 * 4 + 2], acc[mi][ni][e4 * 4 + 3]};
            *reinterpret_cast<f32x4*>(dst + ((size_t)((mi * NI + ni) * 4 + e4) * NT + tid) * 4) = v;
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(p.flags + lw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      continue;
    }
    if (cc1 < p.nk) {
      const int last_unit = tile * p.nk + p.nk - 1;
      const int last_worker = last_unit / p.units_per_worker;
      for (int pw = lw + 1; pw <= last_worker; ++pw) {
        if (tid == 0) {
          int spins = 0;
          while (__hip_atomic_load(p.flags + pw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > SPIN_LIMIT) { atomicOr(p.flags + p.err_index, 1); break; }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        const float* src = p.partials + (size_t)pw * (NT * 32 * NI);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(src + ((size_t)((mi * NI + ni) * 4 + e4) * NT + tid) * 4);
              acc[mi][ni][e4 * 4 + 0] += v[0]; acc[mi][ni][e4 * 4 + 1] += v[1];
              acc[mi][ni][e4 * 4 + 2] += v[2]; acc[mi][ni][e4 * 4 + 3] += v[3];
            }
        __syncthreads();
        if (tid == 0) __hip_atomic_store(p.flags + pw, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }

    // ---- epilogue through LDS: tile row r is patch pixel (r / PW, r % PW)
    float* Cs = reinterpret_cast<float*>(smem_raw);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
          const int col = wn * 32 * NI + ni * 32 + fi;
          Cs[row * CS_STRIDE + col] = acc[mi][ni][e];
        }
    __syncthreads();
    constexpr int C4 = HN / 4;
    constexpr int RPI = NT / C4;
    const int c4 = tid % C4, rsub = tid / C4;
    const int col = n0 + c4 * 4;
    if (col < p.K) {
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
      if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + col);
      if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + col);
      // one copy of the row loop per (residual mode, ReLU): with the modes tested inside, every row ended in the compiler's
      // vmcnt(0) lgkmcnt(0) -- its LDS read and the acknowledgement of the previous row's store one after the other (measured on
      // conv3x3_halo_s1.hip, scripts/probe_halo_timeline.py)
      auto rows = [&](auto rm_tag, auto relu_tag) {
        constexpr int RM = decltype(rm_tag)::value;
        constexpr bool RELU = decltype(relu_tag)::value;
#pragma unroll 4
        for (int it = 0; it < HM / RPI; ++it) {
          const int r = it * RPI + rsub;
          const int py = r / p.PW, px = r - py * p.PW;
          const int yy = y0 + py, xx = x0 + px;
          if (r < p.MP && yy < p.H && xx < p.W) {
            const size_t row = (size_t)(img * p.H + yy) * p.W + xx;
            f32x4 v = *reinterpret_cast<const f32x4*>(Cs + r * CS_STRIDE + c4 * 4);
            v = v * sc + sh;
            if (RM == 1) {
              v += *reinterpret_cast<const f32x4*>(p.res + row * p.ldr + col);
            } else if (RM == 2) {
              const size_t ro = ((size_t)(img * (p.H >> 1) + (yy >> 1)) * (p.W >> 1) + (xx >> 1));
              v += *reinterpret_cast<const f32x4*>(p.res + ro * p.ldr + col);
            }
            if (RELU) {
              v[0] = v[0] > 0.f ? v[0] : 0.f; v[1] = v[1] > 0.f ? v[1] : 0.f;
              v[2] = v[2] > 0.f ? v[2] : 0.f; v[3] = v[3] > 0.f ? v[3] : 0.f;
            }
            *reinterpret_cast<f32x4*>(p.y + row * p.ldy + col) = v;
          }
        }
      };
      using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>; using T2 = std::integral_constant<int, 2>;
      if (p.res_mode == 0) { if (p.relu) rows(T0{}, std::true_type{}); else rows(T0{}, std::false_type{}); }
      else if (p.res_mode == 1) { if (p.relu) rows(T1{}, std::true_type{}); else rows(T1{}, std::false_type{}); }
      else { if (p.relu) rows(T2{}, std::true_type{}); else rows(T2{}, std::false_type{}); }
    }
    __syncthreads();
  }
}

#define LVC_MAX_WORKERS 1024
static int g_cus_halo = 0;

extern "C" int lvc_conv2d_nhwc_bf16x3(const float* x, const unsigned short* w_split, const float* scale,
                                      const float* shift, const float* residual, float* y, int N, int H, int W, int C,
                                      int K, int R, int S, int stride, int pad, int Kg, int relu, int res_mode, int ldy,
                                      int ldr, void* workspace, void* stream);

// Patch shape for an H x W output: PH * PW <= 256 pixels, (PH + 2) * (PW + 2) <= HALO_MAX halo pixels, fewest patches
// (every patch costs a full 256-row MFMA tile whatever its fill); ties go to the smaller halo.
static void pick_patch(int H, int W, int* PH, int* PW) {
  long long best_tiles = -1;
  int best_halo = 0, bh = 1, bw = 8;
  for (int pw = 4; pw <= 128; ++pw)
    for (int ph = 1; ph * pw <= HM; ++ph) {
      const int halo = (ph + 2) * (pw + 2);
      if (halo > HALO_MAX) break;
      const long long tiles = (long long)lvc_cdiv(H, ph) * lvc_cdiv(W, pw);
      if (best_tiles < 0 || tiles < best_tiles || (tiles == best_tiles && halo < best_halo)) {
        best_tiles = tiles; best_halo = halo; bh = ph; bw = pw;
      }
    }
  *PH = bh; *PW = bw;
}

// 3x3 / stride 1 / pad 1 specialisation of lvc_conv2d_nhwc_bf16x3: same arguments minus (R, S, stride, pad), same
// packed weights, same workspace, same result contract.
// Test hooks of this kernel (tests/test_gpu_kernels.py::test_conv3x3_halo_matches_cpu; no environment lookups on the launch path):
// a fixed patch shape ph x pw (0 = the kernel's choice) and `force` = no fallback to the generic kernel on small maps.
static int g_halo_test_ph = 0, g_halo_test_pw = 0, g_halo_test_force = 0;
extern "C" void lvc_set_halo_test_hooks(int ph, int pw, int force) { g_halo_test_ph = ph; g_halo_test_pw = pw; g_halo_test_force = force; }

extern "C" int lvc_conv3x3_nhwc_bf16x3(const float* x, const unsigned short* w_split, const float* scale,
                                       const float* shift, const float* residual, float* y, int N, int H, int W, int C,
                                       int K, int Kg, int relu, int res_mode, int ldy, int ldr, void* workspace,
                                       void* stream) {
  LVC_CHECK_ARG(x && w_split && y && workspace, "null pointer");
  LVC_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && K > 0, "non-positive dimension");
  LVC_CHECK_ARG(C % 32 == 0 && Kg == 9 * C, "needs C % 32 == 0 and Kg == 9*C");
  LVC_CHECK_ARG(res_mode >= 0 && res_mode <= 2 && (res_mode == 0 || residual), "bad residual");
  if (res_mode == 2) LVC_CHECK_ARG(H % 2 == 0 && W % 2 == 0, "upsample-add needs even output size");
  HaloArgs a;
  a.x = x; a.w = w_split; a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
  a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.relu = relu; a.res_mode = res_mode;
  a.ldy = ldy > 0 ? ldy : K; a.ldr = ldr > 0 ? ldr : K;
  LVC_CHECK_ARG((K & 3) == 0 && (a.ldy & 3) == 0 && (res_mode == 0 || (a.ldr & 3) == 0), "K, ldy, ldr must be multiples of 4");
  LVC_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)w_split & 15) == 0 && ((uintptr_t)y & 15) == 0 &&
                    ((uintptr_t)workspace & 15) == 0, "pointers must be 16-byte aligned");
  pick_patch(H, W, &a.PH, &a.PW);
  if (g_halo_test_ph > 0 && g_halo_test_pw > 0 && g_halo_test_ph * g_halo_test_pw <= HM && (g_halo_test_ph + 2) * (g_halo_test_pw + 2) <= HALO_MAX) {
    a.PH = g_halo_test_ph; a.PW = g_halo_test_pw;      // test hook (lvc_set_halo_test_hooks): a fixed patch shape
  }
  a.HW = a.PW + 2; a.HP = (a.PH + 2) * a.HW; a.MP = a.PH * a.PW;
  a.tiles_x = lvc_cdiv(W, a.PW); a.tiles_y = lvc_cdiv(H, a.PH);
  const int ni = K <= 64 ? 1 : 2;   // 64-wide tiles for the 64-channel layers
  const int HN = 64 * ni;
  a.tiles_n = lvc_cdiv(K, HN);
  a.nk = C / 32;
  // small feature maps (p5 / p6 / a single image's res5): too few 256-pixel patches to fill 256 CUs before stream-K
  // splits every tile eight ways -- the generic 128-row kernel measures faster there (scripts/probe_halo.py)
  if ((long long)N * a.tiles_x * a.tiles_y * a.tiles_n < 128 && !g_halo_test_force)
    return lvc_conv2d_nhwc_bf16x3(x, w_split, scale, shift, residual, y, N, H, W, C, K, 3, 3, 1, 1, Kg, relu, res_mode,
                                  ldy, ldr, workspace, stream);
  long long units = (long long)N * a.tiles_x * a.tiles_y * a.tiles_n * a.nk;
  LVC_CHECK_ARG(units < (1ll << 31), "iteration space too large");
  a.total_units = (int)units;
  const long long xb = (long long)N * H * W * C * 4, wb = (long long)(lvc_cdiv(K, 128) * 128) * Kg * 2;   // planes are padded to 128 rows
  LVC_CHECK_ARG(xb < (1ll << 31) && 3 * wb < (1ll << 31), "input / weight tensor must be smaller than 2 GiB");
  a.x_bytes = (int)xb; a.w_plane_bytes = (int)wb;
  if (g_cus_halo == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    g_cus_halo = cus;
  }
  int cap = g_cus_halo;  // one worker per CU: 150 KB of LDS per workgroup
  if (cap > LVC_MAX_WORKERS) cap = LVC_MAX_WORKERS;
  int workers = (int)(units < cap ? units : cap);
  a.units_per_worker = (int)((units + workers - 1) / workers);
  a.nworkers = (int)((units + a.units_per_worker - 1) / a.units_per_worker);
  a.partials = (float*)workspace;
  a.flags = (int*)((char*)workspace + (size_t)LVC_MAX_WORKERS * 256 * 128 * 4);
  a.err_index = LVC_MAX_WORKERS;
  if (ni == 1)
    hipLaunchKernelGGL(conv3x3_halo_kernel<1>, dim3(a.nworkers), dim3(NT), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(conv3x3_halo_kernel<2>, dim3(a.nworkers), dim3(NT), 0, (hipStream_t)stream, a);
  LVC_CHECK_LAUNCH();
  return LVC_OK;
}
