// Implicit-GEMM forward kernel with HALF-STEP LDS stages (round 5): the 128 x 128 tile of igemm.hip, the same v_mfma_f32_32x32x2_f32
// products in the same order, but an LDS stage holds 16 of the 32 floats of a K-step -- 2 x 256 rows x 20 floats = 40 KB instead of
// 72 KB, 16 staging registers instead of 32 -- so that THREE workgroups share a CU instead of two.
//
// Why: the cycle counters of round 5 (profiles/r05_prologue_ab.txt) show a deep-K tile spending ~22 k cycles in front of its K loop and
// 23-33 k behind it (address arithmetic, first operand loads, the LDS-staged epilogue), issued by ONE wave per SIMD between the
// co-resident workgroup's MFMAs; with two workgroups per CU the matrix pipe is fed by a single wave per SIMD for a fifth of the time.
// A third workgroup keeps two waves per SIMD in their K loops while the third is in its prologue or epilogue.
//
// K order: igemm.hip feeds MFMA (group q = 0..3, e = 0..3) of a 32-wide step with k = 4q + e from lanes 0..31 and k = 16 + 4q + e from
// lanes 32..63.  Half-step h of that step therefore holds k = 8h .. 8h + 7 in LDS columns 0..7 and k = 16 + 8h .. 16 + 8h + 7 in columns
// 8..15: its two groups are q = 2h, 2h + 1 of the full step -- the same pairs in the same sequence, the same bits.
#include <stdlib.h>
#include "nbm_common.h"
#include "igemm_params.h"
#include <type_traits>

namespace {

using nbm_igemm::IgemmParams;
constexpr int BM = 128, BN = 128, WM = 64, WN = 64, MT = 2, NT = 2;
constexpr int P16 = 20;      // floats per LDS row (16 + 4 pad: conflict-free ds_read_b128 of 16 consecutive rows)
enum { EPI_STD = 0 };

template <int OCC>
__global__ __launch_bounds__(256, OCC) void igemm_h16_kernel(const IgemmParams p) {
  constexpr int STAGES = 1, EPI = EPI_STD;          // (names the shared epilogue text expects: the tile goes out in two halves)
  constexpr bool ROWS = false;
  __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * P16];
  float* As = lds;                   // [2][BM][P16]
  float* Bs = lds + 2 * BM * P16;    // [2][BN][P16]

  // ---- XCD-aware tile id (bijective for any grid size), as in igemm.hip
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tile_m = wg / p.n_tiles, tile_n = wg - tile_m * p.n_tiles;
  const int bm0 = tile_m * BM, bn0 = tile_n * BN;
  const int g = blockIdx.z;
  const float* __restrict__ xg = p.x + (long long)g * p.x_gs;
  const float* __restrict__ wgp = p.w + (long long)g * p.w_gs;
  auto row_pixel = [&](int m) -> long long { return m; };      // (ROWS form of the epilogue text: unused)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  const int lrow = lane & 31, lh = lane >> 5;

  // ---- staging assignment: thread -> rows r0 + 64 i (i = 0, 1) of the A and of the W tile, 16-byte chunk c of the stage's 16 floats;
  // chunk c holds k = koff .. koff + 3 (+ 8 for the second half-step) of the 32-wide step
  const int c = tid & 3, r0 = tid >> 2;
  const int koff = c < 2 ? 4 * c : 16 + 4 * (c - 2);
  long long blk_base;
  {
    const int m0 = bm0 < p.M ? bm0 : 0;
    const int b = (int)nbm_fdiv((unsigned)m0, p.fd_howo), rem = m0 - b * p.HoWo;
    const int oy = (int)nbm_fdiv((unsigned)rem, p.fd_wo), ox = rem - oy * p.Wo;
    blk_base = ((long long)(b * p.H + oy * p.stride - p.pad) * p.W + (ox * p.stride - p.pad)) * p.x_ld;
  }
  unsigned a_rel[2], b_rel[2];
  unsigned long long a_taps[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = bm0 + r0 + 64 * i;
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    const int b = (int)nbm_fdiv((unsigned)mm, p.fd_howo), rem = mm - b * p.HoWo;
    const int oy = (int)nbm_fdiv((unsigned)rem, p.fd_wo), ox = rem - oy * p.Wo;
    const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
    const long long base = ((long long)(b * p.H + iy0) * p.W + ix0) * p.x_ld;
    a_rel[i] = ((unsigned)(base - blk_base) + koff) * 4u;        // bytes; rows ascend with m: never negative
    unsigned rowm = 0u, colm = 0u;
    unsigned long long mk = 0ull;
    for (int r = 0; r < p.kh; ++r) rowm |= ((unsigned)(iy0 + r) < (unsigned)p.H ? 1u : 0u) << r;
    for (int s2 = 0; s2 < p.kw; ++s2) colm |= ((unsigned)(ix0 + s2) < (unsigned)p.W ? 1u : 0u) << s2;
    for (int r = 0; r < p.kh; ++r) mk |= ((rowm >> r) & 1u) ? (unsigned long long)colm << (r * p.kw) : 0ull;
    a_taps[i] = ok ? mk : 0ull;
    const int n = bn0 + r0 + 64 * i;
    b_rel[i] = n < p.N ? (unsigned)(n * p.w_ld + koff) * 4u : 0x80000000u;
  }
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xg + blk_base), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wgp), 0, 0x7ffffff0, 0x00020000);

  f32x4 ra[2], rb[2];
  // cursor of the next half-step to load: filter tap (channel chunk OUTER, tap INNER like igemm.hip), first channel of the 32-wide step, half
  int cur_r = 0, cur_s = 0, cur_c0 = 0, cur_h = 0;
  auto load_tiles = [&]() {
    const int tap = cur_r * p.kw + cur_s;
    const unsigned a_soff = (unsigned)((((long long)cur_r * p.W + cur_s) * p.x_ld + cur_c0 + 8 * cur_h) * 4);
    const unsigned b_soff = (unsigned)((tap * p.Cin + cur_c0 + 8 * cur_h) * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned vo = ((a_taps[i] >> tap) & 1ull) ? a_rel[i] : 0x80000000u;
      ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo, a_soff, 0));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, b_rel[i], b_soff, 0));
    if (cur_h ^= 1, cur_h == 0) {
      if (++cur_s == p.kw) { cur_s = 0; if (++cur_r == p.kh) { cur_r = 0; cur_c0 += 32; } }
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<f32x4*>(As + (buf * BM + r0 + 64 * i) * P16 + c * 4) = ra[i];
      *reinterpret_cast<f32x4*>(Bs + (buf * BN + r0 + 64 * i) * P16 + c * 4) = rb[i];
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto mfma_group = [&](const float* Ab, const float* Bb, int q) {
    f32x4 a[MT], b[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * P16 + q * 4);
#pragma unroll
    for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * P16 + q * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
  };

  // One half-step (ONE barrier): group 0 + the LDS writes of the next half-step, group 1 + the global loads of the one after it
  auto h_step = [&](int t, auto store_c, auto load_c) {
    constexpr bool STORE = decltype(store_c)::value, LOAD = decltype(load_c)::value;
    const int cur = t & 1;
    __syncthreads();
    const float* Ab = As + (cur * BM + wm0 + lrow) * P16 + lh * 8;
    const float* Bb = Bs + (cur * BN + wn0 + lrow) * P16 + lh * 8;
    mfma_group(Ab, Bb, 0);
    if constexpr (STORE) {
      store_lds(cur ^ 1);
      __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);
#pragma unroll
      for (int z = 0; z < 4; ++z) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(Ab, Bb, 1);
    if constexpr (LOAD) {
      load_tiles();
      __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);
#pragma unroll
      for (int z = 0; z < 4; ++z) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  {
    using T = std::true_type;
    using F = std::false_type;
    const int S = 2 * p.nk;            // >= 2
    load_tiles();
    store_lds(0);
    load_tiles();
    int t = 0;
    for (; t + 2 < S; ++t) h_step(t, T{}, T{});
    h_step(t, T{}, F{});
    ++t;
    h_step(t, F{}, F{});
  }
  __syncthreads();

#define NBM_EPI_LDS_FLOATS (2 * (BM + BN) * P16)
#include "igemm_epilogue.inc"
#undef NBM_EPI_LDS_FLOATS
}

}  // namespace

int nbm_igemm::h16_launch(const IgemmParams& p0, int groups, int occ, hipStream_t st) {
  IgemmParams p = p0;
  p.m_tiles = (p.M + BM - 1) / BM;
  p.n_tiles = (p.N + BN - 1) / BN;
  const dim3 grid(p.m_tiles * p.n_tiles, 1, groups);
  if (occ >= 4) hipLaunchKernelGGL((igemm_h16_kernel<4>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((igemm_h16_kernel<3>), grid, dim3(256), 0, st, p);
  return nbm_launch_status();
}
