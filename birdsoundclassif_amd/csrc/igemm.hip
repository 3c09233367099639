// Implicit-GEMM convolution / batched GEMM on the gfx950 fp32 matrix core
// (v_mfma_f32_32x32x2_f32: exact fp32 fmaf chain, 64 FLOP/clk/SIMD).
//
//   C[m][n] = sum_k A[m][k] * W[n][k]
//
// A is gathered on the fly from the NHWC activation (im2col never materialised), W is KRSC.
// Block tile BM x BN x 32; 4 waves (one per SIMD), each owning (WM/32) x (WN/32) accumulator tiles of
// 32x32.  Operands are staged global -> registers -> LDS ([rows][36] floats: the 4-float pad makes every
// ds_read_b128 lane group hit 16 distinct 16-byte slots), double buffered with ONE barrier per K-step:
// the global loads of step k+1 are issued before the MFMAs of step k and written to the other buffer
// after them.
//
// K order inside a 32-wide step is free as long as A and W agree: lane (row i, half h) feeds k = 16h + j
// to MFMA j, so each lane reads 16 contiguous floats (4 x ds_read_b128) per operand row.
// For kh*kw > 1 the K loop runs channel-chunk OUTER, filter tap INNER, so the 3x3 halo of a block
// (3 rows x 130 px x 128 B) is re-read 9x from L2 instead of streaming the whole 384-channel pixel per tap.
//
// Workgroup -> tile mapping is XCD-aware (bijective chunking of the grid over the 8 L2s) so the N-tiles
// of one M-tile, which share the gathered A rows, run on the same XCD.
#include <stdlib.h>
#include "nbm_common.h"
#include "igemm_params.h"
#include <type_traits>

namespace {

constexpr int BK = 32;
constexpr int PITCH = 36;  // floats per LDS row (32 + 4 pad)

enum { A_FAST = 0, A_GENERIC = 1 };
enum { EPI_STD = 0 };

using nbm_igemm::IgemmParams;

// STAGES = 2: the deep-K pipeline (double-buffered LDS, 2 workgroups per CU).  STAGES = 1: short-K layers (K <= 256: 1x1
// convolutions whose time is output / residual traffic, not MFMA): single LDS buffer (36.8 KB) and an epilogue staged in
// two halves, so THREE workgroups fit a CU and their load / compute / store phases overlap instead of serialising.
// K = 64 layers still sit at ~76 TF/s whatever N is (SQ counters: every wave spends 19 % of its 44 k-cycle life in its own
// MFMAs, 36 % waiting for the matrix pipe behind its two SIMD partners, 28 % in waitcnt / barriers).  Measured and dropped
// for them (scripts/lateral_probe.py, 64 -> 384 @188x512 x 64: 4.04 ms standalone, 6.45 ms with the merge epilogue):
// 64-wide tiles at four workgroups per CU (4.19 / 6.05 ms); a persistent workgroup that keeps the weight tile in LDS and
// has the next pixel tile's loads in flight under the current epilogue (3.99 / 7.96 ms), with or without a start stagger
// of the two co-resident workgroups (3.85-3.91 ms).
// Deep K, measured and dropped (round 3): a 256 x 128 tile (waves of 128 x 64: 128 accumulator registers in the AGPR file, 25 % fewer
// LDS reads and half the barriers per MFMA, 110 KB of LDS -> ONE workgroup = one wave per SIMD) -- 2048 -> 3072 @24576: 2.34 ms
// against 2.23 (132 vs 138.5 TF/s), 1024 -> 1536 @98304: 2.50 / 2.30, 1024 -> 256 @24x64: 0.42 / 0.39, small grids far worse
// (512 -> 2048 @12x32: 0.60 / 0.44).  The second co-resident wave per SIMD is worth more than the larger tile.
// A PERSISTENT form of this kernel (512 workgroups walk their XCD's chunk of tiles; address set-up and the first operand loads of the
// next tile issued before the epilogue of the current one; 189 VGPRs) was built to hide the per-tile ramp (~1.9 K-steps by a fit over
// three shapes) and measured 2 % SLOWER everywhere: 24576 x 3072 x 2048 134.3 -> 131.2 TF/s, 98304 x 1536 x 1024 135.7 -> 133.0,
// 98304 x 1024 x 512 128.5 -> 125.3, 393216 x 128 x 512 121.6 -> 119.1 (scripts/blas_compare.py).  With two workgroups per CU the
// ramp of one is already covered by the other; what the vendor library gains on these shapes comes from stream-K (no partial round
// of workgroups), not from pipelining across tiles.  Not kept.
template <int BM, int BN, int WM, int WN, int AMODE, int EPI, int STAGES = 2, bool ROWS = false>
__global__ __launch_bounds__(256, BM > 128 ? 1 : STAGES == 1 ? 3 : 2) void igemm_kernel(const IgemmParams p) {
  constexpr int MT = WM / 32, NT = WN / 32;
  constexpr int WAVES_N = BN / WN;
  constexpr int AR = BM / 32, BR = BN / 32;  // rows per thread in the staging pass
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");

  __shared__ __attribute__((aligned(16))) float lds[STAGES * (BM + BN) * PITCH];
  if constexpr (STAGES == 2) nbm_stagger_priority();
  float* As = lds;                       // [STAGES][BM][PITCH]
  float* Bs = lds + STAGES * BM * PITCH; // [STAGES][BN][PITCH]

  // ---- XCD-aware tile id (bijective for any grid size)
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  int tile_m = wg / p.n_tiles, tile_n = wg - tile_m * p.n_tiles;
  // dense pixel index of GEMM row m, or -1 (ROWS only)
  auto row_pixel = [&](int m) -> long long {
    if (p.rows_mode == 1) return p.rows[m];
    const int t = p.rows[m >> 4];
    if (t < 0) return -1;
    const int thw = p.rows_TH * p.rows_TW, k = m & 15;
    const int b = (int)nbm_fdiv((unsigned)t, p.fd_rows_thw), rem = t - b * thw;
    const int ty = (int)nbm_fdiv((unsigned)rem, p.fd_rows_tw), tx = rem - ty * p.rows_TW;
    const int y = 2 * ty - 1 + (k >> 2), x = 2 * tx - 1 + (k & 3);
    if ((unsigned)y >= (unsigned)p.H || (unsigned)x >= (unsigned)p.W) return -1;
    return ((long long)b * p.H + y) * p.W + x;
  };
  if constexpr (ROWS) {
    if (p.rows_blocks) {          // device-side count: the filled blocks lead; spread them over the XCDs (see wino_fused.hip)
      const int j = bid >> 3;
      tile_n = j % p.n_tiles;
      tile_m = (j / p.n_tiles) * 8 + xcd;
      if (tile_m >= p.m_tiles || tile_m * (BM / 128) >= *p.rows_blocks * (p.rows_mode == 2 ? 16 : 1)) return;
    }
    if (p.rows[p.rows_mode == 2 ? (tile_m * BM) >> 4 : tile_m * BM] < 0) return;       // an empty block (lists are -1 padded at the end)
  }
  const int bm0 = tile_m * BM, bn0 = tile_n * BN;
  const int g = blockIdx.z;

  const float* __restrict__ xg = p.x + (long long)g * p.x_gs;
  const float* __restrict__ wgp = p.w + (long long)g * p.w_gs;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
  const int lrow = lane & 31, lh = lane >> 5;

  // ---- staging assignment: thread -> (row r0 + 32 i, 16-byte chunk c4)
  const int c4 = tid & 7, r0 = tid >> 3;
  long long a_base[AR];
  int a_iy0[AR], a_ix0[AR];
  bool a_ok[AR];
  // FAST path: per-row 32-bit offsets relative to a block-uniform (scalar) base, and one bit per filter tap that says
  // whether the tap lands inside the image -- the K loop then needs one add and one bit test per row instead of
  // re-deriving coordinates (the address/validity VALU work sat in front of every MFMA burst: 15 % of the kernel).
  unsigned a_rel[AR];
  unsigned long long a_taps[AR];
  long long blk_base = 0;
  if constexpr (ROWS) {           // base = start of the image of the block's first listed entry (a block spans <= 2 images)
    const int e0 = p.rows[p.rows_mode == 2 ? bm0 >> 4 : bm0];
    const int b = (int)nbm_fdiv((unsigned)e0, p.rows_mode == 2 ? p.fd_rows_thw : p.fd_howo);
    blk_base = (long long)b * p.HoWo * p.x_ld;
  } else {
    const int m0 = bm0 < p.M ? bm0 : 0;
    const int b = (int)nbm_fdiv((unsigned)m0, p.fd_howo), rem = m0 - b * p.HoWo;
    const int oy = (int)nbm_fdiv((unsigned)rem, p.fd_wo), ox = rem - oy * p.Wo;
    blk_base = ((long long)(b * p.H + oy * p.stride - p.pad) * p.W + (ox * p.stride - p.pad)) * p.x_ld;
  }
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = bm0 + r0 + 32 * i;
    a_ok[i] = m < p.M;
    long long mm = a_ok[i] ? m : 0;
    if constexpr (ROWS) {
      if (a_ok[i]) { mm = row_pixel(m); a_ok[i] = mm >= 0; }
      if (!a_ok[i]) mm = blk_base / p.x_ld;
    }
    // (pixel indices fit 31 bits: M is an int; a 64-bit division here was ~150 instructions per row)
    const int b = (int)nbm_fdiv((unsigned)mm, p.fd_howo), rem = (int)(mm - (long long)b * p.HoWo);
    const int oy = (int)nbm_fdiv((unsigned)rem, p.fd_wo), ox = rem - oy * p.Wo;
    a_iy0[i] = oy * p.stride - p.pad;
    a_ix0[i] = ox * p.stride - p.pad;
    a_base[i] = ((long long)(b * p.H + a_iy0[i]) * p.W + a_ix0[i]) * p.x_ld;
    a_rel[i] = ((unsigned)(a_base[i] - blk_base) + c4 * 4) * 4u;   // bytes; rows ascend with m: never negative
    unsigned long long mk = 0ull;
    if constexpr (AMODE == A_FAST) {
      // rows and columns of the filter that land inside the image, then their product -- selects only: as a kh x kw nest of data-dependent
      // branches this loop was the longest part of the tile prologue (cycle counters of the data-gradient twin, round 5)
      unsigned rowm = 0u, colm = 0u;
      for (int r = 0; r < p.kh; ++r) rowm |= ((unsigned)(a_iy0[i] + r) < (unsigned)p.H ? 1u : 0u) << r;
      for (int s2 = 0; s2 < p.kw; ++s2) colm |= ((unsigned)(a_ix0[i] + s2) < (unsigned)p.W ? 1u : 0u) << s2;
      for (int r = 0; r < p.kh; ++r) mk |= ((rowm >> r) & 1u) ? (unsigned long long)colm << (r * p.kw) : 0ull;
      mk = a_ok[i] ? mk : 0ull;
    }
    a_taps[i] = mk;
  }
  const float* b_ptr[BR];
  bool b_ok[BR];
  unsigned b_rel[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) {
    const int n = bn0 + r0 + 32 * i;
    b_ok[i] = n < p.N;
    b_ptr[i] = wgp + (long long)(b_ok[i] ? n : 0) * p.w_ld + c4 * 4;
    b_rel[i] = b_ok[i] ? (unsigned)(n * p.w_ld + c4 * 4) * 4u : 0x80000000u;   // bytes, or out of range
  }
  // raw buffer resources (stride 0, 2 GB window): base = block-uniform pointer, range check gives the zero padding
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xg + blk_base), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wgp), 0, 0x7ffffff0, 0x00020000);

  f32x4 ra[AR], rb[BR];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // K-step cursor (FAST: channel chunk outer, tap inner)
  int cur_r = 0, cur_s = 0, cur_c0 = 0;

  auto load_tiles = [&](int kt) {
    if constexpr (AMODE == A_FAST) {
      const int tap = cur_r * p.kw + cur_s;
      // buffer loads: block-uniform resource + scalar tap offset + per-lane 32-bit offset; a tap outside the image (or
      // a row beyond M / N) gets an out-of-range offset and the hardware range check returns zeros -- no branches.
      const unsigned a_soff = (unsigned)((((long long)cur_r * p.W + cur_s) * p.x_ld + cur_c0) * 4);
      const unsigned b_soff = (unsigned)((tap * p.Cin + cur_c0) * 4);
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const unsigned vo = ((a_taps[i] >> tap) & 1ull) ? a_rel[i] : 0x80000000u;
        ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo, a_soff, 0));
      }
#pragma unroll
      for (int i = 0; i < BR; ++i)
        rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, b_rel[i], b_soff, 0));
      // advance cursor
      if (++cur_s == p.kw) { cur_s = 0; if (++cur_r == p.kh) { cur_r = 0; cur_c0 += BK; } }
    } else {
      const int k0 = kt * BK + c4 * 4;
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        f32x4 v = zero4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = k0 + e;
          if (a_ok[i] && k < p.K) {
            const int tap = k / p.Cin, c = k - tap * p.Cin;
            const int r = tap / p.kw, s = tap - r * p.kw;
            const int iy = a_iy0[i] + r, ix = a_ix0[i] + s;
            if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
              v[e] = xg[a_base[i] + ((long long)r * p.W + s) * p.x_ld + c];
          }
        }
        ra[i] = v;
      }
#pragma unroll
      for (int i = 0; i < BR; ++i)
        rb[i] = b_ok[i] ? *reinterpret_cast<const f32x4*>(b_ptr[i] + kt * BK) : zero4;
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AR; ++i)
      *reinterpret_cast<f32x4*>(As + (buf * BM + r0 + 32 * i) * PITCH + c4 * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < BR; ++i)
      *reinterpret_cast<f32x4*>(Bs + (buf * BN + r0 + 32 * i) * PITCH + c4 * 4) = rb[i];
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // Software pipeline (ONE barrier per K-step, loads two tiles ahead):
  //   iteration kt:  barrier | MFMA group 0 + LDS writes of tile kt+1 (from registers, into the buffer nobody reads)
  //                          | MFMA group 1 + global loads of tile kt+2 (into the registers just freed)
  //                          | MFMA groups 2, 3
  // so the global-load / address / ds_write instructions issue BETWEEN this wave's own MFMAs instead of in a separate
  // window in front of them (that window cost 15 % of the kernel: both waves of a SIMD hit it together).
  if constexpr (STAGES == 2) {
    load_tiles(0);
    store_lds(0);
    if (p.nk > 1) load_tiles(1);
  }

  auto mfma_group = [&](const float* Ab, const float* Bb, int q) {
    f32x4 a[MT], b[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * PITCH + q * 4);
#pragma unroll
    for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * PITCH + q * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
  };

  // One K-step.  STORE / LOAD are compile-time so that the steady-state body is ONE basic block: the scheduler is then
  // told (sched_group_barrier) to place one LDS write, resp. one buffer load, after every second MFMA.
  auto k_step = [&](int kt, auto store_c, auto load_c) {
    constexpr bool STORE = decltype(store_c)::value, LOAD = decltype(load_c)::value;
    const int cur = kt & 1;
    __syncthreads();
    const float* Ab = As + (cur * BM + wm0 + lrow) * PITCH + lh * 16;
    const float* Bb = Bs + (cur * BN + wn0 + lrow) * PITCH + lh * 16;
    mfma_group(Ab, Bb, 0);
    if constexpr (STORE) {
      store_lds(cur ^ 1);
      __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);
#pragma unroll
      for (int z = 0; z < AR + BR; ++z) {
        __builtin_amdgcn_sched_group_barrier(0x008, (4 * MT * NT) / (AR + BR) > 0 ? (4 * MT * NT) / (AR + BR) : 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(Ab, Bb, 1);
    if constexpr (LOAD) {
      load_tiles(kt + 2);
      __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);
#pragma unroll
      for (int z = 0; z < AR + BR; ++z) {
        __builtin_amdgcn_sched_group_barrier(0x008, (4 * MT * NT) / (AR + BR) > 0 ? (4 * MT * NT) / (AR + BR) : 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(Ab, Bb, 2);
    mfma_group(Ab, Bb, 3);
  };
  if constexpr (STAGES == 2) {
    using T = std::true_type;
    using F = std::false_type;
    int kt = 0;
    for (; kt + 2 < p.nk; ++kt) k_step(kt, T{}, T{});
    if (p.nk >= 2) { k_step(kt, T{}, F{}); ++kt; }
    k_step(kt, F{}, F{});
  } else {
    // short K: load -> LDS -> MFMA, the next tile's loads in flight during the MFMAs; the other two workgroups of the CU
    // cover the barriers
    load_tiles(0);
    for (int kt = 0; kt < p.nk; ++kt) {
      if (kt) __syncthreads();                       // everyone finished reading the previous tile
      store_lds(0);
      if (kt + 1 < p.nk) load_tiles(kt + 1);
      __syncthreads();
      const float* Ab = As + (wm0 + lrow) * PITCH + lh * 16;
      const float* Bb = Bs + (wn0 + lrow) * PITCH + lh * 16;
#pragma unroll
      for (int q = 0; q < 4; ++q) mfma_group(Ab, Bb, q);
    }
  }
  __syncthreads();

#define NBM_EPI_LDS_FLOATS (STAGES * (BM + BN) * PITCH)
#include "igemm_epilogue.inc"
#undef NBM_EPI_LDS_FLOATS
}

// ---- streaming 1x1 (stride 1) for the layers whose whole weight matrix fits LDS (N * K <= 16 K: ResNet layer1's 64 -> 64 / 64 -> 256 /
// 256 -> 64 and their data gradients).  Those layers are HBM-bound (K = 64: 2.25 KB of activation / residual / output traffic per pixel
// against 32 K MACs) and the tiled kernel above leaves them at ~3.1 TB/s: every K-step of every tile re-stages the weights and meets at
// a workgroup barrier, so load, MFMA and store phases of the 8-12 waves of a CU line up instead of overlapping.  Here the weights are
// written to LDS ONCE per (persistent) workgroup and every WAVE is on its own after that: it streams 32 pixel rows straight from
// global memory into MFMA operand registers (lane (row, half h) holds k = 16 h .. 16 h + 15 of every 32-wide chunk: the same K
// order as the tiled kernel, so the sums are bit-identical), multiplies against all N channels, and sends the 32 x N tile out through
// a private 32 x 32 LDS patch (16-byte residual loads / stores, 128-byte row segments) -- no workgroup barrier after the prologue.
// Measured at B = 64 (scripts/stream1x1_probe.py, tiled -> streaming, outputs bit-identical): 64 -> 256 + residual + ReLU 1.17-1.20 -> 0.90 ms
// (1.05 before the group's residual rows were requested ahead of its MFMAs), 64 -> 64 0.28 -> 0.20, 256 -> 64 0.68 -> 0.62; 64 -> 256
// without residual 0.78 -> 0.78 (left on the tiled kernel).  Groups of 2 blocks 0.97; 12 waves per workgroup spill (25-74 registers):
// 0.91.  What stays: 256 -> 64 takes 0.62 ms where its loads alone take 0.26 (6 TB/s), loads + stores 0.39 and its MFMAs 0.32 -- the
// phases ADD although every wave now has eight chunk loads in flight during its MFMAs (vmcnt(28) in front of every chunk): with the
// matrix pipe and HBM both near their limits the part does not sustain both (the tiled kernels show the same sum).
struct StreamParams {
  const float* x; const float* w; float* y; const float* scale; const float* shift; const float* residual;
  int M, x_ld, w_ld, y_ld, res_ld, m_tiles; float alpha; int act;
  unsigned* bits_out; int n_words;      // optional: (y > 0) bits, n_words = N_total / 32 words per row (nbm_gemm_desc.bits_out)
};
constexpr int SPITCH = 36;
// NTG: 32-wide channel blocks per accumulation group; WAVES per workgroup.  Two schedules, both aimed at keeping every wave's memory
// requests in flight WHILE it multiplies (ablation of the first version, 256 -> 64 at B = 64: 0.64 ms; without the MFMAs 0.39 = 5 TB/s;
// without loads and stores 0.26: the phases of a wave simply added up):
//  * K32 <= 2 (K = 64, up to 256 output channels): the rows of the NEXT tile are requested before the MFMAs (`an`), and each group's
//    residual rows before the group's MFMAs, so the epilogue finds them in registers;
//  * K32 > 2 (K = 256, 64 output channels): all accumulators live at once, K chunks outermost -- the registers of chunk kc are
//    reloaded with the next tile's chunk kc right after its MFMAs were issued: 8 chunk loads in flight per wave at all times, no
//    second register set.
template <int K32, int NT, int NTG, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void stream1x1_kernel(const StreamParams p) {
  constexpr int K = K32 * 32, N = NT * 32, WP = K + 4;
  constexpr bool CHUNKED = K32 > 2;
  static_assert(NT % NTG == 0 && (!CHUNKED || NTG == NT), "groups");
  __shared__ __attribute__((aligned(16))) float Ws[N * WP];
  __shared__ __attribute__((aligned(16))) float Stg[WAVES][32 * SPITCH];
  __shared__ __attribute__((aligned(16))) unsigned Bw[WAVES][32 * NT];       // (y > 0) words of a wave's 32-row tile (bits_out)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  // blockIdx.y = slice of N output channels (layers whose whole weight matrix does not fit LDS: every slice streams the rows again --
  // through L2, the slices of a tile run side by side -- and owns its columns of the output / residual)
  const int n_off = blockIdx.y * N;
  for (int i = tid; i < N * (K / 4); i += WAVES * 64) {
    const int n = i / (K / 4), c = i - n * (K / 4);
    *reinterpret_cast<f32x4*>(Ws + n * WP + c * 4) = *reinterpret_cast<const f32x4*>(p.w + (long long)(n_off + n) * p.w_ld + c * 4);
  }
  __syncthreads();
  float* stg = Stg[wave];
  const int er = lane >> 3, ec = (lane & 7) * 4;            // epilogue: 8 rows x 8 chunks of 4 channels per pass
  f32x4 a[K32][4], an[CHUNKED ? 1 : K32][4];
  auto row_ptr = [&](int tile) {
    int m = tile * 32 + li;
    m = m < p.M ? m : p.M - 1;
    return p.x + (long long)m * p.x_ld + 16 * lh;
  };
  auto load_chunk = [&](const float* xp, int kc, f32x4* dst) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = *reinterpret_cast<const f32x4*>(xp + kc * 32 + 4 * q);
  };
  const int step = gridDim.x * WAVES;
  int tile = blockIdx.x * WAVES + wave;
  if (tile < p.m_tiles) {
    const float* xp = row_ptr(tile);
#pragma unroll
    for (int kc = 0; kc < K32; ++kc) load_chunk(xp, kc, CHUNKED ? a[kc] : an[kc]);
  }
  for (; tile < p.m_tiles; tile += step) {
    const bool more = tile + step < p.m_tiles;
    const float* xn = row_ptr(more ? tile + step : tile);
    if constexpr (!CHUNKED) {
#pragma unroll
      for (int kc = 0; kc < K32; ++kc)
#pragma unroll
        for (int q = 0; q < 4; ++q) a[kc][q] = an[kc][q];
#pragma unroll
      for (int kc = 0; kc < K32; ++kc) load_chunk(xn, kc, an[kc]);       // unconditional (see below): after the last tile, its own rows again
    }
    const int m0 = tile * 32;
    f32x4 rq[NTG][4];
    auto load_res = [&](int jj, f32x4* dst) {
      if (!p.residual) return;
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int m = m0 + pass * 8 + er;
        dst[pass] = *reinterpret_cast<const f32x4*>(p.residual + (long long)(m < p.M ? m : p.M - 1) * p.res_ld + n_off + jj * 32 + ec);
      }
    };
#pragma unroll
    for (int g = 0; g < NT / NTG; ++g) {
#pragma unroll
      for (int j = 0; j < NTG; ++j) load_res(g * NTG + j, rq[j]);          // in flight during this group's MFMAs
      f32x16 acc[NTG];
#pragma unroll
      for (int j = 0; j < NTG; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
      if constexpr (CHUNKED) {
#pragma unroll
        for (int kc = 0; kc < K32; ++kc) {
#pragma unroll
          for (int j = 0; j < NTG; ++j) {
            const float* wb = Ws + (j * 32 + li) * WP + 16 * lh + kc * 32;
            f32x4 b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) b[q] = *reinterpret_cast<const f32x4*>(wb + 4 * q);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kc][q][e], b[q][e], acc[j], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);       // keep the reload HERE (the scheduler sank all eight behind the last MFMA)
          load_chunk(xn, kc, a[kc]);       // the next tile's chunk (or, after the last tile, this one's again: no branch, so the
                                           // compiler counts the loads in flight exactly) into the registers just consumed
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int j = 0; j < NTG; ++j) {
          const float* wb = Ws + ((g * NTG + j) * 32 + li) * WP + 16 * lh;
#pragma unroll
          for (int kc = 0; kc < K32; ++kc) {
            f32x4 b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) b[q] = *reinterpret_cast<const f32x4*>(wb + kc * 32 + 4 * q);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kc][q][e], b[q][e], acc[j], 0, 0, 0);
          }
        }
      }
      // epilogue, one 32 x 32 block at a time through this wave's LDS patch (same arithmetic as the tiled kernel's)
#pragma unroll
      for (int j = 0; j < NTG; ++j) {
        const int jj = g * NTG + j;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int e = 0; e < 16; ++e) stg[((e & 3) + 8 * (e >> 2) + 4 * lh) * SPITCH + li] = acc[j][e];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int n = n_off + jj * 32 + ec;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + n);
        if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + n);
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const int r = pass * 8 + er, m = m0 + r;
          if (m >= p.M) continue;
          f32x4 v = *reinterpret_cast<const f32x4*>(stg + r * SPITCH + ec);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (v[e] * p.alpha) * sc[e] + sh[e] + 0.f;
          if (p.residual) { v[0] += rq[j][pass][0]; v[1] += rq[j][pass][1]; v[2] += rq[j][pass][2]; v[3] += rq[j][pass][3]; }
          if (p.act == NBM_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
          } else if (p.act == NBM_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] / (1.0f + expf(-v[e]));
          } else if (p.act == NBM_ACT_LEAKY) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.01f * v[e];
          }
          if (p.bits_out) {                          // the 8 lanes of a row (er) hold its 32 channels of block jj: one word
            const int grp8 = 8 * er;                 // this row's byte of the wave-wide comparison masks: bit 8 e + q <-> channel 4 q + e
            const unsigned wb = ((unsigned)(__builtin_amdgcn_ballot_w64(v[0] > 0.f) >> grp8) & 0xffu) |
                                (((unsigned)(__builtin_amdgcn_ballot_w64(v[1] > 0.f) >> grp8) & 0xffu) << 8) |
                                (((unsigned)(__builtin_amdgcn_ballot_w64(v[2] > 0.f) >> grp8) & 0xffu) << 16) |
                                (((unsigned)(__builtin_amdgcn_ballot_w64(v[3] > 0.f) >> grp8) & 0xffu) << 24);
            if (ec == 0) Bw[wave][r * NT + jj] = wb;        // written out row-contiguous once the tile is complete (below)
          }
          *reinterpret_cast<f32x4*>(p.y + (long long)m * p.y_ld + n) = v;
        }
      }
    }
    if (p.bits_out) {
      // the tile's 32 x NT words in pieces of NT / 2 consecutive words per lane: word by word from the lanes that formed them, these
      // stores were 4 useful bytes per 32-byte sector (64 -> 256 @94x256 at B = 128: 1.47 -> 1.69 ms with the bits, round 5)
      static_assert(NT >= 2 && NT % 2 == 0, "two lanes per tile row");
      constexpr int WPL = NT / 2;                              // words per lane
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      const int r = lane >> 1, j0 = (lane & 1) * WPL;
      const int m = m0 + r;
      if (m < p.M) {
        unsigned* dst = p.bits_out + (long long)m * p.n_words + (n_off >> 5) + j0;
        const unsigned* src = Bw[wave] + r * NT + j0;
        if constexpr (WPL == 4) *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
        else if constexpr (WPL == 2) *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(src);
        else {
#pragma unroll
          for (int q = 0; q < WPL; ++q) dst[q] = src[q];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  }
}

template <int K32, int NT, int NTG, int WAVES, int WG_PER_CU>
int launch_stream(const StreamParams& p, hipStream_t st, int slices = 1) {
  const int wgs = (p.m_tiles + WAVES - 1) / WAVES;
  int gx = 256 * WG_PER_CU / slices;                    // the persistent grid is shared by the slices
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL((stream1x1_kernel<K32, NT, NTG, WAVES>), dim3(wgs < gx ? wgs : gx, slices), dim3(WAVES * 64), 0, st, p);
  return nbm_launch_status();
}

template <int BM, int BN, int WM, int WN, int AMODE, int EPI>
int launch(const IgemmParams& p, int groups, hipStream_t st) {
  dim3 grid(p.m_tiles * p.n_tiles, 1, groups);
  hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, AMODE, EPI>), grid, dim3(256), 0, st, p);
  return nbm_launch_status();
}

int launch_s1(const IgemmParams& p, int groups, hipStream_t st) {
  dim3 grid(p.m_tiles * p.n_tiles, 1, groups);
  hipLaunchKernelGGL((igemm_kernel<128, 128, 64, 64, A_FAST, EPI_STD, 1>), grid, dim3(256), 0, st, p);
  return nbm_launch_status();
}

int launch_s1_rows(const IgemmParams& p, hipStream_t st) {
  const int mt = p.rows_blocks ? (p.m_tiles + 7) / 8 * 8 : p.m_tiles;
  dim3 grid(mt * p.n_tiles, 1, 1);
  hipLaunchKernelGGL((igemm_kernel<128, 128, 64, 64, A_FAST, EPI_STD, 1, true>), grid, dim3(256), 0, st, p);
  return nbm_launch_status();
}

}  // namespace

extern "C" int nbm_gemm_conv(const nbm_gemm_desc* d, void* stream) {
  if (!d || !d->x || !d->w || !d->y) return NBM_EINVAL;
  if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->N <= 0 || d->kh <= 0 || d->kw <= 0 ||
      d->stride <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->groups <= 0)
    return NBM_EINVAL;
  // host-side shape check: the output geometry must be the one the gather assumes
  if ((d->H + 2 * d->pad - d->kh) / d->stride + 1 != d->Ho || (d->W + 2 * d->pad - d->kw) / d->stride + 1 != d->Wo)
    return NBM_EINVAL;
  if (d->x_ld < d->Cin || d->y_ld < d->N || (d->residual && d->res_ld < d->N)) return NBM_EINVAL;
  // An output width 64 past a multiple of 128 (the cell-domain data-gradient planes of the deferred lateral: 256 -> 448): the last 128-wide
  // tile would be half padding -- the first N - 64 channels on 128-wide tiles, the last 64 on the 64-wide kernel.  Every output element is the
  // same sum in the same order either way (round 5; weight-gradient twin: nbm_conv_wgrad).
  {
    static const int tail_split = getenv("NBM_NT_TAIL") ? atoi(getenv("NBM_NT_TAIL")) : 1;
    if (tail_split && d->N > 128 && (d->N & 127) == 64 && !d->rows && !d->up && !d->bits_out && !d->shift_per_row && (d->Cin % BK) == 0 &&
        d->kh * d->kw * d->Cin > 256) {
      nbm_gemm_desc a = *d, b = *d;
      const int n0 = d->N - 64;
      a.N = n0;
      b.N = 64;
      b.w = d->w + (long long)n0 * d->w_ld;
      b.y = d->y + n0;
      if (d->scale) b.scale = d->scale + n0;
      if (d->shift) b.shift = d->shift + n0;
      if (d->residual) b.residual = d->residual + n0;
      if (d->mask) b.mask = d->mask + n0;
      const int rc = nbm_gemm_conv(&a, stream);
      return rc ? rc : nbm_gemm_conv(&b, stream);
    }
  }
  IgemmParams p{};
  p.x = d->x; p.w = d->w; p.y = d->y; p.scale = d->scale; p.shift = d->shift; p.residual = d->residual;
  p.x_gs = d->x_gs; p.w_gs = d->w_gs; p.y_gs = d->y_gs; p.res_gs = d->res_gs;
  p.M = d->B * d->Ho * d->Wo; p.N = d->N; p.K = d->kh * d->kw * d->Cin;
  p.nk = (p.K + BK - 1) / BK;
  p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pad = d->pad;
  p.Ho = d->Ho; p.Wo = d->Wo; p.HoWo = d->Ho * d->Wo;
  p.fd_howo = nbm_fastdiv_make((unsigned)p.HoWo); p.fd_wo = nbm_fastdiv_make((unsigned)p.Wo);
  p.x_ld = d->x_ld; p.w_ld = d->w_ld; p.y_ld = d->y_ld; p.res_ld = d->res_ld;
  p.alpha = d->alpha; p.act = d->act; p.shift_per_row = d->shift_per_row;
  if (p.w_ld < p.nk * BK) return NBM_EINVAL;  // every W row must hold nk*32 readable floats
  if (d->mask) {
    if (d->rows || d->groups != 1 || d->mask_ld < d->N) return NBM_EUNSUPPORTED;
    p.mask = d->mask; p.mask_ld = d->mask_ld;
  }
  p.vec_epi = ((d->N & 3) == 0 && (d->y_ld & 3) == 0 && (d->y_gs & 3) == 0 && nbm_aligned16(d->y) &&
               (!d->residual || ((d->res_ld & 3) == 0 && (d->res_gs & 3) == 0 && nbm_aligned16(d->residual))) &&
               (!d->scale || nbm_aligned16(d->scale)) && (!d->shift || d->shift_per_row || nbm_aligned16(d->shift)) &&
               (!d->mask || ((d->mask_ld & 3) == 0 && nbm_aligned16(d->mask))))
                  ? 1 : 0;
  if ((d->w_ld & 3) || (d->w_gs & 3) || !nbm_aligned16(d->w)) return NBM_EALIGN;
  if (d->bits_out) {              // (y > 0) bits: written by the vector epilogue, whole 32-channel words
    if (!p.vec_epi || (d->N & 31) || d->groups != 1 || d->rows) return NBM_EUNSUPPORTED;
    p.bits_out = d->bits_out;
  }
  if (d->up) {
    if (!p.vec_epi || d->groups != 1 || d->up_H <= 0 || d->up_W <= 0 || !nbm_aligned16(d->up)) return NBM_EUNSUPPORTED;
    p.up = d->up; p.up_H = d->up_H; p.up_W = d->up_W;
    p.up_sh = d->Ho > 1 ? (float)(d->up_H - 1) / (float)(d->Ho - 1) : 0.f;
    p.up_sw = d->Wo > 1 ? (float)(d->up_W - 1) / (float)(d->Wo - 1) : 0.f;
  }
  const bool fast = (d->Cin % BK) == 0 && (d->x_ld & 3) == 0 && (d->x_gs & 3) == 0 && nbm_aligned16(d->x);
  hipStream_t st = (hipStream_t)stream;
  const int BM = 128;
  p.m_tiles = (p.M + BM - 1) / BM;
  if (d->rows) {                  // listed rows: the short-K 1x1 variant only
    if (d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad != 0 || d->groups != 1 || !fast || !p.vec_epi || d->shift_per_row ||
        p.nk > 8 || d->N <= 64 || (d->rows_mode != 1 && d->rows_mode != 2) || d->rows_count <= 0 || (d->rows_count % 128))
      return NBM_EUNSUPPORTED;
    if (d->rows_mode == 2 && (d->rows_TH != (d->H + 1) / 2 || d->rows_TW != (d->W + 1) / 2)) return NBM_EINVAL;
    if (2ll * d->H * d->W * d->x_ld * 4 > 0x7fffffffll) return NBM_EUNSUPPORTED;
    p.rows = d->rows; p.rows_blocks = d->rows_blocks; p.rows_mode = d->rows_mode; p.rows_TH = d->rows_TH; p.rows_TW = d->rows_TW;
    p.fd_rows_thw = nbm_fastdiv_make((unsigned)(d->rows_TH * d->rows_TW)); p.fd_rows_tw = nbm_fastdiv_make((unsigned)d->rows_TW);
    p.M = d->rows_count;
    p.m_tiles = p.M / BM;
    p.n_tiles = (d->N + 127) / 128;
    return launch_s1_rows(p, st);
  }
  // streaming form for 1x1 / stride 1 layers whose weights fit LDS (see stream1x1_kernel)
  const char* stream_env = getenv("NBM_STREAM1X1");          // read per call: the parity test flips it inside one process
  if (!(stream_env && stream_env[0] == '0') && d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0 && d->groups == 1 && fast && p.vec_epi && !d->up && !d->mask &&
      !d->shift_per_row && (d->N % 32) == 0 && p.M >= 8192) {
    StreamParams sp{d->x, d->w, d->y, d->scale, d->shift, d->residual, p.M, d->x_ld, d->w_ld, d->y_ld, d->res_ld, (p.M + 31) / 32,
                    d->alpha, d->act, d->bits_out, d->N >> 5};
    const int k32 = d->Cin / 32, nt = d->N / 32;
    // 64 -> 256 without a residual is write-bound and gains nothing (0.78 ms either way at B = 64): tiled kernel
    if (k32 == 2 && nt == 8 && d->residual) return launch_stream<2, 8, 4, 8, 1>(sp, st);
    if (k32 == 2 && nt == 2) return launch_stream<2, 2, 2, 8, 2>(sp, st);
    if (k32 == 8 && nt == 2) return launch_stream<8, 2, 2, 8, 1>(sp, st);
    // wider layers in slices of N (experiment switch NBM_STREAM_SLICED, default on)
    static const int sliced = getenv("NBM_STREAM_SLICED") ? atoi(getenv("NBM_STREAM_SLICED")) : 1;
    if (sliced && d->residual) {
      if (k32 == 4 && nt % 4 == 0 && nt <= 32) return launch_stream<4, 4, 4, 8, 1>(sp, st, nt / 4);      // 128 -> 512: 4 slices of 128
      if (k32 == 8 && nt % 2 == 0 && nt <= 64) return launch_stream<8, 2, 2, 8, 1>(sp, st, nt / 2);      // 256 -> 1024: 16 slices of 64
    }
  }
  if (d->N > 64) {
    p.n_tiles = (d->N + 127) / 128;
    // deep K on the bf16 matrix pipe through split fp32 operands (igemm_split.hip); NBM_SPLIT_BF16=0 keeps the fp32 instruction.
    // Read per call: the parity tests flip it inside one process.  The choice depends on the LAYER (K, N), never on the number of rows:
    // a clip's result must not depend on how many clips share its batch (the two kernels sum in different orders).
    const char* split_env = getenv("NBM_SPLIT_BF16");
    static const int split_min_nk = getenv("NBM_SPLIT_MIN_NK") ? atoi(getenv("NBM_SPLIT_MIN_NK")) : 9;      // experiment switch (>= 3)
    if (split_env && split_env[0] == '1' && fast && p.vec_epi && p.nk >= (split_min_nk < 3 ? 3 : split_min_nk) && d->kh * d->kw < 63)
      return nbm_igemm::split_launch(p, d->groups, st);
    // deep K: half-step LDS stages, three workgroups per CU (igemm_h16.hip: the same products in the same order as the two-stage kernel below,
    // 2-14 % faster launch by launch at B = 64, scripts/h16_probe.py).  NBM_H16 = 0: the two-stage kernel; 3 (default) / 4: workgroups per CU;
    // from NBM_H16_MIN_NK K-steps up (default 9: up to 8 the single-stage kernel below stays)
    const char* h16_env = getenv("NBM_H16");                   // read per call: the parity test flips it inside one process
    const int h16 = h16_env ? atoi(h16_env) : 3;
    const char* h16_min_env = getenv("NBM_H16_MIN_NK");
    const int h16_min = h16_min_env ? atoi(h16_min_env) : 9;
    if (h16 && fast && p.vec_epi && p.nk >= h16_min && d->kh * d->kw < 63) return nbm_igemm::h16_launch(p, d->groups, h16, st);
    // short K and a 16-byte epilogue: the three-workgroups-per-CU variant (see the template comment)
    static const int shortk_max = getenv("NBM_SHORTK_MAX") ? atoi(getenv("NBM_SHORTK_MAX")) : 8;   // 0 disables
    if (fast && p.vec_epi && p.nk <= shortk_max)
      return launch_s1(p, d->groups, st);

    return fast ? launch<128, 128, 64, 64, A_FAST, EPI_STD>(p, d->groups, st)
                : launch<128, 128, 64, 64, A_GENERIC, EPI_STD>(p, d->groups, st);
  } else if (d->N > 32) {
    p.n_tiles = 1;
    // (NEGATIVE, round 5: 256 x 64 tiles with a 64 x 64 patch per wave -- the fragment reuse of the 128 x 128 kernel -- on the layer1 3x3
    // 64 -> 64 @94x256: two LDS stages / one workgroup per CU 1.30 ms against 1.14 at B = 64, one stage / two workgroups per CU 1.15:
    // the 64-wide tile's time is not its LDS reads per MFMA)
    // single LDS stage / three workgroups per CU for the 64-wide tile up to K = 32 * s1_n64 (NBM_S1_N64; 0 = never): this tile spends half the
    // MFMA cycles per K-step of the 128-wide one, so its prologue / epilogue weigh double and the third workgroup pays up to K = 576 (layer1's
    // 3x3 64 -> 64 @94x256: 1.075 -> 0.99 ms at B = 64, 2.05 -> 1.83 at B = 128; same K order, same bits)
    static const int s1_n64 = getenv("NBM_S1_N64") ? atoi(getenv("NBM_S1_N64")) : 20;
    if (fast && p.vec_epi && p.nk <= s1_n64) {
      dim3 grid(p.m_tiles * p.n_tiles, 1, d->groups);
      hipLaunchKernelGGL((igemm_kernel<128, 64, 64, 32, A_FAST, EPI_STD, 1>), grid, dim3(256), 0, st, p);
      return nbm_launch_status();
    }
    return fast ? launch<128, 64, 64, 32, A_FAST, EPI_STD>(p, d->groups, st)
                : launch<128, 64, 64, 32, A_GENERIC, EPI_STD>(p, d->groups, st);
  } else {
    p.n_tiles = 1;
    return fast ? launch<128, 32, 32, 32, A_FAST, EPI_STD>(p, d->groups, st)
                : launch<128, 32, 32, 32, A_GENERIC, EPI_STD>(p, d->groups, st);
  }
}
