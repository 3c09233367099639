// Winograd F(2x2,3x3) forward convolution, transforms fused into the GEMM (the FPN output convolutions, reference
// fpn.py:137,145 -- 68 % of the forward FLOPs -- and the ResNet 3x3 / stride-1 layers).
//
//   y[b][2ty+p][2tx+q][n] = epi( sum_{i,j} AT[p][i] AT[q][j] * ( sum_c V[i][j][t][c] U[i][j][n][c] ) ),   V = B^T d B
//
// Round 1 ran three kernels: x -> V[16][T][C] (591 MB per 188x512 image), 16 grouped GEMMs -> M[16][T][N] (394 MB), M -> y.
// Now:
//  * wino23_rows_kernel applies only the ROW half of the input transform: R[i][b][ty][xp][c] = (B^T d)_i for the four row
//    combinations of tile row ty, kept as full-width image rows (xp = x + 1, one zero column on each side).  Neighbouring
//    tiles share their columns there, so R is 2x the input instead of 4x (297 MB per image instead of 591).
//  * wino23_fused_kernel: ONE workgroup owns a [128 tiles x BN channels] block for ALL 16 planes.  The COLUMN half of the
//    input transform happens while the A operand is staged: V[i][j] = R_i[2tx + k1] +- R_i[2tx + k2] (two buffer loads
//    with scalar column offsets, one fma on the way to LDS).  The 16 x (C / 32) K-steps run as one continuous software
//    pipeline (same staging as igemm.hip: raw buffer loads two steps ahead -> registers -> padded LDS rows, one barrier
//    per step); the plane product stays in a scratch accumulator and, after the last K-step of a plane, is added with its
//    +-1 weights into the four output accumulators Y[p][q].  M never exists; the epilogue (scale, bias / FrozenBN shift,
//    ReLU, producer mask) writes y directly, 16 bytes per lane through an LDS-staged tile.  The per-tile fixed cost of the
//    GEMM (prologue + epilogue ~ 1.7 K-steps) is paid once per 192 K-steps instead of once per 12.
//
// Two shapes: BN = 128 (wave tile 64 x 64: 64 scratch + 256 output accumulator registers, ONE wave per SIMD) and BN = 64
// (wave tile 64 x 32: 32 + 128, two workgroups per CU like igemm.hip).
#include "nbm_common.h"
#include <type_traits>

#if defined(__HIP_DEVICE_COMPILE__)
#define NBM_KEEP(x) asm volatile("" : "+v"(x))      // the value stays live and opaque (timing-only ablations)
#else
#define NBM_KEEP(x) (void)(x)
#endif

namespace {

constexpr int BK = 32;
constexpr int PITCH = 36;     // floats per LDS row (32 + 4 pad): conflict-free ds_read_b128
constexpr int LIST_BM = 128;     // tile rows of a workgroup's block where tile LISTS are involved (the lists come in 128-entry blocks)

struct WinoFusedParams {
  const float* R; const float* U; float* y;
  const float* scale; const float* shift; const float* mask;
  long long r_gs;               // floats between the four row-combination images of R (= B * TH * WP * C)
  int u_gs;                     // floats between planes of U (= N * C)
  int T, N, C, nk;
  int H, W, TH, TW, THW, WP;
  nbm_fastdiv fd_thw, fd_tw;      // tile id -> (image, tile row, tile column) without run-time divisions (nbm_common.h)
  int m_tiles, n_tiles, relu;
  // Optional tile list (demand-driven evaluation, see nbm_wino23_conv_fused_tiles): m_tiles * 128 entries, entry = linear
  // tile id (b * THW + ty * TW + tx) ascending inside a 128-entry block, or -1 (only at the end of a block).  n_blocks
  // (device, optional): number of leading blocks that are filled; the other workgroups exit at once.
  const int* tiles; const int* n_blocks;
  // Optional per-block plane / store masks (tile lists only): blk_info[block] = plane mask (bit xi = 4 i + j: the plane is
  // computed) | store mask << 16 (bit 2 p + q: output pixel (p, q) of every tile of the block is stored).  A block whose
  // tiles only need output row p = 1 (A^T row [0 1 -1 -1]) skips the four planes i = 0, likewise for columns: 9 or 12 planes
  // instead of 16.  The planes that are computed are accumulated in the same order as in a full block: a stored pixel is
  // bit-identical either way.
  const unsigned* blk_info;
};

// x [B][H][W][C] -> R [4][B][TH][WP][C], WP = 2 TW + 2: R[i][b][ty][x + 1] = sum_a BT[i][a] x[b][2ty - 1 + a][x] with
// BT = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]; rows / columns outside the image are zero.
__global__ __launch_bounds__(256) void wino23_rows_kernel(const float* __restrict__ x, int B, int H, int W, int C4,
                                                          int TH, int WP, float* __restrict__ R) {
  const long long per_plane = (long long)B * TH * WP * C4;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  f32x4* r4 = reinterpret_cast<f32x4*>(R);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < per_plane; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % C4);
    long long t = idx / C4;
    const int xp = (int)(t % WP);
    t /= WP;
    const int ty = (int)(t % TH), b = (int)(t / TH);
    const int ix = xp - 1;
    f32x4 d[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int iy = 2 * ty - 1 + a;
      const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      d[a] = ok ? x4[(((long long)b * H + iy) * W + ix) * C4 + c] : zero;
    }
    r4[idx] = d[0] - d[2];
    r4[idx + per_plane] = d[1] + d[2];
    r4[idx + 2 * per_plane] = d[2] - d[1];
    r4[idx + 3 * per_plane] = d[1] - d[3];
  }
}

// The same row transform for listed tiles only: entry e of `tiles` (linear tile id or -1) gets the four columns
// xp = 2 tx .. 2 tx + 3 of its tile row written in all four row-combination images.  Neighbouring listed tiles write their two
// shared columns twice (same values).  n_blocks (device, optional) = number of leading 128-entry blocks that are filled.
__global__ __launch_bounds__(256) void wino23_rows_tiles_kernel(const float* __restrict__ x, int B, int H, int W, int C4,
                                                                int TH, int TW, int WP, const int* __restrict__ tiles,
                                                                int n_entries, const int* __restrict__ n_blocks,
                                                                const unsigned* __restrict__ blk_info, float* __restrict__ R,
                                                                int pat_stride) {
  // pat_stride S > 0: the pixels of the 3x3 / stride-S / pad-1 pattern read as zeros (as in wino23_outgrad_kernel: their share of a
  // gradient map is taken by the cell transforms, this pass adds the rest)
  const int S = pat_stride, OHp = S > 0 ? (H + 2 - 3) / S + 1 : 0, OWp = S > 0 ? (W + 2 - 3) / S + 1 : 0;
  const long long per_plane = (long long)B * TH * WP * C4;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  f32x4* r4 = reinterpret_cast<f32x4*>(R);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  if (n_blocks) n_entries = min(n_entries, *n_blocks * LIST_BM);
  const long long total = (long long)n_entries * 4 * C4;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % C4);
    const long long t = idx / C4;
    const int k = (int)(t & 3), e = (int)(t >> 2);
    const int id = tiles[e];
    if (id < 0) continue;
    // planes of the block (blk_info, optional): a row combination i / a column k that no computed plane uses is skipped
    // (k = 0 is only read by the planes j = 0, k = 3 only by j = 3; the input rows behind a skipped combination may not exist)
    const unsigned pm = blk_info ? blk_info[e >> 7] & 0xffffu : 0xffffu;
    if ((k == 0 && !(pm & 0x1111u)) || (k == 3 && !(pm & 0x8888u))) continue;
    const int THW = TH * TW;
    const int b = id / THW, rem = id - b * THW;
    const int ty = rem / TW, tx = rem - ty * TW;
    const int xp = 2 * tx + k, ix = xp - 1;
    f32x4 d[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int iy = 2 * ty - 1 + a;
      bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W &&
                !(a == 0 && !(pm & 0x000fu)) && !(a == 3 && !(pm & 0xf000u));
      if (S > 0 && ok && (iy + 1) % S < 3 && (iy + 1) / S < OHp && (ix + 1) % S < 3 && (ix + 1) / S < OWp) ok = false;
      d[a] = ok ? x4[(((long long)b * H + iy) * W + ix) * C4 + c] : zero;
    }
    const long long o = (((long long)b * TH + ty) * WP + xp) * C4 + c;
    if (pm & 0x000fu) r4[o] = d[0] - d[2];
    r4[o + per_plane] = d[1] + d[2];
    r4[o + 2 * per_plane] = d[2] - d[1];
    if (pm & 0xf000u) r4[o + 3 * per_plane] = d[1] - d[3];
  }
}

// ABL: timing-only ablations for scripts/wino_fused_probe.py (results are wrong): 1 = no flush, 2 = no epilogue,
// 4 = every workgroup reads the same L2-resident A tile
// BM: tile rows of the block (128; 96 for dense launches whose 128-row blocks would leave the last round of resident workgroups
// half empty -- one wave row of 96 x 32 patches, MT = 3, NT = 1; an output sums its planes and K-steps in the same order in every shape)
template <int BM, int BN, int WM, int WN, int ABL = 0>
__global__ __launch_bounds__(256, BN == 128 ? 1 : 2) void wino23_fused_kernel(const WinoFusedParams p) {
  constexpr int MT = WM / 32, NT = WN / 32;
  constexpr int WAVES_N = BN / WN;
  constexpr int AR = BM / 32, BR = BN / 32;
  static_assert((BM / WM) * WAVES_N == 4, "4 waves per workgroup");

  __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * PITCH];
  if constexpr (BN != 128) nbm_stagger_priority();
  float* As = lds;
  float* Bs = lds + 2 * BM * PITCH;

  // ---- XCD-aware tile id (bijective for any grid size): the N tiles of one M tile share an L2
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  int tile_m = wg / p.n_tiles, tile_n = wg - tile_m * p.n_tiles;
  if (p.n_blocks) {
    // device-side block count: only the leading blocks are filled, and the contiguous ranges above would put all of them on
    // one XCD (measured: the RoI launch ran on an eighth of the chip).  Round-robin the M blocks over the XCDs instead; the N
    // tiles of a block stay on its XCD, back to back.  The grid is padded to a multiple of 8 * n_tiles.
    const int j = bid >> 3;
    tile_n = j % p.n_tiles;
    tile_m = (j / p.n_tiles) * 8 + xcd;
    if (tile_m >= p.m_tiles) return;
  }
  const int bm0 = tile_m * BM, bn0 = tile_n * BN;

  if (p.tiles) {                                   // uniform per workgroup
    if (p.n_blocks && tile_m >= *p.n_blocks) return;
    if (p.tiles[bm0] < 0) return;
  }
  auto tile_of = [&](int t) { return p.tiles ? p.tiles[t] : t; };     // t < p.T checked by the callers
  const unsigned info = (p.tiles && p.blk_info) ? p.blk_info[tile_m] : 0x000fffffu;
  const unsigned plane_mask = info & 0xffffu, store_mask = (info >> 16) & 0xfu;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
  const int lrow = lane & 31, lh = lane >> 5;
  const int c4 = tid & 7, r0 = tid >> 3;

  // A rows: tile t = (b, ty, tx) starts at column 2 tx of image row (b, ty) of R; byte offsets relative to the block's
  // first tile.  Rows beyond T (and B rows beyond N) get an out-of-range offset: the buffer range check returns zeros.
  unsigned a_rel[AR], b_rel[BR];
  long long blk_base;
  {
    const int t = bm0 < p.T ? tile_of(bm0) : 0;
    const int bi = (int)nbm_fdiv((unsigned)t, p.fd_thw), rem = t - bi * p.THW;
    const int ty = (int)nbm_fdiv((unsigned)rem, p.fd_tw), tx = rem - ty * p.TW;
    blk_base = (((long long)bi * p.TH + ty) * p.WP + 2 * tx) * p.C;
  }
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int t = bm0 + r0 + 32 * i < p.T ? tile_of(bm0 + r0 + 32 * i) : -1;
    if (t >= 0) {
      const int bi = (int)nbm_fdiv((unsigned)t, p.fd_thw), rem = t - bi * p.THW;
      const int ty = (int)nbm_fdiv((unsigned)rem, p.fd_tw), tx = rem - ty * p.TW;
      const long long off = (((long long)bi * p.TH + ty) * p.WP + 2 * tx) * p.C - blk_base;   // rows ascend with t
      a_rel[i] = (unsigned)((off + c4 * 4) * 4);
    } else {
      a_rel[i] = 0x80000000u;
    }
  }
#pragma unroll
  for (int i = 0; i < BR; ++i) {
    const int n = bn0 + r0 + 32 * i;
    b_rel[i] = n < p.N ? (unsigned)((n * p.C + c4 * 4) * 4) : 0x80000000u;
  }
  const float* a_plane0 = (ABL & 4) ? p.R : p.R + blk_base;
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.U), 0, 0x7ffffff0, 0x00020000);

  f32x4 ra[AR], ra2[AR], rb[BR];
  float a_sign = 1.f;             // sign of the second column of the tile being staged
  unsigned ld_rem = plane_mask;   // planes still to load (lowest set bit = ld_xi), same for the LDS-store cursor below
  int ld_xi = __builtin_ctz(ld_rem), ld_c0 = 0;       // cursor of the NEXT tile to load: plane, first channel

  // V[i][j] = R_i[k1] + s R_i[k2] with (k1, k2, s) = (0,2,-), (1,2,+), (2,1,-), (1,3,-) for j = 0..3 (rows of B^T)
  auto load_tiles = [&]() {
    const int i = ld_xi >> 2, j = ld_xi & 3;
    const int k1 = j == 0 ? 0 : (j == 2 ? 2 : 1), k2 = j == 2 ? 1 : (j == 3 ? 3 : 2);
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a_plane0 + ((ABL & 4) ? 0ll : (long long)i * p.r_gs)), 0, 0x7ffffff0, 0x00020000);
    const unsigned a_soff1 = (unsigned)((k1 * p.C + ld_c0) * 4), a_soff2 = (unsigned)((k2 * p.C + ld_c0) * 4);
    const unsigned b_soff = (unsigned)((ld_xi * p.u_gs + ld_c0) * 4);
#pragma unroll
    for (int r = 0; r < AR; ++r) {
      ra[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, a_rel[r], a_soff1, 0));
      ra2[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, a_rel[r], a_soff2, 0));
    }
#pragma unroll
    for (int r = 0; r < BR; ++r)
      rb[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, b_rel[r], b_soff, 0));
    ld_c0 += BK;
    if (ld_c0 == p.C) { ld_c0 = 0; ld_rem &= ld_rem - 1; ld_xi = ld_rem ? __builtin_ctz(ld_rem) : 0; }
  };
  // sign of the tile that the NEXT store_lds writes; tiles are stored in load order, one step behind the loads
  unsigned st_rem = plane_mask;
  int st_xi = __builtin_ctz(st_rem), st_c0 = 0;
  auto store_lds = [&](int buf) {
    const float sgn = (st_xi & 3) == 1 ? 1.f : -1.f;
#pragma unroll
    for (int r = 0; r < AR; ++r) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaf(sgn, ra2[r][e], ra[r][e]);
      *reinterpret_cast<f32x4*>(As + (buf * BM + r0 + 32 * r) * PITCH + c4 * 4) = v;
    }
#pragma unroll
    for (int r = 0; r < BR; ++r)
      *reinterpret_cast<f32x4*>(Bs + (buf * BN + r0 + 32 * r) * PITCH + c4 * 4) = rb[r];
    st_c0 += BK;
    if (st_c0 == p.C) { st_c0 = 0; st_rem &= st_rem - 1; st_xi = st_rem ? __builtin_ctz(st_rem) : 0; }
  };
  (void)a_sign;

  f32x16 acc[MT][NT];               // product of the current plane
  f32x16 Y[2][2][MT][NT];           // output accumulators, [p][q]
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        acc[i][j][e] = 0.f;
        Y[0][0][i][j][e] = 0.f; Y[0][1][i][j][e] = 0.f; Y[1][0][i][j][e] = 0.f; Y[1][1][i][j][e] = 0.f;
      }

  // One K-step = 4 groups of 4 k-pairs.  The fragments of group q + 1 are read from LDS while the 4 MT NT MFMAs of group q
  // run (register double buffer fa / fb).  sched_group_barrier pins the interleave below: without it the same code runs
  // 19 % slower.  Measured and dropped (scripts/wino_fused_probe.py, 188x512 x 26 images, 16.2 ms as is): three LDS stages
  // with the barrier behind the LDS writes, so that the first fragments of step k+1 are prefetched in G3 of step k
  // (18.9 ms: the compiler bunches G0's MFMAs and spills); plane-dependent scalars (buffer descriptor, column offsets)
  // kept as loop-carried state instead of being re-derived every step (17.9 ms: the conditional update splits the step's
  // scheduling region); the same state kept per plane in an explicit plane loop with the last two steps of every plane
  // peeled (main loop 15.1 instead of 15.4 ms, but the five inlined step bodies push the allocator into 98-128 spilled
  // registers around the flush: 17.2 ms); the step barrier moved between G2 and G3 with the next step's first fragments
  // fetched under G3's MFMAs (legal with two stages; main loop unchanged at 15.3 ms, 32 spilled registers: 16.6 ms).
  f32x4 fa[2][MT], fb[2][NT];
  auto read_frags = [&](const float* Ab, const float* Bb, int q, int s) {
#pragma unroll
    for (int i = 0; i < MT; ++i) fa[s][i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * PITCH + q * 4);
#pragma unroll
    for (int j = 0; j < NT; ++j) fb[s][j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * PITCH + q * 4);
  };
  auto mfma_group = [&](int s) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][i][e], fb[s][j][e], acc[i][j], 0, 0, 0);
  };

  int cur = 0;
  auto k_step = [&](auto store_c, auto load_c) {
    constexpr bool STORE = decltype(store_c)::value, LOAD = decltype(load_c)::value;
    constexpr int NM = 4 * MT * NT;             // MFMAs per group
    __syncthreads();
    const float* Ab = As + (cur * BM + wm0 + lrow) * PITCH + lh * 16;
    const float* Bb = Bs + (cur * BN + wn0 + lrow) * PITCH + lh * 16;
    read_frags(Ab, Bb, 0, 0);
    // ---- group 0 (+ LDS writes of the next tile)
    read_frags(Ab, Bb, 1, 1);
    mfma_group(0);
    if constexpr (STORE) store_lds(cur ^ 1);
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MT + NT), 0);
    if constexpr (STORE) {
#pragma unroll
      for (int z = 0; z < AR + BR; ++z) {
        __builtin_amdgcn_sched_group_barrier(0x008, NM / (AR + BR) > 0 ? NM / (AR + BR) : 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- group 1 (+ global loads two tiles ahead)
    read_frags(Ab, Bb, 2, 0);
    mfma_group(1);
    if constexpr (LOAD) load_tiles();
    __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);
    if constexpr (LOAD) {
#pragma unroll
      for (int z = 0; z < 2 * AR + BR; ++z) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- groups 2, 3
    read_frags(Ab, Bb, 3, 1);
    mfma_group(0);
    __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(1);
    cur ^= 1;
  };

  // Y[p][q] += AT[p][i] * AT[q][j] * acc;  acc = 0.   AT = [1 1 1 0; 0 1 -1 -1]; the weights are wave-uniform +-1 / 0.
  auto flush = [&](int xi) {
    const int i = xi >> 2, j = xi & 3;
    const float ai[2] = {i < 3 ? 1.f : 0.f, i == 0 ? 0.f : (i == 1 ? 1.f : -1.f)};
    const float aj[2] = {j < 3 ? 1.f : 0.f, j == 0 ? 0.f : (j == 1 ? 1.f : -1.f)};
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const float c = ai[pp] * aj[qq];
        if (c != 0.f) {
#pragma unroll
          for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
              for (int e = 0; e < 16; ++e) Y[pp][qq][a][b][e] = fmaf(c, acc[a][b][e], Y[pp][qq][a][b][e]);
        }
      }
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  };

  using TT = std::true_type;
  using FF = std::false_type;
  load_tiles();
  store_lds(0);
  load_tiles();
  for (unsigned rem = plane_mask; rem;) {
    const int xi = __builtin_ctz(rem);
    rem &= rem - 1;
    if (rem) {
      for (int kt = 0; kt < p.nk; ++kt) k_step(TT{}, TT{});
    } else {
      for (int kt = 0; kt + 2 < p.nk; ++kt) k_step(TT{}, TT{});
      k_step(TT{}, FF{});
      k_step(FF{}, FF{});
    }
    if constexpr (!(ABL & 1)) {
      flush(xi);
    } else {                                   // keep the plane product alive (cdna guide rule 17) without the VALU work
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) NBM_KEEP(acc[a][b]);
    }
  }
  __syncthreads();
  if constexpr (ABL & 2) {
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        NBM_KEEP(acc[a][b]);
        NBM_KEEP(Y[0][0][a][b]); NBM_KEEP(Y[0][1][a][b]); NBM_KEEP(Y[1][0][a][b]); NBM_KEEP(Y[1][1][a][b]);
      }
    return;
  }

  // ---- epilogue: four passes (p, q) through an LDS-staged [128 tiles][BN] tile, 16-byte stores along the channels.
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
  constexpr int CP = BN + 4;
  static_assert(BM * CP + 2 * BM <= 2 * (BM + BN) * PITCH, "epilogue tile + row table must fit the operand buffers");
  float* Cs = lds;
  // per tile row of the block: element index of output pixel (2 ty, 2 tx) and which of its 2 x 2 pixels exist
  // (odd sizes: the last tile row / column is half outside) -- computed once, the four passes only look it up
  long long* row_pix = reinterpret_cast<long long*>(lds + BM * CP);          // [BM]; BM * CP * 4 is a multiple of 16
  if (tid < BM) {
    const int t = bm0 + tid < p.T ? tile_of(bm0 + tid) : -1;
    long long v = -1;
    if (t >= 0) {
      const int bi = (int)nbm_fdiv((unsigned)t, p.fd_thw), rem = t - bi * p.THW;
      const int ty = (int)nbm_fdiv((unsigned)rem, p.fd_tw), tx = rem - ty * p.TW;
      const long long pix = (p.relu & 4) ? (long long)(bm0 + tid) * 4 : ((long long)bi * p.H + 2 * ty) * p.W + 2 * tx;
      v = (pix << 2) | (2 * ty + 1 < p.H ? 2 : 0) | (2 * tx + 1 < p.W ? 1 : 0);
    }
    row_pix[tid] = v;
  }
  constexpr int CH = BN / 4;
  constexpr int RPP = 256 / CH;
  const int cc = tid % CH, rr = tid / CH;
  const int n = bn0 + cc * 4;
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (n < p.N) {
    if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + n);
    if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + n);
  }
#pragma unroll
  for (int pp = 0; pp < 2; ++pp)
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      if (!((store_mask >> (2 * pp + qq)) & 1u)) continue;         // uniform: nobody reads this pixel of the block's tiles
      __syncthreads();
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            Cs[(wm0 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * CP + wn0 + b * 32 + lrow] = Y[pp][qq][a][b][e];
      __syncthreads();
      if (n < p.N) {
        const int need = (pp ? 2 : 0) | (qq ? 1 : 0);
#pragma unroll 4
        for (int r = rr; r < BM; r += RPP) {
          const long long rp = row_pix[r];
          if (rp < 0) break;
          if ((rp & need) != need) continue;
          f32x4 v = *reinterpret_cast<const f32x4*>(Cs + r * CP + cc * 4);
          const long long idx = ((rp >> 2) + ((p.relu & 4) ? pp * 2 + qq : pp * p.W + qq)) * p.N + n;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] * sc[e] + sh[e];
          if (p.relu & 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          if (p.relu & 2) {               // listed tiles, accumulate: y += result (every listed tile is listed once)
            const f32x4 o = *reinterpret_cast<const f32x4*>(p.y + idx);
            v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
          }
          if (p.mask) {
            const f32x4 mk = *reinterpret_cast<const f32x4*>(p.mask + idx);
#pragma unroll
            for (int e = 0; e < 4; ++e) if (!(mk[e] > 0.f)) v[e] = 0.f;
          }
          *reinterpret_cast<f32x4*>(p.y + idx) = v;
        }
      }
    }
}

}  // namespace

// x -> R (row half of the input transform) -- see nbm_hip.h.
extern "C" int nbm_wino23_rows(const float* x, int B, int H, int W, int C, float* R, void* stream) {
  if (!x || !R || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return NBM_EINVAL;
  if (!nbm_aligned16(x) || !nbm_aligned16(R)) return NBM_EALIGN;
  const int TH = (H + 1) >> 1, TW = (W + 1) >> 1, WP = 2 * TW + 2;
  const long long n = (long long)B * TH * WP * (C / 4);
  long long g = (n + 255) / 256;
  if (g > 65536) g = 65536;
  hipLaunchKernelGGL(wino23_rows_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, B, H, W, C / 4, TH, WP, R);
  return nbm_launch_status();
}

// Row transform of the listed tiles only -- see nbm_hip.h.
extern "C" int nbm_wino23_rows_tiles(const float* x, int B, int H, int W, int C, const int* tiles, int n_entries,
                                     const int* n_blocks, const unsigned* blk_info, float* R, int skip_pattern_stride,
                                     void* stream) {
  if (!x || !R || !tiles || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || n_entries < 0 || skip_pattern_stride < 0) return NBM_EINVAL;
  if (!nbm_aligned16(x) || !nbm_aligned16(R)) return NBM_EALIGN;
  if (n_entries == 0) return NBM_OK;
  const int TH = (H + 1) >> 1, TW = (W + 1) >> 1, WP = 2 * TW + 2;
  const long long n = (long long)n_entries * 4 * (C / 4);
  long long g = (n + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(wino23_rows_tiles_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, B, H, W, C / 4, TH, TW,
                     WP, tiles, n_entries, n_blocks, blk_info, R, skip_pattern_stride);
  return nbm_launch_status();
}

static int wino23_conv_fused_launch(const float* R, const float* U, const float* scale, const float* shift,
                                    const float* mask, int relu, int B, int H, int W, int C, int N, float* y,
                                    const int* tiles, int n_entries, const int* n_blocks, const unsigned* blk_info,
                                    int variant, void* stream);

// y = epi(conv3x3(x)) from the row-transformed input R [4][B][TH][WP][C] and the weights U [16][N][C] -- see nbm_hip.h.
extern "C" int nbm_wino23_conv_fused(const float* R, const float* U, const float* scale, const float* shift,
                                     const float* mask, int relu, int B, int H, int W, int C, int N, float* y,
                                     int variant, void* stream) {
  return wino23_conv_fused_launch(R, U, scale, shift, mask, relu, B, H, W, C, N, y, nullptr, 0, nullptr, nullptr, variant, stream);
}

// The same for the listed tiles only (pixels of y outside the listed tiles are not written) -- see nbm_hip.h.
extern "C" int nbm_wino23_conv_fused_tiles(const float* R, const float* U, const float* scale, const float* shift,
                                           const float* mask, int relu, int B, int H, int W, int C, int N, float* y,
                                           const int* tiles, int n_entries, const int* n_blocks, const unsigned* blk_info,
                                           void* stream) {
  if (!tiles || n_entries < 0 || (n_entries % LIST_BM)) return NBM_EINVAL;
  if (n_entries == 0) return NBM_OK;
  return wino23_conv_fused_launch(R, U, scale, shift, mask, relu, B, H, W, C, N, y, tiles, n_entries, n_blocks, blk_info, 0,
                                  stream);
}

static int wino23_conv_fused_launch(const float* R, const float* U, const float* scale, const float* shift,
                                    const float* mask, int relu, int B, int H, int W, int C, int N, float* y,
                                    const int* tiles, int n_entries, const int* n_blocks, const unsigned* blk_info,
                                    int variant, void* stream) {
  if (!R || !U || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || N <= 0) return NBM_EINVAL;
  if ((C % BK) || C / BK < 2 || (N & 3)) return NBM_EUNSUPPORTED;
  if (!nbm_aligned16(R) || !nbm_aligned16(U) || !nbm_aligned16(y) || (shift && !nbm_aligned16(shift)) ||
      (scale && !nbm_aligned16(scale)) || (mask && !nbm_aligned16(mask)))
    return NBM_EALIGN;
  WinoFusedParams p{};
  p.TH = (H + 1) >> 1; p.TW = (W + 1) >> 1; p.THW = p.TH * p.TW; p.WP = 2 * p.TW + 2;
  p.fd_thw = nbm_fastdiv_make((unsigned)p.THW); p.fd_tw = nbm_fastdiv_make((unsigned)p.TW);
  const long long T = (long long)B * p.THW;
  // 32-bit byte offsets inside the kernel: the rows of one 128-tile block (<= 128 image rows of R apart), one row
  // combination of R behind the block base, and the whole of U
  if (T > 0x7fffff00ll || (long long)N * C * 16 * 4 > 0x7fffffffll) return NBM_EUNSUPPORTED;
  if ((130ll * p.WP + 4) * C * 4 > 0x7fffffffll) return NBM_EUNSUPPORTED;
  // listed tiles: a 128-entry block may span up to min(B, 8) whole images of one row combination (the callers keep a block
  // inside one image, or a fixed pattern of every image's tile rows)
  if (tiles && (long long)(B < 8 ? B : 8) * p.TH * p.WP * C * 4 > 0x7fffffffll) return NBM_EUNSUPPORTED;
  p.R = R; p.U = U; p.y = y; p.scale = scale; p.shift = shift; p.mask = mask; p.relu = relu;
  p.T = (int)T; p.N = N; p.C = C; p.nk = C / BK; p.H = H; p.W = W;
  p.r_gs = (long long)B * p.TH * p.WP * C; p.u_gs = N * C;
  p.m_tiles = (int)((T + LIST_BM - 1) / LIST_BM);
  p.tiles = tiles; p.n_blocks = n_blocks; p.blk_info = blk_info;
  if (tiles) { p.T = n_entries; p.m_tiles = n_entries / LIST_BM; }
  hipStream_t st = (hipStream_t)stream;
  // `variant`: 0 = automatic, 128 / 64 = channel-tile width; anything else is refused.  The timing-only ablations of
  // scripts/wino_fused_probe.py (variant + 1000 * ABL: WRONG results by design) exist only in a -DNBM_ABLATE build
  // (make ablate -> libnbm_hip_ablate.so), never in the shipped library.
  int abl = 0;
#ifdef NBM_ABLATE
  abl = variant / 1000;
  variant %= 1000;
#endif
  if (variant != 0 && variant != 64 && variant != 128) return NBM_EINVAL;
  const bool wide = variant == 128 || (variant == 0 && N % 128 == 0);
  p.n_tiles = wide ? (N + 127) / 128 : (N + 63) / 64;
  // Dense launches of the wide shape run ONE workgroup per CU, in rounds of `cus` workgroups: 96-row blocks where they finish sooner than
  // 128-row blocks (256 -> 256 @24x64 at B = 64: 384 blocks = 1.5 rounds of 128 rows against 512 blocks = 2 rounds of 96; 512 -> 512
  // @12x32: 192 of 256 CUs busy against all of them with 3/4 of the work each).  The 96-row shape stages 1.17 x the operand rows per
  // MFMA: it has to win by more than `pen` (measured, scripts/wino_fused_one.py).  NBM_WINO_BM = 128 / 96 forces a shape (experiments).
  int bm = LIST_BM;
  if (wide && !tiles) {
    const char* fe = getenv("NBM_WINO_BM");                 // read per call: the parity test flips it inside one process
    const int force = fe ? atoi(fe) : 0;
    static const int cus = [] { int dev = 0, n = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
                                return n > 0 ? n : 256; }();
    const long long nt = p.n_tiles;
    const long long r128 = ((T + 127) / 128 * nt + cus - 1) / cus, r96 = ((T + 95) / 96 * nt + cus - 1) / cus;
    if (force == 96 || (force == 0 && r96 * 96 * 108 < r128 * 128 * 100)) bm = 96;
    if (force == 128) bm = 128;
    p.m_tiles = (int)((T + bm - 1) / bm);
  }
  const dim3 grid((n_blocks ? (p.m_tiles + 7) / 8 * 8 : p.m_tiles) * p.n_tiles), block(256);
#define NBM_WF(BM_, BN_, WM_, WN_, A_) hipLaunchKernelGGL((wino23_fused_kernel<BM_, BN_, WM_, WN_, A_>), grid, block, 0, st, p)
#ifdef NBM_ABLATE
  if (wide) {
    switch (abl) { case 0: NBM_WF(128, 128, 64, 64, 0); break; case 1: NBM_WF(128, 128, 64, 64, 1); break; case 2: NBM_WF(128, 128, 64, 64, 2); break;
                   case 3: NBM_WF(128, 128, 64, 64, 3); break; case 7: NBM_WF(128, 128, 64, 64, 7); break; default: return NBM_EINVAL; }
  } else {
    switch (abl) { case 0: NBM_WF(128, 64, 64, 32, 0); break; case 1: NBM_WF(128, 64, 64, 32, 1); break; case 2: NBM_WF(128, 64, 64, 32, 2); break;
                   case 3: NBM_WF(128, 64, 64, 32, 3); break; case 7: NBM_WF(128, 64, 64, 32, 7); break; default: return NBM_EINVAL; }
  }
#else
  (void)abl;
  if (!wide) NBM_WF(128, 64, 64, 32, 0);
  else if (bm == 96) NBM_WF(96, 128, 96, 32, 0);
  else NBM_WF(128, 128, 64, 64, 0);
#endif
#undef NBM_WF
  return nbm_launch_status();
}
