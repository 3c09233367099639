// Weight-gradient (TN) GEMM with fp32 operands on the gfx950 BF16 matrix core: the pixel-major twin of igemm_split.hip.
//
//   dW[n][k] += alpha * row_scale[n] * sum_m G[m][n] * X[m][k]        (plain GEMM rows: 1x1 / stride 1 / pad 0 convolutions, nn.Linear,
//                                                                      the grouped Winograd- / cell-domain products; split-K over m)
//
// Same arithmetic as igemm_split.hip (every operand value x = hi + mid + lo in bf16, six of the nine partial products, hi*hi in an
// accumulator of its own), same 256 x 128 tile, K16 stages, three LDS plane buffers / three raw register slots, and the SAME gap-by-gap
// stage schedule.  What differs is the orientation of the operands: the reduction index m (pixels) is the ROW index of both tensors in
// memory, so a thread stages chunks of 8 consecutive CHANNELS of one pixel (coalesced 16-byte loads, one 16-byte LDS write per plane),
// the LDS image of a stage is pixel-major -- [16 pixels][128 channels] bf16 tiles of 256-byte rows -- and the MFMA operands (8
// consecutive pixels of one channel per lane) come out of it through `ds_read_b64_tr_b16`, the hardware transpose read: two reads per
// fragment (pixels 8h .. 8h+3 and 8h+4 .. 8h+7).  16-byte chunk c of row r sits at 256 r + 16 (c ^ (((r & 3) << 2) | ((r >> 2) & 3))):
// conflict-free for the ds_write_b128 staging writes and for the transposed reads (cdna_hip_programming.md T10, image (b)).
#include <stdlib.h>
#include "nbm_common.h"
#include "igemm_split_tn.h"
#include <type_traits>
#include <utility>

namespace {

using nbm_igemm::SplitTnParams;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int BM = 256, BN = 128, WM = 128, WN = 64, MT = 4, NT = 2;
constexpr int CH = 3;                               // staging chunks (pixel, 8 channels) per thread and K16 stage: two of G, one of X
constexpr int TILE = 16 * 256;                      // bytes of one [16 pixels][128 channels] bf16 tile
constexpr int PLANE = 3 * TILE;                     // G channels 0..127, G channels 128..255, X channels 0..127
constexpr int STAGE = 3 * PLANE;
constexpr int NBUF = 3;
constexpr int UOPS = CH * 4 * 11;                   // split micro-ops per stage and thread
constexpr int NM = MT * NT * 6;                     // MFMAs per stage and wave

template <int N> using I = std::integral_constant<int, N>;
__device__ inline int off_b(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

template <int REM>
__global__ __launch_bounds__(256, 1) void igemm_split_tn_kernel(const SplitTnParams p) {
  __shared__ __attribute__((aligned(16))) float lds[NBUF * STAGE / 4];
  unsigned char* const ldsb = reinterpret_cast<unsigned char*>(lds);

  // ---- XCD-aware tile id (bijective for any grid size), as in igemm.hip
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tile_m = wg / p.n_tiles, tile_n = wg - tile_m * p.n_tiles;
  const int bm0 = tile_m * BM, bn0 = tile_n * BN;     // first dW row n, first dW column k of the tile
  const int g = blockIdx.z;
  const float* __restrict__ gg = p.g + (long long)g * p.g_gs;
  const float* __restrict__ xg = p.x + (long long)g * p.x_gs;
  const int m_begin = blockIdx.y * p.k_chunk;          // this split's pixels: [m_begin, m_begin + k_chunk), rows >= M read zeros

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  const int lrow = lane & 31, lh = lane >> 5;

  // ---- staging assignment: thread -> pixel kq of the stage, 8-channel chunk j: G columns bm0 + 8 j and bm0 + 128 + 8 j, X columns bn0 + 8 j
  const int kq = tid >> 4, j8 = tid & 15;
  unsigned rel[CH];
  int wofs[CH];
  {
    const int ca0 = bm0 + 8 * j8, ca1 = bm0 + 128 + 8 * j8, cb = bn0 + 8 * j8;
    rel[0] = ca0 < p.N ? (unsigned)(kq * p.g_ld + ca0) * 4u : 0x80000000u;
    rel[1] = ca1 < p.N ? (unsigned)(kq * p.g_ld + ca1) * 4u : 0x80000000u;
    rel[2] = cb < p.K ? (unsigned)(kq * p.x_ld + cb) * 4u : 0x80000000u;
#pragma unroll
    for (int c = 0; c < CH; ++c) wofs[c] = c * TILE + off_b(kq, j8);
  }
  // ---- fragment addresses (transposed reads): lane 4 q + p of a 16-lane group supplies row (pixel) 8 h + q [+ 4], channels 4 p .. 4 p + 3 of
  // the group's 16 channels; group = (channel half `sub`, pixel half h)
  int a_ofs[MT][2], b_ofs[NT][2];
  {
    const int l16 = lane & 15, q = l16 >> 2, pp = l16 & 3, grp = lane >> 4, sub = grp & 1, h = grp >> 1;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int c0 = wm0 + 32 * i, tile = c0 >> 7, ch = ((c0 & 127) >> 3) + 2 * sub + (pp >> 1);
#pragma unroll
      for (int r = 0; r < 2; ++r) a_ofs[i][r] = tile * TILE + off_b(8 * h + q + 4 * r, ch) + 8 * (pp & 1);
    }
#pragma unroll
    for (int jn = 0; jn < NT; ++jn) {
      const int ch = ((wn0 + 32 * jn) >> 3) + 2 * sub + (pp >> 1);
#pragma unroll
      for (int r = 0; r < 2; ++r) b_ofs[jn][r] = 2 * TILE + off_b(8 * h + q + 4 * r, ch) + 8 * (pp & 1);
    }
  }
  const int S = p.k_chunk >> 4;                      // K16 stages (even: k_chunk % 32 == 0)

  // ---- load cursor: 16 pixels per stage; a buffer resource per stage and operand whose size ends at pixel M (rows beyond read zeros)
  int ld_t = 0;
  __amdgpu_buffer_rsrc_t rsrc_a, rsrc_b;
  auto cursor_set = [&]() {
    const long long mb = (long long)m_begin + 16ll * ld_t;
    long long rows = (ld_t < S) ? (long long)p.M - mb : 0;
    rows = rows < 0 ? 0 : (rows > 16 ? 16 : rows);
    const long long mbs = rows > 0 ? mb : 0;         // (an empty resource never touches memory; keep its base inside the tensor anyway)
    rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gg + mbs * p.g_ld), 0, (int)(rows * p.g_ld * 4), 0x00020000);
    rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xg + mbs * p.x_ld), 0, (int)(rows * p.x_ld * 4), 0x00020000);
  };
  auto cursor_next = [&]() { ++ld_t; cursor_set(); };
  float raw[3][CH][8];                               // raw data of stage t lives in slot t % 3
  auto gload1 = [&](auto rc, int c, int q) {         // piece q (4 floats) of chunk c of the cursor's stage
    constexpr int R = decltype(rc)::value;
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(c < 2 ? rsrc_a : rsrc_b, rel[c] + 16u * q, 0, 0));
#pragma unroll
    for (int x = 0; x < 4; ++x) raw[R][c][4 * q + x] = v[x];
  };
  // ---- the split as a stream of single vector instructions (11 per pair of floats)
  unsigned hp[CH][4], mp[CH][4], lp[CH][4];
  float t0[CH * 4], t1[CH * 4];
  auto cvt2 = [](float a, float b) -> unsigned {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
  };
  auto uop = [&](auto rc, auto kc) {
    constexpr int R = decltype(rc)::value, k = decltype(kc)::value;
    constexpr int pr = k / 11, ph = k % 11, c = pr / 4, j = pr % 4;
    float& x0 = raw[R][c][2 * j];
    float& x1 = raw[R][c][2 * j + 1];
    if constexpr (ph == 0) hp[c][j] = cvt2(x0, x1);
    if constexpr (ph == 1) t0[pr] = __builtin_bit_cast(float, hp[c][j] << 16);
    if constexpr (ph == 2) t1[pr] = __builtin_bit_cast(float, hp[c][j] & 0xffff0000u);
    if constexpr (ph == 3) x0 = x0 - t0[pr];
    if constexpr (ph == 4) x1 = x1 - t1[pr];
    if constexpr (ph == 5) mp[c][j] = cvt2(x0, x1);
    if constexpr (ph == 6) t0[pr] = __builtin_bit_cast(float, mp[c][j] << 16);
    if constexpr (ph == 7) t1[pr] = __builtin_bit_cast(float, mp[c][j] & 0xffff0000u);
    if constexpr (ph == 8) x0 = x0 - t0[pr];
    if constexpr (ph == 9) x1 = x1 - t1[pr];
    if constexpr (ph == 10) lp[c][j] = cvt2(x0, x1);
  };
  auto pwrite = [&](int buf, int c, int pl) {       // one plane of one chunk
    const u32x4 v = pl == 0 ? u32x4{hp[c][0], hp[c][1], hp[c][2], hp[c][3]} : pl == 1 ? u32x4{mp[c][0], mp[c][1], mp[c][2], mp[c][3]}
                                                                                 : u32x4{lp[c][0], lp[c][1], lp[c][2], lp[c][3]};
    *reinterpret_cast<u32x4*>(ldsb + buf * STAGE + pl * PLANE + wofs[c]) = v;
  };
  // fragments: plane 0 in two sets, planes 1 / 2 in one; every fragment = two transposed reads
  bf16x8 a0[2][MT], b0[2][NT], a1[MT], b1[NT], a2[MT], b2[NT];
  auto tr2 = [&](const unsigned char* base, const int (&ofs)[2]) -> bf16x8 {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + ofs[0]));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + ofs[1]));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  auto rdA = [&](int buf, int pl, int i) { return tr2(ldsb + buf * STAGE + pl * PLANE, a_ofs[i]); };
  auto rdB = [&](int buf, int pl, int jn) { return tr2(ldsb + buf * STAGE + pl * PLANE, b_ofs[jn]); };

  f32x16 acc[MT][NT], lo[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = lo[i][j][q] = 0.f;
  auto mfma1 = [&](auto par, auto zc) {
    constexpr int P = decltype(par)::value, z = decltype(zc)::value;
    constexpr int t = z / (MT * NT), i = (z % (MT * NT)) / NT, j = z % NT;
    if constexpr (t == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[P][i], b0[P][j], acc[i][j], 0, 0, 0);
    if constexpr (t == 1) lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[P][i], b1[j], lo[i][j], 0, 0, 0);
    if constexpr (t == 2) lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b0[P][j], lo[i][j], 0, 0, 0);
    if constexpr (t == 3) lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b1[j], lo[i][j], 0, 0, 0);
    if constexpr (t == 4) lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[P][i], b2[j], lo[i][j], 0, 0, 0);
    if constexpr (t == 5) lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[i], b0[P][j], lo[i][j], 0, 0, 0);
  };

  // One stage as ONE basic block.  PH = s mod 6 (compile time): fragment set P = PH & 1; raw slot of the data split here (stage s + 2)
  // R = (PH + 2) % 3, reloaded with stage s + 5 (the cursor's); LDS buffers: this stage's PH % 3 (a2 is still read from it), the next
  // stage's (PH + 1) % 3, written (PH + 2) % 3.
  auto stage = [&](auto phc) {
    constexpr int PH = decltype(phc)::value, P = PH & 1, R = (PH + 2) % 3;
    constexpr int bc = PH % 3, bn = (PH + 1) % 3, bw = (PH + 2) % 3;
    __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): this wave's LDS reads / writes are done
    __builtin_amdgcn_s_barrier();
    [&]<int... Z>(std::integer_sequence<int, Z...>) {
      ([&] {
        constexpr int z = Z;
        mfma1(I<P>{}, I<z>{});
        // 16 gaps per chunk: 11 with four split micro-ops (+ at most one fragment read), 3 with one plane write, 2 with one buffer load
        if constexpr (z < 4) a2[z] = rdA(bc, 2, z);
        if constexpr (z >= 4 && z < 8) a0[P ^ 1][z - 4] = rdA(bn, 0, z - 4);
        if constexpr (z >= 8 && z < 10) b0[P ^ 1][z - 8] = rdB(bn, 0, z - 8);
        if constexpr (z >= 32 && z < 34) b1[z - 32] = rdB(bn, 1, z - 32);
        if constexpr (z >= 34 && z < 38) a1[z - 34] = rdA(bn, 1, z - 34);
        if constexpr (z >= 40 && z < 42) b2[z - 40] = rdB(bn, 2, z - 40);
        constexpr int c = z / 16, q = z % 16;
        if constexpr (q < 11)
          [&]<int... U>(std::integer_sequence<int, U...>) { (uop(I<R>{}, I<44 * c + 4 * q + U>{}), ...); }(std::make_integer_sequence<int, 4>{});
        if constexpr (q >= 11 && q < 14) pwrite(bw, c, q - 11);
        if constexpr (q >= 14) gload1(I<R>{}, c, q - 14);
        __builtin_amdgcn_sched_barrier(0);
      }(), ...);
    }(std::make_integer_sequence<int, NM>{});
    cursor_next();
  };

  // ---- prologue: stages 0 and 1 split and written, stages 2, 3, 4 requested, every fragment of stage 0 read
  auto load_all = [&](auto rc) {
#pragma unroll
    for (int c = 0; c < CH; ++c) { gload1(rc, c, 0); gload1(rc, c, 1); }
    cursor_next();
  };
  auto split_all = [&](auto rc, int buf) {
    [&]<int... U>(std::integer_sequence<int, U...>) { (uop(rc, I<U>{}), ...); }(std::make_integer_sequence<int, UOPS>{});
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) pwrite(buf, c, pl);
  };
  cursor_set();
  load_all(I<0>{});
  load_all(I<1>{});
  split_all(I<0>{}, 0);
  split_all(I<1>{}, 1);
  load_all(I<2>{});
  load_all(I<0>{});
  load_all(I<1>{});
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MT; ++i) { a0[0][i] = rdA(0, 0, i); a1[i] = rdA(0, 1, i); a2[i] = rdA(0, 2, i); }
#pragma unroll
  for (int j = 0; j < NT; ++j) { b0[0][j] = rdB(0, 0, j); b1[j] = rdB(0, 1, j); b2[j] = rdB(0, 2, j); }
  // REM = S mod 6 stages peeled in front, so that the steady loop is six whole phases with one exit
  int s = REM;
  [&]<int... Q>(std::integer_sequence<int, Q...>) { (stage(I<Q>{}), ...); }(std::make_integer_sequence<int, REM>{});
  for (; s + 6 <= S; s += 6) {
    stage(I<REM % 6>{}); stage(I<(REM + 1) % 6>{}); stage(I<(REM + 2) % 6>{});
    stage(I<(REM + 3) % 6>{}); stage(I<(REM + 4) % 6>{}); stage(I<(REM + 5) % 6>{});
  }
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] += lo[i][j][q];

  // ---- epilogue: split-K partial sums into the zero-initialised (or partial) dW with fp32 atomics, as igemm_tn_kernel does.
  // C/D layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
  float* __restrict__ og = p.out + (long long)g * p.out_gs;
#pragma unroll
  for (int jn = 0; jn < NT; ++jn) {
    const int col = bn0 + wn0 + jn * 32 + lrow;
    if (col >= p.K) continue;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = bm0 + wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (n >= p.N) continue;
        float v = acc[i][jn][e] * p.alpha;
        if (p.row_scale) v *= p.row_scale[n];
        atomicAdd(og + (long long)n * p.out_ld + col, v);
      }
  }
}

}  // namespace

int nbm_igemm::split_tn_launch(const SplitTnParams& p, int splits, int groups, hipStream_t st) {
  const dim3 grid(((p.N + BM - 1) / BM) * p.n_tiles, splits, groups);
  switch ((p.k_chunk >> 4) % 6) {
    case 0: hipLaunchKernelGGL((igemm_split_tn_kernel<0>), grid, dim3(256), 0, st, p); break;
    case 2: hipLaunchKernelGGL((igemm_split_tn_kernel<2>), grid, dim3(256), 0, st, p); break;
    default: hipLaunchKernelGGL((igemm_split_tn_kernel<4>), grid, dim3(256), 0, st, p); break;
  }
  return nbm_launch_status();
}
