// Implicit-GEMM convolution / batched GEMM with fp32 operands on the gfx950 BF16 matrix core
// (v_mfma_f32_32x32x16_bf16: 1024 FLOP/clk/SIMD against 64 for the fp32 instruction).
//
//   C[m][n] = sum_k A[m][k] * W[n][k],   A, W, C fp32
//
// Every operand value is split into three bf16 terms, x = hi + mid + lo (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid):
// 24 significand bits, the split itself is exact up to the last rounding), and six of the nine partial products are accumulated in
// fp32: hi*hi in one accumulator, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi in a second one (the three dropped products are below
// 2^-24 of the result each).  Measured against float64 (scripts/split_probe/acc.hip, profiles/r03_split_bf16_accuracy.txt): 0.4 x
// the rms error of the fp32 matrix instruction, which adds its K products in one sequential fmaf chain (K roundings where this form
// has K / 16 per accumulator).
//
// 256 x 128 tile, ONE workgroup of 4 waves per CU (128 x 64 per wave: 2 x 128 accumulator registers), K16 stages:
//   * operand rows are split on their way from registers to LDS (three bf16 planes, 32-byte rows, the 16-byte halves swizzled by
//     row bit 3: conflict-free ds_read_b128 and ds_write_b128);
//   * three LDS plane buffers and three register slots of raw fp32 data: the buffer loads of stage s + 5 are issued in stage s;
//   * the six products of a stage run in ONE order (a0b0, a0b1, a1b0, a1b1, a0b2, a2b0), so only plane 0 needs a second fragment
//     set: planes 1 and 2 of the next stage are read once their registers are dead (b1 / a1 after MFMA 32, b2 after 40, a2 at the
//     start of the stage that uses it last);
//   * the split is a stream of single vector instructions (11 per pair of floats); a stage's 48 MFMA gaps are three runs of 16, one per
//     chunk of 8 floats: 11 gaps with four split instructions (and at most one fragment read), 3 with one plane write, 2 with one
//     buffer load -- a 32x32x16 MFMA hides ~5 single-issue fillers of its own wave, a ds_write_b128 costs 13 cycles of issue
//     (/opt/skills/guides/MI355X_MICROARCH.md, cycle constants) -- and `sched_barrier` pins every gap.
// The gather (A from the NHWC activation, taps inner, channel chunks outer; W in KRSC) and the epilogue are those of igemm.hip.
#include <stdlib.h>
#include "nbm_common.h"
#include "igemm_params.h"
#include <type_traits>
#include <utility>

namespace {

using nbm_igemm::IgemmParams;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int PITCH = 36;                           // (epilogue text only)
constexpr int BM = 256, BN = 128, WM = 128, WN = 64, MT = 4, NT = 2;
constexpr int TROWS = BM + BN, CH = TROWS / 128;    // staging chunks (row, 8 floats) per thread and K16 stage
constexpr int PLANE = TROWS * 32;                   // bytes per bf16 plane of one stage
constexpr int STAGE = 3 * PLANE;
constexpr int NBUF = 3;
constexpr int UOPS = CH * 4 * 11;                   // split micro-ops per stage and thread
constexpr int NM = MT * NT * 6;                     // MFMAs per stage and wave
enum { EPI_STD = 0 };

template <int N> using I = std::integral_constant<int, N>;
__device__ inline int slot(int row, int half) { return row * 32 + ((half ^ ((row >> 3) & 1)) << 4); }

template <int REM>
__global__ __launch_bounds__(256, 1) void igemm_split_kernel(const IgemmParams p) {
  constexpr int STAGES = 2, EPI = EPI_STD;          // (names the shared epilogue text expects)
  constexpr bool ROWS = false;
  __shared__ __attribute__((aligned(16))) float lds[NBUF * STAGE / 4];
  unsigned char* const ldsb = reinterpret_cast<unsigned char*>(lds);

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

  // ---- staging assignment: thread -> rows (tid >> 1) + 128 c (c = 0, 1: A rows, c = 2: W row), 8-float chunk e of the stage's 16
  const int e = tid & 1, r0 = tid >> 1;
  long long blk_base;
  {
    const int m0 = bm0 < p.M ? bm0 : 0;
    const int b = (int)nbm_fdiv((unsigned)m0, p.fd_howo), rem = m0 - b * p.HoWo;
    const int oy = (int)nbm_fdiv((unsigned)rem, p.fd_wo), ox = rem - oy * p.Wo;
    blk_base = ((long long)(b * p.H + oy * p.stride - p.pad) * p.W + (ox * p.stride - p.pad)) * p.x_ld;
  }
  unsigned rel[CH];                                  // byte offset of the chunk inside its buffer resource (A: relative to blk_base)
  unsigned long long taps[CH - 1];                   // A rows: one bit per filter tap that lands inside the image
  int wofs[CH];
#pragma unroll
  for (int c = 0; c < CH - 1; ++c) {
    const int m = bm0 + r0 + 128 * c;
    const bool ok = m < p.M;
    const long long mm = ok ? m : 0;
    const int b = (int)nbm_fdiv((unsigned)mm, p.fd_howo), rem = (int)(mm - (long long)b * p.HoWo);
    const int oy = (int)nbm_fdiv((unsigned)rem, p.fd_wo), ox = rem - oy * p.Wo;
    const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
    const long long base = ((long long)(b * p.H + iy0) * p.W + ix0) * p.x_ld;
    rel[c] = ((unsigned)(base - blk_base) + e * 8) * 4u;       // rows ascend with m: never negative
    unsigned long long mk = 0ull;
    unsigned rowm = 0u, colm = 0u;                     // (selects only, as in igemm.hip)
    for (int r = 0; r < p.kh; ++r) rowm |= ((unsigned)(iy0 + r) < (unsigned)p.H ? 1u : 0u) << r;
    for (int s2 = 0; s2 < p.kw; ++s2) colm |= ((unsigned)(ix0 + s2) < (unsigned)p.W ? 1u : 0u) << s2;
    for (int r = 0; r < p.kh; ++r) mk |= ((rowm >> r) & 1u) ? (unsigned long long)colm << (r * p.kw) : 0ull;
    taps[c] = ok ? mk : 0ull;
    wofs[c] = slot(r0 + 128 * c, e);
  }
  {
    const int n = bn0 + r0;
    rel[CH - 1] = n < p.N ? (unsigned)(n * p.w_ld + e * 8) * 4u : 0x80000000u;
    wofs[CH - 1] = slot(BM + r0, e);
  }
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xg + blk_base), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wgp), 0, 0x7ffffff0, 0x00020000);
  int a_ofs[MT], b_ofs[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) a_ofs[i] = slot(wm0 + 32 * i + lrow, lh);
#pragma unroll
  for (int j = 0; j < NT; ++j) b_ofs[j] = slot(BM + wn0 + 32 * j + lrow, lh);
  const int S = 2 * p.nk;                            // K16 stages

  // ---- load cursor: K32 steps run channel chunk OUTER, filter tap INNER (like igemm.hip); a stage is one half of a step
  int ld_t = 0, ld_r = 0, ld_s = 0, ld_c0 = 0;       // stage index, tap row / column, channel of the stage's first float
  unsigned ld_asoff = 0, ld_bsoff = 0; int ld_tap = 0;
  auto cursor_set = [&]() {
    // past the end (the last stages request data nobody splits): A offsets out of range through the tap test (bit 63 is never set; a
    // buffer load's range check covers the per-lane offset only, NOT the scalar one), W reads its first K-slice again
    const bool live = ld_t < S;
    ld_tap = live ? ld_r * p.kw + ld_s : 63;
    ld_asoff = live ? (unsigned)((((long long)ld_r * p.W + ld_s) * p.x_ld + ld_c0) * 4) : 0u;
    ld_bsoff = live ? (unsigned)(((ld_r * p.kw + ld_s) * p.Cin + ld_c0) * 4) : 0u;
  };
  auto cursor_next = [&]() {
    ++ld_t;
    if (ld_t & 1) ld_c0 += 16;
    else {
      ld_c0 -= 16;
      if (++ld_s == p.kw) { ld_s = 0; if (++ld_r == p.kh) { ld_r = 0; ld_c0 += 32; } }
    }
    cursor_set();
  };
  float raw[3][CH][8];                               // raw data of stage t lives in slot t % 3
  auto gload1 = [&](auto rc, int c, int q) {         // piece q (4 floats) of chunk c of the cursor's stage
    constexpr int R = decltype(rc)::value;
    f32x4 v;
    if (c < CH - 1) {
      const unsigned vo = ((taps[c] >> ld_tap) & 1ull) ? rel[c] : 0x80000000u;
      v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo, ld_asoff + 16u * q, 0));
    } else {
      v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, rel[c], ld_bsoff + 16u * q, 0));
    }
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
  // fragments: plane 0 in two sets, planes 1 / 2 in one
  bf16x8 a0[2][MT], b0[2][NT], a1[MT], b1[NT], a2[MT], b2[NT];
  auto rdA = [&](int buf, int pl, int i) { return *reinterpret_cast<const bf16x8*>(ldsb + buf * STAGE + pl * PLANE + a_ofs[i]); };
  auto rdB = [&](int buf, int pl, int j) { return *reinterpret_cast<const bf16x8*>(ldsb + buf * STAGE + pl * PLANE + b_ofs[j]); };

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
  __syncthreads();

#define NBM_EPI_LDS_FLOATS (NBUF * STAGE / 4)
#include "igemm_epilogue.inc"
#undef NBM_EPI_LDS_FLOATS
}

}  // namespace

int nbm_igemm::split_launch(const IgemmParams& p0, int groups, hipStream_t st) {
  IgemmParams p = p0;
  p.m_tiles = (p.M + BM - 1) / BM;
  p.n_tiles = (p.N + BN - 1) / BN;
  const dim3 grid(p.m_tiles * p.n_tiles, 1, groups);
  switch ((2 * p.nk) % 6) {
    case 0: hipLaunchKernelGGL((igemm_split_kernel<0>), grid, dim3(256), 0, st, p); break;
    case 2: hipLaunchKernelGGL((igemm_split_kernel<2>), grid, dim3(256), 0, st, p); break;
    default: hipLaunchKernelGGL((igemm_split_kernel<4>), grid, dim3(256), 0, st, p); break;
  }
  return nbm_launch_status();
}
