// Speed probe 4 (round 4): C[m][n] = sum_k A[m][k] W[n][k] (fp32 in, fp32 out) through bf16 split terms on the bf16 matrix pipe.
// 256 x 128 tile, ONE workgroup of 4 waves per CU (128 x 64 per wave: 2 x 128 accumulator registers), K16 stages:
//   * operand rows split hi / mid / lo on their way from registers to LDS (three bf16 planes, 32-byte rows, halves swizzled by row bit 3);
//   * three LDS plane buffers, three register slots of raw fp32 data: the global loads of stage s + 5 are issued in stage s;
//   * the six products of a stage run in ONE order (a0b0, a0b1, a1b0, a1b1, a0b2, a2b0), so only plane 0 needs a second fragment set:
//     planes 1 and 2 of the next stage are read once their registers are dead (b1 / a1 after MFMA 32, b2 after 40, a2 after 48);
//   * every MFMA gap carries <= 3 split instructions and <= 2 LDS / memory instructions (an MFMA hides ~5 single-issue fillers).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-slp-vectorize gemm4.hip -o gemm4 ; run: ./gemm4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <type_traits>
#include <utility>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define TBM 256
constexpr int BM = 256, BN = 128, TWG = 1;
constexpr int WM = 128, WN = 64, MT = 4, NT = 2;
constexpr int ROWS = BM + BN, CH = ROWS / 128;      // staging chunks (row, 8 floats) per thread and K16 stage
constexpr int PLANE = ROWS * 32;                    // bytes per plane of one stage
constexpr int STAGE = 3 * PLANE;
constexpr int NBUF = 3;

template <int N> using I = std::integral_constant<int, N>;
__device__ inline int slot(int row, int half) { return row * 32 + ((half ^ ((row >> 3) & 1)) << 4); }

template <int TERMS, int ABL, int REM>
__global__ __launch_bounds__(256, 1) void gemm_split(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C,
                                                     int M, int N, int K, int n_tiles, unsigned long long* __restrict__ cyc) {
  const unsigned long long t_start = __builtin_readcyclecounter();
  __shared__ __attribute__((aligned(16))) unsigned char lds[NBUF * STAGE];
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tile_m = wg / n_tiles, tile_n = wg - tile_m * n_tiles;
  const int bm0 = tile_m * BM, bn0 = tile_n * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  const int lrow = lane & 31, lh = lane >> 5;
  const int e = tid & 1, r0 = tid >> 1;             // staging: rows (tid >> 1) + 128 c, 8-float chunk e of the stage's 16
  const float* gp[CH];
  int wofs[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int row = r0 + 128 * c;
    gp[c] = (row < BM ? A + (size_t)(bm0 + row) * K : W + (size_t)(bn0 + row - BM) * K) + e * 8;
    wofs[c] = slot(row, e);
  }
  int a_ofs[MT], b_ofs[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) a_ofs[i] = slot(wm0 + 32 * i + lrow, lh);
#pragma unroll
  for (int j = 0; j < NT; ++j) b_ofs[j] = slot(BM + wn0 + 32 * j + lrow, lh);
  const int S = K / 16;                              // stages

  float raw[3][CH][8];                               // raw data of stage t lives in slot t % 3
  auto gload1 = [&](int t, auto rc, int c, int q) {
    constexpr int R = decltype(rc)::value;
    const int tt = (ABL & 2) ? 0 : (ABL & 16) ? (t & 15) : t < S ? t : S - 1;        // (past the end: the last stage again, never used)
    const f32x4 v = *reinterpret_cast<const f32x4*>(gp[c] + tt * 16 + 4 * q);
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
  constexpr int UOPS = CH * 4 * 11;
  auto uop = [&](auto rc, auto kc) {
    constexpr int R = decltype(rc)::value, k = decltype(kc)::value;
    constexpr int pr = k / 11, ph = k % 11, c = pr / 4, j = pr % 4;
    float& x0 = raw[R][c][2 * j];
    float& x1 = raw[R][c][2 * j + 1];
    if constexpr (ABL & 1) {
      if constexpr (ph == 0) { hp[c][j] = __builtin_bit_cast(unsigned, x0); mp[c][j] = __builtin_bit_cast(unsigned, x1); lp[c][j] = hp[c][j] ^ mp[c][j]; }
    } else {
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
    }
  };
  auto pwrite = [&](int buf, int c, int pl) {       // one plane of one chunk
    const u32x4 v = pl == 0 ? u32x4{hp[c][0], hp[c][1], hp[c][2], hp[c][3]} : pl == 1 ? u32x4{mp[c][0], mp[c][1], mp[c][2], mp[c][3]}
                                                                                 : u32x4{lp[c][0], lp[c][1], lp[c][2], lp[c][3]};
    *reinterpret_cast<u32x4*>(lds + buf * STAGE + pl * PLANE + wofs[c]) = v;
  };
  // fragments: plane 0 in two sets, planes 1 / 2 in one
  bf16x8 a0[2][MT], b0[2][NT], a1[MT], b1[NT], a2[MT], b2[NT];
  auto rdA = [&](int buf, int pl, int i) { return *reinterpret_cast<const bf16x8*>(lds + buf * STAGE + pl * PLANE + a_ofs[i]); };
  auto rdB = [&](int buf, int pl, int j) { return *reinterpret_cast<const bf16x8*>(lds + buf * STAGE + pl * PLANE + b_ofs[j]); };

  f32x16 acc[MT][NT], lo[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = lo[i][j][q] = 0.f;
  constexpr int NPROD = TERMS >= 6 ? 6 : TERMS >= 3 ? 3 : 1, NM = MT * NT * NPROD;
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

  // One stage as ONE basic block.  PH = s mod 6 (compile time): fragment set P = PH & 1, raw slot of the data split here (stage s + 2)
  // R = (PH + 2) % 3; LDS buffers: this stage's s % 3 (a2 is still read from it), the next stage's (s + 1) % 3, written (s + 2) % 3.
  auto stage = [&](int s, auto phc) {
    constexpr int PH = decltype(phc)::value, P = PH & 1, R = (PH + 2) % 3;
    const int bc = (PH % 3), bn = (PH + 1) % 3, bw = (PH + 2) % 3;
    __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): this wave's LDS reads / writes are done
    __builtin_amdgcn_s_barrier();
    [&]<int... Z>(std::integer_sequence<int, Z...>) {
      ([&] {
        constexpr int z = Z;
        mfma1(I<P>{}, I<z>{});
        // 16 gaps per chunk: 11 with four split micro-ops (+ at most one fragment read), 3 with one plane write, 2 with one global load
        if constexpr (!(ABL & 8) && NPROD == 6) {
          if constexpr (z < 4) a2[z] = rdA(bc, 2, z);
          if constexpr (z >= 4 && z < 8) a0[P ^ 1][z - 4] = rdA(bn, 0, z - 4);
          if constexpr (z >= 8 && z < 10) b0[P ^ 1][z - 8] = rdB(bn, 0, z - 8);
          if constexpr (z >= 32 && z < 34) b1[z - 32] = rdB(bn, 1, z - 32);
          if constexpr (z >= 34 && z < 38) a1[z - 34] = rdA(bn, 1, z - 34);
          if constexpr (z >= 40 && z < 42) b2[z - 40] = rdB(bn, 2, z - 40);
        }
        if constexpr (!(ABL & 4) && NPROD == 6) {
          constexpr int c = z / 16, q = z % 16;
          if constexpr (q < 11)
            [&]<int... U>(std::integer_sequence<int, U...>) { (uop(I<R>{}, I<44 * c + 4 * q + U>{}), ...); }(std::make_integer_sequence<int, 4>{});
          if constexpr (q >= 11 && q < 14) pwrite(bw, c, q - 11);
          if constexpr (q >= 14) gload1(s + 5, I<R>{}, c, q - 14);
        }
        __builtin_amdgcn_sched_barrier(0);
      }(), ...);
    }(std::make_integer_sequence<int, NM>{});
  };

  // prologue: stages 0 and 1 split and written, stages 2, 3, 4 requested, every fragment of stage 0 (and a2 is re-read in the stage)
  auto split_all = [&](auto rc, int buf) {
    [&]<int... U>(std::integer_sequence<int, U...>) { (uop(rc, I<U>{}), ...); }(std::make_integer_sequence<int, UOPS>{});
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) pwrite(buf, c, pl);
  };
#pragma unroll
  for (int c = 0; c < CH; ++c) { gload1(0, I<0>{}, c, 0); gload1(0, I<0>{}, c, 1); gload1(1, I<1>{}, c, 0); gload1(1, I<1>{}, c, 1); }
  split_all(I<0>{}, 0);
  split_all(I<1>{}, 1);
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    gload1(2, I<2>{}, c, 0); gload1(2, I<2>{}, c, 1); gload1(3, I<0>{}, c, 0); gload1(3, I<0>{}, c, 1); gload1(4, I<1>{}, c, 0); gload1(4, I<1>{}, c, 1);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MT; ++i) { a0[0][i] = rdA(0, 0, i); a1[i] = rdA(0, 1, i); a2[i] = rdA(0, 2, i); }
#pragma unroll
  for (int j = 0; j < NT; ++j) { b0[0][j] = rdB(0, 0, j); b1[j] = rdB(0, 1, j); b2[j] = rdB(0, 2, j); }
  // REM = S mod 6 stages peeled in front, so that the steady loop is six whole phases with one exit
  int s = 0;
  [&]<int... Q>(std::integer_sequence<int, Q...>) { ((stage(s, I<Q>{}), ++s), ...); }(std::make_integer_sequence<int, REM>{});
  for (; s + 6 <= S; s += 6) {
    stage(s, I<REM % 6>{}); stage(s + 1, I<(REM + 1) % 6>{}); stage(s + 2, I<(REM + 2) % 6>{});
    stage(s + 3, I<(REM + 3) % 6>{}); stage(s + 4, I<(REM + 4) % 6>{}); stage(s + 5, I<(REM + 5) % 6>{});
  }
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * lh;
        C[(size_t)(bm0 + wm0 + 32 * i + row) * N + bn0 + wn0 + 32 * j + lrow] = acc[i][j][q] + lo[i][j][q];
      }
  if (cyc && threadIdx.x == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - t_start;
}

static unsigned long long s = 88172645463325252ull;
static double urand() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (s >> 11) * (1.0 / 9007199254740992.0); }

static unsigned long long* g_cyc = nullptr;
static double g_avg_cyc = 0;
template <int TERMS, int ABL, int REM>
static float run_r(const float* dA, const float* dW, float* dC, int M, int N, int K, int reps) {
  const int mt = M / BM, nt = N / BN;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) gemm_split<TERMS, ABL, REM><<<mt * nt, 256>>>(dA, dW, dC, M, N, K, nt, g_cyc);
  (void)hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) gemm_split<TERMS, ABL, REM><<<mt * nt, 256>>>(dA, dW, dC, M, N, K, nt, g_cyc);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(mt * nt);
  (void)hipMemcpy(h.data(), g_cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double sum = 0; for (auto v : h) sum += (double)v;
  g_avg_cyc = sum / h.size();
  return ms / reps;
}
template <int TERMS, int ABL = 0>
static float run(const float* dA, const float* dW, float* dC, int M, int N, int K, int reps) {
  const int rem = (K / 16) % 6;
  return rem == 0 ? run_r<TERMS, ABL, 0>(dA, dW, dC, M, N, K, reps) : rem == 2 ? run_r<TERMS, ABL, 2>(dA, dW, dC, M, N, K, reps)
                                                                              : run_r<TERMS, ABL, 4>(dA, dW, dC, M, N, K, reps);
}

static void rep(const char* what, float ms, int M, int N, int K) {
  const double tiles = (double)(M / BM) * (N / BN), rounds = tiles / 256.0, st = K / 16;
  printf("      %-34s %.3f ms   %.0f cycles / tile = %.0f / stage (1536 = MFMA only)   clock if the CUs were always busy: %.2f GHz\n", what, ms, g_avg_cyc, g_avg_cyc / st, rounds * g_avg_cyc / (ms * 1e6));
}
int main() {
  (void)hipMalloc(&g_cyc, 1 << 20);
  struct S { int M, N, K; const char* what; } shapes[] = {
    {98304, 256, 1024, "1x1 1024->256 @24x64 B=64 (product: 0.39 ms)"},
    {98304, 1024, 256, "1x1 256->1024 @24x64 B=64 (0.45 ms)"},
    {24576, 3072, 2048, "attention 2048->3072 (2.29 ms)"},
    {98304, 1536, 1024, "attention 1024->1536 (2.32 ms)"},
    {393216, 128, 512, "1x1 512->128 @47x128 B=64 (0.41 ms)"},
    {24576, 512, 4608, "3x3 512->512 @12x32 as a plain GEMM, K = 4608"},
  };
  printf("tile %d x %d, %d workgroup(s) per CU, LDS %d bytes\n", BM, BN, TWG, NBUF * STAGE);
  for (auto& sh : shapes) {
    const int M = sh.M, N = sh.N, K = sh.K;
    std::vector<float> A((size_t)M * K), W((size_t)N * K), C((size_t)M * N);
    for (auto& v : A) { double x = urand() * 2 - 1; v = (float)(x > 0 ? x : 0); }
    for (auto& v : W) v = (float)((urand() * 2 - 1) * sqrt(6.0 / K));
    float *dA, *dW, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dW, W.size() * 4); hipMalloc(&dC, C.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    const double flop = 2.0 * M * N * K;
    const float t6 = run<6>(dA, dW, dC, M, N, K, 10);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double mx = 0, rr = 0;
    for (int t = 0; t < 4096; ++t) {
      const int i = (int)(urand() * M), j = (int)(urand() * N);
      double r = 0; for (int k = 0; k < K; ++k) r += (double)A[(size_t)i * K + k] * W[(size_t)j * K + k];
      mx = fmax(mx, fabs(C[(size_t)i * N + j] - r)); rr += r * r;
    }
    const float t1 = 0.f;
    { float t = run<6, 0>(dA, dW, dC, M, N, K, 10); rep("full", t, M, N, K); }
    { float t = run<6, 1>(dA, dW, dC, M, N, K, 10); rep("no split VALU", t, M, N, K); }
    { float t = run<6, 2>(dA, dW, dC, M, N, K, 10); rep("same K stage reloaded (L1 hits)", t, M, N, K); }
    { float t = run<6, 16>(dA, dW, dC, M, N, K, 10); rep("16 K stages cycled (L2 hits)", t, M, N, K); }
    { float t = run<6, 4>(dA, dW, dC, M, N, K, 10); rep("no split / write / global load", t, M, N, K); }
    { float t = run<6, 12>(dA, dW, dC, M, N, K, 10); rep("MFMA + barriers only", t, M, N, K); }
    printf("M=%6d N=%4d K=%4d  x6: %.3f ms = %.1f TF/s(fp32-equivalent)   hi*hi only: %.3f ms   max err / rms %.2e   %s\n",
           M, N, K, t6, flop / t6 * 1e-9, t1, mx / sqrt(rr / 4096), sh.what);
    hipFree(dA); hipFree(dW); hipFree(dC);
  }
  return 0;
}
