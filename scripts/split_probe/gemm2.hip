// Speed probe 2 (round 4): C[m][n] = sum_k A[m][k] W[n][k] (fp32 in, fp32 out) through bf16 split terms on the bf16 matrix pipe,
// software pipelined.  BM x 128 tile (BM = 128: two workgroups per CU, BM = 256: one), 4 waves (BM/2 x 64 each), K16 stages
// double buffered in LDS (three bf16 planes per operand row, 32-byte rows, 16-byte halves swizzled by row bit 3: conflict-free
// ds_read_b128 and ds_write_b128), fragments double buffered in registers, ONE barrier per stage:
//   stage s:  barrier | read fragments of stage s+1 | MFMAs of stage s (six products) | split + write stage s+2 | global loads
// Build: hipcc --offload-arch=gfx950 -O3 -DTBM=256 gemm2.hip -o gemm2_256 ; run: ./gemm2_256
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <type_traits>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#ifndef TBM
#define TBM 256
#endif
constexpr int BM = TBM, BN = 128;
constexpr int TWG = BM == 128 ? 2 : 1;
constexpr int WM = BM / 2, WN = 64, MT = WM / 32, NT = WN / 32;
constexpr int ROWS = BM + BN, CH = ROWS / 128;      // staging chunks (row, 8 floats) per thread and K16 stage
constexpr int PLANE = ROWS * 32;                    // bytes per plane of one stage
constexpr int STAGE = 3 * PLANE;
#ifndef VPM
#define VPM 3
#endif
#ifndef WZ0
#define WZ0 15
#endif
#ifndef WZS
#define WZS 3
#endif

__device__ inline int slot(int row, int half) { return row * 32 + ((half ^ ((row >> 3) & 1)) << 4); }

__device__ inline void split8(const f32x4 lo4, const f32x4 hi4, u32x4& p0, u32x4& p1, u32x4& p2) {
  float x[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
  bf16x8 h, m, l;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    h[i] = (__bf16)x[i]; float r = x[i] - (float)h[i];
    m[i] = (__bf16)r; r = r - (float)m[i];
    l[i] = (__bf16)r;
  }
  p0 = __builtin_bit_cast(u32x4, h); p1 = __builtin_bit_cast(u32x4, m); p2 = __builtin_bit_cast(u32x4, l);
}

template <int TERMS, int ABL = 0>
__global__ __launch_bounds__(256, TWG) void gemm_split(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C,
                                                       int M, int N, int K, int n_tiles) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tile_m = wg / n_tiles, tile_n = wg - tile_m * n_tiles;
  const int bm0 = tile_m * BM, bn0 = tile_n * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  const int lrow = lane & 31, lh = lane >> 5;
  // staging: thread -> rows (tid >> 1) + 128 i, 8-float chunk e = tid & 1 of each K16 half
  const int e = tid & 1, r0 = tid >> 1;
  const float* gp[CH];
  int wofs[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int row = r0 + 128 * i;
    gp[i] = (row < BM ? A + (size_t)(bm0 + row) * K : W + (size_t)(bn0 + row - BM) * K) + e * 8;
    wofs[i] = slot(row, e);
  }
  int a_ofs[MT], b_ofs[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) a_ofs[i] = slot(wm0 + 32 * i + lrow, lh);
#pragma unroll
  for (int j = 0; j < NT; ++j) b_ofs[j] = slot(BM + wn0 + 32 * j + lrow, lh);

  f32x4 raw[2][CH][2];                 // [K16 half][chunk][4 + 4 floats] of one K32 tile
  auto gload = [&](int kt, int h) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      raw[h][i][0] = *reinterpret_cast<const f32x4*>(gp[i] + kt * 32 + h * 16);
      raw[h][i][1] = *reinterpret_cast<const f32x4*>(gp[i] + kt * 32 + h * 16 + 4);
    }
  };
  auto swrite = [&](int buf, int h) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      u32x4 p0, p1, p2;
      if constexpr (ABL & 1) { p0 = __builtin_bit_cast(u32x4, raw[h][i][0]); p1 = __builtin_bit_cast(u32x4, raw[h][i][1]); p2 = p0 ^ p1; }
      else split8(raw[h][i][0], raw[h][i][1], p0, p1, p2);
      unsigned char* d = lds + buf * STAGE + wofs[i];
      *reinterpret_cast<u32x4*>(d) = p0; *reinterpret_cast<u32x4*>(d + PLANE) = p1; *reinterpret_cast<u32x4*>(d + 2 * PLANE) = p2;
    }
  };
  bf16x8 fa[2][MT][3], fb[2][NT][3];
  auto fread = [&](int buf, auto par) {
    constexpr int P = decltype(par)::value;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int i = 0; i < MT; ++i) fa[P][i][pl] = *reinterpret_cast<const bf16x8*>(lds + buf * STAGE + pl * PLANE + a_ofs[i]);
#pragma unroll
      for (int j = 0; j < NT; ++j) fb[P][j][pl] = *reinterpret_cast<const bf16x8*>(lds + buf * STAGE + pl * PLANE + b_ofs[j]);
    }
  };
  f32x16 acc[MT][NT], lo[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = lo[i][j][q] = 0.f;
  auto mfmas = [&](auto par) {
    constexpr int P = decltype(par)::value;
    // product-outer order: eight independent accumulators between two MFMAs on the same one
    constexpr int pa[6] = {0, 2, 1, 0, 1, 0}, pb[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
    for (int t = (TERMS >= 6 ? 0 : TERMS >= 3 ? 3 : 5); t < 6; ++t)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if (t < 5) lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[P][i][pa[t]], fb[P][j][pb[t]], lo[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[P][i][0], fb[P][j][0], acc[i][j], 0, 0, 0);
        }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;

  const int nk = K / 32;               // K32 tiles; stage s = 2 kt + h lives in LDS buffer h
  // One K16 stage as ONE basic block.  H: which half / LDS buffer / fragment set holds the stage; READ: fetch the next stage's fragments;
  // WRITE: split + store the stage after next into the buffer this stage's fragments came from, then request the tile after that.
  auto stage = [&](int kt, auto hc, auto readc, auto writec) {
    constexpr int H = decltype(hc)::value;
    constexpr bool READ = decltype(readc)::value && !(ABL & 8), WRITE = decltype(writec)::value && !(ABL & 4);
    __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): this wave's LDS reads / writes are done
    __builtin_amdgcn_s_barrier();
    if constexpr (READ) fread(H ^ 1, std::integral_constant<int, H ^ 1>{});
    mfmas(hc);
    if constexpr (WRITE) {
      swrite(H, H);
      const int kn = (ABL & 2) ? 0 : kt + 2 < nk ? kt + 2 : nk - 1;       // (the last two iterations request the last tile again: no branch)
      gload(kn, H);
    }
    constexpr int NM = MT * NT * (TERMS >= 6 ? 6 : TERMS >= 3 ? 3 : 1);
#pragma unroll
    for (int z = 0; z < NM; ++z) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (READ && z < 3 * (MT + NT)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if (WRITE) {
        // the split is a linear stream of ~44 vector instructions per chunk: VPM per gap; a chunk's three LDS writes go into the gaps
        // right after its last instruction, its two global loads (next tile, same registers) behind them
        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
        constexpr int GPC = (44 + VPM - 1) / VPM;        // gaps per chunk
        const int c = z / GPC, q = z - c * GPC;
        if (c >= 1 && c <= CH && q < 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        if (c >= 1 && c <= CH && q >= 3 && q < 5) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  using T = std::true_type;
  using F = std::false_type;
  gload(0, 0); gload(0, 1);
  swrite(0, 0); swrite(1, 1);
  if (nk > 1) { gload(1, 0); gload(1, 1); }
  __syncthreads();
  fread(0, P0{});
  for (int kt = 0; kt + 1 < nk; ++kt) {
    stage(kt, P0{}, T{}, T{});
    stage(kt, P1{}, T{}, T{});
  }
  stage(nk - 1, P0{}, T{}, F{});
  stage(nk - 1, P1{}, F{}, F{});
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * lh;
        C[(size_t)(bm0 + wm0 + 32 * i + row) * N + bn0 + wn0 + 32 * j + lrow] = acc[i][j][q] + lo[i][j][q];
      }
}

static unsigned long long s = 88172645463325252ull;
static double urand() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (s >> 11) * (1.0 / 9007199254740992.0); }

template <int TERMS, int ABL = 0>
static float run(const float* dA, const float* dW, float* dC, int M, int N, int K, int reps) {
  const int mt = M / BM, nt = N / BN;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) gemm_split<TERMS, ABL><<<mt * nt, 256>>>(dA, dW, dC, M, N, K, nt);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) gemm_split<TERMS, ABL><<<mt * nt, 256>>>(dA, dW, dC, M, N, K, nt);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  struct S { int M, N, K; const char* what; } shapes[] = {
    {98304, 256, 1024, "1x1 1024->256 @24x64 B=64 (product: 0.39 ms)"},
    {98304, 1024, 256, "1x1 256->1024 @24x64 B=64 (0.45 ms)"},
    {24576, 3072, 2048, "attention 2048->3072 (2.29 ms)"},
    {98304, 1536, 1024, "attention 1024->1536 (2.32 ms)"},
    {393216, 128, 512, "1x1 512->128 @47x128 B=64 (0.41 ms)"},
    {24576, 512, 4608, "3x3 512->512 @12x32 as a plain GEMM, K = 4608"},
  };
  printf("tile %d x %d, %d workgroup(s) per CU, LDS %d bytes\n", BM, BN, TWG, 2 * STAGE);
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
    const float t1 = run<1>(dA, dW, dC, M, N, K, 10);
    printf("   ablations: no split VALU %.3f   same K tile reloaded %.3f   no write / global load %.3f   MFMA + barriers only %.3f ms\n",
           run<6, 1>(dA, dW, dC, M, N, K, 10), run<6, 2>(dA, dW, dC, M, N, K, 10), run<6, 4>(dA, dW, dC, M, N, K, 10), run<6, 12>(dA, dW, dC, M, N, K, 10));
    printf("M=%6d N=%4d K=%4d  x6: %.3f ms = %.1f TF/s(fp32-equivalent)   hi*hi only: %.3f ms   max err / rms %.2e   %s\n",
           M, N, K, t6, flop / t6 * 1e-9, t1, mx / sqrt(rr / 4096), sh.what);
    hipFree(dA); hipFree(dW); hipFree(dC);
  }
  return 0;
}
