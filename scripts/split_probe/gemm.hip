// Speed probe: C[m][n] = sum_k A[m][k] W[n][k] (fp32 in, fp32 out) through bf16 split terms on the bf16 matrix pipe.
// 128 x 128 tile, 4 waves (64 x 64 each), K-step 32; operands split hi / mid / lo while they are staged into LDS
// (three bf16 planes, 80-byte rows), six products per K-slice, the five small ones in their own accumulator.
// Build: hipcc --offload-arch=gfx950 -O3 gemm.hip -o gemm ; run: ./gemm
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#ifndef TBM
#define TBM 256
#endif
#ifndef TWG
#define TWG 1
#endif
constexpr int BM = TBM, BN = 128, BK = 32;
constexpr int WM = BM / 2, WN = 64, MT = WM / 32, NT = WN / 32, AR = BM / 64, BR = BN / 64;
constexpr int PITCH = 80;                       // bytes per LDS row of one plane (64 + 16 pad)
constexpr int PLANE = (BM + BN) * PITCH;        // bytes per plane

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
__device__ inline void nosplit8(const f32x4 lo4, const f32x4 hi4, u32x4& p0, u32x4& p1, u32x4& p2) {
  p0 = __builtin_bit_cast(u32x4, lo4); p1 = __builtin_bit_cast(u32x4, hi4); p2 = p0 ^ p1;
}

template <int TERMS, int ABL = 0>
__global__ __launch_bounds__(256, TWG) void gemm_split(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C,
                                                    int M, int N, int K, int n_tiles) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * PLANE];
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tile_m = wg / n_tiles, tile_n = wg - tile_m * n_tiles;
  const int bm0 = tile_m * BM, bn0 = tile_n * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  const int lrow = lane & 31, lh = lane >> 5;
  // staging: thread -> (row r0 + 64 i, 8-float chunk c)
  const int c = tid & 3, r0 = tid >> 2;
  const float* ap[AR]; const float* wp[BR];
#pragma unroll
  for (int i = 0; i < AR; ++i) ap[i] = A + (size_t)(bm0 + r0 + 64 * i) * K + c * 8;
#pragma unroll
  for (int i = 0; i < BR; ++i) wp[i] = W + (size_t)(bn0 + r0 + 64 * i) * K + c * 8;
  f32x4 ra[AR][2], rb[BR][2];
  auto load = [&](int kt) {
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      ra[i][0] = *reinterpret_cast<const f32x4*>(ap[i] + kt * BK);
      ra[i][1] = *reinterpret_cast<const f32x4*>(ap[i] + kt * BK + 4);
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      rb[i][0] = *reinterpret_cast<const f32x4*>(wp[i] + kt * BK);
      rb[i][1] = *reinterpret_cast<const f32x4*>(wp[i] + kt * BK + 4);
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      u32x4 p0, p1, p2;
      if constexpr (ABL & 1) nosplit8(ra[i][0], ra[i][1], p0, p1, p2); else split8(ra[i][0], ra[i][1], p0, p1, p2);
      unsigned char* d = lds + (r0 + 64 * i) * PITCH + c * 16;
      *reinterpret_cast<u32x4*>(d) = p0; *reinterpret_cast<u32x4*>(d + PLANE) = p1; *reinterpret_cast<u32x4*>(d + 2 * PLANE) = p2;
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      u32x4 p0, p1, p2;
      if constexpr (ABL & 1) nosplit8(rb[i][0], rb[i][1], p0, p1, p2); else split8(rb[i][0], rb[i][1], p0, p1, p2);
      unsigned char* d = lds + (BM + r0 + 64 * i) * PITCH + c * 16;
      *reinterpret_cast<u32x4*>(d) = p0; *reinterpret_cast<u32x4*>(d + PLANE) = p1; *reinterpret_cast<u32x4*>(d + 2 * PLANE) = p2;
    }
  };
  f32x16 acc[MT][NT], lo[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = lo[i][j][e] = 0.f;

  const int nk = K / BK;
  load(0);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt) __syncthreads();
    store();
    if (kt + 1 < nk) load((ABL & 2) ? 0 : kt + 1);
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      bf16x8 a[MT][3], b[NT][3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
          a[i][pl] = *reinterpret_cast<const bf16x8*>(lds + pl * PLANE + (wm0 + 32 * i + lrow) * PITCH + (2 * g + lh) * 16);
#pragma unroll
        for (int i = 0; i < NT; ++i)
          b[i][pl] = *reinterpret_cast<const bf16x8*>(lds + pl * PLANE + (BM + wn0 + 32 * i + lrow) * PITCH + (2 * g + lh) * 16);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if constexpr (TERMS >= 6) {
            lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], lo[i][j], 0, 0, 0);
            lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], lo[i][j], 0, 0, 0);
            lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], lo[i][j], 0, 0, 0);
          }
          if constexpr (TERMS >= 3) {
            lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], lo[i][j], 0, 0, 0);
            lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], lo[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
        }
    }
  }
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
        C[(size_t)(bm0 + wm0 + 32 * i + row) * N + bn0 + wn0 + 32 * j + lrow] = acc[i][j][e] + lo[i][j][e];
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
    {98304, 1024, 256, "1x1 256->1024 @24x64 B=64 (0.54 ms incl. residual)"},
    {24576, 3072, 2048, "attention 2048->3072 (2.23 ms)"},
    {98304, 1536, 1024, "attention 1024->1536 (2.29 ms)"},
    {393216, 128, 512, "1x1 512->128 @47x128 B=64 (0.41 ms)"},
    {24576, 512, 4608, "3x3 512->512 @12x32 as a plain GEMM, K = 4608"},
  };
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
    // spot check against float64 on 4096 random entries
    double mx = 0, rr = 0;
    for (int t = 0; t < 4096; ++t) {
      const int i = (int)(urand() * M), j = (int)(urand() * N);
      double r = 0; for (int k = 0; k < K; ++k) r += (double)A[(size_t)i * K + k] * W[(size_t)j * K + k];
      mx = fmax(mx, fabs(C[(size_t)i * N + j] - r)); rr += r * r;
    }
    const float t1 = run<1>(dA, dW, dC, M, N, K, 10);
    const float ta = run<6, 1>(dA, dW, dC, M, N, K, 10), tb = run<6, 2>(dA, dW, dC, M, N, K, 10), tc = run<6, 3>(dA, dW, dC, M, N, K, 10), td = run<1, 3>(dA, dW, dC, M, N, K, 10);
    printf("   ablations: no split VALU %.3f   same K-tile reloaded (L1/L2 hits) %.3f   both %.3f   both + one product %.3f ms\n", ta, tb, tc, td);
    printf("M=%6d N=%4d K=%4d  x6: %.3f ms = %.1f TF/s(fp32-equivalent)   hi*hi only: %.3f ms   max err / rms %.2e   %s\n",
           M, N, K, t6, flop / t6 * 1e-9, t1, mx / sqrt(rr / 4096), sh.what);
    hipFree(dA); hipFree(dW); hipFree(dC);
  }
  return 0;
}
