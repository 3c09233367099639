// Accuracy probe: fp32 GEMM through bf16 split terms on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16) against the
// fp32 pipe (v_mfma_f32_32x32x2_f32) and a float64 host result.  One wave per 32x32 output tile, operands straight from
// global memory.  Build: hipcc --offload-arch=gfx950 -O2 acc.hip -o acc ; run: ./acc
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ inline void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x; float r = x - (float)h;
  m = (__bf16)r; r = r - (float)m;
  l = (__bf16)r;
}

// MODE 0: fp32 pipe.  MODE 3 / 6 / 9: number of split products (ordered small terms first inside one K-step of 16)
template <int MODE>
__global__ void gemm(const float* A, const float* W, float* C, int M, int N, int K) {
  const int lane = threadIdx.x & 63;
  const int tm = blockIdx.x * 32, tn = blockIdx.y * 32;
  const int r = lane & 31, h = lane >> 5;
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  if constexpr (MODE == 0) {
    for (int k = 0; k < K; k += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(size_t)(tm + r) * K + k + h], W[(size_t)(tn + r) * K + k + h], acc, 0, 0, 0);
  } else {
    for (int k = 0; k < K; k += 16) {
      bf16x8 a0, a1, a2, b0, b1, b2;
      for (int i = 0; i < 8; ++i) {
        __bf16 x, y, z;
        split3(A[(size_t)(tm + r) * K + k + 8 * h + i], x, y, z); a0[i] = x; a1[i] = y; a2[i] = z;
        split3(W[(size_t)(tn + r) * K + k + 8 * h + i], x, y, z); b0[i] = x; b1[i] = y; b2[i] = z;
      }
      if constexpr (MODE >= 9) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc, 0, 0, 0);
      }
      if constexpr (MODE >= 6) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc, 0, 0, 0);
    }
  }
  for (int e = 0; e < 16; ++e) {
    const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
    C[(size_t)(tm + row) * N + tn + r] = acc[e];
  }
}
// MODE 16: the small terms in their own accumulator, added once at the end
__global__ void gemm_two_acc(const float* A, const float* W, float* C, int M, int N, int K) {
  const int lane = threadIdx.x & 63;
  const int tm = blockIdx.x * 32, tn = blockIdx.y * 32;
  const int r = lane & 31, h = lane >> 5;
  f32x16 acc, lo;
  for (int e = 0; e < 16; ++e) acc[e] = lo[e] = 0.f;
  for (int k = 0; k < K; k += 16) {
    bf16x8 a0, a1, a2, b0, b1, b2;
    for (int i = 0; i < 8; ++i) {
      __bf16 x, y, z;
      split3(A[(size_t)(tm + r) * K + k + 8 * h + i], x, y, z); a0[i] = x; a1[i] = y; a2[i] = z;
      split3(W[(size_t)(tn + r) * K + k + 8 * h + i], x, y, z); b0[i] = x; b1[i] = y; b2[i] = z;
    }
    lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, lo, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc, 0, 0, 0);
  }
  for (int e = 0; e < 16; ++e) {
    const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
    C[(size_t)(tm + row) * N + tn + r] = acc[e] + lo[e];
  }
}

static unsigned long long s = 88172645463325252ull;
static double urand() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (s >> 11) * (1.0 / 9007199254740992.0); }
static double nrand() { return sqrt(-2 * log(urand() + 1e-300)) * cos(6.283185307179586 * urand()); }

int main() {
  const int M = 256, N = 256;
  const int Ks[] = {64, 256, 576, 1024, 3456};
  printf("%6s %-10s %12s %12s %12s  (errors relative to the rms of the exact output)\n", "K", "path", "max", "rms", "mean(signed)");
  for (int K : Ks) {
    std::vector<float> A((size_t)M * K), W((size_t)N * K), C((size_t)M * N);
    for (auto& v : A) { double x = nrand(); v = (float)(x > 0 ? x : 0); }          // post-ReLU activations
    for (auto& v : W) v = (float)(nrand() * sqrt(2.0 / K));
    std::vector<double> R((size_t)M * N);
    double rr = 0;
    for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) {
      double t = 0; for (int k = 0; k < K; ++k) t += (double)A[(size_t)i * K + k] * W[(size_t)j * K + k];
      R[(size_t)i * N + j] = t; rr += t * t; }
    rr = sqrt(rr / (M * N));
    // float32 sequential fmaf chain on the host (what the reference's CPU kernels roughly do)
    float *dA, *dW, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dW, W.size() * 4); hipMalloc(&dC, C.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    const char* names[] = {"fp32 pipe", "bf16 x3", "bf16 x6", "bf16 x9", "x6 2 acc", "host fmaf"};
    for (int mode = 0; mode < 6; ++mode) {
      dim3 g(M / 32, N / 32);
      if (mode == 0) gemm<0><<<g, 64>>>(dA, dW, dC, M, N, K);
      if (mode == 1) gemm<3><<<g, 64>>>(dA, dW, dC, M, N, K);
      if (mode == 2) gemm<6><<<g, 64>>>(dA, dW, dC, M, N, K);
      if (mode == 3) gemm<9><<<g, 64>>>(dA, dW, dC, M, N, K);
      if (mode == 4) gemm_two_acc<<<g, 64>>>(dA, dW, dC, M, N, K);
      if (mode < 5) { hipDeviceSynchronize(); hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost); }
      else for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) {
        float t = 0; for (int k = 0; k < K; ++k) t = fmaf(A[(size_t)i * K + k], W[(size_t)j * K + k], t); C[(size_t)i * N + j] = t; }
      double mx = 0, sq = 0, sg = 0;
      for (size_t i = 0; i < C.size(); ++i) { double e = C[i] - R[i]; mx = fmax(mx, fabs(e)); sq += e * e; sg += e; }
      printf("%6d %-10s %12.3e %12.3e %12.3e\n", K, names[mode], mx / rr, sqrt(sq / C.size()) / rr, sg / C.size() / rr);
    }
    hipFree(dA); hipFree(dW); hipFree(dC);
  }
  return 0;
}
