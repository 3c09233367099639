// How fast does a wave-private 32 x 256 tile of fp32 go out to HBM, depending on the shape of each store instruction?
//   mode 0: one instruction = 8 rows x 128 B (the streaming / tiled GEMM epilogues: a 32-wide block at a time)
//   mode 1: one instruction = 4 rows x 256 B;  mode 2: 2 rows x 512 B;  mode 3: 1 row x 1 KB
// Build: hipcc --offload-arch=gfx950 -O3 store.hip -o store
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int MODE>
__global__ __launch_bounds__(512) void k(float* __restrict__ y, int m_tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int SEG = 32 << MODE;                 // floats per row segment of one instruction
  constexpr int LPR = SEG / 4;                    // lanes per row
  constexpr int RPI = 64 / LPR;                   // rows per instruction
  const f32x4 v = {1.f, 2.f, 3.f, (float)lane};
  for (int tile = blockIdx.x * 8 + wave; tile < m_tiles; tile += gridDim.x * 8) {
    float* base = y + (size_t)tile * 32 * 256;
#pragma unroll
    for (int seg = 0; seg < 256 / SEG; ++seg)
#pragma unroll
      for (int r0 = 0; r0 < 32; r0 += RPI) {
        const int r = r0 + lane / LPR, c = seg * SEG + (lane % LPR) * 4;
        *reinterpret_cast<f32x4*>(base + r * 256 + c) = v;
      }
  }
}
template <int MODE> float run(float* y, int m_tiles) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<256, 512>>>(y, m_tiles);
  hipEventRecord(a);
  for (int i = 0; i < 10; ++i) k<MODE><<<256, 512>>>(y, m_tiles);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 10;
}
int main() {
  const int M = 64 * 94 * 256, m_tiles = M / 32;
  float* y; hipMalloc(&y, (size_t)M * 256 * 4);
  const double gb = (double)M * 256 * 4e-9;
  printf("1.58 GB written per launch\n");
  printf("8 rows x 128 B: %.3f ms = %.2f TB/s\n", run<0>(y, m_tiles), gb / run<0>(y, m_tiles));
  printf("4 rows x 256 B: %.3f ms = %.2f TB/s\n", run<1>(y, m_tiles), gb / run<1>(y, m_tiles));
  printf("2 rows x 512 B: %.3f ms = %.2f TB/s\n", run<2>(y, m_tiles), gb / run<2>(y, m_tiles));
  printf("1 row  x 1 KB : %.3f ms = %.2f TB/s\n", run<3>(y, m_tiles), gb / run<3>(y, m_tiles));
  return 0;
}
