// Torch-free, library-free control experiment for DESIGN 4d: two captured graph execs alive in one process, each a chain of NODES
// kernel nodes (large by-value argument structs, a data dependency from node to node through device memory, memset nodes in
// between), the second exec launched immediately after the first one's launch has completed.  Every kernel checks the checksum of
// its own argument struct (stale / foreign kernel arguments show up as `bad_args`) and the chain's result shows up in `wrong`.
// No kernel takes a pointer argument: a corrupted argument cannot turn into a wild access.
//   hipcc --offload-arch=gfx950 -O2 scripts/graph_pair_repro.hip -o scripts/graph_pair_repro && scripts/graph_pair_repro [nodes] [delay_ms]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
constexpr int N = 1 << 22;
__device__ float g_buf[2][2][N];            // [graph][ping-pong][N]
__device__ unsigned g_bad_args[2];
__device__ unsigned g_scratch[2][1024];         // target of the memset nodes between the kernel nodes
struct Args { unsigned long long magic[30]; unsigned long long sum; int graph, node, n, pad; };

__global__ void step_kernel(Args a) {
  unsigned long long s = 0;
  for (int i = 0; i < 30; ++i) s += a.magic[i] * (i + 1);
  if (s != a.sum || a.graph < 0 || a.graph > 1 || a.n != N) { if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&g_bad_args[a.graph & 1], 1u); return; }
  const float* src = g_buf[a.graph][a.node & 1];
  float* dst = g_buf[a.graph][(a.node + 1) & 1];
  // reads a neighbour that another workgroup (most likely on another XCD) wrote in the previous node
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) dst[i] = src[(i + 4099) & (N - 1)] + 1.0f;
}

int main(int argc, char** argv) {
  int nodes = argc > 1 ? atoi(argv[1]) : 300, delay_ms = argc > 2 ? atoi(argv[2]) : 0;
  int rt = 0; CK(hipRuntimeGetVersion(&rt));
  float* base; unsigned *bad, *scr;
  CK(hipGetSymbolAddress((void**)&base, HIP_SYMBOL(g_buf))); CK(hipGetSymbolAddress((void**)&bad, HIP_SYMBOL(g_bad_args)));
  CK(hipGetSymbolAddress((void**)&scr, HIP_SYMBOL(g_scratch)));
  hipStream_t st[2]; hipGraph_t gr[2]; hipGraphExec_t ex[2];
  for (int g = 0; g < 2; ++g) {
    CK(hipStreamCreate(&st[g]));
    CK(hipStreamBeginCapture(st[g], hipStreamCaptureModeGlobal));
    CK(hipMemsetAsync(base + (size_t)g * 2 * N, 0, sizeof(float) * N, st[g]));
    for (int k = 0; k < nodes; ++k) {
      Args a; a.sum = 0; a.graph = g; a.node = k; a.n = N; a.pad = 0;
      for (int i = 0; i < 30; ++i) { a.magic[i] = 0x9E3779B97F4A7C15ull * (unsigned long long)(g * 100003 + k * 31 + i + 1); a.sum += a.magic[i] * (i + 1); }
      hipLaunchKernelGGL(step_kernel, dim3(2048), dim3(256), 0, st[g], a);
      if (k % 16 == 7) CK(hipMemsetAsync(scr + g * 1024, 0, 1024 * sizeof(unsigned), st[g]));   // memset nodes like the product's counters
    }
    CK(hipStreamEndCapture(st[g], &gr[g]));
    CK(hipGraphInstantiate(&ex[g], gr[g], nullptr, nullptr, 0));
  }
  CK(hipMemset(bad, 0, 2 * sizeof(unsigned)));
  std::vector<float> h(N);
  int fails = 0;
  for (int round = 0; round < 3; ++round) {
    for (int g = 0; g < 2; ++g) {
      CK(hipGraphLaunch(ex[g], st[g])); CK(hipStreamSynchronize(st[g]));
      if (delay_ms) std::this_thread::sleep_for(std::chrono::milliseconds(delay_ms));
    }
    for (int g = 0; g < 2; ++g) {
      CK(hipMemcpy(h.data(), base + ((size_t)g * 2 + (nodes & 1)) * N, sizeof(float) * N, hipMemcpyDeviceToHost));
      long wrong = 0; for (int i = 0; i < N; ++i) wrong += h[i] != (float)nodes;
      unsigned b[2]; CK(hipMemcpy(b, bad, sizeof(b), hipMemcpyDeviceToHost));
      printf("[hip runtime %d, %d nodes, delay %d ms] round %d graph %d: wrong values %ld, kernels with bad arguments %u\n", rt, nodes, delay_ms, round, g, wrong, b[g]);
      fails += wrong != 0 || b[g] != 0;
    }
  }
  // both execs in flight together on their two streams
  for (int r = 0; r < 4; ++r) for (int g = 0; g < 2; ++g) CK(hipGraphLaunch(ex[g], st[g]));
  CK(hipDeviceSynchronize());
  for (int g = 0; g < 2; ++g) {
    CK(hipMemcpy(h.data(), base + ((size_t)g * 2 + (nodes & 1)) * N, sizeof(float) * N, hipMemcpyDeviceToHost));
    long wrong = 0; for (int i = 0; i < N; ++i) wrong += h[i] != (float)nodes;
    printf("[hip runtime %d] concurrent graph %d: wrong values %ld\n", rt, g, wrong); fails += wrong != 0;
  }
  printf(fails ? "FAIL\n" : "PASS\n");
  return fails ? 1 : 0;
}
