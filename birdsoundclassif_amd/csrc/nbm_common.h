// Shared device/host helpers for the NBM HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nbm_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define NBM_WAVE 64

static inline int nbm_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? NBM_OK : (int)e;
}

static inline bool nbm_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// Zero fill as a KERNEL launch.  The library never calls hipMemsetAsync: inside a stream capture that call becomes a MEMSET NODE, and
// the HIP runtime torch 2.10+rocm7.0 bundles (7.0.51831) loses memset nodes of a graph exec from its second launch on (AQL packet
// capture; scripts/graph_pair_repro.hip, profiles/r05_graph_pair.txt) -- the counters of the proposal stage then kept the
// previous replay's values.  A kernel node is replayed reliably.  p: 4-byte aligned, bytes % 4 == 0.
namespace {
__global__ void nbm_zero_words_kernel(uint32_t* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
}  // namespace
static inline hipError_t nbm_zero_async(void* p, size_t bytes, hipStream_t st) {
  if ((((uintptr_t)p) & 3u) || (bytes & 3u)) return hipErrorInvalidValue;
  const size_t n = bytes >> 2;
  if (!n) return hipSuccess;
  const size_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(nbm_zero_words_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, st, (uint32_t*)p, n);
  return hipGetLastError();
}

#include "nbm_fastdiv.h"

// Order-preserving map float -> uint32 (larger float => larger key); NaN sorts above +inf.
__device__ __forceinline__ uint32_t nbm_f2key(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float nbm_key2f(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

__device__ __forceinline__ float nbm_silu_f(float v) { return v / (1.0f + __expf(-v)); }

// torch.round / np.rint: round half to even.
__device__ __forceinline__ float nbm_rint(float v) { return rintf(v); }

__device__ __forceinline__ float nbm_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float nbm_wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float nbm_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Two workgroups share a CU (one wave each per SIMD).  Started together they stay in lockstep: both waves of a SIMD
// reach their load / LDS-write window at the same time and the matrix pipe idles (80 % MFMA busy measured; 92 % with
// the loads ablated).  Pairs that run in lockstep also finish together, so their successors start together again.
// Break the symmetry once per workgroup: half of the workgroups (chosen so that both the first generation, ids b and
// b + 256, and later same-XCD neighbours, ids b and b + 8, differ) start half a K-step late.
__device__ __forceinline__ void nbm_stagger_priority() {
  const unsigned b = blockIdx.x;
  if (((b >> 3) ^ (b >> 8)) & 1u) __builtin_amdgcn_s_sleep(36);      // 36 x 64 cycles ~ half a K-step of 64 MFMAs
}
