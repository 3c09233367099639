// Exact division of a 32-bit unsigned by a launch-invariant divisor without a divide instruction (host part: plain C++, also compiled by
// tests/test_fastdiv_host.py with g++; device part: HIP only).  Included by nbm_common.h.
#pragma once

// Division of a 32-bit unsigned by a LAUNCH-INVARIANT divisor (image width, pixels per image, stride ...): the host prepares a
// multiplier and two shifts (Granlund-Montgomery, round-up method: exact for every 32-bit n and every d >= 1), the kernel spends five
// integer instructions instead of the ~35 of a run-time v_rcp-based division.  The tile prologues decode 4-5 GEMM rows into (image, y, x)
// each and the data-gradient kernel divided by the stride once per (row, tap): 25-55 k cycles per tile before the first MFMA (round 5,
// cycle counters in the `ablate_nn` build), as long as the whole K loop of the short-K layers.
struct nbm_fastdiv { unsigned mul, sh1, sh2; };
static inline nbm_fastdiv nbm_fastdiv_make(unsigned d) {
  nbm_fastdiv f{1u, 0u, 0u};
  if (d <= 1u) return f;                               // q = n
  unsigned l = 0;
  while ((1ull << l) < d) ++l;                         // l = ceil(log2 d), 1 <= l <= 32
  f.mul = (unsigned)((((1ull << l) - d) << 32) / d + 1ull);
  f.sh1 = 1u;
  f.sh2 = l - 1u;
  return f;
}
// the same five operations on the host (tests) and on the device (kernels)
static inline unsigned nbm_fdiv_host(unsigned n, const nbm_fastdiv f) {
  const unsigned t = (unsigned)(((unsigned long long)f.mul * n) >> 32);
  return (t + ((n - t) >> f.sh1)) >> f.sh2;
}
#if defined(__HIPCC__)
__device__ __forceinline__ unsigned nbm_fdiv(unsigned n, const nbm_fastdiv f) {
  const unsigned t = __umulhi(f.mul, n);
  return (t + ((n - t) >> f.sh1)) >> f.sh2;
}
#endif

