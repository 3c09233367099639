/* Stand-alone use of the C ABI (no Python, no torch): a 3x3 / pad 1 convolution with bias + ReLU on NHWC fp32 buffers
 * through nbm_gemm_conv, checked against a scalar CPU loop.  Build and run (on a machine with an MI355X):
 *
 *   gcc -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/cabi_gemm_conv.c -Lbirdsoundclassif_amd -lnbm_hip \
 *       -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/birdsoundclassif_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/cabi_gemm_conv
 *   /tmp/cabi_gemm_conv
 *
 * This is what a binding in any language does: allocate device buffers, fill the POD descriptor, pass a stream. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nbm_hip.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(void) {
  const int B = 2, H = 12, W = 20, Cin = 64, N = 96, K = 9 * Cin;
  const size_t nx = (size_t)B * H * W * Cin, nw = (size_t)N * K, ny = (size_t)B * H * W * N;
  float *x = (float*)malloc(nx * 4), *w = (float*)malloc(nw * 4), *bias = (float*)malloc(N * 4), *y = (float*)malloc(ny * 4);
  unsigned s = 12345u;
  for (size_t i = 0; i < nx; ++i) { s = s * 1664525u + 1013904223u; x[i] = (float)(s >> 8) / 16777216.0f - 0.5f; }
  for (size_t i = 0; i < nw; ++i) { s = s * 1664525u + 1013904223u; w[i] = ((float)(s >> 8) / 16777216.0f - 0.5f) * 0.1f; }
  for (int i = 0; i < N; ++i) bias[i] = 0.01f * (float)(i % 7);
  float *dx, *dw, *db, *dy;
  CHECK(hipMalloc((void**)&dx, nx * 4)); CHECK(hipMalloc((void**)&dw, nw * 4));
  CHECK(hipMalloc((void**)&db, N * 4)); CHECK(hipMalloc((void**)&dy, ny * 4));
  CHECK(hipMemcpy(dx, x, nx * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dw, w, nw * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(db, bias, N * 4, hipMemcpyHostToDevice));
  hipStream_t st;
  CHECK(hipStreamCreate(&st));

  nbm_gemm_desc d;
  memset(&d, 0, sizeof d);
  d.x = dx; d.w = dw; d.y = dy; d.shift = db;            /* KRSC weights: w[n][(r*3+s)*Cin + c] */
  d.groups = 1; d.B = B; d.H = H; d.W = W; d.Cin = Cin; d.N = N;
  d.kh = 3; d.kw = 3; d.stride = 1; d.pad = 1; d.Ho = H; d.Wo = W;
  d.x_ld = Cin; d.w_ld = K; d.y_ld = N; d.alpha = 1.0f; d.act = NBM_ACT_RELU;
  const int rc = nbm_gemm_conv(&d, st);
  if (rc) { fprintf(stderr, "nbm_gemm_conv rc=%d\n", rc); return 3; }
  CHECK(hipStreamSynchronize(st));
  CHECK(hipMemcpy(y, dy, ny * 4, hipMemcpyDeviceToHost));

  double max_err = 0.0;
  for (int b = 0; b < B; ++b)
    for (int oy = 0; oy < H; ++oy)
      for (int ox = 0; ox < W; ++ox)
        for (int n = 0; n < N; n += 5) {
          double acc = bias[n];
          for (int r = 0; r < 3; ++r)
            for (int t = 0; t < 3; ++t) {
              const int iy = oy - 1 + r, ix = ox - 1 + t;
              if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
              for (int c = 0; c < Cin; ++c)
                acc += (double)x[(((size_t)b * H + iy) * W + ix) * Cin + c] * (double)w[(size_t)n * K + (r * 3 + t) * Cin + c];
            }
          if (acc < 0) acc = 0;
          const double e = fabs(acc - (double)y[(((size_t)b * H + oy) * W + ox) * N + n]);
          if (e > max_err) max_err = e;
        }
  printf("%s: 3x3 conv + bias + ReLU, max |err| = %.3g\n", nbm_version(), max_err);
  return max_err < 1e-5 ? 0 : 1;
}
