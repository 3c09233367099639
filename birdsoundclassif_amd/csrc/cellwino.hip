// Cell transforms of the BACKWARD pass of a demand-driven 3x3 / stride-1 / pad-1 convolution whose output is read through a
// 3x3 / stride-S / pad-1 pattern (the finest FPN output map and the RPN's strided depthwise convolution, reference
// fpn.py:145, layers.py:62-65,81).
//
// The gradient wrt that output is non-zero on the pattern pixels only (plus the RoI windows, which the listed-tile kernels
// handle): one 3 x 3 block per S x S cell (rows S*oy-1 .. S*oy+1).  Per cell
//   data gradient    gx[5x5 patch, rows S*oy-2 .. S*oy+2] = FULL linear convolution of the 3x3 block with the 3x3 kernel
//   weight gradient  dW[3x3] = correlation of the 5x5 input patch with the 3x3 block
// i.e. polynomial multiplication / its transpose: Toom-Cook with 5 points per axis, 25 multiplications per (cell, n, c)
// instead of 81 -- and instead of the 16 / 12 / 9 planes of EVERY 2x2 Winograd tile within a pixel of the pattern (56 % of
// the tiles) that the listed F(2x2,3x3) kernel spent on the same gradient.  Points {0, 1, -1, 2, -2}:
//   E    [5 x 3]  evaluation of a 3-coefficient polynomial      (block and kernel side)
//   Vinv [5 x 5]  inverse Vandermonde: interpolation of the 5 product values back to 5 coefficients
//   block   ->  Vg = E g E^T                   nbm_cell_outgrad     [25][cells][N]     (shared by both gradients)
//   kernel  ->  U  = E w E^T                   host (float64, once per step)
//   M_xi = Vg_xi U_xi^T  (25 grouped GEMMs, nbm_gemm_conv)   ->   gx patch = Vinv M Vinv^T      nbm_cell_dgrad_output
//   patch   ->  Vx = Vinv^T x Vinv             nbm_cell_input       [25][cells][C]
//   dU_xi = Vg_xi^T Vx_xi (25 TN GEMMs, nbm_conv_wgrad)      ->   dW = E^T dU E                 host
// fp32 error against float64: 3e-6 relative (the F(2x2,3x3) level), measured in scripts / tests.
// All three kernels are HBM-bound streams: a thread owns (cell, 4 channels), 16-byte accesses, channels fastest.
#include "nbm_common.h"

namespace {

constexpr int NP = 5;
constexpr int MAX_BIAS_N = 2048;      // channels of a bias gradient that cell_outgrad_kernel can reduce in LDS
// compile-time matrices: every use below has constant indices after unrolling, so zeros and ones fold away
__device__ constexpr float CE[5][3] = {{1.f, 0.f, 0.f}, {1.f, 1.f, 1.f}, {1.f, -1.f, 1.f}, {1.f, 2.f, 4.f}, {1.f, -2.f, 4.f}};
__device__ constexpr float CV[5][5] = {{1.f, 0.f, 0.f, 0.f, 0.f},
                                       {0.f, 2.f / 3, -2.f / 3, -1.f / 12, 1.f / 12},
                                       {-5.f / 4, 2.f / 3, 2.f / 3, -1.f / 24, -1.f / 24},
                                       {0.f, -1.f / 6, 1.f / 6, 1.f / 12, -1.f / 12},
                                       {1.f / 4, -1.f / 6, -1.f / 6, 1.f / 24, 1.f / 24}};

struct CellGeom { int B, H, W, C4, S, OH, OW; long long T; };

// 32-bit index arithmetic throughout (the launchers refuse T * C4 >= 2^31): a 64-bit div / mod chain per thread costs these
// streaming kernels more than their loads (measured on the bilinear backward: 4.7 -> 3.6 ms from this alone)
__device__ __forceinline__ void cell_of(const CellGeom& q, unsigned cell, int& b, int& oy, int& ox) {
  ox = (int)(cell % (unsigned)q.OW);
  const unsigned r = cell / (unsigned)q.OW;
  oy = (int)(r % (unsigned)q.OH);
  b = (int)(r / (unsigned)q.OH);
}

// g [B][H][W][N] -> Vg [25][T][N] = E blk E^T per cell (pixels outside the image read as zeros); bias_grad (optional) +=
// sum of the block pixels.  A thread keeps one channel quad over its grid-stride loop (the launch makes the stride a
// multiple of N/4).
__global__ __launch_bounds__(256) void cell_outgrad_kernel(const float* __restrict__ g, const CellGeom q, float* __restrict__ Vg,
                                                           float* __restrict__ bias_grad) {
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
  f32x4* v4 = reinterpret_cast<f32x4*>(Vg);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = zero;
  int my_c = -1;
  const unsigned total = (unsigned)(q.T * q.C4);
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const int c = (int)(i % (unsigned)q.C4);
    const unsigned cell = i / (unsigned)q.C4;
    my_c = c;
    int b, oy, ox;
    cell_of(q, cell, b, oy, ox);
    f32x4 d[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int y = q.S * oy - 1 + a;
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int x = q.S * ox - 1 + e;
        const bool ok = (unsigned)y < (unsigned)q.H && (unsigned)x < (unsigned)q.W;
        d[a][e] = ok ? g4[(((long long)b * q.H + y) * q.W + x) * q.C4 + c] : zero;
        bsum += d[a][e];
      }
    }
    f32x4 t[NP][3];                    // E d
#pragma unroll
    for (int a = 0; a < NP; ++a)
#pragma unroll
      for (int e = 0; e < 3; ++e) t[a][e] = CE[a][0] * d[0][e] + CE[a][1] * d[1][e] + CE[a][2] * d[2][e];
#pragma unroll
    for (int a = 0; a < NP; ++a)
#pragma unroll
      for (int e = 0; e < NP; ++e)
        v4[((long long)(a * NP + e) * q.T + cell) * q.C4 + c] = CE[e][0] * t[a][0] + CE[e][1] * t[a][1] + CE[e][2] * t[a][2];
  }
  // bias gradient: per-thread sums -> LDS (one slot per channel) -> ONE global atomic per channel and workgroup.  (Per-thread global
  // atomics -- 8 M of them on 256 addresses -- were 2/3 of this kernel's time: 3.2 ms at 1.3 TB/s for 4.2 GB.)
  if (bias_grad) {                     // uniform
    __shared__ float red[MAX_BIAS_N];
    const int n = q.C4 * 4;
    for (int k = threadIdx.x; k < n; k += 256) red[k] = 0.f;
    __syncthreads();
    if (my_c >= 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(&red[my_c * 4 + e], bsum[e]);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += 256) atomicAdd(bias_grad + k, red[k]);
  }
}

// x [B][H][W][C] -> Vx [25][T][C] = Vinv^T patch Vinv per cell (5x5 patch, rows S*oy-2 .. S*oy+2; outside the image = the
// convolution's zero padding)
// ld4 / coff4 (in 16-byte units): row pitch of Vx and the channel offset this call writes at -- two sources can fill one operand
// RAW: the patch itself, plane a * 5 + l = patch pixel (row a, column l) -- the operand of a 5x5 / stride S convolution written as 25
// taps of a [25][T][K] tensor (the RPN's strided reader composed with the output convolution in front of it: ondemand.rpn_composite)
template <bool RAW>
__global__ __launch_bounds__(256) void cell_input_kernel(const float* __restrict__ x, const CellGeom q, float* __restrict__ Vx, int ld4,
                                                         int coff4) {
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  f32x4* v4 = reinterpret_cast<f32x4*>(Vx);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const unsigned total = (unsigned)(q.T * q.C4);
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const int c = (int)(i % (unsigned)q.C4);
    const unsigned cell = i / (unsigned)q.C4;
    int b, oy, ox;
    cell_of(q, cell, b, oy, ox);
    f32x4 t[NP][NP];                   // t[a][l] = sum_j Vinv[j][a] patch[j][l], built row by row (25 live values, not 50)
#pragma unroll
    for (int a = 0; a < NP; ++a)
#pragma unroll
      for (int l = 0; l < NP; ++l) t[a][l] = zero;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int y = q.S * oy - 2 + j;
#pragma unroll
      for (int l = 0; l < NP; ++l) {
        const int xx = q.S * ox - 2 + l;
        const bool ok = (unsigned)y < (unsigned)q.H && (unsigned)xx < (unsigned)q.W;
        const f32x4 v = ok ? x4[(((long long)b * q.H + y) * q.W + xx) * q.C4 + c] : zero;
        if constexpr (RAW) {
          v4[((long long)(j * NP + l) * q.T + cell) * ld4 + coff4 + c] = v;
        } else {
#pragma unroll
          for (int a = 0; a < NP; ++a)
            if (CV[j][a] != 0.f) t[a][l] += CV[j][a] * v;
        }
      }
    }
    if constexpr (RAW) continue;
#pragma unroll
    for (int a = 0; a < NP; ++a)
#pragma unroll
      for (int e = 0; e < NP; ++e) {
        f32x4 o = zero;
#pragma unroll
        for (int l = 0; l < NP; ++l)
          if (CV[l][e] != 0.f) o += CV[l][e] * t[a][l];
        v4[((long long)(a * NP + e) * q.T + cell) * ld4 + coff4 + c] = o;
      }
  }
}

// The same transform of a patch that is not in memory: patch pixel = bilinear_align_corners(x1)[pixel] + bias (the top-down merge of
// fpn.py:143-144 without the lateral term, which enters the plane GEMMs through its own 64-channel operand), 0 outside the image.
// x1 [B][Hc][Wc][C]; same interpolation arithmetic as the merge epilogue of igemm.hip.
template <bool RAW>
__global__ __launch_bounds__(256) void cell_input_up_kernel(const float* __restrict__ x1, const float* __restrict__ bias, const CellGeom q,
                                                            int Hc, int Wc, float sh, float sw, float* __restrict__ Vx, int ld4, int coff4) {
#pragma clang fp contract(off)
  const f32x4* s4 = reinterpret_cast<const f32x4*>(x1);
  f32x4* v4 = reinterpret_cast<f32x4*>(Vx);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const unsigned total = (unsigned)(q.T * q.C4);
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const int c = (int)(i % (unsigned)q.C4);
    const unsigned cell = i / (unsigned)q.C4;
    int b, oy, ox;
    cell_of(q, cell, b, oy, ox);
    const f32x4 bv = bias ? reinterpret_cast<const f32x4*>(bias)[c] : zero;
    f32x4 t[NP][NP];
#pragma unroll
    for (int a = 0; a < NP; ++a)
#pragma unroll
      for (int l = 0; l < NP; ++l) t[a][l] = zero;
    const long long rb = (long long)b * Hc;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int y = q.S * oy - 2 + j;
      const bool yin = (unsigned)y < (unsigned)q.H;
      const float fy = sh * (yin ? y : 0);
      const int y0 = (int)fy, y1 = y0 + (y0 < Hc - 1 ? 1 : 0);
      const float ly = fminf(fmaxf(fy - y0, 0.f), 1.f), hy = 1.f - ly;
      // The 20 loads of a patch row go out together, before the first interpolation: no branch around them (a pixel outside the image
      // reads a clamped address and is zeroed afterwards) -- behind one, the four loads of every patch pixel waited for those of the
      // pixel before, 25 dependent round trips per thread
      f32x4 v00[NP], v01[NP], v10[NP], v11[NP];
      float lxs[NP];
#pragma unroll
      for (int l = 0; l < NP; ++l) {
        const float fx = sw * min(max(q.S * ox - 2 + l, 0), q.W - 1);
        const int x0 = (int)fx, x1i = x0 + (x0 < Wc - 1 ? 1 : 0);
        lxs[l] = fminf(fmaxf(fx - x0, 0.f), 1.f);
        v00[l] = s4[((rb + y0) * Wc + x0) * q.C4 + c]; v01[l] = s4[((rb + y0) * Wc + x1i) * q.C4 + c];
        v10[l] = s4[((rb + y1) * Wc + x0) * q.C4 + c]; v11[l] = s4[((rb + y1) * Wc + x1i) * q.C4 + c];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int l = 0; l < NP; ++l) {
        const bool ok = yin && (unsigned)(q.S * ox - 2 + l) < (unsigned)q.W;
        const float lx = lxs[l], hx = 1.f - lx;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[e] = ok ? (hy * (hx * v00[l][e] + lx * v01[l][e]) + ly * (hx * v10[l][e] + lx * v11[l][e])) + bv[e] : 0.f;
        if constexpr (RAW) {
          v4[((long long)(j * NP + l) * q.T + cell) * ld4 + coff4 + c] = v;
        } else {
#pragma unroll
          for (int a = 0; a < NP; ++a)
            if (CV[j][a] != 0.f) t[a][l] += CV[j][a] * v;
        }
      }
    }
    if constexpr (RAW) continue;
#pragma unroll
    for (int a = 0; a < NP; ++a)
#pragma unroll
      for (int e = 0; e < NP; ++e) {
        f32x4 o = zero;
#pragma unroll
        for (int l = 0; l < NP; ++l)
          if (CV[l][e] != 0.f) o += CV[l][e] * t[a][l];
        v4[((long long)(a * NP + e) * q.T + cell) * ld4 + coff4 + c] = o;
      }
  }
}

// M [25][T][C] -> gx [B][H][W][C]: patch = Vinv M Vinv^T written to the 5x5 pixels of the cell that lie inside the image
// (cls < 0, S >= 5: patches of different cells do not overlap); the rest of gx is not touched.
// cls = 0..3 (S >= 3): only the cells with (oy & 1, ox & 1) == (cls >> 1, cls & 1), and the patch is ADDED to gx -- cells of one
// parity class are 2 S >= 6 pixels apart, so four launches accumulate the overlapping patches of a stride-3 / 4 pattern without atomics
__global__ __launch_bounds__(256) void cell_dgrad_output_kernel(const float* __restrict__ M, const CellGeom q, float* __restrict__ gx,
                                                                int cls, int ld4, int coff4, float* __restrict__ bias_grad) {
  const f32x4* m4 = reinterpret_cast<const f32x4*>(M);
  f32x4* o4 = reinterpret_cast<f32x4*>(gx);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = zero;                   // bias_grad (optional) += sum of the patch pixels inside the image (as in cell_outgrad_kernel)
  int my_c = -1;
  // cls >= 0: the threads enumerate the cells of that parity class only (a quarter of the grid: no idle waves)
  const int cy = cls >= 0 ? cls >> 1 : 0, cx = cls >= 0 ? cls & 1 : 0;
  const int OHc = cls >= 0 ? (q.OH - cy + 1) >> 1 : q.OH, OWc = cls >= 0 ? (q.OW - cx + 1) >> 1 : q.OW;
  const unsigned total = (unsigned)((long long)q.B * OHc * OWc * q.C4);
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const int c = (int)(i % (unsigned)q.C4);
    unsigned rest = i / (unsigned)q.C4;
    int ox = (int)(rest % (unsigned)OWc); rest /= (unsigned)OWc;
    int oy = (int)(rest % (unsigned)OHc);
    const int b = (int)(rest / (unsigned)OHc);
    if (cls >= 0) { oy = 2 * oy + cy; ox = 2 * ox + cx; }
    const unsigned cell = ((unsigned)b * q.OH + oy) * q.OW + ox;
    my_c = c;
    f32x4 t[NP][NP];                   // t[j][e] = sum_a Vinv[j][a] M[a][e]
#pragma unroll
    for (int j = 0; j < NP; ++j)
#pragma unroll
      for (int e = 0; e < NP; ++e) t[j][e] = zero;
#pragma unroll
    for (int a = 0; a < NP; ++a)
#pragma unroll
      for (int e = 0; e < NP; ++e) {
        const f32x4 v = m4[((long long)(a * NP + e) * q.T + cell) * ld4 + coff4 + c];
#pragma unroll
        for (int j = 0; j < NP; ++j)
          if (CV[j][a] != 0.f) t[j][e] += CV[j][a] * v;
      }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int y = q.S * oy - 2 + j;
      if ((unsigned)y >= (unsigned)q.H) continue;
#pragma unroll
      for (int l = 0; l < NP; ++l) {
        const int xx = q.S * ox - 2 + l;
        if ((unsigned)xx >= (unsigned)q.W) continue;
        f32x4 o = zero;
#pragma unroll
        for (int e = 0; e < NP; ++e)
          if (CV[l][e] != 0.f) o += CV[l][e] * t[j][e];
        const long long at = (((long long)b * q.H + y) * q.W + xx) * q.C4 + c;
        bsum += o;
        if (cls >= 0) o += o4[at];
        o4[at] = o;
      }
    }
  }
  if (bias_grad) {                     // uniform
    __shared__ float red[MAX_BIAS_N];
    const int n = q.C4 * 4;
    for (int k = threadIdx.x; k < n; k += 256) red[k] = 0.f;
    __syncthreads();
    if (my_c >= 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(&red[my_c * 4 + e], bsum[e]);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += 256) atomicAdd(bias_grad + k, red[k]);
  }
}

// M [25][T][N] (= Vx_xi U_xi^T) -> the 3x3 pattern block of every cell in y [B][H][W][N]: blk = E^T M E + bias (FORWARD: the
// correlation form of the same 25-multiplication algorithm, F(3x3,3x3)); only the pattern pixels inside the image are written
__global__ __launch_bounds__(256) void cell_output_kernel(const float* __restrict__ M, const float* __restrict__ bias, const CellGeom q,
                                                          float* __restrict__ y) {
  const f32x4* m4 = reinterpret_cast<const f32x4*>(M);
  f32x4* o4 = reinterpret_cast<f32x4*>(y);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const unsigned total = (unsigned)(q.T * q.C4);
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const int c = (int)(i % (unsigned)q.C4);
    const unsigned cell = i / (unsigned)q.C4;
    int b, oy, ox;
    cell_of(q, cell, b, oy, ox);
    const f32x4 bv = bias ? reinterpret_cast<const f32x4*>(bias)[c] : zero;
    f32x4 t[3][NP];                    // t[i][e] = sum_a E[a][i] M[a][e]
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int e = 0; e < NP; ++e) t[r][e] = zero;
#pragma unroll
    for (int a = 0; a < NP; ++a)
#pragma unroll
      for (int e = 0; e < NP; ++e) {
        const f32x4 v = m4[((long long)(a * NP + e) * q.T + cell) * q.C4 + c];
#pragma unroll
        for (int r = 0; r < 3; ++r)
          if (CE[a][r] != 0.f) t[r][e] += CE[a][r] * v;
      }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int yy = q.S * oy - 1 + r;
      if ((unsigned)yy >= (unsigned)q.H) continue;
#pragma unroll
      for (int l = 0; l < 3; ++l) {
        const int xx = q.S * ox - 1 + l;
        if ((unsigned)xx >= (unsigned)q.W) continue;
        f32x4 o = bv;
#pragma unroll
        for (int e = 0; e < NP; ++e)
          if (CE[e][l] != 0.f) o += CE[e][l] * t[r][e];
        o4[(((long long)b * q.H + yy) * q.W + xx) * q.C4 + c] = o;
      }
    }
  }
}

// ---- kernel side of the cell transforms (was: float64 torch.einsum calls = Tensile GEMMs on the training path).  Float64
// arithmetic, ONE rounding to fp32 -- the transformed kernels carry every pattern pixel of the forward pass.
__device__ __forceinline__ void cell_e_w_et(const double g[3][3], double u[NP][NP]) {
  double t[NP][3];                     // E g
#pragma unroll
  for (int a = 0; a < NP; ++a)
#pragma unroll
    for (int s = 0; s < 3; ++s) t[a][s] = (double)CE[a][0] * g[0][s] + (double)CE[a][1] * g[1][s] + (double)CE[a][2] * g[2][s];
#pragma unroll
  for (int a = 0; a < NP; ++a)
#pragma unroll
    for (int b = 0; b < NP; ++b) u[a][b] = (double)CE[b][0] * t[a][0] + (double)CE[b][1] * t[a][1] + (double)CE[b][2] * t[a][2];
}

// w [N][C][3][3] -> U = E w E^T.  U_nc (optional): [25][N][ld] at column 0 (B operand of the forward plane GEMMs);
// U_cn (optional): [25][ldt rows >= C][N] rows 0..C-1 (B operand of the data-gradient plane GEMMs).
__global__ __launch_bounds__(256) void cell_weight_kernel(const float* __restrict__ w, int N, int C, float* __restrict__ U_nc, int ld,
                                                          float* __restrict__ U_cn, int ldt) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N * C) return;
  const int n = i / C, c = i - n * C;
  double g[3][3], u[NP][NP];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s2 = 0; s2 < 3; ++s2) g[r][s2] = (double)w[(long long)i * 9 + r * 3 + s2];
  cell_e_w_et(g, u);
#pragma unroll
  for (int a = 0; a < NP; ++a)
#pragma unroll
    for (int b = 0; b < NP; ++b) {
      const int xi = a * NP + b;
      if (U_nc) U_nc[((long long)xi * N + n) * ld + c] = (float)u[a][b];
      if (U_cn) U_cn[((long long)xi * ldt + c) * N + n] = (float)u[a][b];
    }
}

// The deferred lateral folded into the consumer's kernels: [alpha U W_lat] with U = E w E^T ([25][N][C]) and W_lat [C][Cin] (row pitch
// wl_ld) -> columns C .. C + Cin - 1 of U_nc [25][N][ld] and rows C .. C + Cin - 1 of U_cn [25][ldt][N].  Thread = (n, k):
// P[r][s] = sum_c w[n][c][r][s] W_lat[c][k] in float64 (the transform is linear, so E (sum_c ...) E^T = sum_c (E w E^T) W), then E P E^T.
__global__ __launch_bounds__(256) void cell_weight_fold_kernel(const float* __restrict__ w, const float* __restrict__ wl, int wl_ld, int N,
                                                               int C, int Cin, float alpha, float* __restrict__ U_nc, int ld,
                                                               float* __restrict__ U_cn, int ldt) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N * Cin) return;
  const int n = i / Cin, k = i - n * Cin;
  double g[3][3], u[NP][NP];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s2 = 0; s2 < 3; ++s2) g[r][s2] = 0.0;
  const float* wn = w + (long long)n * C * 9;
  for (int c = 0; c < C; ++c) {
    const double l = (double)wl[(long long)c * wl_ld + k];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s2 = 0; s2 < 3; ++s2) g[r][s2] += (double)wn[c * 9 + r * 3 + s2] * l;
  }
  cell_e_w_et(g, u);
#pragma unroll
  for (int a = 0; a < NP; ++a)
#pragma unroll
    for (int b = 0; b < NP; ++b) {
      const int xi = a * NP + b;
      const float v = (float)((double)alpha * u[a][b]);
      if (U_nc) U_nc[((long long)xi * N + n) * ld + C + k] = v;
      if (U_cn) U_cn[((long long)xi * ldt + C + k) * N + n] = v;
    }
}

// dU [25][N][ld] (columns 0..C-1) -> dW [N][C][3][3] = E^T dU E (float64, one rounding): the gradient wrt the checkpoint-layout kernel
__global__ __launch_bounds__(256) void cell_weight_grad_kernel(const float* __restrict__ dU, int N, int C, int ld, float* __restrict__ dW) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N * C) return;
  const int n = i / C, c = i - n * C;
  double t[3][NP];                     // t[r][b] = sum_a E[a][r] dU[a][b]
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int b = 0; b < NP; ++b) t[r][b] = 0.0;
#pragma unroll
  for (int a = 0; a < NP; ++a)
#pragma unroll
    for (int b = 0; b < NP; ++b) {
      const double v = (double)dU[((long long)(a * NP + b) * N + n) * ld + c];
#pragma unroll
      for (int r = 0; r < 3; ++r)
        if (CE[a][r] != 0.f) t[r][b] += (double)CE[a][r] * v;
    }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s2 = 0; s2 < 3; ++s2) {
      double o = 0.0;
#pragma unroll
      for (int b = 0; b < NP; ++b)
        if (CE[b][s2] != 0.f) o += (double)CE[b][s2] * t[r][b];
      dW[(long long)i * 9 + r * 3 + s2] = (float)o;
    }
}

// S >= 3: the 3x3 blocks of different cells are disjoint; the 5x5 patches may overlap (read-only) unless they are written
inline bool cell_geom(int B, int H, int W, int C, int S, CellGeom& q, int min_S = 3) {
  if (B <= 0 || H < 3 || W < 3 || C <= 0 || (C & 3) || S < min_S) return false;
  q.B = B; q.H = H; q.W = W; q.C4 = C / 4; q.S = S;
  q.OH = (H + 2 - 3) / S + 1; q.OW = (W + 2 - 3) / S + 1;          // output size of the 3x3 / stride S / pad 1 reader
  q.T = (long long)B * q.OH * q.OW;
  if (q.T * q.C4 >= 0x7fffffffll) return false;          // 32-bit thread indices (callers cut the batch into chunks far below this)
  return true;
}

inline unsigned stream_grid(long long total, int per) {
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  while ((blocks * 256) % per) ++blocks;      // a thread keeps one channel quad over its grid-stride loop
  return (unsigned)blocks;
}

}  // namespace

extern "C" int nbm_cell_outgrad(const float* g, int B, int H, int W, int N, int stride, float* Vg, float* bias_grad, void* stream) {
  CellGeom q;
  if (!g || !Vg || !cell_geom(B, H, W, N, stride, q)) return NBM_EINVAL;
  if (!nbm_aligned16(g) || !nbm_aligned16(Vg)) return NBM_EALIGN;
  if (bias_grad && N > MAX_BIAS_N) return NBM_EUNSUPPORTED;
  hipLaunchKernelGGL(cell_outgrad_kernel, dim3(stream_grid(q.T * q.C4, q.C4)), dim3(256), 0, (hipStream_t)stream, g, q, Vg, bias_grad);
  return nbm_launch_status();
}

extern "C" int nbm_cell_input(const float* x, int B, int H, int W, int C, int stride, float* Vx, int ld, int c_off, void* stream) {
  CellGeom q;
  if (!x || !Vx || !cell_geom(B, H, W, C, stride, q) || ld < c_off + C || c_off < 0) return NBM_EINVAL;
  if (!nbm_aligned16(x) || !nbm_aligned16(Vx) || (ld & 3) || (c_off & 3)) return NBM_EALIGN;
  hipLaunchKernelGGL(cell_input_kernel<false>, dim3(stream_grid(q.T * q.C4, q.C4)), dim3(256), 0, (hipStream_t)stream, x, q, Vx, ld / 4,
                     c_off / 4);
  return nbm_launch_status();
}

extern "C" int nbm_cell_input_up(const float* x1, const float* bias, int B, int H, int W, int C, int Hc, int Wc, int stride, float* Vx,
                                 int ld, int c_off, void* stream) {
  CellGeom q;
  if (!x1 || !Vx || !cell_geom(B, H, W, C, stride, q) || Hc <= 0 || Wc <= 0 || ld < c_off + C || c_off < 0) return NBM_EINVAL;
  if (!nbm_aligned16(x1) || !nbm_aligned16(Vx) || (bias && !nbm_aligned16(bias)) || (ld & 3) || (c_off & 3)) return NBM_EALIGN;
  const float sh = H > 1 ? (float)(Hc - 1) / (float)(H - 1) : 0.f, sw = W > 1 ? (float)(Wc - 1) / (float)(W - 1) : 0.f;
  hipLaunchKernelGGL(cell_input_up_kernel<false>, dim3(stream_grid(q.T * q.C4, q.C4)), dim3(256), 0, (hipStream_t)stream, x1, bias, q, Hc, Wc, sh,
                     sw, Vx, ld / 4, c_off / 4);
  return nbm_launch_status();
}

extern "C" int nbm_cell_patches(const float* x, int B, int H, int W, int C, int stride, float* Vx, int ld, int c_off, void* stream) {
  CellGeom q;
  if (!x || !Vx || !cell_geom(B, H, W, C, stride, q) || ld < c_off + C || c_off < 0) return NBM_EINVAL;
  if (!nbm_aligned16(x) || !nbm_aligned16(Vx) || (ld & 3) || (c_off & 3)) return NBM_EALIGN;
  hipLaunchKernelGGL(cell_input_kernel<true>, dim3(stream_grid(q.T * q.C4, q.C4)), dim3(256), 0, (hipStream_t)stream, x, q, Vx, ld / 4,
                     c_off / 4);
  return nbm_launch_status();
}

extern "C" int nbm_cell_patches_up(const float* x1, const float* bias, int B, int H, int W, int C, int Hc, int Wc, int stride, float* Vx,
                                   int ld, int c_off, void* stream) {
  CellGeom q;
  if (!x1 || !Vx || !cell_geom(B, H, W, C, stride, q) || Hc <= 0 || Wc <= 0 || ld < c_off + C || c_off < 0) return NBM_EINVAL;
  if (!nbm_aligned16(x1) || !nbm_aligned16(Vx) || (bias && !nbm_aligned16(bias)) || (ld & 3) || (c_off & 3)) return NBM_EALIGN;
  const float sh = H > 1 ? (float)(Hc - 1) / (float)(H - 1) : 0.f, sw = W > 1 ? (float)(Wc - 1) / (float)(W - 1) : 0.f;
  hipLaunchKernelGGL(cell_input_up_kernel<true>, dim3(stream_grid(q.T * q.C4, q.C4)), dim3(256), 0, (hipStream_t)stream, x1, bias, q, Hc, Wc,
                     sh, sw, Vx, ld / 4, c_off / 4);
  return nbm_launch_status();
}

extern "C" int nbm_cell_output(const float* M, const float* bias, int B, int H, int W, int N, int stride, float* y, void* stream) {
  CellGeom q;
  if (!M || !y || !cell_geom(B, H, W, N, stride, q)) return NBM_EINVAL;
  if (!nbm_aligned16(M) || !nbm_aligned16(y) || (bias && !nbm_aligned16(bias))) return NBM_EALIGN;
  hipLaunchKernelGGL(cell_output_kernel, dim3(stream_grid(q.T * q.C4, q.C4)), dim3(256), 0, (hipStream_t)stream, M, bias, q, y);
  return nbm_launch_status();
}

extern "C" int nbm_cell_dgrad_output(const float* M, int B, int H, int W, int C, int stride, float* gx, int parity_class, int ld,
                                     int c_off, float* bias_grad, void* stream) {
  CellGeom q;
  // parity_class < 0: 5x5 patches are WRITTEN, no overlap allowed (stride >= 5); 0..3: one parity class of cells, ADDED (stride >= 3)
  if (!M || !gx || parity_class > 3 || !cell_geom(B, H, W, C, stride, q, parity_class < 0 ? 5 : 3)) return NBM_EINVAL;
  if (ld < c_off + C || c_off < 0 || (ld & 3) || (c_off & 3) || (bias_grad && C > MAX_BIAS_N)) return NBM_EINVAL;
  if (!nbm_aligned16(M) || !nbm_aligned16(gx)) return NBM_EALIGN;
  hipLaunchKernelGGL(cell_dgrad_output_kernel, dim3(stream_grid(q.T * q.C4, q.C4)), dim3(256), 0, (hipStream_t)stream, M, q, gx,
                     parity_class, ld / 4, c_off / 4, bias_grad);
  return nbm_launch_status();
}

extern "C" int nbm_cell_weight(const float* w, int N, int C, float* U_nc, int ld, float* U_cn, int ldt, void* stream) {
  if (!w || (!U_nc && !U_cn) || N <= 0 || C <= 0 || (U_nc && ld < C) || (U_cn && ldt < C)) return NBM_EINVAL;
  hipLaunchKernelGGL(cell_weight_kernel, dim3((unsigned)((N * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, N, C, U_nc, ld, U_cn,
                     ldt);
  return nbm_launch_status();
}

extern "C" int nbm_cell_weight_fold(const float* w, const float* w_lat, int wl_ld, int N, int C, int Cin, float alpha, float* U_nc, int ld,
                                    float* U_cn, int ldt, void* stream) {
  if (!w || !w_lat || (!U_nc && !U_cn) || N <= 0 || C <= 0 || Cin <= 0 || wl_ld < Cin || (U_nc && ld < C + Cin) || (U_cn && ldt < C + Cin))
    return NBM_EINVAL;
  hipLaunchKernelGGL(cell_weight_fold_kernel, dim3((unsigned)((N * Cin + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, w_lat, wl_ld, N,
                     C, Cin, alpha, U_nc, ld, U_cn, ldt);
  return nbm_launch_status();
}

extern "C" int nbm_cell_weight_grad(const float* dU, int N, int C, int ld, float* dW, void* stream) {
  if (!dU || !dW || N <= 0 || C <= 0 || ld < C) return NBM_EINVAL;
  hipLaunchKernelGGL(cell_weight_grad_kernel, dim3((unsigned)((N * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dU, N, C, ld, dW);
  return nbm_launch_status();
}

// ---- rectangle of a [n_outer][.][.] row-major array <-> packed rows (the border-cell classes of the composed RPN reader, DESIGN 4h:
// a rectangle [r0, r1) x [c0, c1) of the OH x OW cell grid of every (plane, image), rows of (c1 - c0) * K floats).  16-byte accesses.
namespace {
__global__ __launch_bounds__(256) void copy_rect_kernel(float4* __restrict__ strided, float4* __restrict__ packed, long long n_outer,
                                                         long long outer_pitch4, int n_rows, long long row_pitch4, long long width4,
                                                         int to_strided, int zero_strided) {
  const long long total = n_outer * n_rows * width4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long w = i % width4, rr = i / width4;
    const long long r = rr % n_rows, o = rr / n_rows;
    float4* sp = strided + o * outer_pitch4 + r * row_pitch4 + w;
    if (to_strided) *sp = packed[i];
    else {
      packed[i] = *sp;
      if (zero_strided) *sp = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}
}  // namespace

extern "C" int nbm_copy_rect(float* strided, float* packed, int64_t n_outer, int64_t outer_pitch, int n_rows, int64_t row_pitch, int64_t width,
                             int to_strided, int zero_strided, void* stream) {
  if (!strided || !packed || n_outer <= 0 || n_rows <= 0 || width <= 0 || row_pitch < width || outer_pitch < 0) return NBM_EINVAL;
  if (!nbm_aligned16(strided) || !nbm_aligned16(packed) || (outer_pitch & 3) || (row_pitch & 3) || (width & 3)) return NBM_EALIGN;
  const long long total = (long long)n_outer * n_rows * (width / 4);
  const long long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(copy_rect_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, (float4*)strided,
                     (float4*)packed, (long long)n_outer, (long long)(outer_pitch / 4), n_rows, (long long)(row_pitch / 4),
                     (long long)(width / 4), to_strided, zero_strided);
  return nbm_launch_status();
}
