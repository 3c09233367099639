// Winograd transforms for the large 3x3 / stride-1 / pad-1 convolutions of the FPN (reference fpn.py:137,145
// and their data gradients): Y = A^T [ (G g G^T) . (B^T d B) ] A turns one 3x3 convolution into 16 independent
// [tiles x Cin] x [Cin x Cout] GEMMs (run by igemm_kernel with groups = 16) with 2.25x fewer multiplies.  The two
// kernels here are the HBM-bound input (B^T d B) and output (A^T m A) transforms; the weight transform (G g G^T) is done
// once per parameter version on the host.  All arithmetic fp32; fp32 error ~4x that of the direct kernel (3e-6 on O(1)
// outputs), far inside the 1e-4 parity tolerance.
#include "nbm_common.h"

namespace {

constexpr int WINO_MAX_BIAS_N = 2048;   // channels of a bias gradient that the output-gradient transforms reduce in LDS first


// x [B][H][W][C] -> V [16][T][C], T = B * ceil(H/2) * ceil(W/2); tile (ty, tx) reads rows 2ty-1 .. 2ty+2, cols 2tx-1 .. 2tx+2.
// With a tile list (tiles != NULL: n_list linear tile ids b * TH * TW + ty * TW + tx, or -1) V is compact: [16][n_list][C],
// row t holds the listed tile t (zeros for -1) -- the weight gradient of a demand-driven map only sums over the tiles
// whose output was ever read.
__global__ __launch_bounds__(256) void wino23_input_kernel(const float* __restrict__ x, int B, int H, int W, int C4,
                                                           float* __restrict__ V, const int* __restrict__ tiles,
                                                           int n_list, const unsigned* __restrict__ blk_info) {
  const int TH = (H + 1) >> 1, TW = (W + 1) >> 1;
  const long long T = tiles ? (long long)n_list : (long long)B * TH * TW;
  const long long total = T * C4;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  f32x4* v4 = reinterpret_cast<f32x4*>(V);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) {      // 32-bit: see idx32_ok
    const int c = (int)(i % (unsigned)C4);
    const unsigned t = i / (unsigned)C4;
    const long long id = tiles ? (long long)tiles[t] : t;
    if (id < 0) {
#pragma unroll
      for (int a = 0; a < 16; ++a) v4[((long long)a * T + t) * C4 + c] = zero;
      continue;
    }
    const unsigned uid = (unsigned)id;
    const int tx = (int)(uid % (unsigned)TW);
    const unsigned r = uid / (unsigned)TW;
    const int ty = (int)(r % (unsigned)TH);
    const int b = (int)(r / (unsigned)TH);
    f32x4 d[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int iy = 2 * ty - 1 + a;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ix = 2 * tx - 1 + q;
        const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        d[a][q] = ok ? x4[(((long long)b * H + iy) * W + ix) * C4 + c] : zero;
      }
    }
    // rows: B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
    f32x4 u[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      u[0][q] = d[0][q] - d[2][q];
      u[1][q] = d[1][q] + d[2][q];
      u[2][q] = d[2][q] - d[1][q];
      u[3][q] = d[1][q] - d[3][q];
    }
    // plane_mask (list mode, optional, one word per entry): planes that are not this entry's to contribute are written as
    // zeros -- the forward pass skipped them for a tile entered through one row / column / pixel (the input pixels only they
    // depend on may never have been computed: 0 x garbage must not reach dU), or another entry of the same tile carries them
    const unsigned pm = blk_info ? blk_info[t] & 0xffffu : 0xffffu;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const f32x4 o0 = u[a][0] - u[a][2], o1 = u[a][1] + u[a][2], o2 = u[a][2] - u[a][1], o3 = u[a][1] - u[a][3];
      v4[((long long)(a * 4 + 0) * T + t) * C4 + c] = ((pm >> (a * 4 + 0)) & 1u) ? o0 : zero;
      v4[((long long)(a * 4 + 1) * T + t) * C4 + c] = ((pm >> (a * 4 + 1)) & 1u) ? o1 : zero;
      v4[((long long)(a * 4 + 2) * T + t) * C4 + c] = ((pm >> (a * 4 + 2)) & 1u) ? o2 : zero;
      v4[((long long)(a * 4 + 3) * T + t) * C4 + c] = ((pm >> (a * 4 + 3)) & 1u) ? o3 : zero;
    }
  }
}

// M [16][T][N] (+ bias[N]) -> y [B][H][W][N]; A^T = [1 1 1 0; 0 1 -1 -1]
// epilogue shared by both output transforms: v * scale + shift, ReLU, producer mask (all optional)
template <typename V>
__device__ __forceinline__ V wino_epilogue(V v, V sc, V sh, bool relu, const V* mask, long long idx) {
  v = v * sc + sh;
  constexpr int n = sizeof(V) / sizeof(float);
  if (relu) {
#pragma unroll
    for (int e = 0; e < n; ++e) v[e] = fmaxf(v[e], 0.f);
  }
  if (mask) {
    const V q = mask[idx];
#pragma unroll
    for (int e = 0; e < n; ++e) if (!(q[e] > 0.f)) v[e] = 0.f;
  }
  return v;
}

__global__ __launch_bounds__(256) void wino23_output_kernel(const float* __restrict__ M, const float* __restrict__ scale,
                                                            const float* __restrict__ bias, const float* __restrict__ mask,
                                                            int relu, int B, int H, int W, int N4, float* __restrict__ y) {
  const int TH = (H + 1) >> 1, TW = (W + 1) >> 1;
  const long long T = (long long)B * TH * TW;
  const long long total = T * N4;
  const f32x4* m4 = reinterpret_cast<const f32x4*>(M);
  f32x4* y4 = reinterpret_cast<f32x4*>(y);
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) {      // 32-bit: see idx32_ok
    const int c = (int)(i % (unsigned)N4);
    const unsigned t = i / (unsigned)N4;
    const int tx = (int)(t % (unsigned)TW);
    const unsigned r = t / (unsigned)TW;
    const int ty = (int)(r % (unsigned)TH);
    const int b = (int)(r / (unsigned)TH);
    f32x4 s[2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 m0 = m4[((long long)(0 + q) * T + t) * N4 + c], m1 = m4[((long long)(4 + q) * T + t) * N4 + c];
      const f32x4 m2 = m4[((long long)(8 + q) * T + t) * N4 + c], m3 = m4[((long long)(12 + q) * T + t) * N4 + c];
      s[0][q] = m0 + m1 + m2;
      s[1][q] = m1 - m2 - m3;
    }
    f32x4 bv = {0.f, 0.f, 0.f, 0.f}, sv = {1.f, 1.f, 1.f, 1.f};
    if (bias) bv = reinterpret_cast<const f32x4*>(bias)[c];
    if (scale) sv = reinterpret_cast<const f32x4*>(scale)[c];
    const f32x4* mk = reinterpret_cast<const f32x4*>(mask);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const f32x4 o0 = s[p][0] + s[p][1] + s[p][2];
      const f32x4 o1 = s[p][1] - s[p][2] - s[p][3];
      if (2 * ty + p >= H) break;                                   // odd H: the last tile row is half outside
      const long long row = ((long long)b * H + 2 * ty + p) * W + 2 * tx;
      y4[row * N4 + c] = wino_epilogue(o0, sv, bv, relu, mk, row * N4 + c);
      if (2 * tx + 1 < W) y4[(row + 1) * N4 + c] = wino_epilogue(o1, sv, bv, relu, mk, (row + 1) * N4 + c);
    }
  }
}

// g [B][H][W][N] (gradient wrt the convolution output) -> dM [16][T][N] = (A g A^T) per 2x2 tile: the operand of the 16
// weight-gradient GEMMs dU[xi] = dM[xi]^T V[xi].  Every thread also accumulates the plain sum of its pixels per channel
// (bias gradient) over its grid-stride items -- its channel chunk is fixed because the stride is a multiple of N4.
__global__ __launch_bounds__(256) void wino23_outgrad_kernel(const float* __restrict__ g, int B, int H, int W, int N4,
                                                             float* __restrict__ dM, float* __restrict__ bias_grad,
                                                             const int* __restrict__ tiles, int n_list,
                                                             const unsigned* __restrict__ entry_info, int pat_stride) {
  const int TH = (H + 1) >> 1, TW = (W + 1) >> 1;
  const long long T = tiles ? (long long)n_list : (long long)B * TH * TW;      // tile list: as in wino23_input_kernel
  // pat_stride S > 0: the pixels of the 3x3 / stride-S / pad-1 pattern (rows S*o-1 .. S*o+1 of the cells that exist) read as
  // zeros -- their share of the gradient is taken by the cell transforms (cellwino.hip), every pixel of g counts once
  const int S = pat_stride, OHp = S > 0 ? (H + 2 - 3) / S + 1 : 0, OWp = S > 0 ? (W + 2 - 3) / S + 1 : 0;
  auto in_pattern = [&](int y, int x) -> bool {
    if (S <= 0) return false;
    return (y + 1) % S < 3 && (y + 1) / S < OHp && (x + 1) % S < 3 && (x + 1) / S < OWp;
  };
  const long long total = T * N4;
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
  f32x4* m4 = reinterpret_cast<f32x4*>(dM);
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  int my_c = -1;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) {      // 32-bit: see idx32_ok
    const int c = (int)(i % (unsigned)N4);
    my_c = c;
    const unsigned t = i / (unsigned)N4;
    const long long id = tiles ? (long long)tiles[t] : t;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    if (id < 0) {
#pragma unroll
      for (int a = 0; a < 16; ++a) m4[((long long)a * T + t) * N4 + c] = zero;
      continue;
    }
    const unsigned uid = (unsigned)id;
    const int tx = (int)(uid % (unsigned)TW);
    const unsigned r = uid / (unsigned)TW;
    const int ty = (int)(r % (unsigned)TH);
    const int b = (int)(r / (unsigned)TH);
    const long long row = ((long long)b * H + 2 * ty) * W + 2 * tx;
    const bool in_y = 2 * ty + 1 < H, in_x = 2 * tx + 1 < W;          // odd sizes: the last tile row / column is half outside
    const f32x4 y00 = in_pattern(2 * ty, 2 * tx) ? zero : g4[row * N4 + c];
    const f32x4 y01 = (in_x && !in_pattern(2 * ty, 2 * tx + 1)) ? g4[(row + 1) * N4 + c] : zero;
    const f32x4 y10 = (in_y && !in_pattern(2 * ty + 1, 2 * tx)) ? g4[(row + W) * N4 + c] : zero;
    const f32x4 y11 = (in_y && in_x && !in_pattern(2 * ty + 1, 2 * tx + 1)) ? g4[(row + W + 1) * N4 + c] : zero;
    if (!entry_info || ((entry_info[t] >> 16) & 1u)) bsum += (y00 + y01) + (y10 + y11);   // a tile listed twice counts once
    // rows of A = [1 0; 1 1; 1 -1; 0 -1]
    const f32x4 r0[2] = {y00, y01}, r1[2] = {y00 + y10, y01 + y11}, r2[2] = {y00 - y10, y01 - y11}, r3[2] = {-y10, -y11};
    const f32x4* rows[4] = {r0, r1, r2, r3};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const f32x4 p = rows[a][0], q = rows[a][1];
      m4[((long long)(a * 4 + 0) * T + t) * N4 + c] = p;
      m4[((long long)(a * 4 + 1) * T + t) * N4 + c] = p + q;
      m4[((long long)(a * 4 + 2) * T + t) * N4 + c] = p - q;
      m4[((long long)(a * 4 + 3) * T + t) * N4 + c] = -q;
    }
  }
  if (bias_grad) {                     // uniform: per-thread sums -> LDS -> one global atomic per channel and workgroup
    __shared__ float red[WINO_MAX_BIAS_N];
    const int n = N4 * 4;
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

// ---------------------------------------------------------------------------------------------------------------
// Winograd F(4x4, 3x3) for the BACKWARD convolutions only (data and weight gradients): 4x fewer multiplies than the
// direct kernel and 2.25x (instead of 4x) operand expansion.  Interpolation points {0, 1, -1, 1/2, -2, inf}: with them
// the fp32 error is 1.7e-5 on O(4) outputs (standard points: 5e-5) -- fine for gradients (parity tolerance 2e-3
// relative), not for the forward pass, whose outputs are rounded to pixel coordinates downstream.
// Tiles: 4x4 outputs from 6x6 inputs; T = B * ceil(H/4) * ceil(W/4); 36 transformed planes.  Two channels per thread.
typedef float f32x2 __attribute__((ext_vector_type(2)));

__constant__ float W43_BT[6][6] = {{1.f, -1.5f, -2.f, 1.5f, 1.f, 0.f},
                                   {0.f, -1.f / 3, 1.f / 6, 5.f / 6, 1.f / 3, 0.f},
                                   {0.f, -1.f / 3, 5.f / 6, -1.f / 6, -1.f / 3, 0.f},
                                   {0.f, 32.f / 15, 16.f / 15, -32.f / 15, -16.f / 15, 0.f},
                                   {0.f, 1.f / 30, -1.f / 15, -1.f / 30, 1.f / 15, 0.f},
                                   {0.f, 1.f, -1.5f, -2.f, 1.5f, 1.f}};
__constant__ float W43_AT[4][6] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f},
                                   {0.f, 1.f, -1.f, 0.5f, -2.f, 0.f},
                                   {0.f, 1.f, 1.f, 0.25f, 4.f, 0.f},
                                   {0.f, 1.f, -1.f, 0.125f, -8.f, 1.f}};

// x [B][H][W][C] -> V [36][T][C] = B^T d B of the 6x6 patch at rows 4ty-1 .. 4ty+4, cols 4tx-1 .. 4tx+4 (zero padded)
__global__ __launch_bounds__(256) void wino43_input_kernel(const float* __restrict__ x, int B, int H, int W, int C2,
                                                           float* __restrict__ V) {
  const int TH = (H + 3) >> 2, TW = (W + 3) >> 2;
  const long long T = (long long)B * TH * TW;
  const long long total = T * C2;
  const f32x2* x2 = reinterpret_cast<const f32x2*>(x);
  f32x2* v2 = reinterpret_cast<f32x2*>(V);
  const f32x2 zero = {0.f, 0.f};
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) {      // 32-bit: see idx32_ok
    const int c = (int)(i % (unsigned)C2);
    const unsigned t = i / (unsigned)C2;
    const int tx = (int)(t % (unsigned)TW);
    const unsigned r = t / (unsigned)TW;
    const int ty = (int)(r % (unsigned)TH);
    const int b = (int)(r / (unsigned)TH);
    f32x2 u[6][6];                                   // u = B^T d, built column by column
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int ix = 4 * tx - 1 + q;
      f32x2 d[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        const int iy = 4 * ty - 1 + a;
        const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        d[a] = ok ? x2[(((long long)b * H + iy) * W + ix) * C2 + c] : zero;
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        f32x2 acc = zero;
#pragma unroll
        for (int a = 0; a < 6; ++a) acc += d[a] * W43_BT[k][a];
        u[k][q] = acc;
      }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        f32x2 acc = zero;
#pragma unroll
        for (int q = 0; q < 6; ++q) acc += u[k][q] * W43_BT[j][q];
        v2[((long long)(k * 6 + j) * T + t) * C2 + c] = acc;
      }
  }
}

// M [36][T][N] (+ bias) -> y [B][H][W][N] = A^T m A per 4x4 tile (partial tiles at the bottom / right edge)
__global__ __launch_bounds__(256) void wino43_output_kernel(const float* __restrict__ M, const float* __restrict__ scale,
                                                            const float* __restrict__ bias, const float* __restrict__ mask,
                                                            int relu, int B, int H, int W, int N2, float* __restrict__ y,
                                                            const float* __restrict__ residual) {
  const int TH = (H + 3) >> 2, TW = (W + 3) >> 2;
  const long long T = (long long)B * TH * TW;
  const long long total = T * N2;
  const f32x2* m2 = reinterpret_cast<const f32x2*>(M);
  f32x2* y2 = reinterpret_cast<f32x2*>(y);
  const f32x2 zero = {0.f, 0.f};
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) {      // 32-bit: see idx32_ok
    const int c = (int)(i % (unsigned)N2);
    const unsigned t = i / (unsigned)N2;
    const int tx = (int)(t % (unsigned)TW);
    const unsigned r = t / (unsigned)TW;
    const int ty = (int)(r % (unsigned)TH);
    const int b = (int)(r / (unsigned)TH);
    f32x2 s[4][6];                                   // s = A^T m, built column by column
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      f32x2 m[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) m[a] = m2[((long long)(a * 6 + q) * T + t) * N2 + c];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        f32x2 acc = zero;
#pragma unroll
        for (int a = 0; a < 6; ++a) acc += m[a] * W43_AT[p][a];
        s[p][q] = acc;
      }
    }
    f32x2 bv = zero, sv = {1.f, 1.f};
    if (bias) bv = reinterpret_cast<const f32x2*>(bias)[c];
    if (scale) sv = reinterpret_cast<const f32x2*>(scale)[c];
    const f32x2* mk = reinterpret_cast<const f32x2*>(mask);
    const f32x2* rs = reinterpret_cast<const f32x2*>(residual);
    // the mask / residual values of all 16 pixels are requested up front (clamped addresses): loaded inside the loop below, behind its
    // bounds tests, they were up to 32 dependent round trips per thread
    f32x2 mq[4][4], rq[4][4];
    if (mk || rs) {                                              // uniform
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const long long idc = (((long long)b * H + min(4 * ty + p, H - 1)) * W + min(4 * tx + o, W - 1)) * N2 + c;
          if (mk) mq[p][o] = mk[idc];
          if (rs) rq[p][o] = rs[idc];
        }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int oy = 4 * ty + p;
      if (oy >= H) break;
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int ox = 4 * tx + o;
        if (ox >= W) break;
        f32x2 acc = zero;
#pragma unroll
        for (int q = 0; q < 6; ++q) acc += s[p][q] * W43_AT[o][q];
        const long long idx = (((long long)b * H + oy) * W + ox) * N2 + c;
        f32x2 v = wino_epilogue(acc, sv, bv, relu, (const f32x2*)nullptr, idx);
        if (mk) {
          if (!(mq[p][o][0] > 0.f)) v[0] = 0.f;
          if (!(mq[p][o][1] > 0.f)) v[1] = 0.f;
        }
        if (rs) v += rq[p][o];                                   // gradient of another consumer of the same tensor
        y2[idx] = v;
      }
    }
  }
}

// g [B][H][W][N] -> dM [36][T][N] = A g A^T per 4x4 tile (A = (A^T)^T, 6x4); bias gradient as in the F(2x2) kernel
__global__ __launch_bounds__(256) void wino43_outgrad_kernel(const float* __restrict__ g, int B, int H, int W, int N2,
                                                             float* __restrict__ dM, float* __restrict__ bias_grad) {
  const int TH = (H + 3) >> 2, TW = (W + 3) >> 2;
  const long long T = (long long)B * TH * TW;
  const long long total = T * N2;
  const f32x2* g2 = reinterpret_cast<const f32x2*>(g);
  f32x2* m2 = reinterpret_cast<f32x2*>(dM);
  const f32x2 zero = {0.f, 0.f};
  f32x2 bsum = zero;
  int my_c = -1;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) {      // 32-bit: see idx32_ok
    const int c = (int)(i % (unsigned)N2);
    my_c = c;
    const unsigned t = i / (unsigned)N2;
    const int tx = (int)(t % (unsigned)TW);
    const unsigned r = t / (unsigned)TW;
    const int ty = (int)(r % (unsigned)TH);
    const int b = (int)(r / (unsigned)TH);
    f32x2 u[6][4];                                   // u = A g, built column by column
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const int ox = 4 * tx + o;
      f32x2 d[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int oy = 4 * ty + p;
        d[p] = (oy < H && ox < W) ? g2[(((long long)b * H + oy) * W + ox) * N2 + c] : zero;
        bsum += d[p];
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        f32x2 acc = zero;
#pragma unroll
        for (int p = 0; p < 4; ++p) acc += d[p] * W43_AT[p][k];
        u[k][o] = acc;
      }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        f32x2 acc = zero;
#pragma unroll
        for (int o = 0; o < 4; ++o) acc += u[k][o] * W43_AT[o][j];
        m2[((long long)(k * 6 + j) * T + t) * N2 + c] = acc;
      }
  }
  if (bias_grad) {                     // uniform: per-thread sums -> LDS -> one global atomic per channel and workgroup
    __shared__ float red[WINO_MAX_BIAS_N];
    const int n = N2 * 2;
    for (int k = threadIdx.x; k < n; k += 256) red[k] = 0.f;
    __syncthreads();
    if (my_c >= 0) {
      atomicAdd(&red[my_c * 2], bsum[0]);
      atomicAdd(&red[my_c * 2 + 1], bsum[1]);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += 256) atomicAdd(bias_grad + k, red[k]);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight-side transforms (once per parameter version in inference, once per optimisation step in training).
// G [(m+2) x 3]: F(2x2,3x3) with points 0, 1, -1, inf (rows scaled by 1/2 as usual); F(4x4,3x3) with 0, 1, -1, 1/2, -2, inf.
__constant__ double WG2[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
__constant__ double WG4[6][3] = {{1.0, 0.0, 0.0}, {1.0, 1.0, 1.0}, {1.0, -1.0, 1.0}, {1.0, 0.5, 0.25}, {1.0, -2.0, 4.0},
                                 {0.0, 0.0, 1.0}};

// g [N][C][3][3] (checkpoint layout) -> U [(m+2)^2][N'][C'] = (G g' G^T)[i][j], float64 arithmetic, one rounding.
// transposed = 0: g' = g, N' = N, C' = C.  transposed = 1 (weights of the data-gradient convolution): g'[c][n][a][b] =
// scale[n] g[n][c][2-a][2-b], N' = C, C' = N.
template <int M>
__global__ void wino_weight_kernel(const float* __restrict__ g, const float* __restrict__ scale, int N, int C,
                                   int transposed, float* __restrict__ U) {
  constexpr int A = M + 2;
  const long long total = (long long)N * C;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int n = (int)(idx / C), c = (int)(idx - (long long)n * C);
    double w[3][3];
    const double sc = scale ? (double)scale[n] : 1.0;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int ia = transposed ? 2 - a : a, ib = transposed ? 2 - b : b;
        w[a][b] = (double)g[idx * 9 + ia * 3 + ib] * sc;
      }
    const int no = transposed ? c : n, co = transposed ? n : c;
    const int No = transposed ? C : N, Co = transposed ? N : C;
    double t[A][3];                                     // t = G w
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        double acc = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) acc += (M == 2 ? WG2[i][a] : WG4[i][a]) * w[a][b];
        t[i][b] = acc;
      }
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
      for (int j = 0; j < A; ++j) {
        double acc = 0.0;
#pragma unroll
        for (int b = 0; b < 3; ++b) acc += t[i][b] * (M == 2 ? WG2[j][b] : WG4[j][b]);
        U[((long long)(i * A + j) * No + no) * Co + co] = (float)acc;
      }
  }
}

// dU [(m+2)^2][N][C] -> dW [N][C][3][3] = row_scale[n] * (G^T dU G)[a][b]   (fp32, fixed summation order)
template <int M>
__global__ void wino_weight_grad_kernel(const float* __restrict__ dU, const float* __restrict__ row_scale, int N, int C,
                                        float* __restrict__ dW) {
  constexpr int A = M + 2;
  const long long total = (long long)N * C;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int n = (int)(idx / C);
    float u[A][A];
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
      for (int j = 0; j < A; ++j) u[i][j] = dU[(long long)(i * A + j) * total + idx];
    const float sc = row_scale ? row_scale[n] : 1.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float t[A];                                       // t[j] = sum_i G[i][a] u[i][j]
#pragma unroll
      for (int j = 0; j < A; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < A; ++i) acc += (float)(M == 2 ? WG2[i][a] : WG4[i][a]) * u[i][j];
        t[j] = acc;
      }
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < A; ++j) acc += t[j] * (float)(M == 2 ? WG2[j][b] : WG4[j][b]);
        dW[idx * 9 + a * 3 + b] = acc * sc;
      }
    }
  }
}

// the transform kernels index their threads with 32 bits (a 64-bit div / mod chain per thread cost them 20-25 %): every launcher
// below refuses a launch with >= 2^31 (tile, channel-chunk) threads -- the callers cut the batch into chunks far below that
inline bool idx32_ok(long long threads) { return threads < 0x7fffffffll; }

inline int grid_for(long long n) {
  long long g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

}  // namespace

extern "C" int nbm_wino_input(const float* x, int B, int H, int W, int C, float* V, int m, void* stream) {
  if (!x || !V || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || (m != 2 && m != 4)) return NBM_EINVAL;
  if (!nbm_aligned16(x) || !nbm_aligned16(V)) return NBM_EALIGN;
  const long long tiles = (long long)B * ((H + m - 1) / m) * ((W + m - 1) / m);
  if (!idx32_ok(tiles * (C / 2))) return NBM_EUNSUPPORTED;
  if (m == 2)
    hipLaunchKernelGGL(wino23_input_kernel, dim3(grid_for(tiles * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, B, H, W, C / 4, V,
                       (const int*)nullptr, 0, (const unsigned*)nullptr);
  else
    hipLaunchKernelGGL(wino43_input_kernel, dim3(grid_for(tiles * (C / 2))), dim3(256), 0, (hipStream_t)stream, x, B, H, W, C / 2, V);
  return nbm_launch_status();
}

extern "C" int nbm_wino_outgrad(const float* g, int B, int H, int W, int N, float* dM, float* bias_grad, int m, void* stream) {
  if (!g || !dM || B <= 0 || H <= 0 || W <= 0 || N <= 0 || (N & 3) || (m != 2 && m != 4)) return NBM_EINVAL;
  if (!nbm_aligned16(g) || !nbm_aligned16(dM)) return NBM_EALIGN;
  if (bias_grad && N > WINO_MAX_BIAS_N) return NBM_EUNSUPPORTED;
  const int per = m == 2 ? N / 4 : N / 2;             // channel chunks per tile
  const long long total = (long long)B * ((H + m - 1) / m) * ((W + m - 1) / m) * per;
  if (!idx32_ok(total)) return NBM_EUNSUPPORTED;
  // the grid stride must be a multiple of `per` so that a thread keeps one channel chunk (bias-gradient accumulation)
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  while ((blocks * 256) % per) ++blocks;
  if (m == 2)
    hipLaunchKernelGGL(wino23_outgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, B, H, W, per, dM, bias_grad,
                       (const int*)nullptr, 0, (const unsigned*)nullptr, 0);
  else
    hipLaunchKernelGGL(wino43_outgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, B, H, W, per, dM, bias_grad);
  return nbm_launch_status();
}

// F(2x2,3x3) input / output-gradient transforms of the listed tiles only (compact operands) -- see nbm_hip.h.
extern "C" int nbm_wino23_input_tiles(const float* x, int B, int H, int W, int C, const int* tiles, int n_list,
                                      const unsigned* blk_info, float* V, void* stream) {
  if (!x || !V || !tiles || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || n_list < 0) return NBM_EINVAL;
  if (!nbm_aligned16(x) || !nbm_aligned16(V)) return NBM_EALIGN;
  if (n_list == 0) return NBM_OK;
  if (!idx32_ok((long long)n_list * (C / 4)) || !idx32_ok((long long)B * ((H + 1) / 2) * ((W + 1) / 2))) return NBM_EUNSUPPORTED;
  hipLaunchKernelGGL(wino23_input_kernel, dim3(grid_for((long long)n_list * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, B, H,
                     W, C / 4, V, tiles, n_list, blk_info);
  return nbm_launch_status();
}

extern "C" int nbm_wino23_outgrad_tiles(const float* g, int B, int H, int W, int N, const int* tiles, int n_list,
                                        const unsigned* plane_mask, float* dM, float* bias_grad, int skip_pattern_stride,
                                        void* stream) {
  if (!g || !dM || !tiles || B <= 0 || H <= 0 || W <= 0 || N <= 0 || (N & 3) || n_list < 0 || skip_pattern_stride < 0) return NBM_EINVAL;
  if (!nbm_aligned16(g) || !nbm_aligned16(dM)) return NBM_EALIGN;
  if (n_list == 0) return NBM_OK;
  if (bias_grad && N > WINO_MAX_BIAS_N) return NBM_EUNSUPPORTED;
  const int per = N / 4;
  if (!idx32_ok((long long)n_list * per) || !idx32_ok((long long)B * ((H + 1) / 2) * ((W + 1) / 2))) return NBM_EUNSUPPORTED;
  long long blocks = ((long long)n_list * per + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  while ((blocks * 256) % per) ++blocks;              // a thread keeps one channel chunk (bias-gradient accumulation)
  hipLaunchKernelGGL(wino23_outgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, B, H, W, per, dM,
                     bias_grad, tiles, n_list, plane_mask, skip_pattern_stride);
  return nbm_launch_status();
}

extern "C" int nbm_wino_output(const float* M, const float* scale, const float* shift, const float* mask, int relu, int B,
                               int H, int W, int N, float* y, int m, const float* residual, void* stream) {
  if (residual && (m != 4 || !nbm_aligned16(residual))) return NBM_EUNSUPPORTED;
  if (!M || !y || B <= 0 || H <= 0 || W <= 0 || N <= 0 || (N & 3) || (m != 2 && m != 4)) return NBM_EINVAL;
  if (!nbm_aligned16(M) || !nbm_aligned16(y) || (shift && !nbm_aligned16(shift)) || (scale && !nbm_aligned16(scale)) ||
      (mask && !nbm_aligned16(mask)))
    return NBM_EALIGN;
  const long long tiles = (long long)B * ((H + m - 1) / m) * ((W + m - 1) / m);
  if (!idx32_ok(tiles * (N / 2))) return NBM_EUNSUPPORTED;
  if (m == 2)
    hipLaunchKernelGGL(wino23_output_kernel, dim3(grid_for(tiles * (N / 4))), dim3(256), 0, (hipStream_t)stream, M, scale, shift,
                       mask, relu, B, H, W, N / 4, y);      // (residual: F(4x4) only, checked above)
  else
    hipLaunchKernelGGL(wino43_output_kernel, dim3(grid_for(tiles * (N / 2))), dim3(256), 0, (hipStream_t)stream, M, scale, shift,
                       mask, relu, B, H, W, N / 2, y, residual);
  return nbm_launch_status();
}

extern "C" int nbm_wino_weight(const float* g, const float* scale, int N, int C, int transposed, int m, float* U, void* stream) {
  if (!g || !U || N <= 0 || C <= 0 || (m != 2 && m != 4)) return NBM_EINVAL;
  if (m == 2)
    hipLaunchKernelGGL(wino_weight_kernel<2>, dim3(grid_for((long long)N * C)), dim3(256), 0, (hipStream_t)stream, g, scale, N, C,
                       transposed, U);
  else
    hipLaunchKernelGGL(wino_weight_kernel<4>, dim3(grid_for((long long)N * C)), dim3(256), 0, (hipStream_t)stream, g, scale, N, C,
                       transposed, U);
  return nbm_launch_status();
}

extern "C" int nbm_wino_weight_grad(const float* dU, const float* row_scale, int N, int C, int m, float* dW, void* stream) {
  if (!dU || !dW || N <= 0 || C <= 0 || (m != 2 && m != 4)) return NBM_EINVAL;
  if (m == 2)
    hipLaunchKernelGGL(wino_weight_grad_kernel<2>, dim3(grid_for((long long)N * C)), dim3(256), 0, (hipStream_t)stream, dU,
                       row_scale, N, C, dW);
  else
    hipLaunchKernelGGL(wino_weight_grad_kernel<4>, dim3(grid_for((long long)N * C)), dim3(256), 0, (hipStream_t)stream, dU,
                       row_scale, N, C, dW);
  return nbm_launch_status();
}
