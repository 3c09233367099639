// Backward / training-only point-wise kernels of the NBM detector (NHWC fp32): activation and pooling
// gradients, bilinear top-down gradient, softmax gradients, depthwise-conv gradients, train-mode BatchNorm,
// RoI-pool gradient, and the fused clip-norm + AdamW step over flat parameter buffers.
// All gather-form (deterministic) except where noted (weight-gradient and RoI reductions use fp32 atomics).
#include "nbm_common.h"

namespace {

constexpr int TPB = 256;
inline int grid_for(long long n, int per_block = TPB, int cap = 256 * 16) {
  long long g = (n + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}
#define GRID_STRIDE(i, n) \
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

__global__ void relu_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y, float* __restrict__ out,
                                long long n) {
  GRID_STRIDE(i, n) out[i] = y[i] > 0.f ? gy[i] : 0.f;
}

__global__ void leaky_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y, float* __restrict__ out,
                                 float slope, long long n) {
  GRID_STRIDE(i, n) out[i] = y[i] > 0.f ? gy[i] : gy[i] * slope;
}

// LayerNorm backward, one wave per row: gx = rstd * (g*w - mean(g*w) - xhat * mean(g*w*xhat)); every wave keeps its
// share of dW = sum g*xhat and dB = sum g in registers over the rows it visits and adds it to gw / gb (zeroed by the
// caller) once at the end.  E <= 1024.
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ g, long long rows, int E, float eps,
                                                            float* __restrict__ gx, float* __restrict__ gw,
                                                            float* __restrict__ gb) {
  const int lane = threadIdx.x & 63;
  float aw[16], ab[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) aw[k] = ab[k] = 0.f;
  for (long long row = blockIdx.x * 4ll + (threadIdx.x >> 6); row < rows; row += (long long)gridDim.x * 4) {
    const float* xr = x + row * E;
    const float* gr = g + row * E;
    float s = 0.f;
    for (int c = lane; c < E; c += 64) s += xr[c];
    const float mean = nbm_wave_sum(s) / (float)E;
    float q = 0.f;
    for (int c = lane; c < E; c += 64) { const float d = xr[c] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(nbm_wave_sum(q) / (float)E + eps);
    float c1 = 0.f, c2 = 0.f;
    for (int c = lane; c < E; c += 64) {
      const float xh = (xr[c] - mean) * rstd, gw_ = gr[c] * w[c];
      c1 += gw_;
      c2 += gw_ * xh;
    }
    c1 = nbm_wave_sum(c1) / (float)E;
    c2 = nbm_wave_sum(c2) / (float)E;
    float* o = gx + row * E;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int c = lane + 64 * k;
      if (c < E) {
        const float xh = (xr[c] - mean) * rstd, gv = gr[c];
        o[c] = rstd * (gv * w[c] - c1 - xh * c2);
        aw[k] += gv * xh;
        ab[k] += gv;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int c = lane + 64 * k;
    if (c < E) { atomicAdd(gw + c, aw[k]); atomicAdd(gb + c, ab[k]); }
  }
}

// Backward of nbm_mha_small (same token layout).  Pass 1, a wave per query row: recompute the probabilities, dP = dO V^T,
// dS = P (dP - sum(P dP)), dQ = scale dS K; P and dS rows go to the workspace.  Pass 2, a wave per key row:
// dK = scale dS^T Q, dV = P^T dO.
#define MHA_SMAX 128
__global__ __launch_bounds__(256) void mha_small_bwd_rows_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, const float* __restrict__ go,
    int q_ld, int k_ld, int v_ld, int go_ld, float* __restrict__ gq, int gq_ld, float* __restrict__ ws, int S, int nhead,
    int hd, long long seq_stride, long long batch_stride, const int* __restrict__ n_valid, float scale) {
  __shared__ float Ks[MHA_SMAX][65];
  __shared__ float Vs[MHA_SMAX][65];
  __shared__ float qs[4][64];
  __shared__ float gs[4][64];
  __shared__ float ds[4][MHA_SMAX];
  const int n = blockIdx.x / nhead, h = blockIdx.x - n * nhead;
  const int nv = n_valid ? min(*n_valid, S) : S;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < nv * hd; i += 256) {
    const int j = i / hd, d = i - j * hd;
    const long long row = j * seq_stride + n * batch_stride;
    Ks[j][d] = k[row * k_ld + h * hd + d];
    Vs[j][d] = v[row * v_ld + h * hd + d];
  }
  __syncthreads();
  float* P = ws + (long long)blockIdx.x * 2 * S * S;
  float* D = P + (long long)S * S;
  for (int r = wave; r < nv; r += 4) {
    const long long row = r * seq_stride + n * batch_stride;
    if (lane < hd) {
      qs[wave][lane] = q[row * q_ld + h * hd + lane] * scale;
      gs[wave][lane] = go[row * go_ld + h * hd + lane];
    }
    __builtin_amdgcn_wave_barrier();
    float sc[MHA_SMAX / 64], dp[MHA_SMAX / 64];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < MHA_SMAX / 64; ++t) {
      const int j = lane + 64 * t;
      float a = -INFINITY, b = 0.f;
      if (j < nv) {
        a = 0.f;
        for (int d = 0; d < hd; ++d) { a += qs[wave][d] * Ks[j][d]; b += gs[wave][d] * Vs[j][d]; }
      }
      sc[t] = a;
      dp[t] = b;
      m = fmaxf(m, a);
    }
    m = nbm_wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < MHA_SMAX / 64; ++t) {
      sc[t] = (lane + 64 * t < nv) ? expf(sc[t] - m) : 0.f;
      sum += sc[t];
    }
    sum = nbm_wave_sum(sum);
    float dot = 0.f;
#pragma unroll
    for (int t = 0; t < MHA_SMAX / 64; ++t) { sc[t] /= sum; dot += sc[t] * dp[t]; }
    dot = nbm_wave_sum(dot);
#pragma unroll
    for (int t = 0; t < MHA_SMAX / 64; ++t) {
      const int j = lane + 64 * t;
      if (j < nv) {
        const float dsv = sc[t] * (dp[t] - dot);
        ds[wave][j] = dsv;
        P[(long long)r * S + j] = sc[t];
        D[(long long)r * S + j] = dsv;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < hd) {
      float o = 0.f;
      for (int j = 0; j < nv; ++j) o += ds[wave][j] * Ks[j][lane];
      gq[row * gq_ld + h * hd + lane] = o * scale;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ __launch_bounds__(256) void mha_small_bwd_keys_kernel(
    const float* __restrict__ q, const float* __restrict__ go, int q_ld, int go_ld, const float* __restrict__ ws,
    float* __restrict__ gk, float* __restrict__ gv, int gk_ld, int gv_ld, int S, int nhead, int hd, long long seq_stride,
    long long batch_stride, const int* __restrict__ n_valid, float scale) {
  const int n = blockIdx.x / nhead, h = blockIdx.x - n * nhead;
  const int nv = n_valid ? min(*n_valid, S) : S;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* P = ws + (long long)blockIdx.x * 2 * S * S;
  const float* D = P + (long long)S * S;
  if (lane >= hd) return;
  for (int j = wave; j < nv; j += 4) {
    float ak = 0.f, av = 0.f;
    for (int r = 0; r < nv; ++r) {
      const long long row = r * seq_stride + n * batch_stride;
      ak += D[(long long)r * S + j] * q[row * q_ld + h * hd + lane];
      av += P[(long long)r * S + j] * go[row * go_ld + h * hd + lane];
    }
    const long long rowj = j * seq_stride + n * batch_stride;
    gk[rowj * gk_ld + h * hd + lane] = ak * scale;
    gv[rowj * gv_ld + h * hd + lane] = av;
  }
}

__global__ void silu_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x, float* __restrict__ out,
                                long long n) {
  GRID_STRIDE(i, n) {
    const float v = x[i], s = 1.0f / (1.0f + expf(-v));
    out[i] = gy[i] * (s * (1.0f + v * (1.0f - s)));
  }
}

__global__ void axpby_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                             float alpha, float beta, long long n, long long b_period) {
  GRID_STRIDE(i, n) out[i] = alpha * a[i] + (b ? beta * b[b_period ? i % b_period : i] : 0.f);
}

// BiFPN FusionModule (reference fpn.py:20-30): out = (sum_i relu(w_i) x_i) / (sum_i relu(w_i) + 1e-4), 2 or 3 inputs, the
// learnable weights read from the device (no host round trip).
__global__ void weighted_sum_kernel(const float* __restrict__ x0, const float* __restrict__ x1, const float* __restrict__ x2,
                                    const float* __restrict__ wraw, float* __restrict__ out, long long n4) {
  const float w0 = fmaxf(wraw[0], 0.f), w1 = fmaxf(wraw[1], 0.f), w2 = x2 ? fmaxf(wraw[2], 0.f) : 0.f;
  const float den = (x2 ? (w0 + w1) + w2 : w0 + w1) + 1e-4f;
  const f32x4 *a = reinterpret_cast<const f32x4*>(x0), *b = reinterpret_cast<const f32x4*>(x1),
              *c = reinterpret_cast<const f32x4*>(x2);
  f32x4* o = reinterpret_cast<f32x4*>(out);
  GRID_STRIDE(i, n4) {
    f32x4 num = a[i] * w0 + b[i] * w1;
    if (x2) num = num + c[i] * w2;
    o[i] = num / den;
  }
}

// gx_i = g relu(w_i)/den; gw_i += [w_i > 0] sum g (x_i - out)/den   (gw zeroed by the caller)
__global__ __launch_bounds__(256) void weighted_sum_bwd_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                               const float* __restrict__ x2, const float* __restrict__ wraw,
                                                               const float* __restrict__ g, float* __restrict__ gx0,
                                                               float* __restrict__ gx1, float* __restrict__ gx2,
                                                               float* __restrict__ gw, long long n4) {
  const float w0 = fmaxf(wraw[0], 0.f), w1 = fmaxf(wraw[1], 0.f), w2 = x2 ? fmaxf(wraw[2], 0.f) : 0.f;
  const float den = (x2 ? (w0 + w1) + w2 : w0 + w1) + 1e-4f;
  const f32x4 *a = reinterpret_cast<const f32x4*>(x0), *b = reinterpret_cast<const f32x4*>(x1),
              *c = reinterpret_cast<const f32x4*>(x2), *gg = reinterpret_cast<const f32x4*>(g);
  f32x4 *ga = reinterpret_cast<f32x4*>(gx0), *gb = reinterpret_cast<f32x4*>(gx1), *gc = reinterpret_cast<f32x4*>(gx2);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  GRID_STRIDE(i, n4) {
    const f32x4 va = a[i], vb = b[i], gv = gg[i];
    f32x4 num = va * w0 + vb * w1, vc = {0.f, 0.f, 0.f, 0.f};
    if (x2) { vc = c[i]; num = num + vc * w2; }
    const f32x4 o = num / den;
    if (ga) ga[i] = gv * (w0 / den);
    if (gb) gb[i] = gv * (w1 / den);
    if (x2 && gc) gc[i] = gv * (w2 / den);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s0 += gv[e] * (va[e] - o[e]);
      s1 += gv[e] * (vb[e] - o[e]);
      s2 += gv[e] * (vc[e] - o[e]);
    }
  }
  __shared__ float red[3][4];
  s0 = nbm_wave_sum(s0); s1 = nbm_wave_sum(s1); s2 = nbm_wave_sum(s2);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s0; red[1][threadIdx.x >> 6] = s1; red[2][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x < 3 && (threadIdx.x < 2 || x2)) {
    const float t = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
    if (wraw[threadIdx.x] > 0.f) atomicAdd(gw + threadIdx.x, t / den);
  }
}

// out[n] += sum_m g[m][n]   (block partial sums in double, one atomic per column per block)
__global__ void colsum_kernel(const float* __restrict__ g, long long M, int N, int ld, float* __restrict__ out) {
  const int n = blockIdx.y * 64 + (threadIdx.x & 63);
  const int sub = threadIdx.x >> 6;                      // 4 row lanes
  __shared__ double part[4][64];
  double acc = 0.0;
  if (n < N)
    for (long long m = blockIdx.x * 4ll + sub; m < M; m += (long long)gridDim.x * 4) acc += (double)g[m * ld + n];
  part[sub][threadIdx.x & 63] = acc;
  __syncthreads();
  if (sub == 0 && n < N) atomicAdd(out + n, (float)(part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]));
}

// gradient of 3x3/s2/p1 max pooling, gather form over the arg-max positions the forward pass recorded (r*3+s, one byte
// per output element): every input pixel looks at the <= 4 windows that contain it.  Reads idx + gy, writes gx once.
__global__ void maxpool_bwd_kernel(const uint8_t* __restrict__ idx, const float* __restrict__ gy, float* __restrict__ gx,
                                   int B, int H, int W, int C4, int Ho, int Wo, const float* __restrict__ residual,
                                   const float* __restrict__ mask) {
  const long long total = (long long)B * H * W * C4;
  const uint32_t* idx4 = reinterpret_cast<const uint32_t*>(idx);
  const f32x4* g4 = reinterpret_cast<const f32x4*>(gy);
  f32x4* o4 = reinterpret_cast<f32x4*>(gx);
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C4);
    long long t = i / C4;
    const int ix = (int)(t % W); t /= W;
    const int iy = (int)(t % H);
    const int b = (int)(t / H);
    f32x4 acc = residual ? reinterpret_cast<const f32x4*>(residual)[i] : f32x4{0.f, 0.f, 0.f, 0.f};   // other consumer's gradient
    for (int oy = (iy >> 1); oy <= ((iy + 1) >> 1); ++oy) {
      if (oy >= Ho) continue;
      const int r = iy - (oy * 2 - 1);
      for (int ox = (ix >> 1); ox <= ((ix + 1) >> 1); ++ox) {
        if (ox >= Wo) continue;
        const uint32_t me = (uint32_t)(r * 3 + ix - (ox * 2 - 1));
        const long long o = ((long long)(b * Ho + oy) * Wo + ox) * C4 + c;
        const uint32_t w4 = idx4[o];
        const f32x4 g = g4[o];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (((w4 >> (8 * e)) & 0xFFu) == me) acc[e] += g[e];
      }
    }
    if (mask) {                         // the pooled tensor is a ReLU output: hand its producer the gradient already masked
      const f32x4 m = reinterpret_cast<const f32x4*>(mask)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = m[e] > 0.f ? acc[e] : 0.f;
    }
    o4[i] = acc;
  }
}

// Contributors of coarse index P along one axis (scale s = (n_coarse - 1) / (n_fine - 1), fine index o -> s * o): the fine
// indices with P - 1 < s * o < P + 1, at most ceil(2 / s) <= NF of them, starting at `lo` = the first o with s * o > P - 1.  Fixed slots
// (index lo + k, weight 0 when the slot does not contribute or lies outside the pattern bands) instead of compacted lists: no
// dynamically indexed registers, ~40 VALU instructions per axis.
template <int NF>
__device__ __forceinline__ void upbwd_axis(int P, float s, float inv_s, int n_fine, int n_coarse, int pat, float inv_pat, int last_cell,
                                           int& lo, float (&w)[NF]) {
  lo = max((int)((float)(P - 1) * inv_s), 0);
  while (lo > 0 && s * (float)(lo - 1) > (float)(P - 1)) --lo;          // float rounding of the estimate: settle on the exact first index
  while (lo < n_fine && s * (float)lo <= (float)(P - 1)) ++lo;
#pragma unroll
  for (int k = 0; k < NF; ++k) {
    const int o = lo + k;
    const float f = s * (float)o;
    const int p0 = (int)f, p1 = p0 + (p0 < n_coarse - 1 ? 1 : 0);
    const float l = fminf(fmaxf(f - (float)p0, 0.f), 1.f);
    float wv = (p0 == P ? 1.f - l : 0.f) + (p1 == P ? l : 0.f);
    bool ok = o < n_fine;
    if (pat) {                                                           // (o + 2) % pat < 5 and the cell exists: exact for o < 2^20
      const int q = o + 2, cell = (int)(((float)q + 0.5f) * inv_pat);
      ok = ok && (q - cell * pat) < 5 && cell <= last_cell;
    }
    w[k] = ok ? wv : 0.f;
  }
}

// The same operator for scales with at most NF contributors per axis (2 / s <= NF: every level of the pyramid, s ~ 1/2).  What bounded the
// general kernel below was neither bandwidth nor latency but VALU issue: ~650 VALU + ~450 SALU instructions per wave to build its
// compacted lists (rocprofv3 --pmc SQ_INSTS_VALU on scripts/upbwd_probe.py: 3.0e9 per launch x 4 cycles per wave64 instruction / 1024
// SIMDs = 5.0 ms, the measured time -- which is why reading 39 % of the bytes did not make it faster).  Here: fixed slots, the row
// slots in scalar registers (Y = blockIdx.y), every load of a thread in flight at once.
// SHARE (C4 >= 32: a workgroup covers <= 9 coarse columns of one coarse row): the slots of the row and of those columns are computed
// ONCE per workgroup -- one lane each, in the first wave -- and handed out through LDS, instead of ~240 VALU instructions in every
// thread.
template <int NF, bool SHARE>
__global__ __launch_bounds__(256) void upsample_bwd_fast_kernel(const float* __restrict__ gy, int Hi, int Wi, int C4, float* __restrict__ gsrc,
                                                                int Ho, int Wo, float sh, float sw, int pat) {
  const f32x4* g4 = reinterpret_cast<const f32x4*>(gy);
  f32x4* o4 = reinterpret_cast<f32x4*>(gsrc);
  const int xc = blockIdx.x * 256 + threadIdx.x;
  const bool active = xc < Wi * C4;
  const int X = min(xc / C4, Wi - 1), c = xc - (xc / C4) * C4;
  const int Y = blockIdx.y, b = blockIdx.z;
  const float inv_pat = pat ? 1.f / (float)pat : 0.f;
  int ylo, xlo;
  float wyu[NF], wx[NF];
  if constexpr (SHARE) {
    constexpr int MAXX = 10;                         // 256 / 32 + 2
    __shared__ int s_lo[1 + MAXX];
    __shared__ float s_w[1 + MAXX][NF];
    const int x_first = (blockIdx.x * 256) / C4;
    const int t = threadIdx.x;
    if (t < 1 + MAXX) {
      int lo;
      float w[NF];
      if (t == 0) upbwd_axis<NF>(Y, sh, 1.f / sh, Ho, Hi, pat, inv_pat, pat ? (Ho - 1) / pat : 0, lo, w);
      else upbwd_axis<NF>(min(x_first + t - 1, Wi - 1), sw, 1.f / sw, Wo, Wi, pat, inv_pat, pat ? (Wo - 1) / pat : 0, lo, w);
      s_lo[t] = lo;
#pragma unroll
      for (int k = 0; k < NF; ++k) s_w[t][k] = w[k];
    }
    __syncthreads();
    if (!active) return;
    ylo = s_lo[0];
    xlo = s_lo[1 + X - x_first];
#pragma unroll
    for (int k = 0; k < NF; ++k) { wyu[k] = s_w[0][k]; wx[k] = s_w[1 + X - x_first][k]; }
  } else {
    if (!active) return;
    float wy[NF];
    upbwd_axis<NF>(Y, sh, 1.f / sh, Ho, Hi, pat, inv_pat, pat ? (Ho - 1) / pat : 0, ylo, wy);
    upbwd_axis<NF>(X, sw, 1.f / sw, Wo, Wi, pat, inv_pat, pat ? (Wo - 1) / pat : 0, xlo, wx);
#pragma unroll
    for (int a = 0; a < NF; ++a) wyu[a] = wy[a];
  }
  ylo = __builtin_amdgcn_readfirstlane(ylo);
#pragma unroll
  for (int a = 0; a < NF; ++a) wyu[a] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wyu[a])));
  const f32x4* base = g4 + ((long long)(b * Ho + ylo) * Wo + xlo) * C4 + c;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // two groups of rows (3 + 2): the loads of a group are all in flight together, and 15 instead of 25 vectors in registers keep
  // 6 waves per SIMD resident (all 25: 110 VGPRs, 4 waves -- measured slower; 3 channel vectors per thread: 9.2 ms, three dependent
  // round trips per wave)
#pragma unroll
  for (int a0 = 0; a0 < NF; a0 += 3) {
    f32x4 g[3][NF];
#pragma unroll
    for (int a = a0; a < a0 + 3 && a < NF; ++a)
      if (wyu[a] != 0.f) {                                               // workgroup-uniform
#pragma unroll
        for (int e = 0; e < NF; ++e)
          if (wx[e] != 0.f) g[a - a0][e] = base[((long long)a * Wo + e) * C4];
      }
#pragma unroll
    for (int a = a0; a < a0 + 3 && a < NF; ++a)
      if (wyu[a] != 0.f) {
#pragma unroll
        for (int e = 0; e < NF; ++e)
          if (wx[e] != 0.f) {
            const float w = wyu[a] * wx[e];
            acc[0] += w * g[a - a0][e][0]; acc[1] += w * g[a - a0][e][1]; acc[2] += w * g[a - a0][e][2]; acc[3] += w * g[a - a0][e][3];
          }
      }
  }
  o4[((long long)(b * Hi + Y) * Wi + X) * C4 + c] = acc;
}

// gradient of bilinear(align_corners) up-sampling wrt the coarse map, gather form
// `pat` = S > 0: gy is known to be zero outside the 5x5 patches around the 3x3 / stride-S pattern (the data gradient of a
// demand-driven level's output convolution, pattern share only) -- fine rows / columns outside the patch bands never enter the
// per-axis lists, i.e. the test is paid once per axis and thread, not once per load (the per-load form was the slower one above)
__global__ void upsample_bwd_kernel(const float* __restrict__ gy, int B, int Hi, int Wi, int C4, float* __restrict__ gsrc,
                                    int Ho, int Wo, float sh, float sw, int pat) {
  const f32x4* g4 = reinterpret_cast<const f32x4*>(gy);
  f32x4* o4 = reinterpret_cast<f32x4*>(gsrc);
  // grid = (ceil(Wi * C4 / 256), Hi, B): 32-bit index arithmetic only (the 64-bit div / mod chain of a flat index cost more than
  // the loads)
  const int xc = blockIdx.x * 256 + threadIdx.x;
  if (xc < Wi * C4) {
    const int X = xc / C4, c = xc - X * C4;
    const int Y = blockIdx.y, b = blockIdx.z;
    const long long i = ((long long)(b * Hi + Y) * Wi + X) * C4 + c;
    // The fine rows / columns whose interpolation touches coarse row Y / column X, with their weights: found ONCE per axis (up to
    // MAXT of them: ~2 / scale + 1), then a dense double loop over the two short lists -- the original form re-derived the column
    // weights for every candidate row and walked ~5 x 5 candidates of which <= 3 x 3 contribute.  (Measured at 188x512 -> 94x256 x 384,
    // B = 64: 4.36 ms flat 64-bit index + candidate walk -> 4.7 ms with the lists alone (!) -> 3.57 ms with 32-bit indices; skipping the
    // 61 % of the fine pixels that are known zeros in the data gradient of the demand-driven level -- pattern test + tile bitmap -- made
    // it SLOWER, 4.26 ms: the reads are L2 hits, the tests are not free.  Not kept.  Two more forms measured at B = 128 (6.86 ms, 18.9 GB
    // read = 2.8 TB/s) and not kept either: one contiguous band of coarse rows per XCD (6.82 ms) and the per-axis lists derived once per
    // workgroup and shared through LDS instead of once per thread (7.03 ms) -- it is neither L2 locality nor the list arithmetic.)
    constexpr int MAXT = 6;
    int ys[MAXT], xs[MAXT], ny = 0, nx = 0;
    float wys[MAXT], wxs[MAXT];
    {
      int lo = sh > 0.f ? (int)floorf((Y - 1) / sh) - 1 : 0, hi = sh > 0.f ? (int)ceilf((Y + 1) / sh) + 1 : Ho - 1;
      lo = max(lo, 0); hi = min(hi, Ho - 1);
      for (int oy = lo; oy <= hi; ++oy) {
        if (pat && ((oy + 2) % pat >= 5 || (oy + 2) / pat > (Ho - 1) / pat)) continue;
        const float fy = sh * oy;
        const int y0 = (int)fy, y1 = y0 + (y0 < Hi - 1 ? 1 : 0);
        const float ly = fminf(fmaxf(fy - y0, 0.f), 1.f);
        float wy = 0.f;
        if (y0 == Y) wy += 1.f - ly;
        if (y1 == Y) wy += ly;
        if (wy == 0.f) continue;
#pragma unroll
        for (int k = 0; k < MAXT; ++k) if (k == ny) { ys[k] = oy; wys[k] = wy; }
        if (ny < MAXT) ++ny;
      }
      lo = sw > 0.f ? (int)floorf((X - 1) / sw) - 1 : 0; hi = sw > 0.f ? (int)ceilf((X + 1) / sw) + 1 : Wo - 1;
      lo = max(lo, 0); hi = min(hi, Wo - 1);
      for (int ox = lo; ox <= hi; ++ox) {
        if (pat && ((ox + 2) % pat >= 5 || (ox + 2) / pat > (Wo - 1) / pat)) continue;
        const float fx = sw * ox;
        const int x0 = (int)fx, x1 = x0 + (x0 < Wi - 1 ? 1 : 0);
        const float lx = fminf(fmaxf(fx - x0, 0.f), 1.f);
        float wx = 0.f;
        if (x0 == X) wx += 1.f - lx;
        if (x1 == X) wx += lx;
        if (wx == 0.f) continue;
#pragma unroll
        for (int k = 0; k < MAXT; ++k) if (k == nx) { xs[k] = ox; wxs[k] = wx; }
        if (nx < MAXT) ++nx;
      }
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < MAXT; ++a) {
      if (a >= ny) break;
      const int oy = ys[a];
      const float wy = wys[a];
#pragma unroll
      for (int e = 0; e < MAXT; ++e) {
        if (e >= nx) break;
        const int ox = xs[e];
        const f32x4 g = g4[((long long)(b * Ho + oy) * Wo + ox) * C4 + c];
        const float w = wy * wxs[e];
        acc[0] += w * g[0]; acc[1] += w * g[1]; acc[2] += w * g[2]; acc[3] += w * g[3];
      }
    }
    o4[i] = acc;
  }
}

// dS = P * (dP - sum(P*dP)) * alpha, one wave per row
__global__ void softmax_bwd_kernel(const float* __restrict__ p, const float* __restrict__ gp, float* __restrict__ out,
                                   long long rows, int cols, float alpha) {
  const int lane = threadIdx.x & 63;
  for (long long row = blockIdx.x * 4ll + (threadIdx.x >> 6); row < rows; row += (long long)gridDim.x * 4) {
    const float* pr = p + row * cols;
    const float* gr = gp + row * cols;
    constexpr int REG = 24;                 // rows of up to 1536 columns in registers: p and g are read once (pointwise.hip, softmax_rows_kernel)
    if (cols <= 64 * REG) {
      float pv[REG], gv[REG];
#pragma unroll
      for (int j = 0; j < REG; ++j) {
        const int c = min(lane + 64 * j, cols - 1);
        pv[j] = pr[c]; gv[j] = gr[c];
      }
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < REG; ++j)
        if (lane + 64 * j < cols) s += pv[j] * gv[j];
      s = nbm_wave_sum(s);
      float* o = out + row * cols;
#pragma unroll
      for (int j = 0; j < REG; ++j)
        if (lane + 64 * j < cols) o[lane + 64 * j] = pv[j] * (gv[j] - s) * alpha;
      continue;
    }
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += pr[c] * gr[c];
    s = nbm_wave_sum(s);
    float* o = out + row * cols;
    for (int c = lane; c < cols; c += 64) o[c] = pr[c] * (gr[c] - s) * alpha;
  }
}

__global__ void pair_softmax_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gy,
                                        float* __restrict__ gx, long long n_pairs) {
  GRID_STRIDE(i, n_pairs) {
    const float p0 = y[2 * i], p1 = y[2 * i + 1], g0 = gy[2 * i], g1 = gy[2 * i + 1];
    const float s = p0 * g0 + p1 * g1;
    gx[2 * i] = p0 * (g0 - s);
    gx[2 * i + 1] = p1 * (g1 - s);
  }
}

// depthwise 3x3 (pad 1, multiplier mult, stride st): data gradient, gather form, 4 input channels per thread.
// Only taps r with (iy + 1 - r) % st == 0 reach an output row: r starts at (iy + 1) % st and steps by st, so with the
// RPN's stride 8 most pixels find no tap and just write zeros.
template <int MULT, bool HOIST = false>
__global__ void dwconv_bwd_data_kernel(const float* __restrict__ g, int B, int H, int W, int Cin, int mult_rt, int stride,
                                       const float* __restrict__ w, float* __restrict__ gx, int Ho, int Wo, int accumulate) {
  // accumulate != 0: gx already holds another consumer's gradient of the same map (the RoI pooling's, see
  // nbm_dwconv3x3_bwd_acc): add this one's on the pixels a tap reaches, touch nothing else.
  // one image row per blockIdx.y (grid-strided): with the RPN's strides 8 / 4 most rows are reached by no tap at all and
  // are plain zero fills; the per-pixel work is one division (was six), 12.3 -> see DESIGN for the measured time
  const int mult = MULT > 0 ? MULT : mult_rt;
  const int Cout = Cin * mult, C4 = Cin >> 2;
  const int row_items = W * C4;
  f32x4* o4 = reinterpret_cast<f32x4*>(gx);
  // MULT == 2: the 72 weights of this thread's 4 input channels in registers (its channel chunk is fixed when the item stride is
  // a multiple of C4): the per-tap scalar weight loads were what this kernel spent its time on
  // (NEGATIVE, end of round 3: the taps enumerated statically for strides 1 / 2 with the six loads of a filter row issued together and
  // the absent taps dropped afterwards -- 2.04 -> 2.49 ms for the three launches of a step: at stride 2 only 2.25 of the 9 taps exist)
  // (HOIST is only instantiated for strides <= 2: with larger strides most pixels are reached by no tap at all and preloading 72
  // weights per thread measured 1.8x slower; the register array also costs occupancy, hence a separate instantiation)
  float wr[HOIST ? 8 : 1][9];
  const bool w_fixed = HOIST && ((gridDim.x * blockDim.x) % C4) == 0;
  if constexpr (HOIST) {
    if (w_fixed) {
      const int c0w = (int)((blockIdx.x * blockDim.x + threadIdx.x) % C4) * 4;
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int k = 0; k < 9; ++k) wr[q][k] = w[((long long)c0w * 2 + q) * 9 + k];
    }
  }
  for (int row = blockIdx.y; row < B * H; row += gridDim.y) {
    const int iy = row % H, b = row / H;
    const int r_first = (iy + 1) % stride;
    bool row_has = false;
    for (int r = r_first; r < 3 && r <= iy + 1; r += stride)
      if ((iy + 1 - r) / stride < Ho) row_has = true;
    f32x4* orow = o4 + (long long)row * row_items;
    if (accumulate && !row_has) continue;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < row_items; j += gridDim.x * blockDim.x) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      bool any = false;
      if (row_has) {
        const int ix = j / C4, c0 = (j - ix * C4) * 4;
        if constexpr (MULT == 2) {
          // the 8 gradient values of this thread's 4 channels are contiguous: two 16-byte loads per tap instead of 8 scalar ones
          for (int r = r_first; r < 3 && r <= iy + 1; r += stride) {
            const int oy = (iy + 1 - r) / stride;
            if (oy >= Ho) continue;
            for (int s = (ix + 1) % stride; s < 3 && s <= ix + 1; s += stride) {
              const int ox = (ix + 1 - s) / stride;
              if (ox >= Wo) continue;
              const f32x4* gp = reinterpret_cast<const f32x4*>(g + ((long long)(b * Ho + oy) * Wo + ox) * Cout + c0 * 2);
              const f32x4 g0 = gp[0], g1 = gp[1];
              any = true;
              if (HOIST && w_fixed) {
                // r, s are not compile-time here (they start at (iy + 1) % stride): select the tap's weights with a small switch
                float w8[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                  float v = wr[HOIST ? q : 0][0];
#pragma unroll
                  for (int k = 1; k < 9; ++k) v = (r * 3 + s == k) ? wr[HOIST ? q : 0][k] : v;
                  w8[q] = v;
                }
                acc[0] += g0[0] * w8[0]; acc[0] += g0[1] * w8[1];
                acc[1] += g0[2] * w8[2]; acc[1] += g0[3] * w8[3];
                acc[2] += g1[0] * w8[4]; acc[2] += g1[1] * w8[5];
                acc[3] += g1[2] * w8[6]; acc[3] += g1[3] * w8[7];
              } else {
                const float* wp = w + (long long)c0 * 2 * 9 + r * 3 + s;
                acc[0] += g0[0] * wp[0];  acc[0] += g0[1] * wp[9];        // same order as the generic loop below
                acc[1] += g0[2] * wp[18]; acc[1] += g0[3] * wp[27];
                acc[2] += g1[0] * wp[36]; acc[2] += g1[1] * wp[45];
                acc[3] += g1[2] * wp[54]; acc[3] += g1[3] * wp[63];
              }
            }
          }
        } else
        for (int r = r_first; r < 3 && r <= iy + 1; r += stride) {      // r <= iy + 1: output row >= 0
          const int oy = (iy + 1 - r) / stride;
          if (oy >= Ho) continue;
          for (int s = (ix + 1) % stride; s < 3 && s <= ix + 1; s += stride) {
            const int ox = (ix + 1 - s) / stride;
            if (ox >= Wo) continue;
            const float* gp = g + ((long long)(b * Ho + oy) * Wo + ox) * Cout + c0 * mult;
            const float* wp = w + (long long)c0 * mult * 9 + r * 3 + s;
            any = true;
#pragma unroll
            for (int k = 0; k < 4; ++k)
              for (int e = 0; e < mult; ++e) acc[k] += gp[k * mult + e] * wp[(k * mult + e) * 9];
          }
        }
      }
      if (!accumulate) {
        orow[j] = acc;
      } else if (any) {
        const f32x4 o = orow[j];
        orow[j] = o + acc;
      }
    }
  }
}

// The same data gradient ADDED into gx, scatter form for strides >= 3: the 3x3 input blocks of different output pixels do not
// overlap, so every (output pixel, tap, 4 input channels) item owns its 16 bytes of gx -- no atomics, and only the 9 / stride^2 of
// the map a tap reaches is touched (the gather form above walks every pixel of every reached row: 2.5 ms for the 12.6 GB map of the
// finest level at B = 128 where 1.8 GB are read and written).  Same products in the same order as the gather form.
template <int MULT>
__global__ void dwconv_bwd_data_scatter_kernel(const float* __restrict__ g, int B, int H, int W, int Cin, int mult_rt, int stride,
                                               const float* __restrict__ w, float* __restrict__ gx, int Ho, int Wo) {
  // one thread = (output pixel, 4 input channels): its gradient values and the 9 taps' weights are read once, then nine 16-byte
  // read-modify-writes (the first form took one tap per thread and re-read both for every tap: 2.5 -> 1.9 ms; this one -> see DESIGN)
  const int mult = MULT > 0 ? MULT : mult_rt;
  const int Cout = Cin * mult, C4 = Cin >> 2;
  const long long total = (long long)B * Ho * Wo * C4;
  f32x4* o4 = reinterpret_cast<f32x4*>(gx);
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C4);
    long long t = i / C4;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const int c0 = c * 4;
    const float* gp = g + ((long long)(b * Ho + oy) * Wo + ox) * Cout + c0 * mult;
    const float* wp = w + (long long)c0 * mult * 9;
    if constexpr (MULT == 2) {
      const f32x4 g0 = reinterpret_cast<const f32x4*>(gp)[0], g1 = reinterpret_cast<const f32x4*>(gp)[1];
      float w8[8][9];
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int k = 0; k < 9; ++k) w8[q][k] = wp[q * 9 + k];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int iy = oy * stride - 1 + tap / 3, ix = ox * stride - 1 + tap % 3;
        if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc[0] += g0[0] * w8[0][tap]; acc[0] += g0[1] * w8[1][tap];
        acc[1] += g0[2] * w8[2][tap]; acc[1] += g0[3] * w8[3][tap];
        acc[2] += g1[0] * w8[4][tap]; acc[2] += g1[1] * w8[5][tap];
        acc[3] += g1[2] * w8[6][tap]; acc[3] += g1[3] * w8[7][tap];
        const long long o = (((long long)b * H + iy) * W + ix) * C4 + c;
        o4[o] = o4[o] + acc;
      }
    } else {
      for (int tap = 0; tap < 9; ++tap) {
        const int iy = oy * stride - 1 + tap / 3, ix = ox * stride - 1 + tap % 3;
        if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          for (int e = 0; e < mult; ++e) acc[k] += gp[k * mult + e] * wp[(k * mult + e) * 9 + tap];
        const long long o = (((long long)b * H + iy) * W + ix) * C4 + c;
        o4[o] = o4[o] + acc;
      }
    }
  }
}

// depthwise 3x3 weight (+bias) gradient: one block column per output channel group, atomics on 9(+1) scalars
__global__ void dwconv_bwd_weight_kernel(const float* __restrict__ x, const float* __restrict__ g, int B, int H, int W,
                                         int Cin, int mult, int stride, float* __restrict__ gw, float* __restrict__ gb,
                                         int Ho, int Wo) {
  const int Cout = Cin * mult;
  const int o = blockIdx.y * 64 + (threadIdx.x & 63);
  const int sub = threadIdx.x >> 6;
  __shared__ float part[4][64][10];
  float acc[10];
#pragma unroll
  for (int e = 0; e < 10; ++e) acc[e] = 0.f;
  if (o < Cout) {
    const int ci = o / mult;
    const long long npix = (long long)B * Ho * Wo;
    for (long long pix = blockIdx.x * 4ll + sub; pix < npix; pix += (long long)gridDim.x * 4) {
      long long t = pix;
      const int ox = (int)(t % Wo); t /= Wo;
      const int oy = (int)(t % Ho);
      const int b = (int)(t / Ho);
      const float gv = g[pix * Cout + o];
      acc[9] += gv;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int iy = oy * stride - 1 + r;
        if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const int ix = ox * stride - 1 + s;
          if ((unsigned)ix >= (unsigned)W) continue;
          acc[r * 3 + s] += gv * x[((long long)(b * H + iy) * W + ix) * Cin + ci];
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 10; ++e) part[sub][threadIdx.x & 63][e] = acc[e];
  __syncthreads();
  if (sub == 0 && o < Cout) {
#pragma unroll
    for (int e = 0; e < 9; ++e)
      atomicAdd(gw + o * 9 + e, part[0][threadIdx.x][e] + part[1][threadIdx.x][e] + part[2][threadIdx.x][e] + part[3][threadIdx.x][e]);
    if (gb) atomicAdd(gb + o, part[0][threadIdx.x][9] + part[1][threadIdx.x][9] + part[2][threadIdx.x][9] + part[3][threadIdx.x][9]);
  }
}

// The same weight gradient, four consecutive output channels per thread (MULT in {1, 2, 4}): one 16-byte load of g and one
// 16 / 8 / 4-byte load of x per tap and pixel instead of 4 + 4 x 9 scalar ones.  Same pixel partition per thread as the scalar
// kernel (stream `sub` of block x), hence the same partial sums; 64 threads x 4 channels = 256 channels per block column.
template <int MULT>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_vec_kernel(const float* __restrict__ x, const float* __restrict__ g, int B,
                                                                    int H, int W, int Cin, int stride, float* __restrict__ gw,
                                                                    float* __restrict__ gb, int Ho, int Wo) {
  constexpr int NIN = 4 / MULT;
  const int Cout = Cin * MULT;
  const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int o0 = (blockIdx.y * 64 + lane) * 4;
  __shared__ float part[4][64][41];
  float acc[4][10];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 10; ++e) acc[j][e] = 0.f;
  if (o0 < Cout) {
    const int ci0 = o0 / MULT;
    const int npix = B * Ho * Wo;                       // < 2^31: checked by the host
    for (int pix = blockIdx.x * 4 + sub; pix < npix; pix += gridDim.x * 4) {
      int t = pix;
      const int ox = t % Wo; t /= Wo;
      const int oy = t % Ho;
      const int b = t / Ho;
      const f32x4 gv = *reinterpret_cast<const f32x4*>(g + (long long)pix * Cout + o0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j][9] += gv[j];
      // all nine taps are loaded before the first one is used (clamped address, value zeroed outside the image): behind a bounds
      // branch each, the loads went out one at a time -- ten memory latencies per pixel and stream, which is what these launches cost
      float xv[9][NIN];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int iy = oy * stride - 1 + r;
        const int iyc = min(max(iy, 0), H - 1);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const int ix = ox * stride - 1 + s;
          const int ixc = min(max(ix, 0), W - 1);
          const float* xp = x + ((long long)(b * H + iyc) * W + ixc) * Cin + ci0;
          float* v = xv[r * 3 + s];
          if constexpr (NIN == 4) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(xp);
            v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
          } else if constexpr (NIN == 2) {
            const float2 q = *reinterpret_cast<const float2*>(xp);
            v[0] = q.x; v[1] = q.y;
          } else {
            v[0] = *xp;
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const bool oky = (unsigned)(oy * stride - 1 + r) < (unsigned)H;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const bool ok = oky && (unsigned)(ox * stride - 1 + s) < (unsigned)W;
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j][r * 3 + s] += gv[j] * (ok ? xv[r * 3 + s][j / MULT] : 0.f);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 10; ++e) part[sub][lane][j * 10 + e] = acc[j][e];
  __syncthreads();
  if (sub == 0 && o0 < Cout) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int e = 0; e < 9; ++e)
        atomicAdd(gw + (o0 + j) * 9 + e, part[0][lane][j * 10 + e] + part[1][lane][j * 10 + e] + part[2][lane][j * 10 + e] + part[3][lane][j * 10 + e]);
      if (gb) atomicAdd(gb + o0 + j, part[0][lane][j * 10 + 9] + part[1][lane][j * 10 + 9] + part[2][lane][j * 10 + 9] + part[3][lane][j * 10 + 9]);
    }
  }
}

// FiLM: y = z*gamma + beta with film[p][0:C] = gamma, film[p][C:2C] = beta
__global__ void film_fwd_kernel(const float* __restrict__ z, const float* __restrict__ film, float* __restrict__ y,
                                long long n_pix, int C) {
  const long long total = n_pix * C;
  GRID_STRIDE(i, total) {
    const long long p = i / C;
    const int c = (int)(i - p * C);
    y[i] = z[i] * film[p * 2 * C + c] + film[p * 2 * C + C + c];
  }
}
__global__ void film_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ z, const float* __restrict__ film,
                                float* __restrict__ gz, float* __restrict__ gfilm, long long n_pix, int C) {
  const long long total = n_pix * C;
  GRID_STRIDE(i, total) {
    const long long p = i / C;
    const int c = (int)(i - p * C);
    const float g = gy[i];
    gz[i] = g * film[p * 2 * C + c];
    gfilm[p * 2 * C + c] = g * z[i];
    gfilm[p * 2 * C + C + c] = g;
  }
}

// ---- train-mode BatchNorm over [M][C] (per-channel statistics over the M rows)
// pass 1: sums in double -> stats[c] = {sum, sumsq}
__global__ void bn_stats_kernel(const float* __restrict__ x, long long M, int C, double* __restrict__ stats) {
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int sub = threadIdx.x >> 6;
  __shared__ double ps[4][64], pq[4][64];
  double s = 0.0, q = 0.0;
  if (c < C)
    for (long long m = blockIdx.x * 4ll + sub; m < M; m += (long long)gridDim.x * 4) {
      const double v = (double)x[m * C + c];
      s += v; q += v * v;
    }
  ps[sub][threadIdx.x & 63] = s; pq[sub][threadIdx.x & 63] = q;
  __syncthreads();
  if (sub == 0 && c < C) {
    atomicAdd(stats + 2 * c, ps[0][threadIdx.x] + ps[1][threadIdx.x] + ps[2][threadIdx.x] + ps[3][threadIdx.x]);
    atomicAdd(stats + 2 * c + 1, pq[0][threadIdx.x] + pq[1][threadIdx.x] + pq[2][threadIdx.x] + pq[3][threadIdx.x]);
  }
}
// finalize: mean, invstd; running stats update (momentum, unbiased variance) -- nn.BatchNorm2d semantics
__global__ void bn_finalize_kernel(const double* __restrict__ stats, long long M, int C, float eps, float momentum,
                                   float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ run_mean,
                                   float* __restrict__ run_var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mu = stats[2 * c] / (double)M;
  double var = stats[2 * c + 1] / (double)M - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)mu;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (run_mean) {
    const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mu;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
  }
}
__global__ void bn_apply_kernel(const float* __restrict__ x, long long M, int C, const float* __restrict__ mean,
                                const float* __restrict__ invstd, const float* __restrict__ w, const float* __restrict__ b,
                                float* __restrict__ y) {
  const long long total = M * C;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C);
    y[i] = (x[i] - mean[c]) * invstd[c] * w[c] + b[c];
  }
}
// backward pass 1: red[c] = {sum g, sum g*xhat}
__global__ void bn_bwd_reduce_kernel(const float* __restrict__ g, const float* __restrict__ x, long long M, int C,
                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                     double* __restrict__ red) {
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int sub = threadIdx.x >> 6;
  __shared__ double ps[4][64], pq[4][64];
  double s = 0.0, q = 0.0;
  if (c < C) {
    const float mu = mean[c], is = invstd[c];
    for (long long m = blockIdx.x * 4ll + sub; m < M; m += (long long)gridDim.x * 4) {
      const double gv = (double)g[m * C + c];
      s += gv; q += gv * (double)((x[m * C + c] - mu) * is);
    }
  }
  ps[sub][threadIdx.x & 63] = s; pq[sub][threadIdx.x & 63] = q;
  __syncthreads();
  if (sub == 0 && c < C) {
    atomicAdd(red + 2 * c, ps[0][threadIdx.x] + ps[1][threadIdx.x] + ps[2][threadIdx.x] + ps[3][threadIdx.x]);
    atomicAdd(red + 2 * c + 1, pq[0][threadIdx.x] + pq[1][threadIdx.x] + pq[2][threadIdx.x] + pq[3][threadIdx.x]);
  }
}
// backward pass 2: gx = w*invstd*(g - sum_g/M - xhat*sum_gxhat/M); gw = sum_gxhat, gb = sum_g
__global__ void bn_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ x, long long M, int C,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ w, const double* __restrict__ red, float* __restrict__ gx,
                                    float* __restrict__ gw, float* __restrict__ gb) {
  const long long total = M * C;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C);
    const float xh = (x[i] - mean[c]) * invstd[c];
    const float sg = (float)(red[2 * c] / (double)M), sgx = (float)(red[2 * c + 1] / (double)M);
    gx[i] = w[c] * invstd[c] * (g[i] - sg - xh * sgx);
    if (i < C) { gw[i] = (float)red[2 * i + 1]; gb[i] = (float)red[2 * i]; }
  }
}

// ---- RoI pooling gradient (scatter with atomics; RoI windows overlap)
struct RoiBwdParams {
  float* gfmap[5]; int fh[5], fw[5]; int C; const float* rois; const int* level; int B, n_roi; const float* gpool;
};
__global__ void roi_pool_bwd_kernel(const RoiBwdParams p) {
  const int slot = blockIdx.x;
  const int b = slot / p.n_roi;
  const float* roi = p.rois + (long long)slot * 4;
  const int lvl = p.level[slot];
  const float stride = (float)(2 << lvl);
  int x1 = (int)rintf(roi[0] / stride), y1 = (int)rintf(roi[1] / stride);
  int x2 = (int)rintf(roi[2] / stride), y2 = (int)rintf(roi[3] / stride);
  const int H = p.fh[lvl], W = p.fw[lvl];
  y2 = min(y2, H - 1);
  while (y2 - y1 + 1 < 2) { y1 = max(0, y1 - 1); y2 = min(H - 1, y2 + 1); }
  while (x2 - x1 + 1 < 2) { x1 = max(0, x1 - 1); x2 = min(W - 1, x2 + 1); }
  const int h = y2 - y1 + 1, w = min(x2, W - 1) - x1 + 1;
  float* gf = p.gfmap[lvl];
  const int C = p.C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ya = (i * h) / 2, yb = ((i + 1) * h + 1) / 2;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int xa = (j * w) / 2, xb = ((j + 1) * w + 1) / 2;
        const float gv = p.gpool[((long long)slot * 4 + i * 2 + j) * C + c] / (float)((yb - ya) * (xb - xa));
        for (int yy = ya; yy < yb; ++yy)
          for (int xx = xa; xx < xb; ++xx)
            atomicAdd(gf + (((long long)b * H + y1 + yy) * W + x1 + xx) * C + c, gv);
      }
    }
  }
}

// ---- targeted re-zeroing of the persistent gradient maps of a demand-driven FPN level (nbm_hip.h: nbm_zero_*): the map is zero
// everywhere except where these three writers left something, so restoring it costs their footprint, not a fill of the whole map.
__global__ void zero_roi_windows_kernel(float* __restrict__ g, int H, int W, int C4, const float* __restrict__ rois,
                                        const int* __restrict__ level, int n_roi, int lvl) {
  const int slot = blockIdx.x;
  if (level[slot] != lvl) return;
  const int b = slot / n_roi;
  const float* roi = rois + (long long)slot * 4;
  const float stride = (float)(2 << lvl);
  int x1 = (int)rintf(roi[0] / stride), y1 = (int)rintf(roi[1] / stride);           // the window of roi_pool_bwd_kernel
  int x2 = (int)rintf(roi[2] / stride), y2 = (int)rintf(roi[3] / stride);
  y2 = min(y2, H - 1);
  while (y2 - y1 + 1 < 2) { y1 = max(0, y1 - 1); y2 = min(H - 1, y2 + 1); }
  while (x2 - x1 + 1 < 2) { x1 = max(0, x1 - 1); x2 = min(W - 1, x2 + 1); }
  const int h = y2 - y1 + 1, w = min(x2, W - 1) - x1 + 1;
  f32x4* g4 = reinterpret_cast<f32x4*>(g);
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < h * w * C4; i += blockDim.x) {
    const int c = i % C4, px = i / C4;
    const int yy = px / w, xx = px - yy * w;
    g4[(((long long)b * H + y1 + yy) * W + x1 + xx) * C4 + c] = z;
  }
}
__global__ void zero_pattern_kernel(float* __restrict__ g, int H, int W, int C4, int stride, int OH, int OW) {
  const int cell = blockIdx.x, b = blockIdx.y;
  const int oy = cell / OW, ox = cell - oy * OW;
  f32x4* g4 = reinterpret_cast<f32x4*>(g);
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < 9 * C4; i += blockDim.x) {
    const int c = i % C4, t = i / C4;
    const int y = oy * stride - 1 + t / 3, x = ox * stride - 1 + t % 3;
    if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) g4[(((long long)b * H + y) * W + x) * C4 + c] = z;
  }
}
__global__ void zero_tiles_kernel(float* __restrict__ g, int H, int W, int C4, const int* __restrict__ tiles,
                                  const int* __restrict__ n_blocks) {
  if (n_blocks && (int)blockIdx.x >= *n_blocks) return;
  const int TH = (H + 1) >> 1, TW = (W + 1) >> 1;
  f32x4* g4 = reinterpret_cast<f32x4*>(g);
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (int e = 0; e < 128; ++e) {
    const int t = tiles[blockIdx.x * 128 + e];
    if (t < 0) continue;                       // block-uniform
    const int b = t / (TH * TW), rem = t - b * (TH * TW);
    const int ty = rem / TW, tx = rem - ty * TW;
    for (int i = threadIdx.x; i < 4 * C4; i += blockDim.x) {
      const int c = i % C4, px = i / C4;
      const int y = 2 * ty + (px >> 1), x = 2 * tx + (px & 1);
      if (y < H && x < W) g4[(((long long)b * H + y) * W + x) * C4 + c] = z;
    }
  }
}

// compact [n_entries][2][2][C] <-> the 2x2 tiles of a list in a map [B][H][W][C].  MODE 0: compact = map (pixels outside the image:
// zeros); MODE 1: map += compact (every tile listed once)
template <int MODE>
__global__ void tiles_copy_kernel(float* __restrict__ map, int H, int W, int C4, const int* __restrict__ tiles,
                                  const int* __restrict__ n_blocks, float* __restrict__ compact) {
  if (n_blocks && (int)(blockIdx.x >> 4) >= *n_blocks) return;       // 8 list entries per workgroup, 16 workgroups per 128-entry block
  const int TH = (H + 1) >> 1, TW = (W + 1) >> 1;
  f32x4* m4 = reinterpret_cast<f32x4*>(map);
  f32x4* c4 = reinterpret_cast<f32x4*>(compact);
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (int e = 0; e < 8; ++e) {
    const int ent = blockIdx.x * 8 + e;
    const int t = tiles[ent];
    if (t < 0) continue;                       // block-uniform
    const int b = t / (TH * TW), rem = t - b * (TH * TW);
    const int ty = rem / TW, tx = rem - ty * TW;
    for (int i = threadIdx.x; i < 4 * C4; i += blockDim.x) {
      const int c = i % C4, px = i / C4;
      const int y = 2 * ty + (px >> 1), x = 2 * tx + (px & 1);
      const bool in = y < H && x < W;
      const long long mi = (((long long)b * H + y) * W + x) * C4 + c, ci = ((long long)ent * 4 + px) * C4 + c;
      if (MODE == 0) c4[ci] = in ? m4[mi] : z;
      else if (in) m4[mi] = m4[mi] + c4[ci];
    }
  }
}

// The same bilinear backward in scatter form for the 2x2 tiles of a list held as compact [n_entries][2][2][C] values (the RoI
// share of a demand-driven level's data gradient): every fine pixel adds its four weighted shares into the coarse map.
__global__ void tiles_upsample_bwd_add_kernel(const float* __restrict__ compact, int Ho, int Wo, int C, const int* __restrict__ tiles,
                                              const int* __restrict__ n_blocks, float* __restrict__ gsrc, int Hi, int Wi, float sh,
                                              float sw) {
  // 8 list entries per workgroup (a list of a few 10 000 tiles is only a few hundred 128-entry blocks: too few workgroups for 256 CUs)
  if (n_blocks && (int)(blockIdx.x >> 4) >= *n_blocks) return;
  const int TH = (Ho + 1) >> 1, TW = (Wo + 1) >> 1;
  for (int e = 0; e < 8; ++e) {
    const int ent = blockIdx.x * 8 + e;
    const int t = tiles[ent];
    if (t < 0) continue;                       // block-uniform
    const int b = t / (TH * TW), rem = t - b * (TH * TW);
    const int ty = rem / TW, tx = rem - ty * TW;
    for (int i = threadIdx.x; i < 4 * C; i += blockDim.x) {
      const int c = i % C, px = i / C;
      const int oy = 2 * ty + (px >> 1), ox = 2 * tx + (px & 1);
      if (oy >= Ho || ox >= Wo) continue;
      const float g = compact[((long long)ent * 4 + px) * C + c];
      const float fy = sh * oy, fx = sw * ox;
      const int y0 = (int)fy, y1 = y0 + (y0 < Hi - 1 ? 1 : 0), x0 = (int)fx, x1 = x0 + (x0 < Wi - 1 ? 1 : 0);
      const float ly = fminf(fmaxf(fy - y0, 0.f), 1.f), lx = fminf(fmaxf(fx - x0, 0.f), 1.f);
      float* row0 = gsrc + ((long long)(b * Hi + y0) * Wi) * C + c;
      float* row1 = gsrc + ((long long)(b * Hi + y1) * Wi) * C + c;
      atomicAdd(row0 + (long long)x0 * C, (1.f - ly) * (1.f - lx) * g);
      atomicAdd(row0 + (long long)x1 * C, (1.f - ly) * lx * g);
      atomicAdd(row1 + (long long)x0 * C, ly * (1.f - lx) * g);
      atomicAdd(row1 + (long long)x1 * C, ly * lx * g);
    }
  }
}

// ---- optimiser: squared gradient norm, then clip + AdamW (torch.optim.AdamW semantics, decoupled decay)
__global__ void sqnorm_kernel(const float* __restrict__ g, long long n, double* __restrict__ out) {
  double acc = 0.0;
  GRID_STRIDE(i, n) { const double v = (double)g[i]; acc += v * v; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  __shared__ double part[TPB / 64];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) { double s = 0.0; for (int i = 0; i < TPB / 64; ++i) s += part[i]; atomicAdd(out, s); }
}
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, long long n, float lr, float beta1, float beta2, float eps, float wd,
                             float bc1, float bc2_sqrt, const double* __restrict__ sqnorm, float max_norm) {
  float coef = 1.f;
  if (sqnorm && max_norm > 0.f) {                                  // torch.nn.utils.clip_grad_norm_
    const float total = (float)sqrt(*sqnorm);
    coef = fminf(max_norm / (total + 1e-6f), 1.0f);
  }
  GRID_STRIDE(i, n) {
    const float gi = g[i] * coef;
    float pi = p[i];
    pi *= 1.f - lr * wd;
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
  }
}

}  // namespace

#define ST ((hipStream_t)stream)

// gradient of nbm_avgpool2x2: every pixel of a 2x2 block receives a quarter of the block's gradient
__global__ void avgpool2x2_bwd_kernel(const float* __restrict__ gy, int B, int Ho, int Wo, int C4, float* __restrict__ gx) {
  const int H = 2 * Ho, W = 2 * Wo;
  const long long total = (long long)B * H * W * C4;
  const f32x4* g4 = reinterpret_cast<const f32x4*>(gy);
  f32x4* o4 = reinterpret_cast<f32x4*>(gx);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long long t = i / C4;
    const int xx = (int)(t % W); t /= W;
    const int yy = (int)(t % H);
    const int b = (int)(t / H);
    const f32x4 g = g4[(((long long)b * Ho + (yy >> 1)) * Wo + (xx >> 1)) * C4 + c];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = g[e] * 0.25f;
    o4[i] = o;
  }
}

extern "C" int nbm_avgpool2x2_bwd(const float* gy, int B, int Ho, int Wo, int C, float* gx, void* stream) {
  if (!gy || !gx || B <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || (C & 3)) return NBM_EINVAL;
  if (!nbm_aligned16(gy) || !nbm_aligned16(gx)) return NBM_EALIGN;
  hipLaunchKernelGGL(avgpool2x2_bwd_kernel, dim3(grid_for((long long)B * 4 * Ho * Wo * (C / 4))), dim3(TPB), 0, (hipStream_t)stream, gy, B,
                     Ho, Wo, C / 4, gx);
  return nbm_launch_status();
}

extern "C" int nbm_relu_bwd(const float* gy, const float* y, float* out, int64_t n, void* stream) {
  if (!gy || !y || !out || n <= 0) return NBM_EINVAL;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n)), dim3(TPB), 0, ST, gy, y, out, (long long)n);
  return nbm_launch_status();
}
extern "C" int nbm_leaky_relu_bwd(const float* gy, const float* y, float* out, float slope, int64_t n, void* stream) {
  if (!gy || !y || !out || n <= 0) return NBM_EINVAL;
  hipLaunchKernelGGL(leaky_bwd_kernel, dim3(grid_for(n)), dim3(TPB), 0, ST, gy, y, out, slope, (long long)n);
  return nbm_launch_status();
}
extern "C" int nbm_layernorm_bwd(const float* x, const float* w, const float* g, int64_t rows, int E, float eps, float* gx,
                                 float* gw, float* gb, void* stream) {
  if (!x || !w || !g || !gx || !gw || !gb || rows <= 0 || E <= 0 || E > 1024) return NBM_EINVAL;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(grid_for(rows, 4, 256)), dim3(256), 0, ST, x, w, g, (long long)rows, E, eps,
                     gx, gw, gb);
  return nbm_launch_status();
}
extern "C" int nbm_mha_small_bwd(const float* q, const float* k, const float* v, const float* go, int q_ld, int k_ld,
                                 int v_ld, int go_ld, float* gq, float* gk, float* gv, int gq_ld, int gk_ld, int gv_ld,
                                 float* workspace, int S, int N, int nhead, int hd, int64_t seq_stride,
                                 int64_t batch_stride, const int32_t* n_valid, float scale, void* stream) {
  if (!q || !k || !v || !go || !gq || !gk || !gv || !workspace) return NBM_EINVAL;
  if (S <= 0 || S > MHA_SMAX || N <= 0 || nhead <= 0 || hd <= 0 || hd > 64) return NBM_EINVAL;
  const int E = nhead * hd;
  if (q_ld < E || k_ld < E || v_ld < E || go_ld < E || gq_ld < E || gk_ld < E || gv_ld < E) return NBM_EINVAL;
  hipLaunchKernelGGL(mha_small_bwd_rows_kernel, dim3(N * nhead), dim3(256), 0, ST, q, k, v, go, q_ld, k_ld, v_ld, go_ld, gq,
                     gq_ld, workspace, S, nhead, hd, (long long)seq_stride, (long long)batch_stride, n_valid, scale);
  hipLaunchKernelGGL(mha_small_bwd_keys_kernel, dim3(N * nhead), dim3(256), 0, ST, q, go, q_ld, go_ld, workspace, gk, gv,
                     gk_ld, gv_ld, S, nhead, hd, (long long)seq_stride, (long long)batch_stride, n_valid, scale);
  return nbm_launch_status();
}
extern "C" int nbm_silu_bwd(const float* gy, const float* x, float* out, int64_t n, void* stream) {
  if (!gy || !x || !out || n <= 0) return NBM_EINVAL;
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(grid_for(n)), dim3(TPB), 0, ST, gy, x, out, (long long)n);
  return nbm_launch_status();
}
extern "C" int nbm_axpby(const float* a, const float* b, float* out, float alpha, float beta, int64_t n, int64_t b_period,
                         void* stream) {
  if (!a || !out || n <= 0 || b_period < 0) return NBM_EINVAL;
  hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(TPB), 0, ST, a, b, out, alpha, beta, (long long)n,
                     (long long)b_period);
  return nbm_launch_status();
}
extern "C" int nbm_weighted_sum(const float* x0, const float* x1, const float* x2, const float* weights, float* out, int64_t n,
                                void* stream) {
  if (!x0 || !x1 || !weights || !out || n <= 0 || (n & 3)) return NBM_EINVAL;
  if (!nbm_aligned16(x0) || !nbm_aligned16(x1) || (x2 && !nbm_aligned16(x2)) || !nbm_aligned16(out)) return NBM_EALIGN;
  hipLaunchKernelGGL(weighted_sum_kernel, dim3(grid_for(n / 4)), dim3(TPB), 0, ST, x0, x1, x2, weights, out, (long long)(n / 4));
  return nbm_launch_status();
}
extern "C" int nbm_weighted_sum_bwd(const float* x0, const float* x1, const float* x2, const float* weights, const float* g,
                                    float* gx0, float* gx1, float* gx2, float* gw, int64_t n, void* stream) {
  if (!x0 || !x1 || !weights || !g || !gw || n <= 0 || (n & 3)) return NBM_EINVAL;
  hipLaunchKernelGGL(weighted_sum_bwd_kernel, dim3(grid_for(n / 4, TPB, 2048)), dim3(256), 0, ST, x0, x1, x2, weights, g, gx0,
                     gx1, gx2, gw, (long long)(n / 4));
  return nbm_launch_status();
}
extern "C" int nbm_colsum(const float* g, int64_t M, int N, int ld, float* out, void* stream) {
  if (!g || !out || M <= 0 || N <= 0 || ld < N) return NBM_EINVAL;
  hipError_t e = nbm_zero_async(out, sizeof(float) * N, ST);
  if (e != hipSuccess) return (int)e;
  dim3 grid(grid_for(M, 4, 1024), (N + 63) / 64);
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, ST, g, (long long)M, N, ld, out);
  return nbm_launch_status();
}
extern "C" int nbm_maxpool3x3s2_bwd(const uint8_t* idx, const float* gy, float* gx, int B, int H, int W, int C, int Ho,
                                    int Wo, const float* residual, const float* mask, void* stream) {
  if (!idx || !gy || !gx || B <= 0 || C <= 0 || (C & 3)) return NBM_EINVAL;
  if ((residual && !nbm_aligned16(residual)) || (mask && !nbm_aligned16(mask))) return NBM_EALIGN;
  if ((H + 2 - 3) / 2 + 1 != Ho || (W + 2 - 3) / 2 + 1 != Wo) return NBM_EINVAL;
  if (!nbm_aligned16(gy) || !nbm_aligned16(gx) || (((uintptr_t)idx) & 3u)) return NBM_EALIGN;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((long long)B * H * W * (C / 4))), dim3(TPB), 0, ST, idx, gy, gx, B, H,
                     W, C / 4, Ho, Wo, residual, mask);
  return nbm_launch_status();
}
extern "C" int nbm_upsample_bilinear_bwd(const float* gy, int B, int Hi, int Wi, int C, float* gsrc, int Ho, int Wo,
                                         int pattern_stride, void* stream) {
  if (!gy || !gsrc || B <= 0 || C <= 0 || (C & 3) || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return NBM_EINVAL;
  if (pattern_stride < 0 || (pattern_stride > 0 && pattern_stride < 5)) return NBM_EINVAL;      // the 5x5 patches must be disjoint
  if (!nbm_aligned16(gy) || !nbm_aligned16(gsrc)) return NBM_EALIGN;
  const float sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
  const float sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  if (Hi > 65535 || B > 65535) return NBM_EUNSUPPORTED;
  const int C4 = C / 4;
  const dim3 grid((unsigned)(((long long)Wi * C4 + 255) / 256), Hi, B);
  constexpr int NF = 5;
  if (sh > 0.f && sw > 0.f && 2.f / sh <= NF - 0.05f && 2.f / sw <= NF - 0.05f && Ho < (1 << 20) && Wo < (1 << 20)) {
    if (C4 >= 32) hipLaunchKernelGGL((upsample_bwd_fast_kernel<NF, true>), grid, dim3(256), 0, ST, gy, Hi, Wi, C4, gsrc, Ho, Wo, sh, sw, pattern_stride);
    else hipLaunchKernelGGL((upsample_bwd_fast_kernel<NF, false>), grid, dim3(256), 0, ST, gy, Hi, Wi, C4, gsrc, Ho, Wo, sh, sw, pattern_stride);
  } else {
    hipLaunchKernelGGL(upsample_bwd_kernel, grid, dim3(256), 0, ST, gy, B, Hi, Wi, C4, gsrc, Ho, Wo, sh, sw, pattern_stride);
  }
  return nbm_launch_status();
}
extern "C" int nbm_tiles_upsample_bilinear_bwd_add(const float* compact, int B, int Ho, int Wo, int C, const int* tiles, int n_entries,
                                                   const int* n_blocks, float* gsrc, int Hi, int Wi, void* stream) {
  if (!compact || !tiles || !gsrc || B <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || n_entries <= 0 || (n_entries & 127))
    return NBM_EINVAL;
  const float sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
  const float sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  hipLaunchKernelGGL(tiles_upsample_bwd_add_kernel, dim3(n_entries / 8), dim3(256), 0, ST, compact, Ho, Wo, C, tiles, n_blocks, gsrc,
                     Hi, Wi, sh, sw);
  return nbm_launch_status();
}
extern "C" int nbm_softmax_rows_bwd(const float* p, const float* gp, float* out, int64_t rows, int cols, float alpha,
                                    void* stream) {
  if (!p || !gp || !out || rows <= 0 || cols <= 0) return NBM_EINVAL;
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(grid_for(rows, 4, 256 * 32)), dim3(256), 0, ST, p, gp, out, (long long)rows,
                     cols, alpha);
  return nbm_launch_status();
}
extern "C" int nbm_pair_softmax_bwd(const float* y, const float* gy, float* gx, int64_t n_pairs, void* stream) {
  if (!y || !gy || !gx || n_pairs <= 0) return NBM_EINVAL;
  hipLaunchKernelGGL(pair_softmax_bwd_kernel, dim3(grid_for(n_pairs)), dim3(TPB), 0, ST, y, gy, gx, (long long)n_pairs);
  return nbm_launch_status();
}
extern "C" int nbm_dwconv3x3_bwd(const float* x, const float* g, const float* w, int B, int H, int W, int Cin, int mult,
                                 int stride, float* gx, float* gw, float* gb, int Ho, int Wo, void* stream) {
  if (!x || !g || !w || B <= 0 || Cin <= 0 || mult <= 0 || stride <= 0) return NBM_EINVAL;
  if ((H + 2 - 3) / stride + 1 != Ho || (W + 2 - 3) / stride + 1 != Wo) return NBM_EINVAL;
  const int Cout = Cin * mult;
  if (gx) {
    if ((Cin & 3) || !nbm_aligned16(gx)) return NBM_EALIGN;
    const int bx = (W * (Cin / 4) + TPB - 1) / TPB;
    const dim3 grid(bx < 64 ? bx : 64, (long long)B * H < 65535 ? B * H : 65535);
    if (mult == 2 && stride <= 2) hipLaunchKernelGGL((dwconv_bwd_data_kernel<2, true>), grid, dim3(TPB), 0, ST, g, B, H, W, Cin, mult, stride, w, gx, Ho, Wo, 0);
    else if (mult == 2) hipLaunchKernelGGL(dwconv_bwd_data_kernel<2>, grid, dim3(TPB), 0, ST, g, B, H, W, Cin, mult, stride, w, gx, Ho, Wo, 0);
    else if (mult == 4) hipLaunchKernelGGL(dwconv_bwd_data_kernel<4>, grid, dim3(TPB), 0, ST, g, B, H, W, Cin, mult, stride, w, gx, Ho, Wo, 0);
    else hipLaunchKernelGGL(dwconv_bwd_data_kernel<0>, grid, dim3(TPB), 0, ST, g, B, H, W, Cin, mult, stride, w, gx, Ho, Wo, 0);
  }
  if (gw) {
    hipError_t e = nbm_zero_async(gw, sizeof(float) * Cout * 9, ST);
    if (e == hipSuccess && gb) e = nbm_zero_async(gb, sizeof(float) * Cout, ST);
    if (e != hipSuccess) return (int)e;
    // 192 workgroup columns: every workgroup ends with 2560 atomics on the same 5120 addresses, and that tail grows with the grid
    // (B = 128, 24 x 64 outputs, 512 channels: 64 -> 0.75, 128 -> 0.48, 192 -> 0.44, 256 -> 0.47, 512 -> 0.63, 1024 -> 0.86 ms)
    dim3 grid(grid_for((long long)B * Ho * Wo, 4, 192), (Cout + 63) / 64);
    const bool vec = (mult == 1 || mult == 2 || mult == 4) && (Cout & 3) == 0 && (Cin % (4 / mult)) == 0 && nbm_aligned16(g) &&
                     nbm_aligned16(x) && (long long)B * Ho * Wo < (1ll << 31) - 4ll * 65536;
    if (vec) {
      const dim3 gv(grid.x, (Cout + 255) / 256);
      if (mult == 1) hipLaunchKernelGGL(dwconv_bwd_weight_vec_kernel<1>, gv, dim3(256), 0, ST, x, g, B, H, W, Cin, stride, gw, gb, Ho, Wo);
      else if (mult == 2) hipLaunchKernelGGL(dwconv_bwd_weight_vec_kernel<2>, gv, dim3(256), 0, ST, x, g, B, H, W, Cin, stride, gw, gb, Ho, Wo);
      else hipLaunchKernelGGL(dwconv_bwd_weight_vec_kernel<4>, gv, dim3(256), 0, ST, x, g, B, H, W, Cin, stride, gw, gb, Ho, Wo);
      return nbm_launch_status();
    }
    hipLaunchKernelGGL(dwconv_bwd_weight_kernel, grid, dim3(256), 0, ST, x, g, B, H, W, Cin, mult, stride, gw, gb, Ho, Wo);
  }
  return nbm_launch_status();
}
// Data gradient of the depthwise 3x3 ADDED into gx -- see nbm_hip.h.
extern "C" int nbm_dwconv3x3_bwd_acc(const float* g, const float* w, int B, int H, int W, int Cin, int mult, int stride,
                                     float* gx, int Ho, int Wo, void* stream) {
  if (!g || !w || !gx || B <= 0 || Cin <= 0 || mult <= 0 || stride <= 0) return NBM_EINVAL;
  if ((H + 2 - 3) / stride + 1 != Ho || (W + 2 - 3) / stride + 1 != Wo) return NBM_EINVAL;
  if ((Cin & 3) || !nbm_aligned16(gx)) return NBM_EALIGN;
  if (stride >= 3 && (mult != 2 || nbm_aligned16(g))) {          // disjoint 3x3 blocks: scatter form
    const dim3 sg(grid_for((long long)B * Ho * Wo * (Cin / 4)));
    if (mult == 2) hipLaunchKernelGGL(dwconv_bwd_data_scatter_kernel<2>, sg, dim3(TPB), 0, ST, g, B, H, W, Cin, mult, stride, w, gx, Ho, Wo);
    else hipLaunchKernelGGL(dwconv_bwd_data_scatter_kernel<0>, sg, dim3(TPB), 0, ST, g, B, H, W, Cin, mult, stride, w, gx, Ho, Wo);
    return nbm_launch_status();
  }
  const int bx = (W * (Cin / 4) + TPB - 1) / TPB;
  const dim3 grid(bx < 64 ? bx : 64, (long long)B * H < 65535 ? B * H : 65535);
  if (mult == 2 && stride <= 2) hipLaunchKernelGGL((dwconv_bwd_data_kernel<2, true>), grid, dim3(TPB), 0, ST, g, B, H, W, Cin, mult, stride, w, gx, Ho, Wo, 1);
  else if (mult == 2) hipLaunchKernelGGL(dwconv_bwd_data_kernel<2>, grid, dim3(TPB), 0, ST, g, B, H, W, Cin, mult, stride, w, gx, Ho, Wo, 1);
  else if (mult == 4) hipLaunchKernelGGL(dwconv_bwd_data_kernel<4>, grid, dim3(TPB), 0, ST, g, B, H, W, Cin, mult, stride, w, gx, Ho, Wo, 1);
  else hipLaunchKernelGGL(dwconv_bwd_data_kernel<0>, grid, dim3(TPB), 0, ST, g, B, H, W, Cin, mult, stride, w, gx, Ho, Wo, 1);
  return nbm_launch_status();
}
extern "C" int nbm_film_fwd(const float* z, const float* film, float* y, int64_t n_pix, int C, void* stream) {
  if (!z || !film || !y || n_pix <= 0 || C <= 0) return NBM_EINVAL;
  hipLaunchKernelGGL(film_fwd_kernel, dim3(grid_for(n_pix * C)), dim3(TPB), 0, ST, z, film, y, (long long)n_pix, C);
  return nbm_launch_status();
}
extern "C" int nbm_film_bwd(const float* gy, const float* z, const float* film, float* gz, float* gfilm, int64_t n_pix,
                            int C, void* stream) {
  if (!gy || !z || !film || !gz || !gfilm || n_pix <= 0 || C <= 0) return NBM_EINVAL;
  hipLaunchKernelGGL(film_bwd_kernel, dim3(grid_for(n_pix * C)), dim3(TPB), 0, ST, gy, z, film, gz, gfilm, (long long)n_pix, C);
  return nbm_launch_status();
}
extern "C" int nbm_bn_train_fwd(const float* x, int64_t M, int C, const float* w, const float* b, float eps,
                                float momentum, float* run_mean, float* run_var, double* stats_ws, float* mean,
                                float* invstd, float* y, void* stream) {
  if (!x || !w || !b || !stats_ws || !mean || !invstd || !y || M <= 0 || C <= 0) return NBM_EINVAL;
  hipError_t e = nbm_zero_async(stats_ws, sizeof(double) * 2 * C, ST);
  if (e != hipSuccess) return (int)e;
  dim3 grid(grid_for(M, 4, 1024), (C + 63) / 64);
  hipLaunchKernelGGL(bn_stats_kernel, grid, dim3(256), 0, ST, x, (long long)M, C, stats_ws);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, ST, stats_ws, (long long)M, C, eps, momentum,
                     mean, invstd, run_mean, run_var);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(M * C)), dim3(TPB), 0, ST, x, (long long)M, C, mean, invstd, w, b, y);
  return nbm_launch_status();
}
extern "C" int nbm_bn_train_bwd(const float* g, const float* x, int64_t M, int C, const float* mean, const float* invstd,
                                const float* w, double* red_ws, float* gx, float* gw, float* gb, void* stream) {
  if (!g || !x || !mean || !invstd || !w || !red_ws || !gx || !gw || !gb || M <= 0 || C <= 0 || M * (int64_t)C < C) return NBM_EINVAL;
  hipError_t e = nbm_zero_async(red_ws, sizeof(double) * 2 * C, ST);
  if (e != hipSuccess) return (int)e;
  dim3 grid(grid_for(M, 4, 1024), (C + 63) / 64);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, grid, dim3(256), 0, ST, g, x, (long long)M, C, mean, invstd, red_ws);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(M * C)), dim3(TPB), 0, ST, g, x, (long long)M, C, mean, invstd, w,
                     red_ws, gx, gw, gb);
  return nbm_launch_status();
}
extern "C" int nbm_roi_pool_bwd(float* const* gfmap, const int* fh, const int* fw, int n_levels, int C, const float* rois,
                                const int* level, int B, int n_roi, const float* gpool, void* stream) {
  if (!gfmap || !fh || !fw || !rois || !level || !gpool || n_levels < 1 || n_levels > 5 || B <= 0 || n_roi <= 0) return NBM_EINVAL;
  RoiBwdParams p;
  for (int i = 0; i < 5; ++i) { p.gfmap[i] = i < n_levels ? gfmap[i] : nullptr; p.fh[i] = i < n_levels ? fh[i] : 0; p.fw[i] = i < n_levels ? fw[i] : 0; }
  p.C = C; p.rois = rois; p.level = level; p.B = B; p.n_roi = n_roi; p.gpool = gpool;
  hipLaunchKernelGGL(roi_pool_bwd_kernel, dim3(B * n_roi), dim3(256), 0, ST, p);
  return nbm_launch_status();
}
extern "C" int nbm_zero_roi_windows(float* g, int H, int W, int C, const float* rois, const int* level, int B, int n_roi, int lvl,
                                    void* stream) {
  if (!g || !rois || !level || B <= 0 || n_roi <= 0 || H < 2 || W < 2 || C <= 0 || (C & 3) || lvl < 0 || lvl > 4) return NBM_EINVAL;
  if (!nbm_aligned16(g)) return NBM_EALIGN;
  hipLaunchKernelGGL(zero_roi_windows_kernel, dim3(B * n_roi), dim3(256), 0, ST, g, H, W, C / 4, rois, level, n_roi, lvl);
  return nbm_launch_status();
}
extern "C" int nbm_zero_pattern(float* g, int B, int H, int W, int C, int stride, void* stream) {
  if (!g || B <= 0 || B > 65535 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || stride < 1) return NBM_EINVAL;
  if (!nbm_aligned16(g)) return NBM_EALIGN;
  const int OH = (H + 2 - 3) / stride + 1, OW = (W + 2 - 3) / stride + 1;
  hipLaunchKernelGGL(zero_pattern_kernel, dim3(OH * OW, B), dim3(256), 0, ST, g, H, W, C / 4, stride, OH, OW);
  return nbm_launch_status();
}
extern "C" int nbm_zero_tiles(float* g, int B, int H, int W, int C, const int* tiles, int n_entries, const int* n_blocks,
                              void* stream) {
  if (!g || !tiles || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || n_entries <= 0 || (n_entries & 127)) return NBM_EINVAL;
  if (!nbm_aligned16(g)) return NBM_EALIGN;
  hipLaunchKernelGGL(zero_tiles_kernel, dim3(n_entries / 128), dim3(256), 0, ST, g, H, W, C / 4, tiles, n_blocks);
  return nbm_launch_status();
}
extern "C" int nbm_tiles_gather(const float* map, int B, int H, int W, int C, const int* tiles, int n_entries, const int* n_blocks,
                                float* compact, void* stream) {
  if (!map || !tiles || !compact || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || n_entries <= 0 || (n_entries & 127)) return NBM_EINVAL;
  if (!nbm_aligned16(map) || !nbm_aligned16(compact)) return NBM_EALIGN;
  hipLaunchKernelGGL(tiles_copy_kernel<0>, dim3(n_entries / 8), dim3(256), 0, ST, const_cast<float*>(map), H, W, C / 4, tiles, n_blocks,
                     compact);
  return nbm_launch_status();
}
extern "C" int nbm_tiles_scatter_add(float* map, int B, int H, int W, int C, const int* tiles, int n_entries, const int* n_blocks,
                                     const float* compact, void* stream) {
  if (!map || !tiles || !compact || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || n_entries <= 0 || (n_entries & 127)) return NBM_EINVAL;
  if (!nbm_aligned16(map) || !nbm_aligned16(compact)) return NBM_EALIGN;
  hipLaunchKernelGGL(tiles_copy_kernel<1>, dim3(n_entries / 8), dim3(256), 0, ST, map, H, W, C / 4, tiles, n_blocks,
                     const_cast<float*>(compact));
  return nbm_launch_status();
}
extern "C" int nbm_sqnorm_accum(const float* g, int64_t n, double* out, void* stream) {
  if (!g || !out || n <= 0) return NBM_EINVAL;
  hipLaunchKernelGGL(sqnorm_kernel, dim3(grid_for(n, TPB, 2048)), dim3(TPB), 0, ST, g, (long long)n, out);
  return nbm_launch_status();
}
extern "C" int nbm_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int step, const double* sqnorm, float max_norm, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || step < 1) return NBM_EINVAL;
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(TPB), 0, ST, p, g, m, v, (long long)n, lr, beta1, beta2, eps,
                     weight_decay, bc1, bc2s, sqnorm, max_norm);
  return nbm_launch_status();
}
