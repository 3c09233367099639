// HBM-bound stages of the NBM hot path: waveform preparation, spectrogram normalise/window, and the
// point-wise / small-window detector layers.  All NHWC fp32; every kernel is a coalesced grid-stride
// sweep with the channel (or time) index fastest across lanes.
#include <stdlib.h>
#include "nbm_common.h"

namespace {

constexpr int TPB = 256;
inline int grid_for(long long n, int per_block = TPB, int cap = 256 * 16) {
  long long g = (n + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ------------------------------------------------------------------ front end
// sample i (0 <= i < n_out) of the 44.1 kHz signal as a 16-bit integer
__device__ __forceinline__ int pcm16_sample(const int16_t* __restrict__ x, int n, int upsample,
                                            const int32_t* __restrict__ hq, long long i) {
  if (!upsample) return x[i];
  if ((i & 1) == 0) return x[i >> 1];
  const int n0 = (int)(i >> 1);
  long long acc = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int ia = n0 - k, ib = n0 + 1 + k;
    // clamped address + select, not a load inside the condition: the 32 taps of a sample are then in flight together (one behind
    // the other, each behind its own branch, they made this 17 MB kernel take 0.2 ms)
    const int va = x[min(max(ia, 0), n - 1)], vb = x[min(max(ib, 0), n - 1)];
    const int xa = (ia >= 0 && ia < n) ? va : 0;
    const int xb = (ib >= 0 && ib < n) ? vb : 0;
    acc += (long long)hq[k] * (xa + xb);
  }
  long long q = (acc + 16384) >> 15;
  return (int)(q < -32768 ? -32768 : (q > 32767 ? 32767 : q));
}

__global__ void pcm16_to_wave_kernel(const int16_t* __restrict__ pcm, long long pcm_ld, int n, int upsample,
                                     const int32_t* __restrict__ hq, long long first, long long count,
                                     float* __restrict__ out, long long out_ld, int lead, int reflect) {
  const int b = blockIdx.y;
  const int16_t* x = pcm + (long long)b * pcm_ld;
  float* o = out + (long long)b * out_ld;
  for (long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x; j < out_ld;
       j += (long long)gridDim.x * blockDim.x) {
    long long i = j - lead;                                            // index inside the piece [first, first + count)
    float v = 0.f;
    if (reflect && i >= -(long long)lead && i < count + lead) {       // np.pad(mode='reflect') of the piece: lead < count
      if (i < 0) i = -i;
      else if (i >= count) i = 2 * (count - 1) - i;
    }
    if (i >= 0 && i < count) v = (float)pcm16_sample(x, n, upsample, hq, first + i) * (1.0f / 32768.0f);
    o[j] = v;
  }
}

// Generic front-end input: fp32 samples at any rate -> the piece [first, first + count) of the 44.1 kHz signal, centre
// padded like pcm16_to_wave_kernel.  L == M == 1: the samples themselves (float / 24- / 32-bit files at 44.1 kHz).
// Otherwise a rational L / M polyphase resampler: y[m] = sum_k taps[(m M) mod L][k] * x[floor(m M / L) - T/2 + 1 + k],
// taps [L][T] float64 (Kaiser-windowed sinc built on the host), accumulated in float64 in a fixed order (reproducible bit
// for bit on any device), then rounded to the 16-bit grid like the `ffmpeg -acodec pcm_s16le -ar 44100` step of the
// reference (prepare_dataset.py:175-178) when quant16 is set.
__global__ void resample_to_wave_kernel(const float* __restrict__ x, long long x_ld, long long n, int L, int M,
                                        const double* __restrict__ taps, int T, long long first, long long count,
                                        float* __restrict__ out, long long out_ld, int lead, int reflect, int quant16) {
  const int b = blockIdx.y;
  const float* xb = x + (long long)b * x_ld;
  float* o = out + (long long)b * out_ld;
  for (long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x; j < out_ld;
       j += (long long)gridDim.x * blockDim.x) {
    long long i = j - lead;
    float v = 0.f;
    if (reflect && i >= -(long long)lead && i < count + lead) {
      if (i < 0) i = -i;
      else if (i >= count) i = 2 * (count - 1) - i;
    }
    if (i >= 0 && i < count) {
      const long long m = first + i;
      if (L == 1 && M == 1) {
        v = xb[m];
      } else {
        const long long pos = m * M;
        const long long n0 = pos / L;
        const int ph = (int)(pos - n0 * L);
        const double* h = taps + (long long)ph * T;
        const long long nb = n0 - T / 2 + 1;
        double acc = 0.0;
        for (int k = 0; k < T; ++k) {
          const long long q = nb + k;
          if (q >= 0 && q < n) acc = fma(h[k], (double)xb[q], acc);
        }
        if (quant16) {
          double r = rint(acc * 32768.0);
          r = r < -32768.0 ? -32768.0 : (r > 32767.0 ? 32767.0 : r);
          v = (float)(r * (1.0 / 32768.0));
        } else {
          v = (float)acc;
        }
      }
    }
    o[j] = v;
  }
}

__global__ void minmax_init_kernel(uint32_t* mm, int batch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < batch) { mm[2 * i] = 0xFFFFFFFFu; mm[2 * i + 1] = 0u; }
}

__global__ void spec_windows_kernel(const float* __restrict__ db, long long db_bs, int db_ld, int n_bins,
                                    const uint32_t* __restrict__ minmax, float* __restrict__ img,
                                    int n_img, int w_pix, int hop_img, const int32_t* __restrict__ last_cols) {
  const int b = blockIdx.z, k = blockIdx.y;
  const float lo = nbm_key2f(minmax[2 * b]), hi = nbm_key2f(minmax[2 * b + 1]);
  const float range = hi - lo;
  const float* src = db + (long long)b * db_bs;
  float* dst = img + ((long long)b * n_img + k) * n_bins * w_pix;
  const int start = k * hop_img;
  const bool last = k == n_img - 1;
  const long long total = (long long)n_bins * w_pix;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int f = (int)(i / w_pix), c = (int)(i - (long long)f * w_pix);
    const int col = last ? last_cols[c] : start + c;
    dst[i] = (src[(long long)f * db_ld + col] - lo) / range;
  }
}

// ------------------------------------------------------------------ detector point-wise stages
__global__ void init_conv_kernel(const float* __restrict__ x, long long n_pix, const float* __restrict__ w,
                                 const float* __restrict__ b, int C, float* __restrict__ y) {
  const long long total = n_pix * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long p = i / C;
    const int c = (int)(i - p * C);
    y[i] = x[p] * w[c] + b[c];
  }
}

// idx (optional, training): position r*3+s of the FIRST maximum in (r, s) scan order, torch's tie rule, one byte per output
template <bool IDX>
__global__ void maxpool3x3s2_kernel(const float* __restrict__ x, int B, int H, int W, int C4,
                                    float* __restrict__ y, int Ho, int Wo, uint8_t* __restrict__ idx) {
  const long long total = (long long)B * Ho * Wo * C4;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  f32x4* y4 = reinterpret_cast<f32x4*>(y);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long long t = i / C4;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    uint32_t where = 0;                                  // 4 x 8 bit
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int iy = oy * 2 - 1 + r;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int ix = ox * 2 - 1 + s;
        if ((unsigned)ix >= (unsigned)W) continue;
        const f32x4 v = x4[((long long)(b * H + iy) * W + ix) * C4 + c];
        if constexpr (IDX) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (v[e] > m[e]) { m[e] = v[e]; where = (where & ~(0xFFu << (8 * e))) | ((uint32_t)(r * 3 + s) << (8 * e)); }
        } else {
          m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
        }
      }
    }
    y4[i] = m;
    if constexpr (IDX) reinterpret_cast<uint32_t*>(idx)[i] = where;
  }
}

__global__ void upsample_add_kernel(const float* __restrict__ src, int B, int Hi, int Wi, int C4,
                                    const float* __restrict__ add, float* __restrict__ y, int Ho, int Wo,
                                    float sh, float sw) {
  const long long total = (long long)B * Ho * Wo * C4;
  const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
  const f32x4* a4 = reinterpret_cast<const f32x4*>(add);
  f32x4* y4 = reinterpret_cast<f32x4*>(y);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
#pragma clang fp contract(off)   // see the merge epilogue of igemm.hip: the two must agree bit for bit
    const int c = (int)(i % C4);
    long long t = i / C4;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const float fy = sh * oy, fx = sw * ox;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hi - 1 ? 1 : 0), x1 = x0 + (x0 < Wi - 1 ? 1 : 0);
    const float ly = fminf(fmaxf(fy - y0, 0.f), 1.f), lx = fminf(fmaxf(fx - x0, 0.f), 1.f);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const long long rb = (long long)b * Hi;
    const f32x4 v00 = s4[((rb + y0) * Wi + x0) * C4 + c], v01 = s4[((rb + y0) * Wi + x1) * C4 + c];
    const f32x4 v10 = s4[((rb + y1) * Wi + x0) * C4 + c], v11 = s4[((rb + y1) * Wi + x1) * C4 + c];
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = hy * (hx * v00[e] + lx * v01[e]) + ly * (hx * v10[e] + lx * v11[e]);
    if (add) { const f32x4 a = a4[i]; r[0] += a[0]; r[1] += a[1]; r[2] += a[2]; r[3] += a[3]; }
    y4[i] = r;
  }
}

// one wave per row; 4 rows per 256-thread block
constexpr int SOFTMAX_REG = 24;         // rows of up to 1536 columns (the attention maps of a 24 x 64 level) are held in registers
__global__ void softmax_rows_kernel(float* __restrict__ x, long long rows, int cols, long long ld) {
  const int lane = threadIdx.x & 63;
  for (long long row = blockIdx.x * 4ll + (threadIdx.x >> 6); row < rows; row += (long long)gridDim.x * 4) {
    float* r = x + row * ld;
    if (cols <= 64 * SOFTMAX_REG) {
      // the row stays in registers (same element -> lane mapping, same order of the sum: bit-identical): one read and one write of
      // the map instead of three reads and two writes; all loads of the lane in flight together
      float v[SOFTMAX_REG];
#pragma unroll
      for (int j = 0; j < SOFTMAX_REG; ++j) v[j] = r[min(lane + 64 * j, cols - 1)];
      float m = -INFINITY;
#pragma unroll
      for (int j = 0; j < SOFTMAX_REG; ++j)
        if (lane + 64 * j < cols) m = fmaxf(m, v[j]);
      m = nbm_wave_max(m);
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < SOFTMAX_REG; ++j)
        if (lane + 64 * j < cols) { v[j] = expf(v[j] - m); s += v[j]; }
      s = nbm_wave_sum(s);
#pragma unroll
      for (int j = 0; j < SOFTMAX_REG; ++j)
        if (lane + 64 * j < cols) r[lane + 64 * j] = v[j] / s;
      continue;
    }
    float m = -INFINITY;
    for (int c = lane; c < cols; c += 64) m = fmaxf(m, r[c]);
    m = nbm_wave_max(m);
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) { const float e = expf(r[c] - m); r[c] = e; s += e; }
    s = nbm_wave_sum(s);
    for (int c = lane; c < cols; c += 64) r[c] = r[c] / s;
  }
}

__global__ void dwconv3x3_kernel(const float* __restrict__ x, int B, int H, int W, int Cin, int mult, int stride,
                                 const float* __restrict__ w, const float* __restrict__ bias,
                                 const float* __restrict__ film, long long film_ld, float* __restrict__ y,
                                 int Ho, int Wo) {
  const int Cout = Cin * mult;
  const long long total = (long long)B * Ho * Wo * Cout;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int o = (int)(i % Cout);
    const long long pix = i / Cout;
    long long t = pix;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const int ci = o / mult;
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int iy = oy * stride - 1 + r;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int ix = ox * stride - 1 + s;
        if ((unsigned)ix >= (unsigned)W) continue;
        acc += x[((long long)(b * H + iy) * W + ix) * Cin + ci] * w[o * 9 + r * 3 + s];
      }
    }
    if (bias) acc += bias[o];
    if (film) acc = acc * film[pix * film_ld + o] + film[pix * film_ld + Cout + o];
    y[i] = acc;
  }
}

// The same depthwise convolution, four consecutive output channels per thread (MULT in {1, 2, 4}: 4 / MULT input channels, one
// 16 / 8 / 4-byte load per tap; 16-byte bias / FiLM / output accesses).  The 36 weights of the thread's channels stay in
// registers when the grid stride keeps its channel chunk fixed.  Per output element the arithmetic is the scalar kernel's:
// acc += x * w over the taps in (r, s) order, + bias, * gamma + beta.
template <int MULT>
__global__ void dwconv3x3_vec_kernel(const float* __restrict__ x, int B, int H, int W, int Cin, int stride,
                                     const float* __restrict__ w, const float* __restrict__ bias,
                                     const float* __restrict__ film, long long film_ld, float* __restrict__ y,
                                     int Ho, int Wo) {
  constexpr int NIN = 4 / MULT;
  const int Cout = Cin * MULT, C4 = Cout >> 2;
  const long long total = (long long)B * Ho * Wo * C4;
  const long long gstride = (long long)gridDim.x * blockDim.x;
  const bool fixed_chunk = (gstride % C4) == 0;
  float wr[4][9];
  int w_chunk = -1;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += gstride) {
    const int c4 = (int)(i % C4);
    const long long pix = i / C4;
    long long t = pix;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const int o0 = c4 * 4, ci0 = o0 / MULT;
    if (!fixed_chunk || w_chunk != c4) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 9; ++k) wr[j][k] = w[(o0 + j) * 9 + k];
      w_chunk = c4;
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // all nine taps in flight before the first is used (clamped address, zeroed outside the image): a bounds branch in front of
    // every load makes them go out one at a time, nine memory latencies per output vector (see dwconv_bwd_weight_vec_kernel)
    float xv[9][NIN];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int iyc = min(max(oy * stride - 1 + r, 0), H - 1);
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int ixc = min(max(ox * stride - 1 + s, 0), W - 1);
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
        for (int j = 0; j < 4; ++j) acc[j] += (ok ? xv[r * 3 + s][j / MULT] : 0.f) * wr[j][r * 3 + s];
      }
    }
    f32x4 out = {acc[0], acc[1], acc[2], acc[3]};
    if (bias) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + o0);
      out += bv;
    }
    if (film) {
      const f32x4 ga = *reinterpret_cast<const f32x4*>(film + pix * film_ld + o0);
      const f32x4 be = *reinterpret_cast<const f32x4*>(film + pix * film_ld + Cout + o0);
#pragma unroll
      for (int j = 0; j < 4; ++j) out[j] = out[j] * ga[j] + be[j];
    }
    *reinterpret_cast<f32x4*>(y + i * 4) = out;
  }
}

// one wave per row
__global__ void layernorm_kernel(const float* __restrict__ x, long long rows, int E, const float* __restrict__ w,
                                 const float* __restrict__ b, float eps, float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  for (long long row = blockIdx.x * 4ll + (threadIdx.x >> 6); row < rows; row += (long long)gridDim.x * 4) {
    const float* r = x + row * E;
    float s = 0.f;
    for (int c = lane; c < E; c += 64) s += r[c];
    const float mean = nbm_wave_sum(s) / (float)E;
    float q = 0.f;
    for (int c = lane; c < E; c += 64) { const float d = r[c] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(nbm_wave_sum(q) / (float)E + eps);
    float* o = y + row * E;
    for (int c = lane; c < E; c += 64) o[c] = (r[c] - mean) * rstd * w[c] + b[c];
  }
}

// Multi-head attention over SHORT sequences (Transformer_RCNN: S = images per batch or RoIs per image, <= 128; head
// dim <= 64).  One workgroup per (batch entry n, head); K and V of the group live in LDS, each wave owns query rows.
// Token (s, n) is row s * seq_stride + n * batch_stride of q/k/v/out; keys >= *n_valid (device counter) are masked.
#define MHA_SMAX 128
__global__ __launch_bounds__(256) void mha_small_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, int q_ld, int k_ld, int v_ld,
                                                        float* __restrict__ out, int out_ld, int S, int nhead, int hd,
                                                        long long seq_stride, long long batch_stride,
                                                        const int* __restrict__ n_valid, float scale) {
  __shared__ float Ks[MHA_SMAX][65];
  __shared__ float Vs[MHA_SMAX][64];
  __shared__ float qs[4][64];
  __shared__ float ps[4][MHA_SMAX];
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
  for (int r = wave; r < nv; r += 4) {
    const long long row = r * seq_stride + n * batch_stride;
    if (lane < hd) qs[wave][lane] = q[row * q_ld + h * hd + lane] * scale;
    __builtin_amdgcn_wave_barrier();
    float sc[MHA_SMAX / 64];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < MHA_SMAX / 64; ++t) {
      const int j = lane + 64 * t;
      float a = -INFINITY;
      if (j < nv) {
        a = 0.f;
        for (int d = 0; d < hd; ++d) a += qs[wave][d] * Ks[j][d];
      }
      sc[t] = a;
      m = fmaxf(m, a);
    }
    m = nbm_wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < MHA_SMAX / 64; ++t) {
      const int j = lane + 64 * t;
      const float e = j < nv ? expf(sc[t] - m) : 0.f;
      if (j < nv) ps[wave][j] = e;
      sum += e;
    }
    sum = nbm_wave_sum(sum);
    __builtin_amdgcn_wave_barrier();
    if (lane < hd) {
      float o = 0.f;
      for (int j = 0; j < nv; ++j) o += ps[wave][j] * Vs[j][lane];
      out[row * out_ld + h * hd + lane] = o / sum;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ void silu_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float v = x[i];
    y[i] = v / (1.0f + expf(-v));
  }
}

__global__ void pair_softmax_kernel(const float* __restrict__ x, long long n_pix, int n_anchor, int x_ld,
                                    float* __restrict__ y, int y_ld) {
  const long long total = n_pix * n_anchor;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long p = i / n_anchor;
    const int a = (int)(i - p * n_anchor);
    const float v0 = x[p * x_ld + 2 * a], v1 = x[p * x_ld + 2 * a + 1];
    const float m = fmaxf(v0, v1);
    const float e0 = expf(v0 - m), e1 = expf(v1 - m);
    const float s = e0 + e1;
    y[p * y_ld + 2 * a] = e0 / s;
    y[p * y_ld + 2 * a + 1] = e1 / s;
  }
}

}  // namespace

extern "C" const char* nbm_version(void) { return "nbm_hip 0.4 (gfx950)"; }

extern "C" int nbm_graph_census(void* graph, long long counts[6]) {
  if (!graph || !counts) return NBM_EINVAL;
  for (int i = 0; i < 6; ++i) counts[i] = 0;
  size_t n = 0;
  hipError_t e = hipGraphGetNodes((hipGraph_t)graph, nullptr, &n);
  if (e != hipSuccess) return (int)e;
  if (!n) return NBM_OK;
  hipGraphNode_t* nodes = (hipGraphNode_t*)malloc(n * sizeof(hipGraphNode_t));
  if (!nodes) return NBM_EINVAL;
  e = hipGraphGetNodes((hipGraph_t)graph, nodes, &n);
  for (size_t i = 0; e == hipSuccess && i < n; ++i) {
    hipGraphNodeType t;
    e = hipGraphNodeGetType(nodes[i], &t);
    if (e != hipSuccess) break;
    const int k = t == hipGraphNodeTypeKernel ? 0 : t == hipGraphNodeTypeMemcpy ? 1 : t == hipGraphNodeTypeMemset ? 2
                : t == hipGraphNodeTypeHost ? 3 : t == hipGraphNodeTypeEmpty ? 4 : 5;
    counts[k] += 1;
  }
  free(nodes);
  return e == hipSuccess ? NBM_OK : (int)e;
}

extern "C" int nbm_pcm16_to_wave(const int16_t* pcm, int64_t pcm_ld, int batch, int n, int upsample,
                                 const int32_t* hq, int64_t first, int64_t count, float* out, int64_t out_ld, int lead,
                                 int pad_mode, void* stream) {
  if (!pcm || !out || batch <= 0 || n <= 0 || lead < 0 || (upsample && !hq)) return NBM_EINVAL;
  const int64_t n_out = upsample ? 2 * (int64_t)n : n;
  if (first < 0 || count <= 0 || first + count > n_out) return NBM_EINVAL;
  if (pad_mode != 0 && pad_mode != 1) return NBM_EINVAL;
  if ((int64_t)lead + count + (pad_mode ? lead : 0) > out_ld) return NBM_EINVAL;
  if (pad_mode == 1 && lead >= count) return NBM_EINVAL;            // librosa raises for such short signals too
  dim3 grid(grid_for(out_ld, TPB, 1024), batch);
  hipLaunchKernelGGL(pcm16_to_wave_kernel, grid, dim3(TPB), 0, (hipStream_t)stream, pcm, (long long)pcm_ld, n,
                     upsample, hq, (long long)first, (long long)count, out, (long long)out_ld, lead, pad_mode);
  return nbm_launch_status();
}

extern "C" int nbm_resample_to_wave(const float* x, int64_t x_ld, int batch, int64_t n, int L, int M, const double* taps,
                                    int T, int64_t first, int64_t count, float* out, int64_t out_ld, int lead, int pad_mode,
                                    int quant16, void* stream) {
  if (!x || !out || batch <= 0 || n <= 0 || L <= 0 || M <= 0 || lead < 0) return NBM_EINVAL;
  const bool copy = L == 1 && M == 1;
  if (!copy && (!taps || T <= 0 || (T & 1))) return NBM_EINVAL;
  const int64_t n_out = copy ? n : (n * L + M - 1) / M;            // output samples whose centre lies inside the input
  if (first < 0 || count <= 0 || first + count > n_out) return NBM_EINVAL;
  if (pad_mode != 0 && pad_mode != 1) return NBM_EINVAL;
  if ((int64_t)lead + count + (pad_mode ? lead : 0) > out_ld) return NBM_EINVAL;
  if (pad_mode == 1 && lead >= count) return NBM_EINVAL;
  dim3 grid(grid_for(out_ld, TPB, 1024), batch);
  hipLaunchKernelGGL(resample_to_wave_kernel, grid, dim3(TPB), 0, (hipStream_t)stream, x, (long long)x_ld, (long long)n, L,
                     M, taps, T, (long long)first, (long long)count, out, (long long)out_ld, lead, pad_mode, quant16);
  return nbm_launch_status();
}

extern "C" int nbm_minmax_init(uint32_t* minmax, int batch, void* stream) {
  if (!minmax || batch <= 0) return NBM_EINVAL;
  hipLaunchKernelGGL(minmax_init_kernel, dim3((batch + 63) / 64), dim3(64), 0, (hipStream_t)stream, minmax, batch);
  return nbm_launch_status();
}

extern "C" int nbm_spec_windows(const float* db, int64_t db_bs, int db_ld, int batch, int n_bins, int n_frames,
                                const uint32_t* minmax, float* img, int n_img, int w_pix, int hop_img,
                                const int32_t* last_cols, void* stream) {
  if (!db || !minmax || !img || !last_cols || batch <= 0 || n_img <= 0 || n_frames <= 0) return NBM_EINVAL;
  if ((n_img - 1) * hop_img >= n_frames) return NBM_EINVAL;  // every window must own >= 1 real column
  if ((n_img - 2) * (long long)hop_img + w_pix > n_frames && n_img > 1) return NBM_EINVAL;  // only the last may be short
  dim3 grid(grid_for((long long)n_bins * w_pix, TPB, 512), n_img, batch);
  hipLaunchKernelGGL(spec_windows_kernel, grid, dim3(TPB), 0, (hipStream_t)stream, db, (long long)db_bs, db_ld,
                     n_bins, minmax, img, n_img, w_pix, hop_img, last_cols);
  return nbm_launch_status();
}

extern "C" int nbm_init_conv(const float* x, int64_t n_pix, const float* w, const float* b, int C, float* y,
                             void* stream) {
  if (!x || !w || !b || !y || n_pix <= 0 || C <= 0) return NBM_EINVAL;
  hipLaunchKernelGGL(init_conv_kernel, dim3(grid_for(n_pix * C)), dim3(TPB), 0, (hipStream_t)stream, x,
                     (long long)n_pix, w, b, C, y);
  return nbm_launch_status();
}

extern "C" int nbm_maxpool3x3s2(const float* x, int B, int H, int W, int C, float* y, int Ho, int Wo, uint8_t* idx,
                                void* stream) {
  if (!x || !y || B <= 0 || C <= 0 || (C & 3)) return NBM_EINVAL;
  if ((H + 2 - 3) / 2 + 1 != Ho || (W + 2 - 3) / 2 + 1 != Wo) return NBM_EINVAL;
  if (!nbm_aligned16(x) || !nbm_aligned16(y) || (((uintptr_t)idx) & 3u)) return NBM_EALIGN;
  if (idx)
    hipLaunchKernelGGL(maxpool3x3s2_kernel<true>, dim3(grid_for((long long)B * Ho * Wo * (C / 4))), dim3(TPB), 0,
                       (hipStream_t)stream, x, B, H, W, C / 4, y, Ho, Wo, idx);
  else
    hipLaunchKernelGGL(maxpool3x3s2_kernel<false>, dim3(grid_for((long long)B * Ho * Wo * (C / 4))), dim3(TPB), 0,
                       (hipStream_t)stream, x, B, H, W, C / 4, y, Ho, Wo, idx);
  return nbm_launch_status();
}

// ---- `--dilation` (reference backbone.py:129-131: torchvision's replace_stride_with_dilation for layer4).  A 3x3 / dilation-2 / pad-2
// convolution on an even-sized map IS four ordinary 3x3 / pad-1 convolutions on the four parity classes of its pixels, and everything
// else in a bottleneck is point-wise: the dilated blocks run the ordinary kernels on the space-to-batch form of the map.
//   forward (inverse = 0): y[(2a+b)*B + n][u][v][c] = x[n][2u+a][2v+b][c]   (x [B][H][W][C], y [4B][H/2][W/2][C])
//   inverse = 1: the same index map read the other way (x is the [4B][H/2][W/2][C] side, y the [B][H][W][C] side)
__global__ void space_to_batch2_kernel(const float* __restrict__ x, int B, int H, int W, int C4, float* __restrict__ y, int inverse) {
  const int Hh = H >> 1, Wh = W >> 1;
  const long long total = (long long)B * H * W * C4;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  f32x4* y4 = reinterpret_cast<f32x4*>(y);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long long t = i / C4;                               // dense pixel index (n, yy, xx)
    const int xx = (int)(t % W); t /= W;
    const int yy = (int)(t % H);
    const int n = (int)(t / H);
    const long long packed = ((((long long)((yy & 1) * 2 + (xx & 1)) * B + n) * Hh + (yy >> 1)) * Wh + (xx >> 1)) * C4 + c;
    if (inverse) y4[i] = x4[packed]; else y4[packed] = x4[i];
  }
}

// AdaptiveAvgPool2d to exactly half the size (layers.py:84,94 with the RPN map of the dilated level: 48x128 -> 24x64): the mean of each
// 2x2 block, summed in torch's CPU order (row by row), divided by 4
__global__ void avgpool2x2_kernel(const float* __restrict__ x, int B, int Ho, int Wo, int C4, float* __restrict__ y) {
  const long long total = (long long)B * Ho * Wo * C4;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  f32x4* y4 = reinterpret_cast<f32x4*>(y);
  const int W = 2 * Wo;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long long t = i / C4;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const long long r0 = (((long long)b * 2 * Ho + 2 * oy) * W + 2 * ox) * C4 + c;
    const f32x4 a = x4[r0], b1 = x4[r0 + C4], c1 = x4[r0 + (long long)W * C4], d = x4[r0 + (long long)W * C4 + C4];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (((a[e] + b1[e]) + c1[e]) + d[e]) * 0.25f;
    y4[i] = o;
  }
}

extern "C" int nbm_space_to_batch2(const float* x, int B, int H, int W, int C, float* y, int inverse, void* stream) {
  if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || (H & 1) || (W & 1)) return NBM_EINVAL;
  if (!nbm_aligned16(x) || !nbm_aligned16(y)) return NBM_EALIGN;
  hipLaunchKernelGGL(space_to_batch2_kernel, dim3(grid_for((long long)B * H * W * (C / 4))), dim3(TPB), 0, (hipStream_t)stream, x, B, H, W,
                     C / 4, y, inverse ? 1 : 0);
  return nbm_launch_status();
}

extern "C" int nbm_avgpool2x2(const float* x, int B, int Ho, int Wo, int C, float* y, void* stream) {
  if (!x || !y || B <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || (C & 3)) return NBM_EINVAL;
  if (!nbm_aligned16(x) || !nbm_aligned16(y)) return NBM_EALIGN;
  hipLaunchKernelGGL(avgpool2x2_kernel, dim3(grid_for((long long)B * Ho * Wo * (C / 4))), dim3(TPB), 0, (hipStream_t)stream, x, B, Ho, Wo,
                     C / 4, y);
  return nbm_launch_status();
}

extern "C" int nbm_upsample_bilinear_add(const float* src, int B, int Hi, int Wi, int C, const float* add, float* y,
                                         int Ho, int Wo, void* stream) {
  if (!src || !y || B <= 0 || C <= 0 || (C & 3) || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return NBM_EINVAL;
  if (!nbm_aligned16(src) || !nbm_aligned16(y) || (add && !nbm_aligned16(add))) return NBM_EALIGN;
  const float sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
  const float sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  hipLaunchKernelGGL(upsample_add_kernel, dim3(grid_for((long long)B * Ho * Wo * (C / 4))), dim3(TPB), 0,
                     (hipStream_t)stream, src, B, Hi, Wi, C / 4, add, y, Ho, Wo, sh, sw);
  return nbm_launch_status();
}

extern "C" int nbm_softmax_rows(float* x, int64_t rows, int cols, int64_t ld, void* stream) {
  if (!x || rows <= 0 || cols <= 0 || ld < cols) return NBM_EINVAL;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(grid_for(rows, 4, 256 * 32)), dim3(256), 0, (hipStream_t)stream, x,
                     (long long)rows, cols, (long long)ld);
  return nbm_launch_status();
}

extern "C" int nbm_dwconv3x3(const float* x, int B, int H, int W, int Cin, int mult, int stride, const float* w,
                             const float* bias, const float* film, int64_t film_ld, float* y, int Ho, int Wo,
                             void* stream) {
  if (!x || !w || !y || B <= 0 || Cin <= 0 || mult <= 0 || stride <= 0) return NBM_EINVAL;
  if ((H + 2 - 3) / stride + 1 != Ho || (W + 2 - 3) / stride + 1 != Wo) return NBM_EINVAL;
  if (film && film_ld < 2ll * Cin * mult) return NBM_EINVAL;
  const int Cout = Cin * mult;
  const bool vec = (mult == 1 || mult == 2 || mult == 4) && (Cout & 3) == 0 && (Cin % (4 / mult)) == 0 && nbm_aligned16(y) &&
                   nbm_aligned16(x) && (!bias || nbm_aligned16(bias)) && (!film || (nbm_aligned16(film) && (film_ld & 3) == 0));
  if (vec) {
    const dim3 grid(grid_for((long long)B * Ho * Wo * (Cout / 4))), blk(TPB);
    hipStream_t st = (hipStream_t)stream;
    if (mult == 1) hipLaunchKernelGGL(dwconv3x3_vec_kernel<1>, grid, blk, 0, st, x, B, H, W, Cin, stride, w, bias, film, (long long)film_ld, y, Ho, Wo);
    else if (mult == 2) hipLaunchKernelGGL(dwconv3x3_vec_kernel<2>, grid, blk, 0, st, x, B, H, W, Cin, stride, w, bias, film, (long long)film_ld, y, Ho, Wo);
    else hipLaunchKernelGGL(dwconv3x3_vec_kernel<4>, grid, blk, 0, st, x, B, H, W, Cin, stride, w, bias, film, (long long)film_ld, y, Ho, Wo);
    return nbm_launch_status();
  }
  hipLaunchKernelGGL(dwconv3x3_kernel, dim3(grid_for((long long)B * Ho * Wo * Cin * mult)), dim3(TPB), 0,
                     (hipStream_t)stream, x, B, H, W, Cin, mult, stride, w, bias, film, (long long)film_ld, y, Ho, Wo);
  return nbm_launch_status();
}

extern "C" int nbm_layernorm(const float* x, int64_t rows, int E, const float* w, const float* b, float eps, float* y,
                             void* stream) {
  if (!x || !w || !b || !y || rows <= 0 || E <= 0) return NBM_EINVAL;
  hipLaunchKernelGGL(layernorm_kernel, dim3(grid_for(rows, 4, 256 * 32)), dim3(256), 0, (hipStream_t)stream, x,
                     (long long)rows, E, w, b, eps, y);
  return nbm_launch_status();
}

extern "C" int nbm_mha_small(const float* q, const float* k, const float* v, int q_ld, int k_ld, int v_ld, float* out,
                             int out_ld, int S, int N, int nhead, int hd, int64_t seq_stride, int64_t batch_stride,
                             const int32_t* n_valid, float scale, void* stream) {
  if (!q || !k || !v || !out || S <= 0 || S > MHA_SMAX || N <= 0 || nhead <= 0 || hd <= 0 || hd > 64) return NBM_EINVAL;
  if (q_ld < nhead * hd || k_ld < nhead * hd || v_ld < nhead * hd || out_ld < nhead * hd) return NBM_EINVAL;
  hipLaunchKernelGGL(mha_small_kernel, dim3(N * nhead), dim3(256), 0, (hipStream_t)stream, q, k, v, q_ld, k_ld, v_ld, out,
                     out_ld, S, nhead, hd, (long long)seq_stride, (long long)batch_stride, n_valid, scale);
  return nbm_launch_status();
}

extern "C" int nbm_silu(const float* x, float* y, int64_t n, void* stream) {
  if (!x || !y || n <= 0) return NBM_EINVAL;
  hipLaunchKernelGGL(silu_kernel, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream, x, y, (long long)n);
  return nbm_launch_status();
}

extern "C" int nbm_pair_softmax(const float* x, int64_t n_pix, int n_anchor, int x_ld, float* y, int y_ld,
                                void* stream) {
  if (!x || !y || n_pix <= 0 || n_anchor <= 0 || x_ld < 2 * n_anchor || y_ld < 2 * n_anchor) return NBM_EINVAL;
  hipLaunchKernelGGL(pair_softmax_kernel, dim3(grid_for(n_pix * n_anchor)), dim3(TPB), 0, (hipStream_t)stream, x,
                     (long long)n_pix, n_anchor, x_ld, y, y_ld);
  return nbm_launch_status();
}
