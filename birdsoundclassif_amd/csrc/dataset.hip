// Training-input stage on the device: PNG scanline reconstruction, per-image statistics and the
// `Img_dataset` augmentation of reference nbm_datasets/image_dataset.py:36-96 for a whole batch.  HBM-bound byte /
// element-wise work: each kernel reads its inputs once and writes its outputs once.
#include "nbm_common.h"

#pragma clang fp contract(off)   // torch evaluates `a + c * b` as separate fp32 mul and add; keep the same roundings

namespace {

// ---------------------------------------------------------------------------------------------------------------
// PNG filter reconstruction (PNG spec 9.2) for 8-bit greyscale.  Pixel (r, c) needs (r, c-1), (r-1, c), (r-1, c-1):
// an anti-diagonal wavefront.  One workgroup per image, thread r owns scanline r and works on column t - r at step t;
// the row above publishes its newest pixel through a parity-double-buffered LDS slot, so there is ONE barrier per step.
__global__ __launch_bounds__(1024) void png_unfilter_kernel(const uint8_t* __restrict__ raw, long long raw_bs,
                                                           int H, int W, uint8_t* __restrict__ out,
                                                           long long out_bs, int* __restrict__ status) {
  __shared__ int newest[2][1024];
  const int r = threadIdx.x;
  const uint8_t* line = raw + blockIdx.x * raw_bs + (long long)r * (W + 1);
  uint8_t* dst = out + blockIdx.x * out_bs + (long long)r * W;
  const bool live = r < H;
  const int ft = live ? line[0] : 0;
  if (live && ft > 4) atomicExch(status, 1 + r);
  int a = 0, ul = 0;                                  // left pixel, up-left pixel
  const int steps = W + H - 1;
  for (int t = 0; t < steps; ++t) {
    const int c = t - r;
    if (live && c >= 0 && c < W) {
      const int b = r > 0 ? newest[(t - 1) & 1][r - 1] : 0;
      const int f = line[1 + c];
      int p;
      switch (ft) {
        case 1: p = a; break;
        case 2: p = b; break;
        case 3: p = (a + b) >> 1; break;
        case 4: {
          const int q = a + b - ul;
          const int pa = abs(q - a), pb = abs(q - b), pc = abs(q - ul);
          p = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : ul);
          break;
        }
        default: p = 0;
      }
      a = (f + p) & 0xFF;
      ul = b;
      newest[t & 1][r] = a;
      dst[c] = (uint8_t)a;
    }
    __syncthreads();
  }
}

// Per-image unbiased standard deviation of fp32(u8 / 255) -- `img.std()` of image_dataset.py:66 -- from a 256-bin
// histogram (exact in float64), written as noise scale = std / 2.
__global__ __launch_bounds__(256) void image_std_kernel(const uint8_t* __restrict__ img, long long bs, long long n,
                                                        float* __restrict__ half_std) {
  __shared__ unsigned hist[256];
  __shared__ double red[256];
  hist[threadIdx.x] = 0;
  __syncthreads();
  const uint8_t* p = img + blockIdx.x * bs;
  for (long long i = threadIdx.x; i < n; i += 256) atomicAdd(&hist[p[i]], 1u);
  __syncthreads();
  const double v = (double)(float)((double)threadIdx.x / 255.0);
  const double cnt = (double)hist[threadIdx.x];
  red[threadIdx.x] = cnt * v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const double mean = red[0] / (double)n;
  __syncthreads();
  red[threadIdx.x] = cnt * (v - mean) * (v - mean);
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float sd = (float)sqrt(red[0] / (double)(n - 1));      // torch returns the fp32 std; .item()/2 on the host
    half_std[blockIdx.x] = (float)((double)sd / 2.0);
  }
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// Counter-based N(0,1): element i of stream `seed` (Box-Muller on two 24-bit uniforms of one splitmix64 draw).
__device__ __forceinline__ float counter_randn(uint64_t seed_mixed, uint64_t i) {
  const uint64_t h = splitmix64(seed_mixed + i);
  const float u1 = ((float)(uint32_t)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = ((float)(uint32_t)((h >> 16) & 0xFFFFFF) + 0.5f) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

__global__ void randn_fill_kernel(uint64_t seed, long long n, float* __restrict__ out) {
  const uint64_t sm = splitmix64(seed);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = counter_randn(sm, (uint64_t)i);
}

struct AugParams {            // mirrors struct nbm_augment_params
  float gain, coef, denom, neg_coef, neg_denom;
  int32_t flags;              // bit 0: hard-negative mix, bit 1: low-pass curve
  int32_t hard_index;         // row of `hard` to mix in
  int32_t pad;
  uint64_t noise_seed;
};

__device__ __forceinline__ float u8_to_unit(uint8_t v) { return (float)((double)v / 255.0); }

__global__ __launch_bounds__(256) void augment_kernel(const uint8_t* __restrict__ pos, const uint8_t* __restrict__ neg,
                                                      const uint8_t* __restrict__ hard, int H, int W,
                                                      const AugParams* __restrict__ prm,
                                                      const float* __restrict__ half_std,
                                                      const float* __restrict__ noise_unit,
                                                      const float* __restrict__ curve, float* __restrict__ img_out,
                                                      float* __restrict__ neg_out) {
  const int b = blockIdx.y;
  const AugParams P = prm[b];
  const long long n = (long long)H * W;
  const uint8_t* ps = pos + b * n;
  const uint8_t* ns = neg + b * n;
  const uint8_t* hs = (P.flags & 1) ? hard + (long long)P.hard_index * n : nullptr;
  const float scale = half_std[b];
  const uint64_t sm = splitmix64(P.noise_seed);
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float x = u8_to_unit(ps[i]);
    float g = u8_to_unit(ns[i]);
    const float z = noise_unit ? noise_unit[b * n + i] : counter_randn(sm, (uint64_t)i);
    const float nz = fminf(fmaxf(z * scale, -0.5f), 0.5f);
    x = x + P.gain;
    x = x + nz;
    if (hs) {
      const float h = u8_to_unit(hs[i]);
      x = (x + P.coef * h) / P.denom;
      g = (g + P.neg_coef * h) / P.neg_denom;
    }
    if (P.flags & 2) x = x + curve[b * (long long)H + (int)(i / W)];
    img_out[b * n + i] = x;
    neg_out[b * n + i] = g;
  }
}

__global__ void u8_to_float_kernel(const uint8_t* __restrict__ in, long long n, float* __restrict__ out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = u8_to_unit(in[i]);
}

}  // namespace

extern "C" int nbm_png_unfilter_gray8(const uint8_t* raw, int64_t raw_bs, int batch, int H, int W, uint8_t* out,
                                      int64_t out_bs, int32_t* status, void* stream) {
  if (!raw || !out || !status || batch <= 0 || H <= 0 || H > 1024 || W <= 0) return NBM_EINVAL;
  if (raw_bs < (int64_t)H * (W + 1) || out_bs < (int64_t)H * W) return NBM_EINVAL;
  const int threads = ((H + 63) / 64) * 64;
  hipLaunchKernelGGL(png_unfilter_kernel, dim3(batch), dim3(threads), 0, (hipStream_t)stream, raw,
                     (long long)raw_bs, H, W, out, (long long)out_bs, status);
  return nbm_launch_status();
}

extern "C" int nbm_image_half_std_u8(const uint8_t* img, int64_t bs, int batch, int64_t n, float* half_std,
                                     void* stream) {
  if (!img || !half_std || batch <= 0 || n < 2 || bs < n) return NBM_EINVAL;
  hipLaunchKernelGGL(image_std_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, img, (long long)bs,
                     (long long)n, half_std);
  return nbm_launch_status();
}

extern "C" int nbm_randn_fill(uint64_t seed, int64_t n, float* out, void* stream) {
  if (!out || n <= 0) return NBM_EINVAL;
  const long long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(randn_fill_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0,
                     (hipStream_t)stream, seed, (long long)n, out);
  return nbm_launch_status();
}

extern "C" int nbm_augment_batch(const uint8_t* pos, const uint8_t* neg, const uint8_t* hard, int batch, int H, int W,
                                 const struct nbm_augment_params* params, const float* half_std,
                                 const float* noise_unit, const float* curve, float* img_out, float* neg_out,
                                 void* stream) {
  static_assert(sizeof(AugParams) == sizeof(struct nbm_augment_params), "params layout");
  if (!pos || !neg || !params || !half_std || !curve || !img_out || !neg_out || batch <= 0 || H <= 0 || W <= 0)
    return NBM_EINVAL;
  const long long n = (long long)H * W;
  const long long blocks = (n + 255) / 256;
  dim3 grid((unsigned)(blocks < 512 ? blocks : 512), batch);
  hipLaunchKernelGGL(augment_kernel, grid, dim3(256), 0, (hipStream_t)stream, pos, neg, hard, H, W,
                     (const AugParams*)params, half_std, noise_unit, curve, img_out, neg_out);
  return nbm_launch_status();
}

extern "C" int nbm_u8_to_unit(const uint8_t* in, int64_t n, float* out, void* stream) {
  if (!in || !out || n <= 0) return NBM_EINVAL;
  const long long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(u8_to_float_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0,
                     (hipStream_t)stream, in, (long long)n, out);
  return nbm_launch_status();
}
