// Network stem: init_conv (1x1, 1 -> 3 channels, + bias; reference backbone.py:104-105,110-113) folded into torchvision's
// conv1 (7x7 / stride 2 / pad 3, 3 -> 64) + FrozenBN + ReLU -- one kernel on the single-channel spectrogram image.
//
//   u_c = a_c x + b_c inside the image, 0 in conv1's zero padding
//   z_o(p) = sum_{r,s} (sum_c W[o][c][r][s] a_c) x(p + (r,s)) + sum_{(r,s) inside} (sum_c W[o][c][r][s] b_c)
//          = sum_k  weff[o][k] x_k(p)                           +  bias_o(p)
//   y_o = relu(z_o scale_o + shift_o)
//
// The generic implicit-GEMM path ran this layer on the 3-channel map with K = 147 padded to 160 through per-element
// gathers: 2.8 ms at B = 64 (42 TF/s), plus 0.13 ms for init_conv and 0.6 GB of 3-channel traffic.  Here K = 7 rows x 8
// (7 taps + one zero) = 56: a workgroup owns 128 output pixels of one output row; their 7 x 261 input samples are ONE
// 7.4 KB LDS image, and lane (pixel p, k-slot) reads its A operand of k-pair j straight from it at a compile-time offset
// (r * pitch + s0) + 2 p + slot -- no im2col anywhere.  The folded weights [56][64] sit in LDS as the B operand.  bias_o(p)
// is a per-channel constant for interior pixels and a masked 49-term sum on the three-pixel border.
#include "nbm_common.h"

namespace {

constexpr int SP = 264;        // LDS pitch of an image row: 2 * 128 + 5 samples, padded
constexpr int KR = 8;          // k slots per filter row (7 taps + 1 zero)
constexpr int KK = 7 * KR;     // 56
constexpr int OP = 68;         // LDS pitch of an output pixel (64 channels + 4)
static_assert(7 * SP + KK * 64 <= 128 * OP, "operands must fit the output tile");

struct StemParams {
  const float* img; const float* weff; const float* wb; const float* wb_full; const float* scale; const float* shift;
  float* y;
  int B, H, W, Ho, Wo;
};

__global__ __launch_bounds__(256) void stem7x7_kernel(const StemParams p) {
  // one LDS block: image rows + folded weights during the MFMA loop, then the [128 pixels][64 channels] output tile
  __shared__ __attribute__((aligned(16))) float smem[128 * OP];
  float* img_s = smem;
  float* w_s = smem + 7 * SP;
  __shared__ float cb_s[64 * 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ox0 = blockIdx.x * 128, oy = blockIdx.y, b = blockIdx.z;

  // ---- stage: 7 input rows x 261 columns (zeros outside the image), folded weights
  const int ix0 = 2 * ox0 - 3;
  // (all loads of a thread go out before the first LDS write: clamped addresses, zeros selected afterwards -- with a bounds branch
  // around each load they were eight dependent round trips in front of 56 MFMAs)
  constexpr int NST = (7 * SP + 255) / 256;
  float stv[NST];
#pragma unroll
  for (int k = 0; k < NST; ++k) {
    const int i = min(tid + 256 * k, 7 * SP - 1);
    const int r = i / SP, c = i - r * SP;
    const int iy = min(max(2 * oy - 3 + r, 0), p.H - 1), ix = min(max(ix0 + c, 0), p.W - 1);
    stv[k] = p.img[((long long)b * p.H + iy) * p.W + ix];
  }
#pragma unroll
  for (int k = 0; k < NST; ++k) {
    const int i = tid + 256 * k;
    const int r = i / SP, c = i - r * SP;
    const int iy = 2 * oy - 3 + r, ix = ix0 + c;
    if (i < 7 * SP) img_s[i] = ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && c < 261) ? stv[k] : 0.f;
  }
  for (int i = tid; i < KK * 64 / 4; i += 256)
    reinterpret_cast<f32x4*>(w_s)[i] = reinterpret_cast<const f32x4*>(p.weff)[i];
  // init_conv bias seen through the filter rows inside the image: cb_s[n][s] = sum_{r inside} wb[n][r][s] (only workgroups
  // on the image border read it)
  unsigned rmask = 0;
#pragma unroll
  for (int r = 0; r < 7; ++r)
    if ((unsigned)(2 * oy - 3 + r) < (unsigned)p.H) rmask |= 1u << r;
  const bool row_border = rmask != 0x7fu;
  const bool col_border = ox0 < 2 || 2 * (ox0 + 127) + 3 >= p.W;
  if (row_border || col_border)
    for (int i = tid; i < 64 * 7; i += 256) {
      const int n = i / 7, sx = i - n * 7;
      float a = 0.f;
      for (int r = 0; r < 7; ++r)
        if ((rmask >> r) & 1u) a += p.wb[n * 49 + r * 7 + sx];
      cb_s[n * 8 + sx] = a;
    }
  __syncthreads();

  // ---- 28 k-pairs; wave tile = 32 pixels x 64 channels (two 32x32 accumulators)
  const int pl = lane & 31, slot = lane >> 5;
  const float* a_base = img_s + 2 * (wave * 32 + pl) + slot;
  const float* b_base = w_s + slot * 64 + pl;
  f32x16 acc0, acc1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
  for (int j = 0; j < KK / 2; ++j) {
    constexpr int dummy = 0; (void)dummy;
    const int r = (2 * j) / KR, s0 = (2 * j) % KR;                  // compile-time after unrolling
    const float a = a_base[r * SP + s0];                            // slot 1 of the zero column (s = 7) has weight 0
    const float b0 = b_base[(2 * j) * 64], b1 = b_base[(2 * j) * 64 + 32];
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
  }

  // ---- epilogue.  C/D layout: col (channel) = lane & 31, row (pixel) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
  // Row / column taps inside the image: bit r of rmask, bit s of cmask.  Values go through LDS so that every lane stores
  // 4 channels (16 bytes) and a wave-instruction covers 4 whole pixels = 1 KB contiguous.
  __syncthreads();                                                  // every wave is done with img_s / w_s
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int n = 32 * t + pl;
    const float sc = p.scale[n], sh = p.shift[n];
    float bfull = p.wb_full[n];
    if (row_border) {
      bfull = 0.f;
#pragma unroll
      for (int s = 0; s < 7; ++s) bfull += cb_s[n * 8 + s];
    }
    const f32x16& acc = t == 0 ? acc0 : acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int px = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * slot;
      const int ox = ox0 + px;
      float bias = bfull;
      if (col_border && (ox < 2 || 2 * ox + 3 >= p.W)) {            // only the columns inside the image
        bias = 0.f;
#pragma unroll
        for (int s = 0; s < 7; ++s)
          if ((unsigned)(2 * ox - 3 + s) < (unsigned)p.W) bias += cb_s[n * 8 + s];
      }
      smem[px * OP + n] = fmaxf((acc[e] + bias) * sc + sh, 0.f);
    }
  }
  __syncthreads();
  float* yrow = p.y + (((long long)b * p.Ho + oy) * p.Wo + ox0) * 64;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int px = it * 16 + (tid >> 4), c4 = (tid & 15) * 4;
    if (ox0 + px < p.Wo)
      *reinterpret_cast<f32x4*>(yrow + px * 64 + c4) = *reinterpret_cast<const f32x4*>(smem + px * OP + c4);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight-side gradient of the stem.  With U[n][t] = sum_px g[px][n] x[pix(px) + t] and V[n][t] = sum_px g[px][n] [pix(px) + t
// inside the image] (t = 7x7 tap) the three gradients of the stem follow on the host side (functional.Stem.backward):
//   dW1[n][c][t] = a_c U + b_c V,   da_c = sum W1 U,   db_c = sum W1 V.
// The generic weight-gradient kernel gathered the 2-channel image (x, 1) tap by tap: 4.8 ms at B = 128 (0.66 TB/s on the
// 3.15 GB of g).  Here a persistent workgroup walks 128-pixel strips of output rows: the strip's 7 x 261 input samples and its
// [128 px][64 ch] gradient tile sit in LDS, and D[n][k] += sum_px g[px][n] X[px][k] is an MFMA GEMM with M = 64 channels,
// N = 64 (56 tap slots k = 8 r + s, slot 56 = a column of ones: the plain column sum S[n] of g), K = 128 pixels; wave w owns
// the 32 x 32 tile (w / 2, w % 2).  V = S - C with C[n][t] = sum over the pixels whose tap t falls outside the image (the
// three-pixel border only): stem_vborder_kernel.  One atomic flush per workgroup.
struct StemWgradParams {
  const float* img; const float* g; float* D;          // D [64][64] accumulators: columns 0..55 = U (k = 8 r + s), 56 = S
  int B, H, W, Ho, Wo, n_strips, tiles_x;
};

__global__ __launch_bounds__(256) void stem_wgrad_kernel(const StemWgradParams p) {
  __shared__ __attribute__((aligned(16))) float img_s[7 * SP];
  __shared__ __attribute__((aligned(16))) float g_s[128 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mt = wave >> 1, nt = wave & 1;
  const int pl = lane & 31, slot = lane >> 5;
  const int col = nt * 32 + pl;                          // tap slot of this lane's B operand
  const int r = col >> 3, sx = col & 7;
  const bool tap = col < 56 && sx < 7;
  const float ones = col == 56 ? 1.f : 0.f;
  const float* b_base = img_s + (tap ? r * SP + sx : 0) + 2 * slot;
  const float* a_base = g_s + slot * 64 + mt * 32 + pl;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  for (int strip = blockIdx.x; strip < p.n_strips; strip += gridDim.x) {
    const int tx = strip % p.tiles_x, rowi = strip / p.tiles_x;
    const int oy = rowi % p.Ho, b = rowi / p.Ho;
    const int ox0 = tx * 128, ix0 = 2 * ox0 - 3;
    __syncthreads();                                     // the previous strip's operands are consumed
    // all 16 global loads of a thread in flight together (clamped addresses, zeros selected afterwards), as in stem7x7_kernel
    constexpr int NST = (7 * SP + 255) / 256;
    float stv[NST];
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int i = min(tid + 256 * k, 7 * SP - 1);
      const int rr = i / SP, c = i - rr * SP;
      const int iy = min(max(2 * oy - 3 + rr, 0), p.H - 1), ix = min(max(ix0 + c, 0), p.W - 1);
      stv[k] = p.img[((long long)b * p.H + iy) * p.W + ix];
    }
    const float* gsrc = p.g + (((long long)b * p.Ho + oy) * p.Wo + ox0) * 64;
    const int npx = min(128, p.Wo - ox0);
    f32x4 gv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = tid + 256 * k;
      gv[k] = *reinterpret_cast<const f32x4*>(gsrc + (long long)(min(i >> 4, npx - 1) * 16 + (i & 15)) * 4);
    }
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int i = tid + 256 * k;
      const int rr = i / SP, c = i - rr * SP;
      const int iy = 2 * oy - 3 + rr, ix = ix0 + c;
      if (i < 7 * SP) img_s[i] = ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && c < 261) ? stv[k] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = tid + 256 * k;
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      reinterpret_cast<f32x4*>(g_s)[i] = (i >> 4) < npx ? gv[k] : zero4;
    }
    __syncthreads();
#pragma unroll 8
    for (int j = 0; j < 64; ++j) {                       // k-pair j: pixels 2 j + slot
      const float a = a_base[(2 * j) * 64];
      const float bv = tap ? b_base[4 * j] : ones;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
    }
  }
  // C/D layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int n = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * slot;
    atomicAdd(p.D + n * 64 + col, acc[e]);
  }
}

// C[n][r * 7 + s] = sum over the output pixels whose tap (r, s) falls outside the image of g[px][n].  Only pixels of the rows
// oy < top / oy >= bot0 and of the columns ox < left / ox >= right0 can have such taps: they are enumerated (all columns of the
// border rows, then the border columns of the other rows); thread = (channel, one of 4 pixel streams), 49 running sums in
// registers, one LDS reduction and 64 x 49 atomics per workgroup.
__global__ __launch_bounds__(256) void stem_vborder_kernel(const float* __restrict__ g, int H, int W, int Ho, int Wo, int top,
                                                           int bot0, int left, int right0, float* __restrict__ Cb) {
  __shared__ float red[4][64][49];
  const int n = threadIdx.x & 63, sub = threadIdx.x >> 6, b = blockIdx.y;
  const int nrb = top + (Ho - bot0), ncb = left + (Wo - right0), nmid = bot0 - top;
  const int nitems = nrb * Wo + ncb * nmid;
  float acc[49];
#pragma unroll
  for (int e = 0; e < 49; ++e) acc[e] = 0.f;
  for (int q = blockIdx.x * 4 + sub; q < nitems; q += gridDim.x * 4) {
    int oy, ox;
    if (q < nrb * Wo) {
      const int k = q / Wo;
      ox = q - k * Wo;
      oy = k < top ? k : bot0 + (k - top);
    } else {
      const int q2 = q - nrb * Wo, k = q2 / nmid;
      oy = top + (q2 - k * nmid);
      ox = k < left ? k : right0 + (k - left);
    }
    const float gv = g[(((long long)b * Ho + oy) * Wo + ox) * 64 + n];
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      const bool rin = (unsigned)(2 * oy - 3 + r) < (unsigned)H;
#pragma unroll
      for (int s2 = 0; s2 < 7; ++s2) {
        const bool cin = (unsigned)(2 * ox - 3 + s2) < (unsigned)W;
        acc[r * 7 + s2] += (rin && cin) ? 0.f : gv;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 49; ++e) red[sub][n][e] = acc[e];
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 49; i += 256) {
    const int nn = i / 49, e = i - nn * 49;
    const float v = red[0][nn][e] + red[1][nn][e] + red[2][nn][e] + red[3][nn][e];
    if (v != 0.f) atomicAdd(Cb + i, v);
  }
}

}  // namespace

// U / S accumulators and the border correction of V -- see nbm_hip.h.
extern "C" int nbm_stem7x7_wgrad(const float* img, const float* g, int B, int H, int W, float* D, float* Cb, void* stream) {
  if (!img || !g || !D || !Cb || B <= 0 || H <= 0 || W <= 0) return NBM_EINVAL;
  if (!nbm_aligned16(g)) return NBM_EALIGN;
  StemWgradParams p{};
  p.img = img; p.g = g; p.D = D; p.B = B; p.H = H; p.W = W;
  p.Ho = (H + 6 - 7) / 2 + 1; p.Wo = (W + 6 - 7) / 2 + 1;
  p.tiles_x = (p.Wo + 127) / 128;
  const long long strips = (long long)B * p.Ho * p.tiles_x;
  if (strips > 0x7fffffffll) return NBM_EUNSUPPORTED;
  p.n_strips = (int)strips;
  hipStream_t st = (hipStream_t)stream;
  if (nbm_zero_async(D, sizeof(float) * 64 * 64, st) != hipSuccess || nbm_zero_async(Cb, sizeof(float) * 64 * 49, st) != hipSuccess)
    return (int)hipGetLastError();
  const int grid = (int)(strips < 1024 ? strips : 1024);
  hipLaunchKernelGGL(stem_wgrad_kernel, dim3(grid), dim3(256), 0, st, p);
  // border rows / columns: taps reach outside for oy < 2 or 2 oy + 3 >= H (same for columns)
  int top = p.Ho < 2 ? p.Ho : 2, bot0 = top, left = p.Wo < 2 ? p.Wo : 2, right0 = left;
  while (bot0 < p.Ho && 2 * bot0 + 3 < H) ++bot0;
  while (right0 < p.Wo && 2 * right0 + 3 < W) ++right0;
  hipLaunchKernelGGL(stem_vborder_kernel, dim3(8, B), dim3(256), 0, st, g, H, W, p.Ho, p.Wo, top, bot0, left, right0, Cb);
  return nbm_launch_status();
}

// y = relu(bn(conv1(init_conv(img)))) -- see nbm_hip.h.
extern "C" int nbm_stem7x7(const float* img, int B, int H, int W, const float* weff, const float* wb, const float* wb_full,
                           const float* scale, const float* shift, float* y, void* stream) {
  if (!img || !weff || !wb || !wb_full || !scale || !shift || !y || B <= 0 || H <= 0 || W <= 0) return NBM_EINVAL;
  if (!nbm_aligned16(weff)) return NBM_EALIGN;
  StemParams p{};
  p.img = img; p.weff = weff; p.wb = wb; p.wb_full = wb_full; p.scale = scale; p.shift = shift; p.y = y;
  p.B = B; p.H = H; p.W = W; p.Ho = (H + 6 - 7) / 2 + 1; p.Wo = (W + 6 - 7) / 2 + 1;
  dim3 grid((p.Wo + 127) / 128, p.Ho, B);
  hipLaunchKernelGGL(stem7x7_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
  return nbm_launch_status();
}
