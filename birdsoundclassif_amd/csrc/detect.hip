// Proposal / RoI / detection stages of the NBM detector on device: box decoding, top-N selection,
// greedy NMS, RoI pooling (+ separable positional encoding) and the FastRCNN eval post-processing.
// Integer / sort / index work: results are bit-exact against the oracle for identical inputs, so
// floating-point contraction is disabled wherever a value is rounded to a pixel coordinate or compared
// with a threshold (torch CPU evaluates a*b+c as two rounded operations).
#include "nbm_common.h"

#pragma clang fp contract(off)

namespace {

// bbox_reg_to_coord (reference nets_utils.py:169-186) + clip (layers.py:279-280)
__device__ __forceinline__ void decode_clip(const float* d, const float* a, int img_w, int img_h, float* o) {
  const float wa = (a[2] - a[0]) + 1.0f, ha = (a[3] - a[1]) + 1.0f;
  const float xa = a[0] + 0.5f * wa, ya = a[1] + 0.5f * ha;
  const float x = (d[0] * wa) + xa, y = (d[1] * ha) + ya;
  const float w = expf(d[2]) * wa, h = expf(d[3]) * ha;
  const float xm = (float)(img_w - 1), ym = (float)(img_h - 1);
  o[0] = fminf(fmaxf(rintf(x - 0.5f * w), 0.f), xm);
  o[1] = fminf(fmaxf(rintf(y - 0.5f * h), 0.f), ym);
  o[2] = fminf(fmaxf(rintf(x + 0.5f * w), 0.f), xm);
  o[3] = fminf(fmaxf(rintf(y + 0.5f * h), 0.f), ym);
}

// IoU with the inclusive-pixel convention (reference nets_utils.py:189-207)
__device__ __forceinline__ float iou_incl(const float* a, const float* b) {
  const float xi = fmaxf((fminf(a[2], b[2]) - fmaxf(a[0], b[0])) + 1.0f, 0.f);
  const float yi = fmaxf((fminf(a[3], b[3]) - fmaxf(a[1], b[1])) + 1.0f, 0.f);
  const float inter = xi * yi;
  const float aa = ((a[2] - a[0]) + 1.0f) * ((a[3] - a[1]) + 1.0f);
  const float ab = ((b[2] - b[0]) + 1.0f) * ((b[3] - b[1]) + 1.0f);
  return inter / ((aa + ab) - inter);
}

// ------------------------------------------------------------------ RPN decode
__global__ void rpn_decode_kernel(const float* __restrict__ cls, const float* __restrict__ reg,
                                  const float* __restrict__ anchors, int KA, int n_anchor, int img_w, int img_h,
                                  float min_size, float* __restrict__ boxes, uint32_t* __restrict__ keys,
                                  int* __restrict__ keep_count) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int kept = 0;
  if (i < KA) {
    const int k = i / n_anchor, a = i - k * n_anchor;
    const long long pix = (long long)b * (KA / n_anchor) + k;
    const float score = cls[pix * (2 * n_anchor) + 2 * a + 1];
    float d[4], an[4], o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { d[e] = reg[pix * (4 * n_anchor) + 4 * a + e]; an[e] = anchors[i * 4 + e]; }
    decode_clip(d, an, img_w, img_h, o);
    const bool keep = ((o[2] - o[0]) + 1.0f >= min_size) && ((o[3] - o[1]) + 1.0f >= min_size);
    float* ob = boxes + ((long long)b * KA + i) * 4;
    ob[0] = o[0]; ob[1] = o[1]; ob[2] = o[2]; ob[3] = o[3];
    keys[(long long)b * KA + i] = keep ? nbm_f2key(score) : 0u;
    kept = keep ? 1 : 0;
  }
  const unsigned long long m = __ballot(kept);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(keep_count + b, __popcll(m));
}

// ------------------------------------------------------------------ top-N selection (radix select + bitonic sort)
// composite key = (score_key << 32) | ~index : unique, descending order == (score desc, index asc)
constexpr int SEL_THREADS = 1024;

__global__ __launch_bounds__(SEL_THREADS) void rpn_select_kernel(
    const float* __restrict__ boxes, const uint32_t* __restrict__ keys, const int* __restrict__ keep_count, int B,
    int KA, int top_n, int fail_below, int cap, float* __restrict__ sel_boxes, float* __restrict__ sel_scores,
    int* __restrict__ n_sel, int per_image) {
  extern __shared__ unsigned long long sm[];  // [cap] sort buffer, then 256 histogram counters + scalars
  unsigned long long* buf = sm;
  unsigned int* hist = reinterpret_cast<unsigned int*>(sm + cap);
  __shared__ unsigned long long s_prefix;
  __shared__ int s_need, s_count, s_N;
  const int b = blockIdx.x, tid = threadIdx.x;
  const uint32_t* kb = keys + (long long)b * KA;

  if (tid == 0) {
    // layers.py:287: pre_nms_topN = min(topN, min over the BATCH of the kept-anchor counts).  per_image: every image is a
    // batch of its own (bulk inference: the reference CLI runs one file per model call), n_sel[b]
    int mn = 0x7fffffff;
    if (per_image) mn = keep_count[b];
    else for (int i = 0; i < B; ++i) mn = min(mn, keep_count[i]);
    int N = min(top_n, mn);
    if (N < fail_below) N = 0;
    s_N = N; s_need = N; s_prefix = 0ull; s_count = 0;
    if (per_image) n_sel[b] = N; else if (b == 0) n_sel[0] = N;
  }
  __syncthreads();
  const int N = s_N;
  for (int i = tid; i < cap; i += SEL_THREADS) buf[i] = 0ull;
  if (N > 0) {
    // 8 passes of 8 bits, MSB first: find the N-th largest composite key
    for (int pass = 7; pass >= 0; --pass) {
      for (int i = tid; i < 256; i += SEL_THREADS) hist[i] = 0u;
      __syncthreads();
      const int shift = pass * 8;
      const unsigned long long prefix = s_prefix;
      const unsigned long long himask = pass == 7 ? 0ull : (~0ull << (shift + 8));
      for (int i = tid; i < KA; i += SEL_THREADS) {
        const unsigned long long c = ((unsigned long long)kb[i] << 32) | (unsigned int)(~(unsigned int)i);
        if ((c & himask) == prefix) atomicAdd(&hist[(c >> shift) & 255ull], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        int need = s_need, d = 255;
        for (; d > 0; --d) { if ((int)hist[d] >= need) break; need -= (int)hist[d]; }
        s_need = need;
        s_prefix = prefix | ((unsigned long long)d << shift);
      }
      __syncthreads();
    }
    const unsigned long long thr = s_prefix;  // exactly N composites are >= thr
    for (int i = tid; i < KA; i += SEL_THREADS) {
      const unsigned long long c = ((unsigned long long)kb[i] << 32) | (unsigned int)(~(unsigned int)i);
      if (c >= thr) { const int pos = atomicAdd(&s_count, 1); if (pos < cap) buf[pos] = c; }
    }
  }
  __syncthreads();
  // bitonic sort, descending
  for (int k = 2; k <= cap; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < cap; i += SEL_THREADS) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long x = buf[i], y = buf[ixj];
          const bool desc = (i & k) == 0;
          if (desc ? (x < y) : (x > y)) { buf[i] = y; buf[ixj] = x; }
        }
      }
      __syncthreads();
    }
  }
  for (int r = tid; r < cap; r += SEL_THREADS) {
    float* ob = sel_boxes + ((long long)b * cap + r) * 4;
    if (r < N) {
      const unsigned long long c = buf[r];
      const int idx = (int)(~(unsigned int)(c & 0xffffffffull));
      const float* ib = boxes + ((long long)b * KA + idx) * 4;
      ob[0] = ib[0]; ob[1] = ib[1]; ob[2] = ib[2]; ob[3] = ib[3];
      sel_scores[(long long)b * cap + r] = nbm_key2f((uint32_t)(c >> 32));
    } else {
      ob[0] = ob[1] = ob[2] = ob[3] = 0.f;
      sel_scores[(long long)b * cap + r] = 0.f;
    }
  }
}

// ------------------------------------------------------------------ NMS
__global__ void nms_mask_kernel(const float* __restrict__ boxes, const int* __restrict__ n_in, int n_stride, int cap, int words,
                                float thresh, unsigned long long* __restrict__ mask) {
  const int b = blockIdx.z, rb = blockIdx.y, cb = blockIdx.x;
  const int n = n_in[b * n_stride];
  if (rb * 64 >= n || cb * 64 >= n || cb < rb) return;
  __shared__ float cbox[64][4];
  const int t = threadIdx.x;
  const float* bb = boxes + (long long)b * cap * 4;
  {
    const int j = cb * 64 + t;
#pragma unroll
    for (int e = 0; e < 4; ++e) cbox[t][e] = j < n ? bb[j * 4 + e] : 0.f;
  }
  __syncthreads();
  const int i = rb * 64 + t;
  if (i >= n) return;
  float me[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) me[e] = bb[i * 4 + e];
  unsigned long long bits = 0ull;
  const int jn = min(64, n - cb * 64);
  for (int jj = 0; jj < jn; ++jj) {
    const int j = cb * 64 + jj;
    if (j > i && iou_incl(me, cbox[jj]) >= thresh) bits |= 1ull << jj;
  }
  mask[((long long)b * cap + i) * words + cb] = bits;
}

// one wave per image: lane w owns removed-word w (cap <= 4096)
__global__ __launch_bounds__(64) void nms_scan_kernel(const unsigned long long* __restrict__ mask,
                                                      const int* __restrict__ n_in, int n_stride, int cap, int words,
                                                      int* __restrict__ keep_idx, int* __restrict__ keep_cnt) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int n = n_in[b * n_stride];
  const int nwords = (n + 63) >> 6;
  unsigned long long removed = 0ull;
  int cnt = 0;
  const unsigned long long* mb = mask + (long long)b * cap * words;
  for (int i = 0; i < n; ++i) {
    const int wi = i >> 6;
    const unsigned long long rw = __shfl(removed, wi);
    if (!((rw >> (i & 63)) & 1ull)) {
      if (lane == 0) keep_idx[(long long)b * cap + cnt] = i;
      ++cnt;
      // row i only has valid words for column blocks >= i/64 (upper triangle)
      if (lane < nwords && lane >= wi) removed |= mb[(long long)i * words + lane];
    }
  }
  if (lane == 0) keep_cnt[b] = cnt;
}

__global__ void nms_gather_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                  const int* __restrict__ keep_idx, const int* __restrict__ keep_cnt, int B, int cap,
                                  int post_n, float* __restrict__ rois, float* __restrict__ roi_scores,
                                  int* __restrict__ n_out, int per_image) {
  const int b = blockIdx.x;
  // nets_utils.py:236: post_nms_topN = min(topN, min over the BATCH of the survivor counts); per_image: n_out[b]
  int mn = 0x7fffffff;
  if (per_image) mn = keep_cnt[b];
  else for (int i = 0; i < B; ++i) mn = min(mn, keep_cnt[i]);
  const int R = min(post_n, mn);
  if (threadIdx.x == 0) { if (per_image) n_out[b] = R; else if (b == 0) n_out[0] = R; }
  for (int r = threadIdx.x; r < post_n; r += blockDim.x) {
    float* o = rois + ((long long)b * post_n + r) * 4;
    if (r < R) {
      const int idx = keep_idx[(long long)b * cap + r];
      const float* ib = boxes + ((long long)b * cap + idx) * 4;
      o[0] = ib[0]; o[1] = ib[1]; o[2] = ib[2]; o[3] = ib[3];
      roi_scores[(long long)b * post_n + r] = scores[(long long)b * cap + idx];
    } else {
      o[0] = o[1] = o[2] = o[3] = 0.f;
      roi_scores[(long long)b * post_n + r] = 0.f;
    }
  }
}

// ------------------------------------------------------------------ RoI pooling (+ positional encoding)
struct RoiParams {
  const float* fmap[5]; int fh[5], fw[5]; int n_levels, C;
  const float* rois; const int* n_roi; int n_stride, B, roi_cap;
  const float* pe_f; const float* pe_t; int img_h, img_w;
  float* pool; float* pe; int* level;
};

// Pyramid level and feature-map window of one RoI, layers.py:408-417 (sizes WITHOUT the +1), :424-427 (round), :450-462
// (y2 clamp, growth to at least 2 x 2).  The rows read are y1..y2, the columns x1..min(x2, W-1).
__device__ __forceinline__ void roi_window(const float* roi, int n_levels, const int* fh, const int* fw, int& lvl, int& x1,
                                           int& y1, int& x2, int& y2) {
  const float size = sqrtf((roi[2] - roi[0]) * (roi[3] - roi[1]));
  const float lf = logf(size * 0.1f) / 0.6931471805599453f;
  lvl = (lf != lf) ? (int)0x80000000 : (lf <= -2147483648.f ? (int)0x80000000 : (lf >= 2147483648.f ? (int)0x80000000 : (int)lf));
  lvl = min(max(lvl, 0), n_levels - 1);
  const float stride = (float)(2 << lvl);
  x1 = (int)rintf(roi[0] / stride); y1 = (int)rintf(roi[1] / stride);
  x2 = (int)rintf(roi[2] / stride); y2 = (int)rintf(roi[3] / stride);
  const int H = fh[lvl], W = fw[lvl];
  y2 = min(y2, H - 1);
  while (y2 - y1 + 1 < 2) { y1 = max(0, y1 - 1); y2 = min(H - 1, y2 + 1); }
  while (x2 - x1 + 1 < 2) { x1 = max(0, x1 - 1); x2 = min(W - 1, x2 + 1); }
}

__global__ void roi_pool_kernel(const RoiParams p) {
  const int slot = blockIdx.x;                 // b * roi_cap + r
  const int b = slot / p.roi_cap, r = slot - b * p.roi_cap;
  if (r >= p.n_roi[b * p.n_stride]) return;
  const float* roi = p.rois + (long long)slot * 4;
  int lvl, x1, y1, x2, y2;
  roi_window(roi, p.n_levels, p.fh, p.fw, lvl, x1, y1, x2, y2);
  const int H = p.fh[lvl], W = p.fw[lvl];
  if (threadIdx.x == 0) p.level[slot] = lvl;
  const int C = p.C, half = C >> 1;
  const float* fm = p.fmap[lvl];
  // the reference slices fmap[..., y1:y2+1, x1:x2+1]: python slicing silently clamps x2 (== W when the RoI touches
  // the right border: 1023/2 = 511.5 -> 512) while the positional encoding below keeps the unclamped x2
  const int h = y2 - y1 + 1, w = min(x2, W - 1) - x1 + 1;
  const int s = 2 << lvl;
  const int f0 = s * y1, f1 = min(s * y2, p.img_h), hf = f1 - f0;         // pe_frequency[s*y1 : s*y2]
  const int wt = min(s * (x2 - x1), p.img_w);                              // pe_time[: s*(x2-x1)]
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ya = (i * h) / 2, yb = ((i + 1) * h + 1) / 2;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int xa = (j * w) / 2, xb = ((j + 1) * w + 1) / 2;
        float sum = 0.f;
        for (int yy = ya; yy < yb; ++yy)
          for (int xx = xa; xx < xb; ++xx)
            sum += fm[(((long long)b * H + y1 + yy) * W + x1 + xx) * C + c];
        p.pool[((long long)slot * 4 + i * 2 + j) * C + c] = sum / (float)((yb - ya) * (xb - xa));
      }
    }
    if (c < half) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ra = (i * hf) / 2, rb = ((i + 1) * hf + 1) / 2;
        double acc = 0.0;
        for (int q = ra; q < rb; ++q) acc += (double)p.pe_f[(long long)(f0 + q) * half + c];
        const float v = (float)(acc / (double)(rb - ra));
        p.pe[((long long)slot * 4 + i * 2 + 0) * C + c] = v;
        p.pe[((long long)slot * 4 + i * 2 + 1) * C + c] = v;
      }
    } else {
      const int cc = c - half;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ca = (j * wt) / 2, cb = ((j + 1) * wt + 1) / 2;
        double acc = 0.0;
        for (int q = ca; q < cb; ++q) acc += (double)p.pe_t[(long long)q * half + cc];
        const float v = (float)(acc / (double)(cb - ca));
        p.pe[((long long)slot * 4 + 0 * 2 + j) * C + c] = v;
        p.pe[((long long)slot * 4 + 1 * 2 + j) * C + c] = v;
      }
    }
  }
}

// ------------------------------------------------------------------ demand-driven pyramid level: tiles under the RoIs
// The finest FPN output map is read by two consumers only: the RPN's stride-8 depthwise convolution (a fixed pixel pattern)
// and the RoI pooling of the RoIs assigned to that level.  This kernel lists the 2 x 2 Winograd output tiles of level `level`
// that the RoI windows touch (minus the tiles in `skip`, which the fixed pattern has computed already): one workgroup per
// image, an LDS bitmap, then an ordered compaction into 128-entry blocks (entries = b * TH * TW + ty * TW + tx, ascending,
// -1 padded, a block never mixes images); blocks are handed out by an atomic counter, so the filled blocks are the leading
// *n_blocks ones of `tiles` (capacity B * ceil(TH * TW / 128) blocks: cannot overflow).
struct RoiTilesParams {
  const float* rois; const int* n_roi; int B, roi_cap, n_levels, level; int fh[5], fw[5];
  const unsigned char* skip; int* tiles; int* n_blocks; int dilate, n_stride;
};

__global__ __launch_bounds__(256) void roi_tiles_kernel(const RoiTilesParams p) {
  extern __shared__ unsigned bits[];
  __shared__ int part[256];
  __shared__ int base_s;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int H = p.fh[p.level], W = p.fw[p.level];
  const int TH = (H + 1) >> 1, TW = (W + 1) >> 1, THW = TH * TW;
  const int n_words = (THW + 31) >> 5;
  for (int i = tid; i < n_words; i += 256) bits[i] = 0u;
  __syncthreads();
  const int n = min(p.n_roi[b * p.n_stride], p.roi_cap);
  for (int r = tid; r < n; r += 256) {
    int lvl, x1, y1, x2, y2;
    roi_window(p.rois + ((long long)b * p.roi_cap + r) * 4, p.n_levels, p.fh, p.fw, lvl, x1, y1, x2, y2);
    if (lvl != p.level) continue;
    const int dl = p.dilate;           // > 0: the tiles within `dilate` pixels of the window (support of a 3x3 data gradient)
    const int ty0 = max(y1 - dl, 0) >> 1, ty1 = min(y2 + dl, H - 1) >> 1, tx0 = max(x1 - dl, 0) >> 1, tx1 = min(min(x2, W - 1) + dl, W - 1) >> 1;
    for (int ty = ty0; ty <= ty1; ++ty)
      for (int tx = tx0; tx <= tx1; ++tx) {
        const int id = ty * TW + tx;
        if (p.skip && p.skip[id]) continue;
        atomicOr(&bits[id >> 5], 1u << (id & 31));
      }
  }
  __syncthreads();
  // ordered compaction: thread t owns the words [t * wpt, (t + 1) * wpt)
  const int wpt = (n_words + 255) / 256;
  int cnt = 0;
  for (int i = tid * wpt; i < min((tid + 1) * wpt, n_words); ++i) cnt += __popc(bits[i]);
  part[tid] = cnt;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int i = 0; i < 256; ++i) { const int c = part[i]; part[i] = run; run += c; }
    base_s = run ? atomicAdd(p.n_blocks, (run + 127) >> 7) : 0;
    part[0] = 0;
    bits[n_words] = (unsigned)run;                               // total, one spare word
  }
  __syncthreads();
  const int total = (int)bits[n_words];
  if (!total) return;
  int* out = p.tiles + (long long)base_s * 128;
  int o = part[tid];
  for (int i = tid * wpt; i < min((tid + 1) * wpt, n_words); ++i) {
    unsigned wbits = bits[i];
    while (wbits) {
      const int bit = __ffs(wbits) - 1;
      wbits &= wbits - 1;
      out[o++] = b * THW + (i << 5) + bit;
    }
  }
  const int padded = ((total + 127) >> 7) << 7;
  for (int i = total + tid; i < padded; i += 256) out[i] = -1;
}

// ------------------------------------------------------------------ FastRCNN eval post-processing
constexpr int POST_MAX = 1024;

__global__ __launch_bounds__(256) void rcnn_post_kernel(const float* __restrict__ rois, const int* __restrict__ n_roi,
                                                        int roi_cap, const float* __restrict__ bbox_reg,
                                                        const float* __restrict__ bbox_cls, int n_cls1, int img_w,
                                                        int img_h, float nms_thresh, float min_score,
                                                        int proposal_number, float* __restrict__ det,
                                                        int* __restrict__ n_det, int n_stride) {
  __shared__ unsigned long long skey[POST_MAX];
  __shared__ float sbox[POST_MAX][4];
  __shared__ float sscore[POST_MAX];
  __shared__ int scls[POST_MAX];
  __shared__ unsigned char alive[POST_MAX];
  __shared__ int s_cnt;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int R = min(n_roi[b * n_stride], roi_cap);
  int P = 1;
  while (P < R) P <<= 1;
  // 1. class arg-max (first maximum), score
  for (int r = tid; r < P; r += blockDim.x) {
    unsigned long long key = 0ull;
    if (r < R) {
      const float* pc = bbox_cls + ((long long)b * roi_cap + r) * n_cls1;
      float best = pc[0]; int bi = 0;
      for (int c = 1; c < n_cls1; ++c) { const float v = pc[c]; if (v > best) { best = v; bi = c; } }
      key = ((unsigned long long)nbm_f2key(best) << 32) | (unsigned int)(~(unsigned int)r);
      scls[r] = bi;  // temporarily indexed by roi
    }
    skey[r] = key;
  }
  __syncthreads();
  // 2. sort by (score desc, roi index asc)
  for (int k = 2; k <= P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < P; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long x = skey[i], y = skey[ixj];
          const bool desc = (i & k) == 0;
          if (desc ? (x < y) : (x > y)) { skey[i] = y; skey[ixj] = x; }
        }
      }
      __syncthreads();
    }
  // 3. decode the arg-max class deltas against the RoI, clip (sorted position s <- roi r)
  __shared__ int ocls[POST_MAX];
  for (int s = tid; s < R; s += blockDim.x) {
    const unsigned long long key = skey[s];
    const int r = (int)(~(unsigned int)(key & 0xffffffffull));
    const int c = scls[r];
    const float* pd = bbox_reg + ((long long)b * roi_cap + r) * (4 * n_cls1) + 4 * c;
    const float* pr = rois + ((long long)b * roi_cap + r) * 4;
    float d[4] = {pd[0], pd[1], pd[2], pd[3]}, a[4] = {pr[0], pr[1], pr[2], pr[3]};
    decode_clip(d, a, img_w, img_h, sbox[s]);
    sscore[s] = nbm_key2f((uint32_t)(key >> 32));
    ocls[s] = c;
    alive[s] = c > 0 ? 1 : 0;                      // background dropped before the first NMS (layers.py:734)
  }
  __syncthreads();
  // 4. class-agnostic greedy NMS in score order (layers.py:742)
  for (int i = 0; i < R; ++i) {
    if (alive[i]) {
      for (int j = i + 1 + tid; j < R; j += blockDim.x)
        if (alive[j] && iou_incl(sbox[i], sbox[j]) >= nms_thresh) alive[j] = 0;
    }
    __syncthreads();
  }
  // 5. per-class greedy NMS (layers.py:761).  After step 4 no surviving pair overlaps >= thresh, so this
  //    only reproduces the per-class truncation to `proposal_number`; kept for faithfulness.
  for (int i = 0; i < R; ++i) {
    if (alive[i]) {
      for (int j = i + 1 + tid; j < R; j += blockDim.x)
        if (alive[j] && ocls[j] == ocls[i] && iou_incl(sbox[i], sbox[j]) >= nms_thresh) alive[j] = 0;
    }
    __syncthreads();
  }
  // 6. rank inside class (truncate to proposal_number), min_score filter, emit sorted by (class, score desc)
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  __shared__ unsigned char emit[POST_MAX];
  for (int s = tid; s < R; s += blockDim.x) {
    int rank_in_class = 0;
    if (alive[s])
      for (int j = 0; j < s; ++j) rank_in_class += (alive[j] && ocls[j] == ocls[s]) ? 1 : 0;
    emit[s] = (alive[s] && rank_in_class < proposal_number && sscore[s] > min_score) ? 1 : 0;
  }
  __syncthreads();
  for (int s = tid; s < R; s += blockDim.x) {
    if (!emit[s]) continue;
    int pos = 0;
    for (int j = 0; j < R; ++j)
      if (emit[j] && (ocls[j] < ocls[s] || (ocls[j] == ocls[s] && j < s))) ++pos;
    float* o = det + ((long long)b * roi_cap + pos) * 6;
    o[0] = (float)ocls[s]; o[1] = sbox[s][0]; o[2] = sbox[s][1]; o[3] = sbox[s][2]; o[4] = sbox[s][3];
    o[5] = sscore[s];
    atomicAdd(&s_cnt, 1);
  }
  __syncthreads();
  if (tid == 0) n_det[b] = s_cnt;
}


// Proposal targets, the arithmetic half (layers.py:320-330 -> nets_utils.py:103-126): IoU with the inclusive-pixel (+1) convention
// of every proposal and every ground-truth box of an image against the image's ground-truth boxes, best overlap and FIRST best
// box per row.  One thread per row; the fp32 operations and their order are those of the reference's torch-CPU expressions (this
// file is compiled without contraction, the division is correctly rounded), so the thresholds the host applies afterwards
// resolve exactly as they do there.
__global__ void proposal_iou_kernel(const float* __restrict__ rois, const float* __restrict__ gt, const int* __restrict__ n_gt,
                                    int B, int R, int G, float* __restrict__ mx, int* __restrict__ asg) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (j >= R + G) return;
  const float* a = j < R ? rois + ((size_t)b * R + j) * 4 : gt + ((size_t)b * G + (j - R)) * 4;
  const float a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
  const float area_a = (a2 - a0 + 1.f) * (a3 - a1 + 1.f);
  const int ng = n_gt[b];
  float best = 0.f;
  int arg = 0;
  for (int g = 0; g < G; ++g) {
    float ov = -1.f;                                     // padded column: can never win
    if (g < ng) {
      const float* q = gt + ((size_t)b * G + g) * 4;
      const float g0 = q[0], g1 = q[1], g2 = q[2], g3 = q[3];
      float xi = fminf(a2, g2) - fmaxf(a0, g0) + 1.f;
      xi = fmaxf(xi, 0.f);
      float yi = fminf(a3, g3) - fmaxf(a1, g1) + 1.f;
      yi = fmaxf(yi, 0.f);
      const float inter = xi * yi;
      const float area_g = (g2 - g0 + 1.f) * (g3 - g1 + 1.f);
      const float den = (area_a + area_g) - inter;
      ov = __fdiv_rn(inter, den);
    }
    // numpy max / argmax: a NaN wins and stays, otherwise the first strictly greater value
    if (g == 0 || (!(best != best) && (ov > best || ov != ov))) { best = ov; arg = g; }
  }
  mx[(size_t)b * (R + G) + j] = best;
  asg[(size_t)b * (R + G) + j] = arg;
}

// AnchorTargetLayer, arithmetic half (reference layers.py:150-179, nets_utils.py:103-126): IoU of the image's ground-truth boxes with
// every anchor that lies inside the image, best overlap / first best box per anchor, best overlap per box, and the label each anchor
// has BEFORE the random subsampling: 0 if best < neg_t, 1 if best >= pos_t or the anchor is a best anchor (ties included) of a box
// whose best overlap is > 0, else -1.  One workgroup per image; the IoU uses the reference's fp32 operations in its order (no
// contraction in this file, correctly rounded division), evaluated twice with the same code, so `ov == gmx` means what it means on
// the host.  A NaN or negative overlap (degenerate box) raises the image's flag: the host then recomputes that image with NumPy.
constexpr int ANCHOR_MAXG = 256;
__device__ __forceinline__ float anchor_ov(const float* a, const float* q) {
  const float area_a = (a[2] - a[0] + 1.f) * (a[3] - a[1] + 1.f);
  float xi = fminf(a[2], q[2]) - fmaxf(a[0], q[0]) + 1.f;
  xi = fmaxf(xi, 0.f);
  float yi = fminf(a[3], q[3]) - fmaxf(a[1], q[1]) + 1.f;
  yi = fmaxf(yi, 0.f);
  const float inter = xi * yi;
  const float area_g = (q[2] - q[0] + 1.f) * (q[3] - q[1] + 1.f);
  const float den = (area_a + area_g) - inter;
  return __fdiv_rn(inter, den);
}

__global__ __launch_bounds__(1024) void anchor_targets_kernel(const float* __restrict__ anchors, int n_in, const float* __restrict__ gt,
                                                              const int* __restrict__ n_gt, int G, float neg_t, float pos_t,
                                                              signed char* __restrict__ lab, short* __restrict__ amx,
                                                              int* __restrict__ flag) {
  __shared__ float sg[ANCHOR_MAXG * 4];
  __shared__ int gmx[ANCHOR_MAXG];          // float bits of the best overlap of every box (overlaps are >= +0: integer order == float order)
  __shared__ int bad;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int ng = min(n_gt[b], G);
  for (int i = tid; i < ng * 4; i += 1024) sg[i] = gt[(size_t)b * G * 4 + i];
  for (int i = tid; i < ng; i += 1024) gmx[i] = 0;
  if (tid == 0) bad = 0;
  __syncthreads();
  for (int a = tid; a < n_in; a += 1024) {
    float an[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) an[e] = anchors[(size_t)a * 4 + e];
    for (int g = 0; g < ng; ++g) {
      const float ov = anchor_ov(an, sg + 4 * g);
      if (!(ov >= 0.f)) bad = 1;             // NaN or negative: benign race, every writer stores 1
      else if (ov > 0.f) atomicMax(&gmx[g], __float_as_int(ov));
    }
  }
  __syncthreads();
  for (int a = tid; a < n_in; a += 1024) {
    float an[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) an[e] = anchors[(size_t)a * 4 + e];
    float best = 0.f;
    int arg = 0;
    bool top = false;
    for (int g = 0; g < ng; ++g) {
      const float ov = anchor_ov(an, sg + 4 * g);
      if (g == 0 || ov > best) { best = ov; arg = g; }       // numpy max / argmax: the first strictly greater value
      const float gm = __int_as_float(gmx[g]);
      top = top || (gm > 0.f && ov == gm);
    }
    signed char l = -1;
    if (best < neg_t) l = 0;
    if (best >= pos_t) l = 1;
    if (top) l = 1;
    lab[(size_t)b * n_in + a] = l;
    amx[(size_t)b * n_in + a] = (short)arg;
  }
  if (tid == 0) flag[b] = bad;
}

}  // namespace

extern "C" int nbm_rpn_decode(const float* cls, const float* reg, const float* anchors, int B, int KA,
                                int n_anchor, int img_w, int img_h, int min_size, float* boxes, uint32_t* keys,
                                int* keep_count, void* stream) {
  if (!cls || !reg || !anchors || !boxes || !keys || !keep_count || B <= 0 || KA <= 0 || n_anchor <= 0 ||
      KA % n_anchor)
    return NBM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = nbm_zero_async(keep_count, sizeof(int) * B, st);
  if (e != hipSuccess) return (int)e;
  dim3 grid((KA + 255) / 256, B);
  hipLaunchKernelGGL(rpn_decode_kernel, grid, dim3(256), 0, st, cls, reg, anchors, KA, n_anchor, img_w, img_h,
                     (float)min_size, boxes, keys, keep_count);
  return nbm_launch_status();
}

extern "C" int nbm_rpn_select(const float* boxes, const uint32_t* keys, const int* keep_count, int B, int KA,
                              int top_n, int fail_below, int cap, float* sel_boxes, float* sel_scores, int* n_sel, int per_image,
                              void* stream) {
  if (!boxes || !keys || !keep_count || !sel_boxes || !sel_scores || !n_sel || B <= 0 || KA <= 0) return NBM_EINVAL;
  if (cap < top_n || cap > 4096 || (cap & (cap - 1))) return NBM_EINVAL;
  const size_t shmem = (size_t)cap * 8 + 256 * 4;
  hipLaunchKernelGGL(rpn_select_kernel, dim3(B), dim3(SEL_THREADS), shmem, (hipStream_t)stream, boxes, keys,
                     keep_count, B, KA, top_n, fail_below, cap, sel_boxes, sel_scores, n_sel, per_image ? 1 : 0);
  return nbm_launch_status();
}

extern "C" int nbm_nms_batched(const float* boxes, const float* scores, const int* n_in, int B, int cap,
                               float thresh, int post_n, uint64_t* mask_ws, int* keep_ws, float* rois,
                               float* roi_scores, int* n_out, int per_image, void* stream) {
  if (!boxes || !scores || !n_in || !mask_ws || !keep_ws || !rois || !roi_scores || !n_out || B <= 0) return NBM_EINVAL;
  if (cap <= 0 || cap > 4096 || (cap & 63) || post_n <= 0 || post_n > cap) return NBM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int words = cap / 64;
  int* keep_idx = keep_ws;
  int* keep_cnt = keep_ws + (size_t)B * cap;
  const int ns = per_image ? 1 : 0;
  hipLaunchKernelGGL(nms_mask_kernel, dim3(words, words, B), dim3(64), 0, st, boxes, n_in, ns, cap, words, thresh,
                     reinterpret_cast<unsigned long long*>(mask_ws));
  hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(64), 0, st, reinterpret_cast<const unsigned long long*>(mask_ws),
                     n_in, ns, cap, words, keep_idx, keep_cnt);
  hipLaunchKernelGGL(nms_gather_kernel, dim3(B), dim3(256), 0, st, boxes, scores, keep_idx, keep_cnt, B, cap, post_n,
                     rois, roi_scores, n_out, ns);
  return nbm_launch_status();
}

extern "C" int nbm_roi_pool(const nbm_roi_desc* d, void* stream) {
  if (!d || !d->rois || !d->n_roi || !d->pool || !d->pe || !d->level || !d->pe_f || !d->pe_t) return NBM_EINVAL;
  if (d->n_levels < 1 || d->n_levels > 5 || d->C <= 0 || (d->C & 1) || d->B <= 0 || d->roi_cap <= 0) return NBM_EINVAL;
  RoiParams p;
  for (int i = 0; i < 5; ++i) {
    p.fmap[i] = d->fmap[i]; p.fh[i] = d->fh[i]; p.fw[i] = d->fw[i];
    if (i < d->n_levels && (!d->fmap[i] || d->fh[i] < 2 || d->fw[i] < 2)) return NBM_EINVAL;
  }
  p.n_levels = d->n_levels; p.C = d->C; p.rois = d->rois; p.n_roi = d->n_roi; p.n_stride = d->n_roi_per_image ? 1 : 0; p.B = d->B; p.roi_cap = d->roi_cap;
  p.pe_f = d->pe_f; p.pe_t = d->pe_t; p.img_h = d->img_h; p.img_w = d->img_w;
  p.pool = d->pool; p.pe = d->pe; p.level = d->level;
  hipLaunchKernelGGL(roi_pool_kernel, dim3(d->B * d->roi_cap), dim3(256), 0, (hipStream_t)stream, p);
  return nbm_launch_status();
}

// RoI windows of pyramid level `level` -> list of the 2 x 2 output tiles they touch -- see nbm_hip.h.
extern "C" int nbm_roi_tiles(const float* rois, const int* n_roi, int B, int roi_cap, int n_levels, int level,
                             const int* fh, const int* fw, const unsigned char* skip, int dilate, int* tiles, int* n_blocks,
                             int per_image, void* stream) {
  if (!rois || !n_roi || !fh || !fw || !tiles || !n_blocks || B <= 0 || roi_cap <= 0 || n_levels < 1 || n_levels > 5 ||
      level < 0 || level >= n_levels || dilate < 0 || dilate > 2)
    return NBM_EINVAL;
  RoiTilesParams p{};
  p.dilate = dilate;
  p.n_stride = per_image ? 1 : 0;
  for (int i = 0; i < n_levels; ++i) {
    if (fh[i] < 2 || fw[i] < 2) return NBM_EINVAL;
    p.fh[i] = fh[i]; p.fw[i] = fw[i];
  }
  p.rois = rois; p.n_roi = n_roi; p.B = B; p.roi_cap = roi_cap; p.n_levels = n_levels; p.level = level;
  p.skip = skip; p.tiles = tiles; p.n_blocks = n_blocks;
  const int thw = ((fh[level] + 1) >> 1) * ((fw[level] + 1) >> 1);
  const size_t lds = (size_t)(((thw + 31) >> 5) + 1) * 4;
  if (lds > 60000) return NBM_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (nbm_zero_async(n_blocks, sizeof(int), st) != hipSuccess) return (int)hipGetLastError();
  hipLaunchKernelGGL(roi_tiles_kernel, dim3(B), dim3(256), lds, st, p);
  return nbm_launch_status();
}

extern "C" int nbm_rcnn_post(const float* rois, const int* n_roi, int B, int roi_cap, const float* bbox_reg,
                             const float* bbox_cls, int n_cls1, int img_w, int img_h, float nms_thresh,
                             float min_score, int proposal_number, float* det, int* n_det, int per_image, void* stream) {
  if (!rois || !n_roi || !bbox_reg || !bbox_cls || !det || !n_det || B <= 0 || roi_cap <= 0 || roi_cap > POST_MAX ||
      n_cls1 < 2)
    return NBM_EINVAL;
  hipLaunchKernelGGL(rcnn_post_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, rois, n_roi, roi_cap, bbox_reg,
                     bbox_cls, n_cls1, img_w, img_h, nms_thresh, min_score, proposal_number, det, n_det, per_image ? 1 : 0);
  return nbm_launch_status();
}

// IoU / best ground-truth box of the proposal target layer -- see nbm_hip.h.
extern "C" int nbm_proposal_iou(const float* rois, const float* gt, const int* n_gt, int B, int R, int G, float* mx, int* asg,
                                void* stream) {
  if (!rois || !gt || !n_gt || !mx || !asg || B <= 0 || R < 0 || G <= 0 || B > 65535) return NBM_EINVAL;
  hipLaunchKernelGGL(proposal_iou_kernel, dim3((R + G + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, rois, gt, n_gt, B, R, G,
                     mx, asg);
  return nbm_launch_status();
}

// AnchorTargetLayer: labels before subsampling + first best box of every inside anchor -- see nbm_hip.h.
extern "C" int nbm_anchor_targets(const float* anchors, int n_in, const float* gt, const int* n_gt, int B, int G, float neg_t,
                                  float pos_t, signed char* lab, short* amx, int* flag, void* stream) {
  if (!anchors || !gt || !n_gt || !lab || !amx || !flag || n_in <= 0 || B <= 0 || G <= 0) return NBM_EINVAL;
  if (G > ANCHOR_MAXG) return NBM_EUNSUPPORTED;
  hipLaunchKernelGGL(anchor_targets_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, anchors, n_in, gt, n_gt, G, neg_t, pos_t, lab,
                     amx, flag);
  return nbm_launch_status();
}
