// Backward implicit GEMMs on the gfx950 fp32 matrix core (v_mfma_f32_32x32x2_f32).
//
//   NN  (data gradient):   dX[m][c] = sum_{r,s,n} G[pix(m,r,s)][n] * a_scale[n] * W[n][r][s][c]
//        m enumerates the INPUT pixels of the forward convolution; a forward stride > 1 is handled as a dilated gather
//        of G (only taps with (iy+pad-r) % stride == 0 contribute).  A operand: gathered G rows, n contiguous;
//        B operand "K-major": the KRSC weight rows W[n][r][s][:] are already contiguous along c, no transposed copy
//        of the weights is ever made.  Also serves plain C = A * B ("NN") GEMMs (attention dQ, P.V, dX of linears).
//   TN  (weight gradient): dW[n][r][s][c] += row_scale[n] * sum_m G[m][n] * X[pix(m,r,s)][c]
//        reduction over the output pixels, split over gridDim.y chunks, fp32 atomics into a zeroed dW (KRSC).  Both
//        operands are K-major (pixel-major, exactly as they lie in HBM), so neither is transposed: the MFMA fragments
//        are read from LDS with 16 ds_read_b32 (conflict-free: 32 consecutive floats per half-wave).
//
// Same structure as igemm.hip: 128 x BN x 32 tiles, 4 waves, double-buffered LDS with one barrier per K-step, operands
// fetched with branch-free raw buffer loads (invalid taps / rows get an out-of-range offset and read zeros), global
// loads issued two tiles ahead and, like the LDS writes, scheduled between the MFMAs of the running K-step.
#include <stdlib.h>
#include "nbm_common.h"
#include "igemm_split_tn.h"
#include <type_traits>

namespace {

constexpr int BK = 32;
constexpr int PITCH = 36;     // row-major-K tile: [rows][32 + 4]
constexpr unsigned OOB = 0x80000000u;

struct BwdParams {
  const float* g; const float* w; const float* x; float* out;
  const float* a_scale;      // NN: per-n multiplier of G (FrozenBN scale) or null
  const float* row_scale;    // TN: per-n multiplier of dW rows or null
  const float* residual;     // NN: added to dX (gradient accumulation) or null
  const float* mask;         // NN: forward output of the producer; dX is zeroed where mask <= 0 (ReLU) or null
  const unsigned* mask_bits; // NN: the same mask as bits, [M][Cin / 32] words (nbm_gemm_desc.bits_out of the producer), or null
  const float* residual2;    // NN: a HALF-resolution map [B][ceil(H/2)][ceil(W/2)][Cin] added at the pixels with even iy and ix, or null
  int res2_ld;
  long long g_gs, w_gs, x_gs, out_gs, res_gs;
  int B, H, W, Cin, N, kh, kw, stride, pad, Ho, Wo;
  int g_ld, w_ld, x_ld, out_ld, res_ld, mask_ld;
  int M;                     // NN: B*H*W input pixels; TN: B*Ho*Wo output pixels
  int m_tiles, n_tiles;
  int k_chunk;               // TN: pixels per split
  float alpha;
  int b_generic;             // TN: gather the X tile element-wise (Cin % 4 != 0)
  int w_row;                 // NN: floats between W[n] and W[n+1] (= taps*Cin, or the padded pitch)
  int vec_epi;               // NN: 16-byte epilogue accesses are legal
  float* bias_grad;          // TN: += sum_m g[m][n] (added by the workgroups of the first N-tile) or null
  int plain;                 // TN: 1x1 / stride 1 / pad 0 -- im2col row m IS input pixel m, no bounds to track
  int phased;                // NN, stride 2: M tiles are grouped by the parity class of (iy + pad, ix + pad)
  int ph_tiles[4];           //   M tiles of each class; tile_m = 4 * (tile within class) + class, so the four classes of
                             //   one image region run side by side and fill the same DRAM pages together
  int ablate;                // NN, timing-only build (-DNBM_ABLATE_NN, `make ablate_nn`; never shipped): NBM_NN_ABLATE bits, see ABL_*
  // NN: launch-invariant divisors of the row decode (nbm_fastdiv, nbm_common.h): pixels per image and row width of the (parity class's)
  // pixel grid, the stride, and the full grid again for the half-resolution residual
  nbm_fastdiv fd_hw[4], fd_w[4], fd_st, fd_HW, fd_W;
  nbm_fastdiv fd_howo, fd_wo;   // TN: output pixels per image, output row width
};

// Attribution of igemm_nn_kernel's time (scripts/dgrad_ablate.py -> profiles/r05_dgrad_attribution.txt).  Each bit
// removes ONE component from the kernel; results are wrong by design.  The shipped build compiles none of this (`abl()` is constant 0).
enum { ABL_NO_MASK = 1, ABL_NO_RESIDUAL = 2, ABL_NO_ASCALE = 4, /* 8: was the tap select -- unmasked taps read outside the tensor */ ABL_NO_STORE = 16, ABL_NO_EPILOGUE = 32,
       ABL_NO_LOADS = 64, ABL_NO_LDS_WRITES = 128 };
#ifdef NBM_ABLATE_NN
#define NBM_ABL(p, bit) (((p).ablate & (bit)) != 0)
// per parity class (0 when not phased): cycles in prologue / K loop / epilogue, tiles (thread 0 of every workgroup; clock64)
__device__ unsigned long long nbm_nn_dbg[24];   // [class][address arithmetic | first loads + LDS write | K loop | epilogue], [16 + class] tiles; [20..23] weight-gradient kernel: prologue | K loop | epilogue (atomics) | tiles
#define NBM_DBG_T(var) const long long var = clock64()
#define NBM_DBG_ADD(slot, v) do { if (threadIdx.x == 0) atomicAdd(&nbm_nn_dbg[slot], (unsigned long long)(v)); } while (0)
#else
#define NBM_ABL(p, bit) false
#define NBM_DBG_T(var)
#define NBM_DBG_ADD(slot, v)
#endif

__device__ __forceinline__ int xcd_tile(int nwg, int bid) {
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
}

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, bytes, 0x00020000);
}

// ---------------------------------------------------------------------------------------------------- NN
// STAGES = 2: the deep-K pipeline.  STAGES = 1 (round 3): short-K layers (N * taps <= 256: the 1x1 data gradients of the ResNet
// bottlenecks, whose time is the dX / shortcut-gradient / ReLU-mask traffic of the epilogue, not MFMA): one LDS buffer (35 KB)
// and an epilogue staged in two halves, so THREE workgroups fit a CU and their load / compute / store phases overlap -- the same
// trade as igemm.hip's STAGES = 1 forward variant.
// HS (round 5, STAGES = 2 only): HALF-STEP LDS stages like igemm_h16.hip -- a stage holds 16 of the 32 K values of a step (A rows of 20
// floats, 16 B rows), the epilogue goes out in two halves like the single-stage form's, and THREE workgroups share a CU.  Half-step h
// holds k = 8h .. 8h + 7 (A columns / B rows 0..7, fed by lanes 0..31) and k = 16 + 8h .. (columns / rows 8..15, lanes 32..63): its two
// MFMA groups are groups 2h, 2h + 1 of the full step -- the same products in the same order, the same bits.
template <int BN, int STAGES = 2, bool HS = false>
__global__ __launch_bounds__(256, (STAGES == 1 || HS) ? 3 : 2) void igemm_nn_kernel(const BwdParams p) {
  static_assert(!HS || STAGES == 2, "half-step stages are a form of the two-stage pipeline");
  constexpr int BM = 128, WM = 64, WN = BN / 2, MT = 2, NT = WN / 32;
  constexpr int BP = BN + 4;                                   // K-major B tile pitch
  constexpr int KS = HS ? 16 : BK;                             // K values per LDS stage
  constexpr int AP = HS ? 20 : PITCH;                          // A tile pitch (16 + 4 / 32 + 4 floats)
  constexpr int NG = KS / 8;                                   // MFMA groups (4 k-pairs each) per stage
  __shared__ __attribute__((aligned(16))) float lds[STAGES * (BM * AP + KS * BP)];
  float* As = lds;
  float* Bs = lds + STAGES * BM * AP;
  if constexpr (STAGES == 2) nbm_stagger_priority();

  NBM_DBG_T(dbg_t0);
  const int wg = xcd_tile(gridDim.x, blockIdx.x);
  const int tile_m = wg / p.n_tiles, tile_n = wg - tile_m * p.n_tiles;
  const int bm0 = tile_m * BM, bn0 = tile_n * BN;
  const int grp = blockIdx.z;
  const float* __restrict__ gg = p.g + (long long)grp * p.g_gs;
  const float* __restrict__ wg_ = p.w + (long long)grp * p.w_gs;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  const int lrow = lane & 31, lh = lane >> 5;
  const int st = p.stride;
  // v / stride for v >= 0 through a host-prepared multiplier: a 32-bit division by a run-time value is ~35 vector instructions, and the
  // tap loop of the prologue did 4 of them per (row, tap) (144 per thread for a 3x3), `load_tiles` 2 per K-step, the row decode 2 per row
  auto divst = [&](int v) -> int { return (int)nbm_fdiv((unsigned)v, p.fd_st); };
  const int frm = divst(p.kh - 1), fsm = divst(p.kw - 1);      // largest tap shift in G rows / columns

  // Row -> input pixel.  Plain: row q = bm0 + r enumerates (b, iy, ix) row-major.  Phased (stride 2): a pixel only
  // receives the taps with r = (iy + pad) mod 2, s = (ix + pad) mod 2, so the M tiles are grouped by that parity class;
  // a tile enumerates its class's sub-grid (b, u, v) -> (2u + y0, 2v + x0) and its K loop visits the class's taps only
  // (1, 2, 2 or 4 of the 9 taps of a 3x3 instead of all 9 with 3/4 of the loads masked to zero).
  int y0 = 0, x0 = 0, Hp = p.H, Wp = p.W, rows_here = p.M, q0 = bm0, r_begin = 0, s_begin = 0, t_step = 1;
  if (p.phased) {
    const int ph = tile_m & 3;
    if ((tile_m >> 2) >= p.ph_tiles[ph]) return;               // classes differ by at most a row / column of tiles
    q0 = (tile_m >> 2) * BM;
    r_begin = ph >> 1; s_begin = ph & 1; t_step = 2;
    y0 = (r_begin + p.pad) & 1; x0 = (s_begin + p.pad) & 1;
    Hp = (p.H - y0 + 1) >> 1; Wp = (p.W - x0 + 1) >> 1;
    rows_here = p.B * Hp * Wp;
  }
  const nbm_fastdiv dv_hw = p.fd_hw[p.phased ? (tile_m & 3) : 0], dv_w = p.fd_w[p.phased ? (tile_m & 3) : 0];
  auto decode = [&](int q, int& b, int& iy, int& ix) -> bool {
    const bool ok = q < rows_here;
    const int qq = ok ? q : 0, hw = Hp * Wp;
    b = (int)nbm_fdiv((unsigned)qq, dv_hw);
    const int rem = qq - b * hw, u = (int)nbm_fdiv((unsigned)rem, dv_w), v = rem - u * Wp;
    iy = p.phased ? 2 * u + y0 : u;
    ix = p.phased ? 2 * v + x0 : v;
    return ok;
  };

  // A staging: 128 rows x 8 (half-step: 4) chunks of 16 B.  Row i gathers G at (b, fy - r/st, fx - s/st) for the taps whose parity
  // matches; offsets are relative to a block-uniform base shifted by the largest tap so that they stay non-negative.
  constexpr int ACH = KS / 4, ARPP = 256 / ACH, AR = BM / ARPP;      // chunks per row, rows per pass, rows per thread
  const int c4 = tid % ACH, r0 = tid / ACH;
  // first K value (of the 32 of a step) of this thread's chunk: 4 c4, or -- half-step -- 0, 4, 16, 20 (+ 8 h)
  const int koff = HS ? (c4 < 2 ? 4 * c4 : 16 + 4 * (c4 - 2)) : 4 * c4;
  unsigned a_rel[AR];
  unsigned long long a_taps[AR];
  long long blk_base;
  {
    int b, iy, ix;
    decode(q0, b, iy, ix);
    // column 0 of the first row's G row: with stride > 1 two consecutive input rows can map to the SAME G row, so the
    // first pixel of the tile is not necessarily the smallest address (offsets must stay non-negative)
    (void)ix;
    blk_base = ((long long)(b * p.Ho + divst(iy + p.pad)) * p.Wo) * p.g_ld;
  }
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    int b, iy, ix;
    const bool ok = decode(q0 + r0 + ARPP * i, b, iy, ix);
    const int fy = divst(iy + p.pad), fx = divst(ix + p.pad);
    const int py = (iy + p.pad) - fy * st, px = (ix + p.pad) - fx * st;
    const long long base = ((long long)(b * p.Ho + fy) * p.Wo + fx) * p.g_ld;
    a_rel[i] = ((unsigned)(base - blk_base) + koff) * 4u;
    // taps that reach this pixel: filter rows r with r = py (mod stride) whose G row fy - r / stride exists, likewise columns, then their
    // product -- selects only (as a kh x kw nest of data-dependent branches this was the longest part of the tile prologue: 28 k of the
    // 37 k cycles in front of the first load of a 3x3 tile, cycle counters of the `ablate_nn` build, round 5)
    unsigned long long mk = 0ull;
    unsigned rowm = 0u, colm = 0u;
    for (int r = 0; r < p.kh; ++r) {
      const int rq = divst(r);
      rowm |= ((r - rq * st == py && (unsigned)(fy - rq) < (unsigned)p.Ho) ? 1u : 0u) << r;
    }
    for (int s = 0; s < p.kw; ++s) {
      const int sq = divst(s);
      colm |= ((s - sq * st == px && (unsigned)(fx - sq) < (unsigned)p.Wo) ? 1u : 0u) << s;
    }
    for (int r = 0; r < p.kh; ++r) mk |= ((rowm >> r) & 1u) ? (unsigned long long)colm << (r * p.kw) : 0ull;
    a_taps[i] = ok ? mk : 0ull;
  }
  const long long maxoff = ((long long)frm * p.Wo + fsm) * p.g_ld;
  const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(gg + blk_base - maxoff, 0x7ffffff0u);

  // B staging (K-major): 32 rows x BN/4 chunks; thread -> rows rk0 + (256/(BN/4)) * i
  constexpr int BCH = BN / 4, BROWS = 256 / BCH, BPASS = KS / BROWS;
  const int bc = tid % BCH, rk0 = tid / BCH;
  unsigned b_rel[BPASS];
#pragma unroll
  for (int i = 0; i < BPASS; ++i) {
    const int j = rk0 + BROWS * i;                               // LDS row; half-step: rows 8..15 hold k = 16 + (row - 8) (+ 8 h)
    const int kk = HS ? (j < 8 ? j : j + 8) : j;
    b_rel[i] = (bn0 + bc * 4 < p.Cin) ? (unsigned)(kk * p.w_row + bn0 + bc * 4) * 4u : OOB;
  }
  // rows n >= N fall outside the resource and read zeros
  const long long wbytes = (long long)p.N * p.w_row * 4;
  const __amdgpu_buffer_rsrc_t rsrc_b = make_rsrc(wg_, (unsigned)(wbytes < 0x7ffffff0ll ? wbytes : 0x7ffffff0ll));

  f32x4 ra[AR], rb[BPASS], rsc = {1.f, 1.f, 1.f, 1.f};
  int cur_r = r_begin, cur_s = s_begin, cur_n0 = 0, cur_h = 0;

  int abl_loads_done = 0;
  auto load_tiles = [&]() {
    if (NBM_ABL(p, ABL_NO_LOADS) && abl_loads_done >= 2) return;           // operands stay what the first two loads fetched
    ++abl_loads_done;
    const int tap = cur_r * p.kw + cur_s;
    const int n0h = cur_n0 + 8 * cur_h;                                    // (half-step: the second half starts 8 K values further)
    const unsigned a_soff = (unsigned)((maxoff - ((long long)divst(cur_r) * p.Wo + divst(cur_s)) * p.g_ld + n0h) * 4);
    const unsigned b_soff = (unsigned)(((long long)n0h * p.w_row + (long long)tap * p.Cin) * 4);
#pragma unroll
    for (int i = 0; i < AR; ++i) ra[i] = buf_load4(rsrc_a, ((a_taps[i] >> tap) & 1ull) ? a_rel[i] : OOB, a_soff);
#pragma unroll
    for (int i = 0; i < BPASS; ++i) rb[i] = buf_load4(rsrc_b, b_rel[i], b_soff);
    if (p.a_scale && !NBM_ABL(p, ABL_NO_ASCALE)) rsc = *reinterpret_cast<const f32x4*>(p.a_scale + n0h + koff);
    if (HS && (cur_h ^= 1) != 0) return;                                   // the other half of the same (tap, 32-wide K) step comes next
    cur_s += t_step;
    if (cur_s >= p.kw) { cur_s = s_begin; cur_r += t_step; if (cur_r >= p.kh) { cur_r = r_begin; cur_n0 += BK; } }
  };
  int abl_writes_done = 0;
  auto store_lds = [&](int buf) {
    if (NBM_ABL(p, ABL_NO_LDS_WRITES) && abl_writes_done >= 2) return;
    ++abl_writes_done;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      f32x4 v = ra[i];
      if (!NBM_ABL(p, ABL_NO_ASCALE)) { v[0] *= rsc[0]; v[1] *= rsc[1]; v[2] *= rsc[2]; v[3] *= rsc[3]; }
      *reinterpret_cast<f32x4*>(As + (buf * BM + r0 + ARPP * i) * AP + c4 * 4) = v;
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i)
      *reinterpret_cast<f32x4*>(Bs + (buf * KS + rk0 + BROWS * i) * BP + bc * 4) = rb[i];
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto mfma_group = [&](const float* Ab, const float* Bb, int q) {
    f32x4 a[MT];
    float b[NT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * AP + q * 4);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) b[j][e] = Bb[(q * 4 + e) * BP + j * 32];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
  };

  const int n_r = r_begin < p.kh ? (p.kh - r_begin + t_step - 1) / t_step : 0;
  const int n_s = s_begin < p.kw ? (p.kw - s_begin + t_step - 1) / t_step : 0;
  const int nk = ((p.N + BK - 1) / BK) * n_r * n_s;          // 0: no tap reaches this parity class, dX = residual
  NBM_DBG_T(dbg_t0b);
  const int n_st = HS ? 2 * nk : nk;                             // LDS stages of the K loop
  if constexpr (STAGES == 2) {
    if (n_st > 0) {
      load_tiles();
      store_lds(0);
    }
    if (n_st > 1) load_tiles();
  }

  NBM_DBG_T(dbg_t1);
  auto k_step = [&](int kt, auto store_c, auto load_c) {
    constexpr bool STORE = decltype(store_c)::value, LOAD = decltype(load_c)::value;
    const int cur = kt & 1;
    __syncthreads();
    const float* Ab = As + (cur * BM + wm0 + lrow) * AP + lh * (KS / 2);
    const float* Bb = Bs + (cur * KS + lh * (KS / 2)) * BP + wn0 + lrow;
    mfma_group(Ab, Bb, 0);
    if constexpr (STORE) {
      store_lds(cur ^ 1);
      __builtin_amdgcn_sched_group_barrier(0x100, MT + 4 * NT, 0);
#pragma unroll
      for (int z = 0; z < AR + BPASS; ++z) {
        __builtin_amdgcn_sched_group_barrier(0x008, (4 * MT * NT) / (AR + BPASS), 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(Ab, Bb, 1);
    if constexpr (LOAD) {
      load_tiles();
      __builtin_amdgcn_sched_group_barrier(0x100, MT + 4 * NT, 0);
#pragma unroll
      for (int z = 0; z < AR + BPASS; ++z) {
        __builtin_amdgcn_sched_group_barrier(0x008, (4 * MT * NT) / (AR + BPASS), 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NG == 4) {
      mfma_group(Ab, Bb, 2);
      mfma_group(Ab, Bb, 3);
    }
  };
  if constexpr (STAGES == 2) {
    using T = std::true_type;
    using F = std::false_type;
    int kt = 0;
    for (; kt + 2 < n_st; ++kt) k_step(kt, T{}, T{});
    if (n_st >= 2) { k_step(kt, T{}, F{}); ++kt; }
    if (n_st >= 1) k_step(kt, F{}, F{});
  } else {
    // short K: load -> LDS -> MFMA, the next tile's loads in flight during the MFMAs; the other two workgroups of the CU cover
    // the barriers
    if (nk > 0) load_tiles();
    for (int kt = 0; kt < nk; ++kt) {
      if (kt) __syncthreads();                       // everyone finished reading the previous tile
      store_lds(0);
      if (kt + 1 < nk) load_tiles();
      __syncthreads();
      const float* Ab = As + (wm0 + lrow) * PITCH + lh * 16;
      const float* Bb = Bs + (lh * 16) * BP + wn0 + lrow;
#pragma unroll
      for (int q = 0; q < 4; ++q) mfma_group(Ab, Bb, q);
    }
  }
  __syncthreads();
  NBM_DBG_T(dbg_t2);
#ifdef NBM_ABLATE_NN
  const int dbg_c = p.phased ? 4 * (tile_m & 3) : 0;
  NBM_DBG_ADD(dbg_c, dbg_t0b - dbg_t0); NBM_DBG_ADD(dbg_c + 1, dbg_t1 - dbg_t0b); NBM_DBG_ADD(dbg_c + 2, dbg_t2 - dbg_t1); NBM_DBG_ADD(16 + dbg_c / 4, 1);
#endif

  float* __restrict__ og = p.out + (long long)grp * p.out_gs;
  const float* __restrict__ rg = (p.residual && !NBM_ABL(p, ABL_NO_RESIDUAL)) ? p.residual + (long long)grp * p.res_gs : nullptr;
  const unsigned* __restrict__ mb_ = NBM_ABL(p, ABL_NO_MASK) ? nullptr : p.mask_bits;     // the mask as bits takes precedence
  const float* __restrict__ mk_ = (NBM_ABL(p, ABL_NO_MASK) || mb_) ? nullptr : p.mask;
  const float* __restrict__ r2_ = NBM_ABL(p, ABL_NO_RESIDUAL) ? nullptr : p.residual2;
  if (NBM_ABL(p, ABL_NO_EPILOGUE)) {                                       // keep the accumulators alive: a store that never happens
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) t += acc[i][j][e];
    if (t == 1.2345e-30f) og[0] = t;
    return;
  }
  auto out_pixel = [&](int q) -> long long {                  // output row of tile row q (q < rows_here)
    if (!p.phased) return q;
    int b, iy, ix;
    decode(q, b, iy, ix);
    return ((long long)b * p.H + iy) * p.W + ix;
  };
  if (p.vec_epi) {
    // accumulator tile -> LDS -> 16-byte residual / mask loads and stores (see igemm.hip)
    constexpr int CP = BN + 4;
    constexpr int HALVES = (STAGES == 1 || HS) ? BM / WM : 1;  // the single-stage / half-step LDS holds WM rows of the tile at a time
    constexpr int HROWS = BM / HALVES;
    static_assert(HROWS * CP + 2 * HROWS <= STAGES * (BM * AP + KS * BP), "epilogue tile + row table must fit the operand buffers");
    float* Cs = lds;
    // stride 2: the output pixel of a tile row (parity-class sub-grid -> image) is decoded ONCE per row into LDS, not by each of the row's
    // BN / 4 threads in the prefetch AND in the store loop (two fast divisions + the parity arithmetic per call; round 5)
    long long* const Rm = reinterpret_cast<long long*>(lds + HROWS * CP);     // [HROWS]; HROWS * CP is even
    constexpr int CH = BN / 4, RPP = 256 / CH;
    const int cc = tid % CH, rr = tid / CH;
    const int c = bn0 + cc * 4;
    constexpr int NR = (HROWS + RPP - 1) / RPP;                 // tile rows a thread finishes
#pragma unroll
    for (int half = 0; half < HALVES; ++half) {
      if (HALVES > 1 && half) __syncthreads();                  // the previous half has been read
      if (p.phased) {
        if (tid < HROWS) Rm[tid] = out_pixel(min(q0 + half * HROWS + tid, rows_here - 1));
        __syncthreads();
      }
      auto row_out = [&](int rl) -> long long {                 // output pixel of row rl of this half (clamped to the last row of the class)
        return p.phased ? Rm[rl] : (long long)min(q0 + half * HROWS + rl, rows_here - 1);
      };
      // residual / mask values of this thread's rows: requested before the accumulators go through LDS (inside the row loop, behind
      // its exit test, they were 2 NR dependent round trips at the end of every tile -- igemm.hip)
      f32x4 rq[NR], mq[NR];
      unsigned mw[NR];                                          // mask words (Cin % 32 == 0: the word of channels c .. c + 3 of the row)
      const bool pre = c < p.Cin && (rg || mk_ || mb_);
      if (pre) {
#pragma unroll
        for (int k = 0; k < NR; ++k) {
          const long long m = row_out(min(rr + k * RPP, HROWS - 1));
          if (rg) rq[k] = *reinterpret_cast<const f32x4*>(rg + m * p.res_ld + c);
          if (mk_) mq[k] = *reinterpret_cast<const f32x4*>(mk_ + m * p.mask_ld + c);
          if (mb_) mw[k] = mb_[m * (p.Cin >> 5) + (c >> 5)];
        }
      }
      if (HALVES == 1 || wm0 == half * HROWS) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
              Cs[(wm0 - half * HROWS + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * CP + wn0 + j * 32 + lrow] = acc[i][j][e];
      }
      __syncthreads();
      if (c >= p.Cin) continue;
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        const int rl = rr + k * RPP;
        if (rl >= HROWS) break;
        const int r = half * HROWS + rl;
        if (q0 + r >= rows_here) break;
        const long long m = row_out(rl);
        f32x4 v = *reinterpret_cast<const f32x4*>(Cs + rl * CP + cc * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
        if (rg) {
          const f32x4 q = rq[k];
          v[0] += q[0]; v[1] += q[1]; v[2] += q[2]; v[3] += q[3];
        }
        if (r2_) {          // the data gradient of a 1x1 / stride-2 shortcut, kept at its own (half) resolution
          const int hw = p.H * p.W, b_ = (int)nbm_fdiv((unsigned)m, p.fd_HW), rem_ = (int)(m - (long long)b_ * hw);
          const int iy_ = (int)nbm_fdiv((unsigned)rem_, p.fd_W), ix_ = rem_ - iy_ * p.W;
          if (!((iy_ | ix_) & 1)) {
            const long long m2 = ((long long)b_ * ((p.H + 1) >> 1) + (iy_ >> 1)) * ((p.W + 1) >> 1) + (ix_ >> 1);
            const f32x4 q = *reinterpret_cast<const f32x4*>(r2_ + m2 * p.res2_ld + c);
            v[0] += q[0]; v[1] += q[1]; v[2] += q[2]; v[3] += q[3];
          }
        }
        if (mk_) {
          const f32x4 q = mq[k];
#pragma unroll
          for (int e = 0; e < 4; ++e) if (!(q[e] > 0.f)) v[e] = 0.f;
        }
        if (mb_) {
          const unsigned by = mw[k] >> ((c >> 2) & 7);        // bit 8 e + q of the word <-> channel 4 q + e (nbm_hip.h)
#pragma unroll
          for (int e = 0; e < 4; ++e) if (!((by >> (8 * e)) & 1u)) v[e] = 0.f;
        }
        if (!NBM_ABL(p, ABL_NO_STORE) || v[0] == 1.2345e-30f) *reinterpret_cast<f32x4*>(og + (long long)m * p.out_ld + c) = v;
      }
    }
#ifdef NBM_ABLATE_NN
    __syncthreads();
    NBM_DBG_ADD(dbg_c + 3, clock64() - dbg_t2);
#endif
    return;
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int c = bn0 + wn0 + j * 32 + lrow;
    if (c >= p.Cin) continue;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int q = q0 + wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (q >= rows_here) continue;
        const long long m = out_pixel(q);
        float v = acc[i][j][e] * p.alpha;
        if (rg) v += rg[(long long)m * p.res_ld + c];
        if (r2_) {
          const int hw = p.H * p.W, b_ = (int)nbm_fdiv((unsigned)m, p.fd_HW), rem_ = (int)(m - (long long)b_ * hw);
          const int iy_ = (int)nbm_fdiv((unsigned)rem_, p.fd_W), ix_ = rem_ - iy_ * p.W;
          if (!((iy_ | ix_) & 1))
            v += r2_[(((long long)b_ * ((p.H + 1) >> 1) + (iy_ >> 1)) * ((p.W + 1) >> 1) + (ix_ >> 1)) * p.res2_ld + c];
        }
        if (mk_ && !(mk_[(long long)m * p.mask_ld + c] > 0.f)) v = 0.f;
        if (mb_ && !((mb_[m * (p.Cin >> 5) + (c >> 5)] >> (8 * (c & 3) + ((c >> 2) & 7))) & 1u)) v = 0.f;
        og[(long long)m * p.out_ld + c] = v;
      }
  }
}

// ---------------------------------------------------------------------------------------------------- TN
// BMODE: how the X tile (im2col rows of one tap) is fetched
//   B_SAME    stride-1 'same' convolution with Wo >= 32: input pixel = output pixel + constant, so every row has a
//             constant byte offset from a per-K-step scalar base -> raw buffer loads, padding / tail rows read zeros
//   B_STRIDED any stride / small maps, 16-byte channel chunks through pointers
//   B_GENERIC unaligned or Cin % 4 != 0: scalar gather over the flattened (tap, c) axis
enum { B_SAME = 0, B_STRIDED = 1, B_GENERIC = 2 };

// BM: rows of the dW tile = output channels n of the convolution.  128 by default; 64 for layers with N <= 64 (ResNet layer1:
// with the 128-row tile half of every MFMA multiplied the zero padding of the G tile -- 59 TF/s "executed" on 3x3 64 -> 64).
template <int BN, int BMODE, int SCHED = 0, int BM = 128>
__global__ __launch_bounds__(256, 2) void igemm_tn_kernel(const BwdParams p) {
  constexpr int WM = BM / 2, WN = BN / 2, MT = WM / 32, NT = WN / 32;
  constexpr int AP = BM + 4, BP = BN + 4;
  __shared__ __attribute__((aligned(16))) float lds[2 * BK * AP + 2 * BK * BP];
  nbm_stagger_priority();
  float* As = lds;                    // [2][32][AP]   G tile:  pixel-major, n contiguous
  float* Bs = lds + 2 * BK * AP;      // [2][32][BP]   X tile:  pixel-major, c contiguous

  NBM_DBG_T(dbg_t0);
  const int wg = xcd_tile(gridDim.x, blockIdx.x);
  const int tile_m = wg / p.n_tiles, tile_n = wg - tile_m * p.n_tiles;
  const int bm0 = tile_m * BM;                      // n0
  const int grp = blockIdx.z;
  const int taps = p.kh * p.kw;
  // N-tile -> (tap, c0)  [fast]   or   j0 over (tap, c) flattened  [generic]
  constexpr bool GEN = BMODE == B_GENERIC;
  const int ctiles = GEN ? 1 : (p.Cin + BN - 1) / BN;
  const int tap = GEN ? 0 : tile_n / ctiles;
  const int c0 = GEN ? 0 : (tile_n - tap * ctiles) * BN;
  const int j0 = tile_n * BN;                       // generic only
  const int tr = tap / p.kw, ts = tap - tr * p.kw;

  const float* __restrict__ gg = p.g + (long long)grp * p.g_gs;
  const float* __restrict__ xg = p.x + (long long)grp * p.x_gs;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  const int lrow = lane & 31, lh = lane >> 5;

  constexpr int ACH = BM / 4, AROWS = 256 / ACH, APASS = BK / AROWS;   // 32 chunks, 8 rows/pass, 4 passes
  constexpr int BCH = BN / 4, BROWS = 256 / BCH, BPASS = BK / BROWS;
  const int ac = tid % ACH, ar0 = tid / ACH;
  const int bc = tid % BCH, br0 = tid / BCH;

  const int m_begin = blockIdx.y * p.k_chunk;
  const int m_end = min(p.M, m_begin + p.k_chunk);
  const int nk = (m_end - m_begin + BK - 1) / BK;
  if (nk <= 0) return;

  f32x4 ra[APASS], rb[BPASS];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const int HoWo = p.Ho * p.Wo;

  // A rows: pixel-major G; rows beyond m_end fall outside the per-step resource and read zeros
  unsigned a_rel[APASS];
#pragma unroll
  for (int i = 0; i < APASS; ++i)
    a_rel[i] = (bm0 + ac * 4 < p.N) ? (unsigned)((ar0 + AROWS * i) * p.g_ld + bm0 + ac * 4) * 4u : OOB;
  // B rows (fast path): output pixel -> input pixel, tracked incrementally (+32 pixels per K-step)
  int b_b[BPASS], b_oy[BPASS], b_ox[BPASS];
  const bool c_ok = c0 + bc * 4 < p.Cin;
#pragma unroll
  for (int i = 0; i < BPASS; ++i) {
    const int m = m_begin + br0 + BROWS * i;
    b_b[i] = (int)nbm_fdiv((unsigned)m, p.fd_howo);
    const int rem = m - b_b[i] * HoWo;
    b_oy[i] = (int)nbm_fdiv((unsigned)rem, p.fd_wo);
    b_ox[i] = rem - b_oy[i] * p.Wo;
  }
  int kt_load = 0;
  // B_SAME: constant row offsets relative to the K-step base pixel (+ the tap's displacement)
  unsigned b_rel[BPASS];
#pragma unroll
  for (int i = 0; i < BPASS; ++i) b_rel[i] = c_ok ? (unsigned)((br0 + BROWS * i) * p.x_ld + c0 + bc * 4) * 4u : OOB;
  const long long dpix = (long long)(tr - p.pad) * p.W + (ts - p.pad);

  auto load_tiles = [&]() {
    const int mbase = m_begin + kt_load * BK;
    const int rows = min(BK, m_end - mbase);
    const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(gg + (long long)mbase * p.g_ld, (unsigned)(rows * p.g_ld * 4));
#pragma unroll
    for (int i = 0; i < APASS; ++i) ra[i] = buf_load4(rsrc_a, a_rel[i], 0);
    if constexpr (BMODE == B_SAME) {
      const __amdgpu_buffer_rsrc_t rsrc_b = make_rsrc(xg + ((long long)mbase + dpix) * p.x_ld, (unsigned)(rows * p.x_ld * 4));
#pragma unroll
      for (int i = 0; i < BPASS; ++i) {
        if (p.plain) {                                                  // plain GEMM rows: always inside
          rb[i] = buf_load4(rsrc_b, b_rel[i], 0);
          continue;
        }
        const bool ok = (unsigned)(b_oy[i] + tr - p.pad) < (unsigned)p.H && (unsigned)(b_ox[i] + ts - p.pad) < (unsigned)p.W;
        rb[i] = buf_load4(rsrc_b, ok ? b_rel[i] : OOB, 0);
        b_ox[i] += BK;                                                  // Wo >= BK on this path
        if (b_ox[i] >= p.Wo) { b_ox[i] -= p.Wo; if (++b_oy[i] == p.Ho) b_oy[i] = 0; }
      }
    } else if constexpr (BMODE == B_STRIDED) {
#pragma unroll
      for (int i = 0; i < BPASS; ++i) {
        const int m = mbase + br0 + BROWS * i;
        const int iy = b_oy[i] * p.stride - p.pad + tr, ix = b_ox[i] * p.stride - p.pad + ts;
        const bool ok = c_ok && m < m_end && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const float* src = xg + ((long long)(b_b[i] * p.H + (ok ? iy : 0)) * p.W + (ok ? ix : 0)) * p.x_ld + c0 + bc * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? src : xg);
        rb[i] = ok ? v : zero4;
        // advance this row by 32 output pixels
        if (p.Wo >= BK) {
          b_ox[i] += BK;
          if (b_ox[i] >= p.Wo) { b_ox[i] -= p.Wo; if (++b_oy[i] == p.Ho) { b_oy[i] = 0; ++b_b[i]; } }
        } else {
          const int mn = m + BK;
          b_b[i] = (int)nbm_fdiv((unsigned)mn, p.fd_howo);
          const int rem = mn - b_b[i] * HoWo;
          b_oy[i] = (int)nbm_fdiv((unsigned)rem, p.fd_wo);
          b_ox[i] = rem - b_oy[i] * p.Wo;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < BPASS; ++i) {
        const int m = mbase + br0 + BROWS * i;
        f32x4 v = zero4;
        if (m < m_end) {
          const int b = (int)nbm_fdiv((unsigned)m, p.fd_howo), rem = m - b * HoWo;
          const int oy = (int)nbm_fdiv((unsigned)rem, p.fd_wo), ox = rem - oy * p.Wo;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int j = j0 + bc * 4 + e;
            if (j < taps * p.Cin) {
              const int t = j / p.Cin, c = j - t * p.Cin;
              const int r = t / p.kw, s = t - r * p.kw;
              const int iy = oy * p.stride - p.pad + r, ix = ox * p.stride - p.pad + s;
              if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                v[e] = xg[((long long)(b * p.H + iy) * p.W + ix) * p.x_ld + c];
            }
          }
        }
        rb[i] = v;
      }
    }
    ++kt_load;
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < APASS; ++i)
      *reinterpret_cast<f32x4*>(As + (buf * BK + ar0 + AROWS * i) * AP + ac * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < BPASS; ++i)
      *reinterpret_cast<f32x4*>(Bs + (buf * BK + br0 + BROWS * i) * BP + bc * 4) = rb[i];
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const bool do_bias = p.bias_grad != nullptr && tile_n == 0;
  float bsum = 0.f;

  auto mfma_group = [&](const float* Ab, const float* Bb, int q) {
    float a[MT][4], b[NT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) a[i][e] = Ab[(q * 4 + e) * AP + i * 32];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) b[j][e] = Bb[(q * 4 + e) * BP + j * 32];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
  };

  load_tiles();
  store_lds(0);
  if (nk > 1) load_tiles();

  auto k_step = [&](int kt, auto store_c, auto load_c) {
    constexpr bool STORE = decltype(store_c)::value, LOAD = decltype(load_c)::value;
    const int cur = kt & 1;
    __syncthreads();
    if (do_bias) {            // workgroup-uniform: column sums of the G tile (bias gradient), 32 * BM / 256 pixels per thread
      constexpr int BT = 256 / BM, PX = BK / BT;
      const float* Ar = As + (cur * BK + (tid / BM) * PX) * AP + (tid % BM);
#pragma unroll
      for (int e = 0; e < PX; ++e) bsum += Ar[e * AP];
    }
    const float* Ab = As + (cur * BK + lh * 16) * AP + wm0 + lrow;
    const float* Bb = Bs + (cur * BK + lh * 16) * BP + wn0 + lrow;
    mfma_group(Ab, Bb, 0);
    if constexpr (STORE) {
      store_lds(cur ^ 1);
      if constexpr (SCHED != 2) {
        __builtin_amdgcn_sched_group_barrier(0x100, (SCHED == 1 ? 2 : 4) * (MT + NT), 0);
#pragma unroll
        for (int z = 0; z < APASS + BPASS; ++z) {
          __builtin_amdgcn_sched_group_barrier(0x008, (4 * MT * NT) / (APASS + BPASS) > 0 ? (4 * MT * NT) / (APASS + BPASS) : 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(Ab, Bb, 1);
    if constexpr (LOAD) {
      load_tiles();
      if constexpr (SCHED != 2) {
        __builtin_amdgcn_sched_group_barrier(0x100, (SCHED == 1 ? 2 : 4) * (MT + NT), 0);
#pragma unroll
        for (int z = 0; z < 4 * MT * NT; ++z) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
          if (z < APASS + BPASS) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(Ab, Bb, 2);
    mfma_group(Ab, Bb, 3);
  };
  NBM_DBG_T(dbg_t1);
  {
    using T = std::true_type;
    using F = std::false_type;
    int kt = 0;
    for (; kt + 2 < nk; ++kt) k_step(kt, T{}, T{});
    if (nk >= 2) { k_step(kt, T{}, F{}); ++kt; }
    k_step(kt, F{}, F{});
  }
  NBM_DBG_T(dbg_t2);

  if (do_bias) {
    const int n = bm0 + (tid % BM);
    if (n < p.N) atomicAdd(p.bias_grad + (long long)grp * p.N + n, bsum);
  }
  float* __restrict__ og = p.out + (long long)grp * p.out_gs;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int jj = wn0 + j * 32 + lrow;
    int col;
    if constexpr (!GEN) { const int c = c0 + jj; if (c >= p.Cin) continue; col = tap * p.Cin + c; }
    else { col = j0 + jj; if (col >= taps * p.Cin) continue; }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = bm0 + wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (n >= p.N) continue;
        float v = acc[i][j][e] * p.alpha;
        if (p.row_scale) v *= p.row_scale[n];
        atomicAdd(og + (long long)n * p.out_ld + col, v);
      }
  }
#ifdef NBM_ABLATE_NN
  __syncthreads();
  NBM_DBG_ADD(20, dbg_t1 - dbg_t0); NBM_DBG_ADD(21, dbg_t2 - dbg_t1); NBM_DBG_ADD(22, clock64() - dbg_t2); NBM_DBG_ADD(23, 1);
#endif
}

}  // namespace

#ifdef NBM_ABLATE_NN
// timing-only build: read and clear the per-class cycle counters of igemm_nn_kernel (scripts/dgrad_ablate.py)
extern "C" int nbm_nn_dbg_read(unsigned long long* host24) {
  unsigned long long z[24] = {0};
  if (hipMemcpyFromSymbol(host24, HIP_SYMBOL(nbm_nn_dbg), sizeof(z)) != hipSuccess) return -4;
  if (hipMemcpyToSymbol(HIP_SYMBOL(nbm_nn_dbg), z, sizeof(z)) != hipSuccess) return -4;
  return NBM_OK;
}
#endif


static int fill_common(const nbm_bwd_desc* d, BwdParams& p) {
  if (!d || !d->g || !d->out) return NBM_EINVAL;
  if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->N <= 0 || d->kh <= 0 || d->kw <= 0 || d->stride <= 0 ||
      d->groups <= 0 || d->kh * d->kw > 64)
    return NBM_EINVAL;
  if ((d->H + 2 * d->pad - d->kh) / d->stride + 1 != d->Ho || (d->W + 2 * d->pad - d->kw) / d->stride + 1 != d->Wo)
    return NBM_EINVAL;
  p.g = d->g; p.w = d->w; p.x = d->x; p.out = d->out;
  p.a_scale = d->a_scale; p.row_scale = d->row_scale; p.residual = d->residual; p.mask = d->mask;
  p.residual2 = d->residual2; p.res2_ld = d->res2_ld;
  p.g_gs = d->g_gs; p.w_gs = d->w_gs; p.x_gs = d->x_gs; p.out_gs = d->out_gs; p.res_gs = d->res_gs;
  p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.N = d->N; p.kh = d->kh; p.kw = d->kw;
  p.stride = d->stride; p.pad = d->pad; p.Ho = d->Ho; p.Wo = d->Wo;
  p.fd_howo = nbm_fastdiv_make((unsigned)(d->Ho * d->Wo)); p.fd_wo = nbm_fastdiv_make((unsigned)d->Wo);
  p.g_ld = d->g_ld; p.w_ld = d->w_ld; p.x_ld = d->x_ld; p.out_ld = d->out_ld; p.res_ld = d->res_ld; p.mask_ld = d->mask_ld;
  p.alpha = d->alpha;
  p.bias_grad = d->bias_grad;
  return NBM_OK;
}

extern "C" int nbm_conv_dgrad(const nbm_bwd_desc* d, void* stream) {
  BwdParams p{};
  int rc = fill_common(d, p);
  if (rc) return rc;
  if (!d->w) return NBM_EINVAL;
  // G rows must hold ceil(N/32)*32 readable floats (zero padded), 16-byte aligned; W rows are read along c
  if ((d->g_ld & 3) || d->g_ld < ((d->N + 31) / 32) * 32 || (d->Cin & 3) || (d->w_ld & 3) || (d->g_gs & 3) || (d->w_gs & 3) ||
      !nbm_aligned16(d->g) || !nbm_aligned16(d->w) || (d->a_scale && !nbm_aligned16(d->a_scale)))
    return NBM_EALIGN;
  if (d->out_ld < d->Cin || (d->residual && d->res_ld < d->Cin) || (d->mask && d->mask_ld < d->Cin)) return NBM_EINVAL;
  if (d->residual2 && (d->stride != 1 || d->groups != 1 || d->res2_ld < d->Cin)) return NBM_EINVAL;
  if (d->a_scale && (d->N & 31)) return NBM_EINVAL;
  if (d->mask_bits) {                    // the ReLU mask as bits: whole 32-channel words, one group
    if ((d->Cin & 31) || d->groups != 1) return NBM_EUNSUPPORTED;
    p.mask_bits = d->mask_bits;
  }
  // the gather window of one 128-row tile (+ one image boundary) must stay inside the 2 GB buffer resource
  {
    const long long row = (long long)d->Wo * d->g_ld * 4;                  // bytes per G image row
    const long long span = (d->W == 1 && d->kh == 1) ? 130ll * d->g_ld * 4   // plain GEMM: 128 consecutive rows
                                                      : (d->kh + 130) * row + (long long)d->Ho * row;
    if (span > 0x70000000ll) return NBM_EUNSUPPORTED;
  }
  p.M = d->B * d->H * d->W;
  p.w_row = d->w_ld;
  p.vec_epi = ((d->out_ld & 3) == 0 && (d->out_gs & 3) == 0 && nbm_aligned16(d->out) &&
               (!d->residual || ((d->res_ld & 3) == 0 && (d->res_gs & 3) == 0 && nbm_aligned16(d->residual))) &&
               (!d->mask || ((d->mask_ld & 3) == 0 && nbm_aligned16(d->mask))) &&
               (!d->residual2 || ((d->res2_ld & 3) == 0 && nbm_aligned16(d->residual2)))) ? 1 : 0;
  p.m_tiles = (p.M + 127) / 128;
  p.fd_st = nbm_fastdiv_make((unsigned)d->stride);
  p.fd_HW = p.fd_hw[0] = nbm_fastdiv_make((unsigned)(d->H * d->W));
  p.fd_W = p.fd_w[0] = nbm_fastdiv_make((unsigned)d->W);
  if (d->stride == 2) {                  // group the M tiles by parity class (see igemm_nn_kernel)
    p.phased = 1;
    int tmax = 0;
    for (int ph = 0; ph < 4; ++ph) {
      const int y0 = ((ph >> 1) + d->pad) & 1, x0 = ((ph & 1) + d->pad) & 1;
      const int Hp = (d->H - y0 + 1) >> 1, Wp = (d->W - x0 + 1) >> 1;
      const long long rows = (long long)d->B * Hp * Wp;
      p.ph_tiles[ph] = (int)((rows + 127) / 128);
      if (p.ph_tiles[ph] > tmax) tmax = p.ph_tiles[ph];
      p.fd_hw[ph] = nbm_fastdiv_make((unsigned)(Hp * Wp));     // (an empty class: no tile of it gets past its early exit)
      p.fd_w[ph] = nbm_fastdiv_make((unsigned)Wp);
    }
    p.m_tiles = 4 * tmax;
  }
#ifdef NBM_ABLATE_NN
  p.ablate = getenv("NBM_NN_ABLATE") ? atoi(getenv("NBM_NN_ABLATE")) : 0;       // read per call: the probe switches it between launches
#endif
  hipStream_t st = (hipStream_t)stream;
  // short K (<= 8 steps of 32) and a 16-byte epilogue: the three-workgroups-per-CU variant (see the template comment)
  static const int shortk_max = getenv("NBM_NN_SHORTK_MAX") ? atoi(getenv("NBM_NN_SHORTK_MAX")) : 8;   // 0 disables
  // stride 2 (phased): a tile's K loop visits its parity class's taps only -- (N / 32) x {1, 2, 2, 4} steps for a 3x3
  // (the largest class's step count decides: 128 -> 128 @94x256, steps 4 / 8 / 8 / 16: 3.61 -> 3.46 ms at B = 128 on the single-stage kernel)
  static const int shortk_ph = getenv("NBM_NN_SHORTK_PHASED") ? atoi(getenv("NBM_NN_SHORTK_PHASED")) : 16;
  const int ph_max = ((d->N + BK - 1) / BK) * ((d->kh + 1) / 2) * ((d->kw + 1) / 2);
  // the 64-wide tile spends half the MFMA cycles per K-step: its prologue / epilogue weigh double, and the third workgroup pays up to
  // K = 576 (layer1's 3x3 64 -> 64 @94x256 at B = 128: 2.39 -> 2.15 ms, round 5)
  const int shortk_lim = d->Cin <= 64 && shortk_max ? (shortk_max > 20 ? shortk_max : 20) : shortk_max;
  const bool shortk = p.vec_epi && (p.phased ? ph_max <= shortk_ph : ((d->N + BK - 1) / BK) * d->kh * d->kw <= shortk_lim);
  // deep K: the half-step form of the two-stage kernel (three workgroups per CU, same bits; NBM_NN_H16=0: the two-stage kernel).  Read per
  // call: the parity test flips it inside one process.
  const char* hs_env = getenv("NBM_NN_H16");
  const bool hs = !(hs_env && hs_env[0] == '0') && p.vec_epi;
  if (d->Cin > 64) {
    p.n_tiles = (d->Cin + 127) / 128;
    const dim3 grid(p.m_tiles * p.n_tiles, 1, d->groups);
    if (shortk) hipLaunchKernelGGL((igemm_nn_kernel<128, 1>), grid, dim3(256), 0, st, p);
    else if (hs) hipLaunchKernelGGL((igemm_nn_kernel<128, 2, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((igemm_nn_kernel<128, 2>), grid, dim3(256), 0, st, p);
  } else {
    p.n_tiles = 1;
    const dim3 grid(p.m_tiles, 1, d->groups);
    if (shortk) hipLaunchKernelGGL((igemm_nn_kernel<64, 1>), grid, dim3(256), 0, st, p);
    else if (hs) hipLaunchKernelGGL((igemm_nn_kernel<64, 2, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((igemm_nn_kernel<64, 2>), grid, dim3(256), 0, st, p);
  }
  return nbm_launch_status();
}

extern "C" int nbm_conv_wgrad(const nbm_bwd_desc* d, void* stream) {
  BwdParams p{};
  int rc = fill_common(d, p);
  if (rc) return rc;
  if (!d->x) return NBM_EINVAL;
  if ((d->g_ld & 3) || (d->g_gs & 3) || !nbm_aligned16(d->g) || d->g_ld < ((d->N + 3) / 4) * 4) return NBM_EALIGN;
  p.b_generic = ((d->Cin & 3) || (d->x_ld & 3) || (d->x_gs & 3) || !nbm_aligned16(d->x)) ? 1 : 0;
  const int taps = d->kh * d->kw;
  if (d->out_ld < taps * d->Cin) return NBM_EINVAL;
  // Plain GEMMs whose column count is 64 past a multiple of 128 (the cell-domain planes of the deferred lateral: [T][256]^T x [T][448]): the
  // 128-wide tiles would multiply 64 columns of padding in the last tile (1 / 8 of the MFMA work of 448 columns: 110 against 122 TF/s for
  // the 384-column twin, round 5) -- the first Cin - 64 columns on 128-wide tiles, the last 64 on the 64-wide kernel.  Same sums per element
  // (the split over the pixels is chosen per launch; the atomics make the order free anyway).
  {
    static const int tail_split = getenv("NBM_TN_TAIL") ? atoi(getenv("NBM_TN_TAIL")) : 1;
    if (tail_split && taps == 1 && d->stride == 1 && d->pad == 0 && !p.b_generic && d->Cin > 128 && (d->Cin & 127) == 64) {
      nbm_bwd_desc a = *d, b = *d;
      a.Cin = d->Cin - 64;
      b.Cin = 64;
      b.x = d->x + (d->Cin - 64);
      b.out = d->out + (d->Cin - 64);
      b.bias_grad = nullptr;                                   // (the column sums of G come from the first launch)
      rc = nbm_conv_wgrad(&a, stream);
      return rc ? rc : nbm_conv_wgrad(&b, stream);
    }
  }
  p.M = d->B * d->Ho * d->Wo;
  const bool narrow_m = d->N <= 64 && !p.b_generic;            // 64-row dW tiles: no MFMA spent on the zero half of a 128-row G tile
  p.m_tiles = narrow_m ? 1 : (d->N + 127) / 128;
  hipStream_t st = (hipStream_t)stream;
  const bool wide = !p.b_generic && d->Cin > 64;
  const int BN = wide ? 128 : 64;
  p.n_tiles = p.b_generic ? (taps * d->Cin + BN - 1) / BN : taps * ((d->Cin + BN - 1) / BN);
  // Split the pixel reduction so that the grid fills the chip in WHOLE rounds: 512 workgroups are resident at once
  // (256 CUs x 2), all of equal length, so a grid of 4.01 rounds costs 5 (the old ">= 2048 workgroups" rule hit exactly
  // that on the largest layer: 54 tiles x 38 splits = 2052).  Pick the split count with the best fill of its last
  // round among those with >= 8 K-steps per split and <= 16 rounds; ties go to fewer splits (fewer atomics).
  const int tiles = p.m_tiles * p.n_tiles * d->groups;
  const int max_splits = (p.M + 8 * BK - 1) / (8 * BK);
  // (512 = 256 CUs x the two workgroups of the 128 x 128 instantiation.  The narrower instantiations hold 3 or 4 per CU; sizing the rounds
  // for 768 / 1024 was measured in round 5 and changes nothing -- 2.161 vs 2.155 ms on layer1's 3x3: a CU with fewer workgroups left runs
  // them faster, the matrix pipe is what they share -- scripts/wgrad_cycles.py)
  const int slots = 512;
  int splits = 1;
  double best = -1.0;
  for (int sp = 1; sp <= max_splits && (long long)sp * tiles <= 16ll * slots; ++sp) {
    const long long wg = (long long)sp * tiles;
    const long long rounds = (wg + slots - 1) / slots;
    const double fill = (double)wg / (double)(rounds * slots);
    if (fill > best + 0.005) { best = fill; splits = sp; }
  }
  p.plain = (d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0) ? 1 : 0;
  // opt-in (NBM_SPLIT_BF16=1): plain weight-gradient GEMMs (1x1 convolutions, nn.Linear, the grouped Winograd- / cell-domain
  // products) with >= 192 rows and > 64 columns on the bf16 matrix pipe through split fp32 operands (igemm_split_tn.hip).  Read per call;
  // chosen by the LAYER, never by the number of pixels.  The bias gradient is not produced there (the caller sums the columns of G).
  {
    const char* se = getenv("NBM_SPLIT_BF16");
    const char* te = getenv("NBM_SPLIT_TN");
    if (se && se[0] == '1' && !(te && te[0] == '0') && p.plain && !p.b_generic && !d->bias_grad && d->N >= 192 && d->Cin > 64 &&
        (d->out_ld >= d->Cin) && (long long)32 * d->g_ld * 4 < 0x40000000ll && (long long)32 * d->x_ld * 4 < 0x40000000ll) {
      nbm_igemm::SplitTnParams q{};
      q.g = d->g; q.x = d->x; q.out = d->out; q.row_scale = d->row_scale;
      q.g_gs = d->g_gs; q.x_gs = d->x_gs; q.out_gs = d->out_gs;
      q.M = p.M; q.N = d->N; q.K = d->Cin; q.g_ld = d->g_ld; q.x_ld = d->x_ld; q.out_ld = d->out_ld; q.alpha = d->alpha;
      q.n_tiles = (d->Cin + 127) / 128;
      const int tiles_s = ((d->N + 255) / 256) * q.n_tiles * d->groups;
      const int slots_s = 256;                                       // one workgroup per CU
      const int max_sp = (p.M + 16 * 32 - 1) / (16 * 32);            // >= 32 K16 stages per split
      int sp_best = 1;
      double fill_best = -1.0;
      for (int sp = 1; sp <= max_sp && (long long)sp * tiles_s <= 16ll * slots_s; ++sp) {
        const long long wgs = (long long)sp * tiles_s;
        const long long rounds = (wgs + slots_s - 1) / slots_s;
        const double fill = (double)wgs / (double)(rounds * slots_s);
        if (fill > fill_best + 0.005) { fill_best = fill; sp_best = sp; }
      }
      q.k_chunk = (((p.M + sp_best - 1) / sp_best) + 31) / 32 * 32;
      const int sp_n = (p.M + q.k_chunk - 1) / q.k_chunk;
      return nbm_igemm::split_tn_launch(q, sp_n, d->groups, st);
    }
  }
  p.k_chunk = (((p.M + splits - 1) / splits) + BK - 1) / BK * BK;
  splits = (p.M + p.k_chunk - 1) / p.k_chunk;
  dim3 grid(p.m_tiles * p.n_tiles, splits, d->groups);
  const bool same = !p.b_generic && d->stride == 1 && d->Ho == d->H && d->Wo == d->W && (d->Wo >= BK || p.plain) &&
                    (long long)BK * d->x_ld * 4 < 0x40000000ll;
  if (p.b_generic) hipLaunchKernelGGL((igemm_tn_kernel<64, B_GENERIC>), grid, dim3(256), 0, st, p);
  else if (narrow_m) {
    if (wide && same) hipLaunchKernelGGL((igemm_tn_kernel<128, B_SAME, 0, 64>), grid, dim3(256), 0, st, p);
    else if (wide) hipLaunchKernelGGL((igemm_tn_kernel<128, B_STRIDED, 0, 64>), grid, dim3(256), 0, st, p);
    else if (same) hipLaunchKernelGGL((igemm_tn_kernel<64, B_SAME, 0, 64>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((igemm_tn_kernel<64, B_STRIDED, 0, 64>), grid, dim3(256), 0, st, p);
  }
  else if (wide && same) {
    static const int sched = getenv("NBM_TN_SCHED") ? atoi(getenv("NBM_TN_SCHED")) : 0;
    if (sched == 1) hipLaunchKernelGGL((igemm_tn_kernel<128, B_SAME, 1>), grid, dim3(256), 0, st, p);
    else if (sched == 2) hipLaunchKernelGGL((igemm_tn_kernel<128, B_SAME, 2>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((igemm_tn_kernel<128, B_SAME>), grid, dim3(256), 0, st, p);
  }
  else if (wide) hipLaunchKernelGGL((igemm_tn_kernel<128, B_STRIDED>), grid, dim3(256), 0, st, p);
  else if (same) hipLaunchKernelGGL((igemm_tn_kernel<64, B_SAME>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((igemm_tn_kernel<64, B_STRIDED>), grid, dim3(256), 0, st, p);
  return nbm_launch_status();
}
