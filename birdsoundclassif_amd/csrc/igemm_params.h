// Launch parameters of the implicit-GEMM forward kernels (igemm.hip: fp32 matrix pipe; igemm_split.hip: bf16 matrix pipe on split
// fp32 operands).  Internal to the library -- the C ABI is nbm_gemm_desc (include/nbm_hip.h).
#pragma once
#include "nbm_common.h"

namespace nbm_igemm {

struct IgemmParams {
  const float* x; const float* w; float* y;
  const float* scale; const float* shift; const float* residual;
  long long x_gs, w_gs, y_gs, res_gs;
  int M, N, K;           // K = kh*kw*Cin (valid)
  int nk;                // number of 32-wide K steps
  int H, W, Cin, kh, kw, stride, pad, Ho, Wo, HoWo;
  int x_ld, w_ld, y_ld, res_ld;
  int m_tiles, n_tiles;
  float alpha; int act; int shift_per_row;
  int vec_epi;           // epilogue may use 16-byte accesses (N % 4 == 0, pitches % 4 == 0, 16-byte aligned bases)
  // fused top-down merge: y += bilinear_align_corners(up [B][up_H][up_W][N]) (vector epilogue only)
  const float* up; int up_H, up_W; float up_sh, up_sw;
  // ROWS instantiation only (1x1 / stride 1 / pad 0, one group, vector epilogue): GEMM row m is pixel row_pixel(m) of the dense
  // NHWC maps x / y / residual (and of the `up` geometry) instead of pixel m -- the lateral convolution of a demand-driven FPN
  // level is only evaluated where its consumer will read (nbm_gemm_conv, `rows`).
  //   rows_mode 1: rows[m] = pixel index b*H*W + y*W + x, ascending, -1 = none (only at the end);
  //   rows_mode 2: rows[m >> 4] = linear 2x2-tile id b*TH*TW + ty*TW + tx (or -1), row m = pixel (m & 15) of the tile's 4x4
  //                input patch (rows 2ty-1.., columns 2tx-1..; outside the image = none).
  //   rows_blocks (device, optional): number of leading 128-entry list blocks that are filled.
  const int* rows; const int* rows_blocks; int rows_mode, rows_TH, rows_TW;
  // producer mask (vector and scalar epilogue, not ROWS): y = 0 where mask[m][n] <= 0 -- a 1x1 data gradient run as this forward GEMM
  const float* mask; int mask_ld;
  // one bit per stored output element, (y > 0), 32 channels per word (vector epilogue, N % 32 == 0, one group, not ROWS) -- or null
  unsigned* bits_out;
  // launch-invariant divisors of the row decode (nbm_fastdiv, nbm_common.h): output pixels per image, output row width, and the tile grid
  // of rows_mode 2
  nbm_fastdiv fd_howo, fd_wo, fd_rows_thw, fd_rows_tw;
};

// igemm_split.hip: 256 x 128 tiles on the bf16 matrix pipe (fast gather, 16-byte epilogue, N > 64, nk > 8 only; the caller checks)
int split_launch(const IgemmParams& p, int groups, hipStream_t st);

// igemm_h16.hip: the 128 x 128 tile with half-step LDS stages, `occ` (3 or 4) workgroups per CU (fast gather, 16-byte epilogue, N > 64; same bits as
// igemm_kernel<128,128,...>)
int h16_launch(const IgemmParams& p, int groups, int occ, hipStream_t st);

}  // namespace nbm_igemm
