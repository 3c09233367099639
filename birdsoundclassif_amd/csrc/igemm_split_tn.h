// Launch parameters of the split-bf16 weight-gradient kernel (igemm_split_tn.hip).  Internal to the library -- the C ABI is nbm_bwd_desc.
#pragma once
#include "nbm_common.h"

namespace nbm_igemm {

struct SplitTnParams {
  const float* g; const float* x; float* out; const float* row_scale;   // G [M][g_ld] (N valid), X [M][x_ld] (K valid), dW [N][out_ld]
  long long g_gs, x_gs, out_gs;
  int M, N, K;           // pixels (reduction), rows of dW, columns of dW
  int g_ld, x_ld, out_ld;
  int k_chunk;           // pixels per split, a multiple of 32
  int n_tiles;           // ceil(K / 128)
  float alpha;
};

// 256 x 128 tiles, one workgroup per CU; grid = (m_tiles * n_tiles, splits, groups)
int split_tn_launch(const SplitTnParams& p, int splits, int groups, hipStream_t st);

}  // namespace nbm_igemm
