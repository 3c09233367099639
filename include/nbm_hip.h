/* nbm_hip.h — C ABI of the MI355X-native NBM hot path (libnbm_hip.so).
 *
 * The reference (LouisBearing/BirdSoundClassif) is 100 % Python: it has no FFI layer, its "native"
 * boundary is torch's own CPU/CUDA operators.  Every entry point below replaces the torch / numpy /
 * librosa operator call(s) named in its comment (reference file:line, relative to /root/reference).
 * The host side (birdsoundclassif_amd/ops.py) binds them with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless the name ends in _host; no ownership transfer,
 *     the caller allocates outputs and workspaces;
 *   - activations are NHWC fp32 ("pixel pitch" = *_ld floats between consecutive pixels);
 *   - `stream` is a hipStream_t passed as void*; all functions are asynchronous on it, re-entrant,
 *     and keep no global state;
 *   - return value: 0 = success, otherwise a negative NBM_E* code (bad argument) or a positive
 *     hipError_t from the launch.
 */
#ifndef NBM_HIP_H
#define NBM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NBM_OK 0
#define NBM_EINVAL (-1)
#define NBM_EALIGN (-2)
#define NBM_EUNSUPPORTED (-3)

#define NBM_ACT_NONE 0
#define NBM_ACT_RELU 1
#define NBM_ACT_SILU 2
#define NBM_ACT_LEAKY 3 /* LeakyReLU, negative slope 0.01 (nn.LeakyReLU default) */

const char* nbm_version(void);

/* Census of a captured graph (a `hipGraph_t`, e.g. torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph()): counts[0..5] = kernel,
 * memcpy, memset, host, empty (fork / join), other nodes.  No counterpart in the reference (nbm_detect.py:23-28 runs one eager model
 * call per file); it is the fence of the captured bulk loop (bulk.GraphedDetector): the HIP runtime torch 2.10+rocm7.0 bundles
 * (7.0.51831) loses MEMSET nodes of a graph exec from its second launch on (scripts/graph_pair_repro.hip), so a step is
 * only replayed when its graph holds kernel (and empty) nodes alone.  This library itself never creates memset nodes. */
int nbm_graph_census(void* graph, long long counts[6]);

/* ------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution / batched GEMM on the fp32 MFMA (v_mfma_f32_32x32x2_f32).
 *
 *   y[g][m][n] = act( alpha * sum_k A[g][m][k] * W[g][n][k] * scale[n] + shift[n|m] + residual[g][m][n] )
 *
 *   A is the im2col view of x (NHWC): m = (b, oy, ox), k = (r, s, c) -> x[b][oy*stride-pad+r][ox*stride-pad+s][c]
 *   (zero outside the image).  W is [N][w_ld] with k contiguous (KRSC weights).  A plain GEMM is the
 *   case H = M, W = 1, kh = kw = 1.  `groups` (blockIdx.z) strides x / w / y / residual by *_gs floats
 *   (attention: one GEMM per clip; 0 = operand shared by all groups).
 *
 * Replaces: F.conv2d / nn.Linear / torch.matmul call sites of the detector forward
 *   backbone.py:104-113 (+ torchvision ResNet-50 convs), self_attention.py:43-49, fpn.py:132-146,
 *   layers.py:25-46,79-99,574-579; FrozenBatchNorm2d affine (backbone.py:52-62), bias, ReLU/SiLU and
 *   the residual add are fused in the epilogue.
 */
typedef struct nbm_gemm_desc {
  const float* x;        /* A source, NHWC [B][H][W][Cin] with pixel pitch x_ld            */
  const float* w;        /* [N][w_ld], K = kh*kw*Cin valid columns (rest must be finite)   */
  float* y;              /* [M][y_ld] (NHWC output, N valid channels)                      */
  const float* scale;    /* [N] or NULL                                                    */
  const float* shift;    /* [N] (or [M] when shift_per_row) or NULL                        */
  const float* residual; /* [M][res_ld] or NULL                                            */
  int64_t x_gs, w_gs, y_gs, res_gs; /* per-group strides in floats                        */
  int groups;
  int B, H, W, Cin;      /* input geometry                                                 */
  int N;                 /* output channels                                                */
  int kh, kw, stride, pad;
  int Ho, Wo;            /* output geometry; M = B*Ho*Wo                                   */
  int x_ld, w_ld, y_ld, res_ld;
  float alpha;
  int act;               /* NBM_ACT_*                                                      */
  int shift_per_row;     /* 1: shift is indexed by m (row) instead of n                    */
  const float* up;       /* optional [B][up_H][up_W][N]: y += bilinear_align_corners(up -> Ho x Wo), added after
                            shift/residual and before act -- the FPN top-down merge fused into the lateral 1x1
                            (fpn.py:143-144); needs the 16-byte epilogue (N % 4 == 0, aligned) and groups == 1  */
  int up_H, up_W;
  /* Listed rows (optional; 1x1 / stride 1 / pad 0, one group, Cin % 32 == 0, Cin <= 256, N > 64, 16-byte epilogue): the GEMM
   * runs over rows_count (% 128 == 0) listed pixels of the dense maps instead of all B*H*W of them; x, y, residual and `up`
   * keep their dense addressing, pixels that are not listed are neither read nor written.  Used for the lateral 1x1 +
   * top-down merge that feeds a demand-driven FPN output convolution (nbm_wino23_conv_fused_tiles).
   *   rows_mode 1: rows[m] = pixel index b*H*W + y*W + x, ascending, -1 = none (only at the end of the list);
   *   rows_mode 2: rows[m / 16] = linear 2x2-tile id b*TH*TW + ty*TW + tx (-1 = none), row m = pixel m % 16 of that tile's
   *                4x4 input patch (the list nbm_roi_tiles builds; rows_count = 16 x its entries);
   *   rows_blocks: optional device scalar, number of leading 128-entry blocks of `rows` that are filled. */
  const int* rows;
  const int* rows_blocks;
  int rows_mode, rows_count, rows_TH, rows_TW;
  /* Optional producer mask [M][mask_ld] (not with `rows`): y is zeroed where mask <= 0, after everything else -- the ReLU of the layer
   * whose data gradient this launch computes (what autograd derives for reference train.py:212 through torchvision's Bottleneck): with
   * the transposed, BatchNorm-scaled weights as `w` and the incoming gradient as `x`, a 1x1 data gradient IS this forward GEMM, which
   * lets it take the split-bf16 form below (NBM_SPLIT_BF16=1; ops.conv_dgrad routes it).  One group; the 16-byte epilogue needs it
   * 16-byte aligned with mask_ld % 4 == 0 (otherwise the scalar epilogue runs). */
  const float* mask;
  int mask_ld;
  /* Optional (round 5): one BIT per output element, (stored y > 0), 32 channels per word -- bits_out[m * (N / 32) + n / 32], bit
   * 8 (n % 4) + (n % 32) / 4 (the four comparison masks of a 16-byte lane group side by side: no bit shuffling in the epilogue) --
   * written by the same epilogue (vector epilogue, N % 32 == 0, one group, not with `rows`; otherwise NBM_EUNSUPPORTED).  The ReLU mask of
   * the consumer's data gradient (nbm_bwd_desc.mask_bits) at 1/32 of the bytes of re-reading y. */
  unsigned* bits_out;
} nbm_gemm_desc;

/* Environment switch, read on every call: NBM_SPLIT_BF16=1 runs the deep-K launches (K = kh*kw*Cin > 256, N > 64, Cin % 32 == 0,
 * 16-byte epilogue; chosen by the LAYER, never by the number of rows) on the bf16 matrix pipe with each fp32 operand split into three
 * bf16 terms and six products accumulated in fp32 (csrc/igemm_split.hip): same interface, fp32-accurate results in another
 * summation order.  Default: the fp32 matrix instruction. */
int nbm_gemm_conv(const nbm_gemm_desc* d, void* stream);

/* Winograd transforms for 3x3 / stride 1 / pad 1 convolutions (the FPN output convolutions, fpn.py:137,145, and their data /
 * weight gradients).  m = 2: F(2x2,3x3), 16 transformed planes, used by the forward pass (error ~3e-6, proposals stay
 * bit-identical to the reference); m = 4: F(4x4,3x3) with interpolation points {0, 1, -1, 1/2, -2, inf}, 36 planes, used by
 * the two backward convolutions only (error ~2e-5).  V[xi][t][c] = (B^T d B)[xi] of the (m+2)x(m+2) input patch of tile t
 * (zero padded); the (m+2)^2 GEMMs M[xi] = V[xi] x U[xi]^T run through nbm_gemm_conv with groups = (m+2)^2;
 * y = act((A^T M A) * scale + shift), zeroed where mask <= 0 (scale / shift / mask may be NULL: FrozenBN + ReLU of the ResNet
 * 3x3 layers, the producer's ReLU mask of a data gradient).  T = B*ceil(H/m)*ceil(W/m) tiles; V: [(m+2)^2][T][C], M: [(m+2)^2][T][N].
 * residual [B][H][W][N] (optional, m = 4 only): added after the epilogue -- the gradient that another consumer of the same
 * tensor has already produced (what autograd would otherwise add in a pass of its own). */
int nbm_wino_input(const float* x, int B, int H, int W, int C, float* V, int m, void* stream);
int nbm_wino_output(const float* M, const float* scale, const float* shift, const float* mask, int relu, int B, int H, int W,
                    int N, float* y, int m, const float* residual, void* stream);
/* weight gradient side: dM[xi][t][n] = (A g A^T)[xi] of the m x m output-gradient tile; dU[xi] = dM[xi]^T V[xi] through
 * nbm_conv_wgrad (groups = (m+2)^2) and dW = G^T dU G on the host.  bias_grad [N] (may be NULL): += sum over pixels of g. */
int nbm_wino_outgrad(const float* g, int B, int H, int W, int N, float* dM, float* bias_grad, int m, void* stream);

/* Weight side of the Winograd convolutions (csrc/winograd.hip).
 *   nbm_wino_weight:      g [N][C][3][3] (checkpoint layout) -> U [(m+2)^2][N'][C'] = G g' G^T in float64, rounded once.
 *                         transposed = 0: g' = g.  transposed = 1: the data-gradient convolution's weights, g'[c][n][a][b] =
 *                         scale[n] g[n][c][2-a][2-b] (kernel rotated by 180 degrees, channel roles swapped, optional
 *                         per-output-channel FrozenBN scale folded in), N' = C, C' = N.
 *   nbm_wino_weight_grad: dU [(m+2)^2][N][C] -> dW [N][C][3][3] = row_scale[n] G^T dU G (row_scale optional). */
int nbm_wino_weight(const float* g, const float* scale, int N, int C, int transposed, int m, float* U, void* stream);
int nbm_wino_weight_grad(const float* dU, const float* row_scale, int N, int C, int m, float* dW, void* stream);

/* Winograd F(2x2,3x3) forward convolution in two launches (csrc/wino_fused.hip):
 *   nbm_wino23_rows:       x [B][H][W][C] -> R [4][B][TH][WP][C], TH = ceil(H/2), WP = 2 ceil(W/2) + 2: the four row
 *                          combinations (B^T d)_i of every tile row as zero-bordered image rows (2x the input);
 *   nbm_wino23_conv_fused: R, U [16][N][C] (G g G^T) -> y [B][H][W][N] = mask(relu(conv * scale + shift)); the column half
 *                          of the input transform, the 16 transformed-domain GEMMs (fp32 MFMA) and the output transform
 *                          run in one kernel.  scale / shift / mask optional.  C % 32 == 0, C >= 64, N % 4 == 0.
 *                          variant: 0 = automatic, 128 / 64 = channel-tile width. */
int nbm_wino23_rows(const float* x, int B, int H, int W, int C, float* R, void* stream);
int nbm_wino23_conv_fused(const float* R, const float* U, const float* scale, const float* shift, const float* mask,
                          int relu, int B, int H, int W, int C, int N, float* y, int variant, void* stream);

/* Demand-driven evaluation of a 3x3 convolution: only the listed 2 x 2 output tiles are computed.  The finest FPN output map
 * (fpn.py:145, level P1 at 188x512) has two consumers in the reference: the RPN's stride-8 depthwise convolution
 * (layers.py:62-65,81: a fixed 3x3-every-8 pixel pattern) and the RoI pooling of the RoIs assigned to that level
 * (layers.py:408-417,464-467); every other pixel of the map is computed by the reference and never read.
 *   tiles:     n_entries int32 (n_entries % 128 == 0): linear tile id b * TH * TW + ty * TW + tx, ascending inside each
 *              128-entry block, -1 = none (only at the end of a block); the tiles of a block lie within 8 consecutive images.
 *   n_blocks:  optional device scalar: number of leading 128-entry blocks that are filled (the rest is not read).
 *   blk_info:  optional, one word per 128-entry block: bits 0-15 = the transformed-domain planes xi = 4 i + j to compute,
 *              bits 16-19 = the output pixels 2 p + q of every tile of the block to store.  A tile of which the reader only
 *              wants the second output row (the stride-8 pattern enters most tiles through one row or one pixel) needs the
 *              planes i = 1..3 only: 12 or 9 GEMMs instead of 16; the pixels it does store are bit-identical.
 *   nbm_wino23_rows_tiles:       row half of the input transform for the four columns of every listed tile;
 *   nbm_wino23_conv_fused_tiles: as nbm_wino23_conv_fused, writes ONLY the pixels of the listed tiles;
 *   nbm_roi_tiles:               builds such a list on the device from the RoI windows (exactly the windows nbm_roi_pool
 *                                reads) of pyramid level `level`, without the tiles flagged in skip [TH * TW] (optional);
 *                                dilate > 0: the tiles within `dilate` pixels of a window instead (where the data gradient of
 *                                a 3x3 convolution of those windows is non-zero);
 *                                tiles must hold B * ceil(TH * TW / 128) * 128 entries; writes *n_blocks. */
int nbm_wino23_rows_tiles(const float* x, int B, int H, int W, int C, const int* tiles, int n_entries, const int* n_blocks,
                          const unsigned* blk_info, float* R,
                          int skip_pattern_stride /* S > 0: the pixels of the 3x3 / stride-S / pad-1 pattern read as zeros (x is a
                                                     gradient map whose pattern share the cell transforms take, see below) */,
                          void* stream);
int nbm_wino23_conv_fused_tiles(const float* R, const float* U, const float* scale, const float* shift, const float* mask,
                                int relu /* bit 0: ReLU; bit 1: ADD the result to y instead of storing it (a tile may then be
                                            listed only once); bit 2: y is COMPACT, [n_entries][2][2][N] in list order (pixels
                                            outside the image are not written) instead of the map */,
                                int B, int H, int W, int C, int N, float* y, const int* tiles, int n_entries,
                                const int* n_blocks, const unsigned* blk_info, void* stream);
int nbm_roi_tiles(const float* rois, const int* n_roi, int B, int roi_cap, int n_levels, int level, const int* fh,
                  const int* fw, const unsigned char* skip, int dilate, int* tiles, int* n_blocks,
                  int per_image /* n_roi[b] instead of n_roi[0]: see nbm_rpn_select */, void* stream);
/* Weight gradient of such a demand-driven convolution: the gradient wrt its output is zero outside the tiles that were
 * read, so only those tiles enter dU[xi] = dM[xi]^T V[xi].  F(2x2,3x3) transforms of the listed tiles (entry -1 = a zero
 * row) into COMPACT operands V [16][n_list][C] and dM [16][n_list][N] (+ bias gradient [N], optional, accumulated); the 16
 * TN GEMMs are nbm_conv_wgrad with groups = 16 over n_list rows.  plane_mask (optional, one word per ENTRY, bits 0-15 = planes
 * xi): the other planes of that entry are written as zeros -- planes the forward pass skipped (their input pixels may never
 * have been computed), or planes that a second entry of the same tile contributes; bit 16 = this entry's pixels enter the
 * bias gradient (a tile that is listed twice sets it once).  skip_pattern_stride S > 0: the pixels of the 3x3 / stride-S /
 * pad-1 reader pattern are read as zeros (their share of the gradient goes through the cell transforms below). */
int nbm_wino23_input_tiles(const float* x, int B, int H, int W, int C, const int* tiles, int n_list, const unsigned* plane_mask,
                           float* V, void* stream);
int nbm_wino23_outgrad_tiles(const float* g, int B, int H, int W, int N, const int* tiles, int n_list, const unsigned* plane_mask,
                             float* dM, float* bias_grad, int skip_pattern_stride, void* stream);
/* Backward pass of the same convolution over the PATTERN pixels of its output gradient (what autograd derives for
 * fpn.py:145 from the RPN's stride-S depthwise convolution, layers.py:62-65,81): the gradient is one 3x3 block per S x S cell
 * (rows S*oy-1 .. S*oy+1, cells = output pixels of the 3x3 / stride-S / pad-1 reader), so per cell the data gradient is the
 * FULL convolution block * kernel (a 5x5 patch) and the weight gradient the correlation of the 5x5 input patch with the
 * block: Toom-Cook with the points {0, 1, -1, 2, -2}, 25 multiplications per (cell, n, c) instead of 81 (csrc/cellwino.hip):
 *   nbm_cell_outgrad       g [B][H][W][N] -> Vg [25][cells][N] = E blk E^T (+ bias_grad [N] += sum of the block pixels, optional)
 *   nbm_cell_input         x [B][H][W][C] -> Vx [25][cells][C] = Vinv^T patch Vinv
 *   nbm_cell_dgrad_output  M [25][cells][C] (= Vg_xi U_xi^T, nbm_gemm_conv with groups = 25, U = E w E^T from the host)
 *                          -> the 5x5 pixels of every cell in gx [B][H][W][C] (parity_class < 0: stored, patches do not overlap:
 *                          S >= 5, other pixels are not written; parity_class 0..3, S >= 3: the cells with
 *                          (oy & 1, ox & 1) == (class >> 1, class & 1) only, ADDED to gx -- four launches accumulate the
 *                          overlapping patches of a stride-3 / 4 pattern)
 * the weight gradient's 25 TN GEMMs dU_xi = Vg_xi^T Vx_xi are nbm_conv_wgrad with groups = 25; dW = E^T dU E on the host. */
int nbm_cell_outgrad(const float* g, int B, int H, int W, int N, int stride, float* Vg, float* bias_grad, void* stream);
/* Kernel side of the same algorithm (float64 arithmetic on the device, one rounding; replaces the reference-side autograd of
 * fpn.py:137,145's weights through torch.einsum):
 *   nbm_cell_weight       w [N][C][3][3] (checkpoint layout) -> U = E w E^T as U_nc [25][N][ld] (columns 0..C-1: B operand of the
 *                         FORWARD plane GEMMs) and / or U_cn [25][ldt][N] (rows 0..C-1: B operand of the data-gradient GEMMs); either
 *                         pointer may be NULL
 *   nbm_cell_weight_fold  the deferred lateral (fpn.py:143-144: x = alpha W_lat t + b + up(x1)) folded into those operands:
 *                         alpha U W_lat, W_lat [C][Cin] with row pitch wl_ld, -> columns C..C+Cin-1 of U_nc / rows C..C+Cin-1 of U_cn
 *   nbm_cell_weight_grad  dU [25][N][ld] (columns 0..C-1) -> dW [N][C][3][3] = E^T dU E */
int nbm_cell_weight(const float* w, int N, int C, float* U_nc, int ld, float* U_cn, int ldt, void* stream);
int nbm_cell_weight_fold(const float* w, const float* w_lat, int wl_ld, int N, int C, int Cin, float alpha, float* U_nc, int ld,
                         float* U_cn, int ldt, void* stream);
int nbm_cell_weight_grad(const float* dU, int N, int C, int ld, float* dW, void* stream);
int nbm_cell_input(const float* x, int B, int H, int W, int C, int stride, float* Vx, int ld /* row pitch of Vx, >= c_off + C */,
                   int c_off /* channel offset written */, void* stream);
/* the same transform of patches that are not in memory: pixel = bilinear_align_corners(x1 [B][Hc][Wc][C])[pixel] + bias[C] (the
 * top-down merge of fpn.py:143-144 without its lateral term), 0 outside the image */
int nbm_cell_input_up(const float* x1, const float* bias, int B, int H, int W, int C, int Hc, int Wc, int stride, float* Vx, int ld,
                      int c_off, void* stream);
/* The untransformed forms: Vx [25][cells][ld], plane 5 a + l = pixel (row a, column l) of every cell's 5x5 patch (0 outside the image)
 * -- the operand of a 5x5 / stride S convolution written as 25 taps.  The RPN's strided reader (layers.py:62-65: depthwise 3x3 / stride S
 * -> 1x1 -> BatchNorm -> SiLU) composed with the linear output convolution in front of it (fpn.py:137,145) is ONE such convolution: in
 * evaluation mode the pattern pixels of the on-demand FPN levels are never formed (ondemand.rpn_composite). */
int nbm_cell_patches(const float* x, int B, int H, int W, int C, int stride, float* Vx, int ld, int c_off, void* stream);
int nbm_cell_patches_up(const float* x1, const float* bias, int B, int H, int W, int C, int Hc, int Wc, int stride, float* Vx, int ld,
                        int c_off, void* stream);

/* Rectangle copy between a strided array and packed rows (no counterpart in the reference: plumbing of the composed RPN reader in
 * training mode, DESIGN 4h -- the border-cell classes of reference layers.py:22-29's depthwise padding are rectangles [r0, r1) x [c0, c1)
 * of the OH x OW cell grid of every (plane, image)): for o < n_outer, r < n_rows:  packed[(o * n_rows + r) * width ...] <->
 * strided[o * outer_pitch + r * row_pitch ...], `width` floats each.  to_strided = 0: packed <- strided (zero_strided = 1 also clears
 * the source rectangle); to_strided = 1: strided <- packed.  All counts in floats, multiples of 4; 16-byte aligned pointers. */
int nbm_copy_rect(float* strided, float* packed, int64_t n_outer, int64_t outer_pitch, int n_rows, int64_t row_pitch, int64_t width,
                  int to_strided, int zero_strided, void* stream);
int nbm_cell_dgrad_output(const float* M, int B, int H, int W, int C, int stride, float* gx, int parity_class,
                          int ld /* row pitch of M, >= c_off + C */, int c_off /* first channel of M read */,
                          float* bias_grad /* optional [C]: += sum of the patch pixels written (inside the image) */, void* stream);
/* FORWARD of the pattern pixels with the same 25 products per cell (the correlation form, F(3x3,3x3)): nbm_cell_input of the
 * input, M_xi = Vx_xi U_xi^T (U = E w E^T as [25][N][C]), then blk = E^T M E + bias into the 3x3 pattern block of every cell of
 * y [B][H][W][N] (other pixels are not written). */
int nbm_cell_output(const float* M, const float* bias, int B, int H, int W, int N, int stride, float* y, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Spectrogram front end.  Replaces File_Processor.load/spectrogram/split_power_spec
 * (nbm_model/nbm_datasets/prepare_dataset.py:160-184, 228-252, 255-294) incl. librosa.stft and the
 * ffmpeg 2x resample (third-party).
 */

/* PCM16 -> centre-padded fp32 waveform rows of one STFT chunk.  The 44.1 kHz signal y of row b is pcm/32768
 * (upsample=0, length n) or its 2x half-band integer interpolation (upsample=1, length 2n; hq = 16 Q15 odd-phase taps,
 * oracle/frontend_ref.py:upsample2x_coeffs); the piece y[first, first+count) is written to out[b][lead ...].
 * pad_mode 0 ('constant', librosa >= 0.10): everything else in the row [0, out_ld) is 0.  pad_mode 1 ('reflect',
 * librosa <= 0.9): the `lead` samples in front of and behind the piece mirror it (np.pad(mode='reflect'), needs
 * lead < count), the rest of the row is 0. */
int nbm_pcm16_to_wave(const int16_t* pcm, int64_t pcm_ld, int batch, int n, int upsample,
                      const int32_t* hq, int64_t first, int64_t count, float* out, int64_t out_ld, int lead,
                      int pad_mode, void* stream);

/* fp32 samples at any rate -> centre-padded fp32 waveform rows of one STFT chunk (same row layout and padding as
 * nbm_pcm16_to_wave).  L = M = 1: the piece [first, first+count) of the samples themselves (float / 24- / 32-bit files at
 * 44.1 kHz, librosa.load, prepare_dataset.py:162).  Otherwise the 44.1 kHz signal is the rational L / M polyphase
 * resampling of x (44100 / sr = L / M reduced): y[m] = sum_k taps[(m M) mod L][k] x[floor(m M / L) - T/2 + 1 + k], taps
 * float64 [L][T] (prepare_dataset.py:resample_taps), float64 accumulation in index order; quant16 = 1 rounds y to the
 * 16-bit grid like the reference's `ffmpeg ... -acodec pcm_s16le -ar 44100` (prepare_dataset.py:175-178).  The signal has
 * ceil(n L / M) samples. */
int nbm_resample_to_wave(const float* x, int64_t x_ld, int batch, int64_t n, int L, int M, const double* taps, int T,
                         int64_t first, int64_t count, float* out, int64_t out_ld, int lead, int pad_mode, int quant16,
                         void* stream);

/* STFT magnitude in dB for n_bins bins on the fp64 MFMA (librosa.stft + np.abs + amp_to_db + crop,
 * prepare_dataset.py:228-247):
 *   db[b][f][t] = 20 log10(max(floor, | sum_n w[n] wave[b][t*hop + n] exp(-2 pi i (low_bin + f) n / n_fft) |))
 * for a symmetric window with w[0] = 0 (periodic Hann), evaluated as two real GEMMs over k = 0 .. n_fft/2 on
 * s[k] = x[k] + x[N-k] and d[k] = x[k] - x[N-k] (csrc/stft.hip).  basis: float64, MFMA fragment order
 * [basis_bin_tiles][basis_ksteps][64 lanes]{cos, sin}, lane l of (tile bt, step ks) = bin bt*16 + (l & 15),
 * k = 4 ks + (l >> 4); window folded in, the k = n_fft/2 column halved, zero beyond n_fft/2 and beyond n_bins
 * (prepare_dataset.py:dft_basis_f64 builds it).  basis_bin_tiles % 8 == 0.  wave rows are already centre-padded;
 * hop % 4 == 0.  minmax[b] = {min, max} over the row as order-preserving uint32 keys, accumulated into
 * (initialise with nbm_minmax_init). */
int nbm_minmax_init(uint32_t* minmax, int batch, void* stream);
int nbm_stft_db(const float* wave, int64_t wave_ld, int batch, int n_frames, int hop, int n_fft,
                const double* basis, int basis_bin_tiles, int basis_ksteps, int n_bins, float floor_amp,
                float* db, int64_t db_bs, int db_ld, uint32_t* minmax, void* stream);

/* (x - min)/(max - min) + window split (File_Processor.split_power_spec, prepare_dataset.py:255-294):
 * img[b][k][f][c] = norm(db[b][f][k*hop_img + c]) for k < n_img - 1; the LAST window takes its columns from
 * last_cols[c] (w_pix absolute column indices < n_frames, device memory): the host lists there the reference's chunk-end
 * cut and its stepwise reflect padding (prepare_dataset.py:window_columns). */
int nbm_spec_windows(const float* db, int64_t db_bs, int db_ld, int batch, int n_bins, int n_frames,
                     const uint32_t* minmax, float* img, int n_img, int w_pix, int hop_img,
                     const int32_t* last_cols, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Point-wise / small-window detector stages (NHWC fp32).
 */

/* y[p][c] = x[p]*w[c] + b[c]  -- BackboneBase.init_conv (backbone.py:104-105,110-113), 1 -> C. */
int nbm_init_conv(const float* x, int64_t n_pix, const float* w, const float* b, int C, float* y, void* stream);

/* Network stem in one kernel (csrc/stem.hip): y [B][Ho][Wo][64] = relu(scale * conv7x7_s2_p3(init_conv(img)) + shift) on the
 * single-channel image img [B][H][W]: BackboneBase.init_conv (1x1, 1 -> 3, + bias; backbone.py:104-105,110-113) folded into
 * torchvision's conv1 + FrozenBN + ReLU (backbone.py:131).  weff [56][64]: k = 8 r + s (s = 7: zero), weff[k][o] =
 * sum_c W1[o][c][r][s] w_init[c]; wb [64][49] = sum_c W1[o][c][r][s] b_init[c] (summed over the taps INSIDE the image for
 * border pixels), wb_full [64] its sum over all 49 taps (prepared by nets/_prep.py:stem_fold). */
int nbm_stem7x7(const float* img, int B, int H, int W, const float* weff, const float* wb, const float* wb_full,
                const float* scale, const float* shift, float* y, void* stream);

/* Weight-side gradient of the same stem (csrc/stem.hip): g [B][Ho][Wo][64] = gradient wrt the stem output, already masked by the
 * ReLU.  D [64][64] (zeroed here): D[n][8 r + s] = U[n][r][s] = sum_px g[px][n] img[pix(px) + (r, s)], D[n][56] = S[n] =
 * sum_px g[px][n];  Cb [64][49] (zeroed here): Cb[n][7 r + s] = sum of g[px][n] over the pixels whose tap (r, s) lies outside
 * the image, so that V[n][r][s] = S[n] - Cb[n][7 r + s] counts the init_conv bias only inside conv1's zero padding.  The
 * caller forms dW1 = a U + b V, da = sum W1 U, db = sum W1 V (backbone.py:104-113,131). */
int nbm_stem7x7_wgrad(const float* img, const float* g, int B, int H, int W, float* D, float* Cb, void* stream);

/* 3x3 / stride 2 / pad 1 max pooling -- torchvision ResNet `maxpool` (backbone.py:131).  idx (may be NULL; training):
 * one byte per output element = r*3+s of the first maximum in scan order, consumed by nbm_maxpool3x3s2_bwd. */
int nbm_maxpool3x3s2(const float* x, int B, int H, int W, int C, float* y, int Ho, int Wo, uint8_t* idx, void* stream);
/* `--dilation` (backbone.py:129-131, torchvision replace_stride_with_dilation on layer4): a 3x3 / dilation-2 / pad-2 convolution on an
 * even-sized map is four ordinary 3x3 / pad-1 convolutions on the parity classes of its pixels, so the dilated bottlenecks run the
 * ordinary kernels on the space-to-batch form  y[(2a+b)*B + n][u][v][c] = x[n][2u+a][2v+b][c]  (x [B][H][W][C] -> y [4B][H/2][W/2][C];
 * inverse != 0: the same map read the other way, x the packed side).  H, W even. */
int nbm_space_to_batch2(const float* x, int B, int H, int W, int C, float* y, int inverse, void* stream);
/* nn.AdaptiveAvgPool2d to exactly half the size (layers.py:84,94 on the 48x128 RPN map of the dilated level): x [B][2Ho][2Wo][C] ->
 * y [B][Ho][Wo][C], mean of each 2x2 block; and its gradient (gx = gy / 4 on the block's four pixels). */
int nbm_avgpool2x2(const float* x, int B, int Ho, int Wo, int C, float* y, void* stream);
int nbm_avgpool2x2_bwd(const float* gy, int B, int Ho, int Wo, int C, float* gx, void* stream);

/* y = bilinear_align_corners(src -> Ho x Wo) [+ add]  -- fpn.py:143-144, layers.py:35-37. */
int nbm_upsample_bilinear_add(const float* src, int B, int Hi, int Wi, int C, const float* add,
                              float* y, int Ho, int Wo, void* stream);

/* in-place row softmax of [rows][cols] with pitch ld -- self_attention.py:47, layers.py:579. */
int nbm_softmax_rows(float* x, int64_t rows, int cols, int64_t ld, void* stream);

/* depthwise 3x3, pad 1, channel multiplier `mult` (out channel o reads in channel o/mult), + bias;
 * optional FiLM: y = y*gamma + beta with film[p][0:Cout]=gamma, film[p][Cout:2Cout]=beta
 * -- DepthwiseSepConv2d.depth_wise / FiLM (layers.py:25-26,38-42). */
int nbm_dwconv3x3(const float* x, int B, int H, int W, int Cin, int mult, int stride,
                  const float* w /*[Cin*mult][9]*/, const float* bias, const float* film, int64_t film_ld,
                  float* y, int Ho, int Wo, void* stream);

/* y = x * sigmoid(x) -- nn.SiLU (layers.py:31,41). */
int nbm_silu(const float* x, float* y, int64_t n, void* stream);

/* y[r][:] = (x[r][:] - mean_r) / sqrt(var_r + eps) * w + b over rows of E floats (biased variance) -- nn.LayerNorm of the
 * Transformer_RCNN head (reference layers.py:589-651, self_attention.py:110-131). */
int nbm_layernorm(const float* x, int64_t rows, int E, const float* w, const float* b, float eps, float* y, void* stream);

/* out = softmax(scale * Q K^T) V per (batch entry, head) for short sequences (S <= 128, hd <= 64): token (s, n) is row
 * s*seq_stride + n*batch_stride of q/k/v/out (row pitches *_ld floats, head h at columns [h*hd, (h+1)*hd)); keys with
 * s >= *n_valid (device counter, may be NULL) are masked -- nn.MultiheadAttention inside the Transformer_RCNN encoder
 * (reference layers.py:613-621,645-646; self_attention.py:110-126). */
int nbm_mha_small(const float* q, const float* k, const float* v, int q_ld, int k_ld, int v_ld, float* out, int out_ld,
                  int S, int N, int nhead, int hd, int64_t seq_stride, int64_t batch_stride, const int32_t* n_valid,
                  float scale, void* stream);

/* softmax over the (bg, fg) pair of every anchor: x[p][2a], x[p][2a+1] (pitch ld) -> y (pitch y_ld)
 * -- layers.py:90. */
int nbm_pair_softmax(const float* x, int64_t n_pix, int n_anchor, int x_ld, float* y, int y_ld, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Proposal / RoI / detection stages (integer + sort work; bit-exact vs the oracle).
 */

/* ProposalLayer decode (layers.py:266-285, nets_utils.py:169-186): per image and anchor (K-major):
 * boxes[b][i][4] (rounded, clipped), keys[b][i] = order-preserving uint32 of the fg score or 0 when the
 * box is smaller than min_size; keep_count[b] = #kept. cls: [B][K][2A] softmaxed, reg: [B][K][4A]. */
int nbm_rpn_decode(const float* cls, const float* reg, const float* anchors /*[K*A][4]*/, int B, int KA,
                   int n_anchor /*A: anchors per location*/, int img_w, int img_h, int min_size, float* boxes,
                   uint32_t* keys, int* keep_count, void* stream);

/* Top-N selection by (score desc, index asc) among kept anchors, N = min(top_n, min_b keep_count[b]);
 * writes sel_boxes[b][cap][4], sel_scores[b][cap], n_sel[0] = N (0 if N < fail_below: "RPN failed"
 * layers.py:287-290).  -- layers.py:292-297.  cap >= top_n, cap power of two <= 4096.
 * per_image != 0: the B images are B independent batches of one (bulk inference over files that the reference CLI
 * runs one per model call, nbm_detect.py:24-28): N_b = min(top_n, keep_count[b]), n_sel[b] = N_b -- no image's
 * proposal count depends on its launch-mates.  The same flag on nbm_nms_batched / nbm_roi_pool (desc field) /
 * nbm_roi_tiles / nbm_rcnn_post makes them read their count as n[b] instead of n[0]. */
int nbm_rpn_select(const float* boxes, const uint32_t* keys, const int* keep_count, int B, int KA,
                   int top_n, int fail_below, int cap, float* sel_boxes, float* sel_scores, int* n_sel,
                   int per_image, void* stream);

/* Greedy NMS in the given order (suppress IoU >= thresh, +1 pixel convention) then the batch-coupled
 * truncation R = min(post_n, min_b #keep_b) -- nets_utils.py:189-245.  n_in[0] boxes per image.
 * Workspaces: mask_ws B*cap*(cap/64) uint64, keep_ws B*(cap+1) int32.  Writes rois[b][post_n][4],
 * roi_scores[b][post_n], n_out[0] = R.  cap: multiple of 64, <= 4096.
 * per_image != 0: n_in[b] boxes in image b, R_b = min(post_n, #keep_b), n_out[b] = R_b. */
int nbm_nms_batched(const float* boxes, const float* scores, const int* n_in, int B, int cap, float thresh,
                    int post_n, uint64_t* mask_ws, int* keep_ws, float* rois, float* roi_scores, int* n_out,
                    int per_image, void* stream);

/* ROIPooling (layers.py:406-497): level assignment, window, 2x2 adaptive average of the FPN map and of
 * the separable positional encoding.  fmaps: 5 device pointers (NHWC, C channels); pe_f [img_h][C/2],
 * pe_t [img_w][C/2].  n_roi[0] RoIs per image out of roi_cap slots.  Outputs NHWC [B*roi_cap][2][2][C]. */
typedef struct nbm_roi_desc {
  const float* fmap[5];
  int fh[5], fw[5];
  int n_levels, C;
  const float* rois;     /* [B][roi_cap][4] */
  const int* n_roi;      /* device scalar, or [B] with n_roi_per_image */
  int B, roi_cap;
  const float* pe_f; const float* pe_t; int img_h, img_w;
  float* pool; float* pe; int* level;
  int n_roi_per_image;   /* != 0: n_roi[b] RoIs in image b (see nbm_rpn_select) */
} nbm_roi_desc;
int nbm_roi_pool(const nbm_roi_desc* d, void* stream);

/* FastRCNN eval post-processing (layers.py:688-776) for B images, n_roi[0] RoIs each:
 * class arg-max, per-class delta gather, decode+clip, sort by score, drop background, class-agnostic
 * NMS, per-class NMS (top proposal_number) and score > min_score.  Output rows sorted by (class asc,
 * score desc): det[b][cap][6] = {class, x1, y1, x2, y2, score}, n_det[b]. */
int nbm_rcnn_post(const float* rois, const int* n_roi, int B, int roi_cap, const float* bbox_reg,
                  const float* bbox_cls, int n_cls1 /*1+num_classes*/, int img_w, int img_h,
                  float nms_thresh, float min_score, int proposal_number, float* det, int* n_det,
                  int per_image /* n_roi[b] */, void* stream);

/* Training, AnchorTargetLayer (layers.py:150-179, nets_utils.py:103-126): IoU (inclusive-pixel convention, the reference's fp32
 * operations in its order) of every anchor inside the image with the n_gt[b] boxes of gt[b]; amx[b][a] = index of the FIRST best box,
 * lab[b][a] = the anchor's label BEFORE the random subsampling (0: best overlap < neg_t; 1: best overlap >= pos_t, or the anchor is a
 * best anchor -- ties included -- of a box whose best overlap is > 0; else -1); flag[b] != 0: a NaN / negative overlap occurred
 * (degenerate box), the caller recomputes that image on the host.  The NumPy RNG draws stay on the host (layers.py:183-197).
 * nbm_proposal_iou below: ProposalTargetLayer (layers.py:320-330, nets_utils.py:103-126): rows j < R = proposals rois[b][j], rows R + i = the
 * image's own ground-truth boxes gt[b][i] (the reference appends them to the proposals); for every row the IoU (inclusive-pixel
 * convention, the reference's fp32 operations in its order) with the n_gt[b] <= G boxes of gt[b]: mx[b][j] = best overlap,
 * asg[b][j] = index of the FIRST best box.  The thresholds and the NumPy RNG draws stay on the host (layers.py:332-376). */
int nbm_anchor_targets(const float* anchors /*[n_in][4]: the anchors inside the image*/, int n_in, const float* gt /*[B][G][4]*/,
                       const int* n_gt /*[B]*/, int B, int G /* <= 256 */, float neg_t, float pos_t, signed char* lab /*[B][n_in]*/,
                       short* amx /*[B][n_in]*/, int* flag /*[B]*/, void* stream);
int nbm_proposal_iou(const float* rois /*[B][R][4]*/, const float* gt /*[B][G][4]*/, const int* n_gt /*[B]*/, int B, int R, int G,
                     float* mx /*[B][R+G]*/, int* asg /*[B][R+G]*/, void* stream);

/* ================================================================================================
 * Training path: backward implicit GEMMs, point-wise gradients, train-mode BatchNorm, optimiser.
 * Replaces what torch.autograd derives for the reference's train step (train.py:205-217):
 * conv/linear/matmul backward, ReLU/SiLU/max-pool/bilinear/softmax/BatchNorm backward,
 * torch.nn.utils.clip_grad_norm_ (train.py:213-214) and torch.optim.AdamW.step (train.py:215,302-303).
 */
typedef struct nbm_bwd_desc {
  const float* g;        /* upstream gradient, NHWC [B][Ho][Wo][N] rows (pitch g_ld)                         */
  const float* w;        /* dgrad: KRSC weights [N][w_ld]                                                    */
  const float* x;        /* wgrad: forward input NHWC [B][H][W][Cin] (pitch x_ld)                            */
  float* out;            /* dgrad: dX [B*H*W][out_ld];  wgrad: dW [N][out_ld] (KRSC, += with fp32 atomics)   */
  const float* a_scale;  /* dgrad: per-n multiplier of g (needs N % 32 == 0) or NULL                         */
  const float* row_scale;/* wgrad: per-n multiplier of the dW rows or NULL                                   */
  const float* residual; /* dgrad: added to dX (gradient accumulation) or NULL                               */
  const float* mask;     /* dgrad: dX := 0 where mask <= 0 (ReLU of the producing layer) or NULL             */
  int64_t g_gs, w_gs, x_gs, out_gs, res_gs;
  int groups;
  int B, H, W, Cin, N, kh, kw, stride, pad, Ho, Wo;   /* geometry of the FORWARD convolution                */
  int g_ld, w_ld, x_ld, out_ld, res_ld, mask_ld;
  float alpha;
  float* bias_grad;      /* wgrad: [groups][N], += sum_m g[m][n] (the bias gradient rides along; zeroed by the caller) or NULL */
  const float* residual2;/* dgrad (groups == 1): a HALF-resolution map [B][ceil(H/2)][ceil(W/2)][Cin] (pitch res2_ld), added to dX at
                            the pixels with even iy and ix -- the data gradient of the block's 1x1 / stride-2 shortcut
                            (torchvision Bottleneck.downsample), which is non-zero only there -- or NULL                    */
  int res2_ld;
  const unsigned* mask_bits;/* dgrad (round 5): the ReLU mask as BITS instead of `mask` -- [M][Cin / 32] words, bit c % 32 of word
                            m * (Cin / 32) + c / 32 (bit order of nbm_gemm_desc.bits_out) set where the producer's output was > 0 -- or NULL.
                            Cin % 32 == 0, one group; takes precedence over `mask`.                                          */
} nbm_bwd_desc;

/* dX[b][iy][ix][c] = alpha * sum_{r,s,n} g[b][(iy+pad-r)/stride][(ix+pad-s)/stride][n] * a_scale[n] * W[n][r][s][c]
 * (+ residual, ReLU mask).  stride 2: the pixels are processed by parity class of (iy+pad, ix+pad), each class visiting
 * only the taps that reach it (no multiply-by-structural-zero work).  g rows must be readable (zero padded) up to ceil(N/32)*32 floats.  Also the plain
 * C[M][Cin] = A[M][N] * B[N][Cin] ("NN") GEMM with H = M, W = 1, 1x1. */
int nbm_conv_dgrad(const nbm_bwd_desc* d, void* stream);
/* dW[n][(r,s,c)] += alpha * row_scale[n] * sum_m g[m][n] * x[pix(m)+(r,s)][c]; dW must be zeroed by the caller.
 * Also the plain C[N][Cin] += A[M][N]^T * B[M][Cin] ("TN") GEMM. */
int nbm_conv_wgrad(const nbm_bwd_desc* d, void* stream);

int nbm_relu_bwd(const float* gy, const float* y, float* out, int64_t n, void* stream);
int nbm_silu_bwd(const float* gy, const float* x, float* out, int64_t n, void* stream);
/* out[i] = alpha*a[i] + beta*b[i % b_period] (b may be NULL; b_period 0: b has n elements): gradient accumulation, the
 * doubled identity levels of SAPyramid (self_attention.py:76), `features + pos` of --add_posenc (nbm_model.py:45-46) */
int nbm_axpby(const float* a, const float* b, float* out, float alpha, float beta, int64_t n, int64_t b_period, void* stream);
/* BiFPN FusionModule (fpn.py:20-30): out = (sum_i relu(w_i) x_i) / (sum_i relu(w_i) + 1e-4), 2 inputs (x2 NULL) or 3;
 * `weights` is the raw learnable parameter on the device.  Backward: gx_i (any may be NULL) and gw[i] += d/dw_i (zeroed
 * by the caller). */
int nbm_weighted_sum(const float* x0, const float* x1, const float* x2, const float* weights, float* out, int64_t n,
                     void* stream);
int nbm_weighted_sum_bwd(const float* x0, const float* x1, const float* x2, const float* weights, const float* g, float* gx0,
                         float* gx1, float* gx2, float* gw, int64_t n, void* stream);
/* Targeted re-zeroing of a gradient map [B][H][W][C] (C % 4 == 0) that is zero except for known footprints -- the gradient of
 * a demand-driven FPN map (training): what RoiPooling's backward scattered into the windows of level `lvl` (the windows of
 * nbm_roi_pool_bwd), what the stride-`stride` depthwise convolution's backward added on its 3x3 blocks (layers.py:62-65), and
 * the 2x2 tiles of a list (layout of nbm_roi_tiles; n_blocks optional device count).  The map can then be kept across steps
 * instead of being filled (12.6 + 18.9 GB per step at B = 128) -- there is no reference counterpart, autograd allocates. */
int nbm_zero_roi_windows(float* g, int H, int W, int C, const float* rois /*[B][n_roi][4]*/, const int* level /*[B][n_roi]*/,
                         int B, int n_roi, int lvl, void* stream);
int nbm_zero_pattern(float* g, int B, int H, int W, int C, int stride, void* stream);
int nbm_zero_tiles(float* g, int B, int H, int W, int C, const int* tiles, int n_entries, const int* n_blocks, void* stream);
/* the 2x2 tiles of such a list <-> compact [n_entries][2][2][C] in list order (the compact output layout of
 * nbm_wino23_conv_fused_tiles): gather reads the map (zeros outside the image), scatter_add adds the compact values into it (every
 * tile listed once).  The RoI share of the finest level's lateral gradients runs on these compact operands. */
int nbm_tiles_gather(const float* map, int B, int H, int W, int C, const int* tiles, int n_entries, const int* n_blocks, float* compact,
                     void* stream);
int nbm_tiles_scatter_add(float* map, int B, int H, int W, int C, const int* tiles, int n_entries, const int* n_blocks,
                          const float* compact, void* stream);
/* out[n] = sum_m g[m][n] (bias gradients) */
int nbm_colsum(const float* g, int64_t M, int N, int ld, float* out, void* stream);
int nbm_maxpool3x3s2_bwd(const uint8_t* idx, const float* gy, float* gx, int B, int H, int W, int C, int Ho, int Wo,
                         const float* residual /* [B][H][W][C], optional: added to gx (another consumer's gradient) */,
                         const float* mask /* [B][H][W][C], optional: gx *= (mask > 0), after the addition -- the pooled tensor is a
                                              ReLU output (backbone.py:131) and its producer need not mask the gradient again */,
                         void* stream);
/* out = gy * (y > 0 ? 1 : slope) -- nn.LeakyReLU backward (Transformer_RCNN embeddings / DETR-style feed-forward) */
int nbm_leaky_relu_bwd(const float* gy, const float* y, float* out, float slope, int64_t n, void* stream);
/* nn.LayerNorm backward over rows of E <= 1024 floats; gw, gb (E floats each) must be zeroed, they are added to */
int nbm_layernorm_bwd(const float* x, const float* w, const float* g, int64_t rows, int E, float eps, float* gx, float* gw,
                      float* gb, void* stream);
/* backward of nbm_mha_small: gq/gk/gv get every VALID token row written (pre-zero them when *n_valid < S);
 * workspace: 2*S*S floats per (batch entry, head) */
int nbm_mha_small_bwd(const float* q, const float* k, const float* v, const float* go, int q_ld, int k_ld, int v_ld,
                      int go_ld, float* gq, float* gk, float* gv, int gq_ld, int gk_ld, int gv_ld, float* workspace, int S,
                      int N, int nhead, int hd, int64_t seq_stride, int64_t batch_stride, const int32_t* n_valid,
                      float scale, void* stream);
/* gradient of the bilinear (align_corners) up-sampling of fpn.py:143-144 wrt the coarse map [B][Hi][Wi][C], gather form.
 * pattern_stride = S >= 5: the caller guarantees that gy [B][Ho][Wo][C] is zero outside the 5x5 patches around the 3x3 / stride-S /
 * pad-1 pattern (rows and columns o*S - 2 .. o*S + 2): those rows / columns are not read (61 % of the map at S = 8); 0: every
 * pixel is read.  nbm_tiles_upsample_bilinear_bwd_add: the same operator in scatter form for the 2x2 tiles of a list (layout of
 * nbm_roi_tiles, ids relative to [B][Ho][Wo]) held as compact [n_entries][2][2][C] values (nbm_wino23_conv_fused_tiles), ADDED
 * into gsrc with fp32 atomics -- the RoI share of the same gradient, a few % of the map. */
int nbm_upsample_bilinear_bwd(const float* gy, int B, int Hi, int Wi, int C, float* gsrc, int Ho, int Wo, int pattern_stride,
                              void* stream);
int nbm_tiles_upsample_bilinear_bwd_add(const float* compact, int B, int Ho, int Wo, int C, const int* tiles, int n_entries,
                                        const int* n_blocks, float* gsrc, int Hi, int Wi, void* stream);
int nbm_softmax_rows_bwd(const float* p, const float* gp, float* out, int64_t rows, int cols, float alpha, void* stream);
int nbm_pair_softmax_bwd(const float* y, const float* gy, float* gx, int64_t n_pairs, void* stream);
/* depthwise 3x3 gradients: gx (may be NULL), gw [Cout][9] and gb [Cout] (gw NULL = skip both) */
int nbm_dwconv3x3_bwd(const float* x, const float* g, const float* w, int B, int H, int W, int Cin, int mult, int stride,
                      float* gx, float* gw, float* gb, int Ho, int Wo, void* stream);
/* the data gradient of the same depthwise convolution ADDED into gx, which already holds the gradient another consumer of the
 * map sent (the RoI pooling's scatter map, nbm_roi_pool_bwd): only the pixels a tap reaches are touched -- replaces a dense
 * write, a dense zero fill and autograd's dense add for an FPN map with these two consumers */
int nbm_dwconv3x3_bwd_acc(const float* g, const float* w, int B, int H, int W, int Cin, int mult, int stride, float* gx, int Ho,
                          int Wo, void* stream);
int nbm_film_fwd(const float* z, const float* film, float* y, int64_t n_pix, int C, void* stream);
int nbm_film_bwd(const float* gy, const float* z, const float* film, float* gz, float* gfilm, int64_t n_pix, int C,
                 void* stream);
/* nn.BatchNorm2d in training mode over rows [M][C]: batch statistics (biased variance for the output, unbiased for the
 * running estimate), running stats updated in place with `momentum`.  stats_ws / red_ws: 2*C doubles. */
int nbm_bn_train_fwd(const float* x, int64_t M, int C, const float* w, const float* b, float eps, float momentum,
                     float* run_mean, float* run_var, double* stats_ws, float* mean, float* invstd, float* y, void* stream);
int nbm_bn_train_bwd(const float* g, const float* x, int64_t M, int C, const float* mean, const float* invstd,
                     const float* w, double* red_ws, float* gx, float* gw, float* gb, void* stream);
/* ROIPooling backward: scatter-add of gpool [B*n_roi][2][2][C] into the (zeroed) FPN gradient maps */
int nbm_roi_pool_bwd(float* const* gfmap_host, const int* fh_host, const int* fw_host, int n_levels, int C,
                     const float* rois, const int* level, int B, int n_roi, const float* gpool, void* stream);
/* out[0] += sum g^2 (double); then AdamW on a flat range with the clip coefficient min(1, max_norm/(sqrt(*sqnorm)+1e-6))
 * evaluated on device (sqnorm NULL or max_norm <= 0: no clipping). */
int nbm_sqnorm_accum(const float* g, int64_t n, double* out, void* stream);
int nbm_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, const double* sqnorm, float max_norm, void* stream);

/* ---- training-input stage: Img_dataset (reference nbm_datasets/image_dataset.py:36-96) ------------------------- */
/* per-image augmentation draws, filled by the host in the reference's RNG call order (image_dataset.py:59,68,71-81,90) */
struct nbm_augment_params {
  float gain;                 /* np.random.uniform(-0.1, 0.35)                                   :68 */
  float coef, denom;          /* hard-negative weight c and 1 + c for the positive image         :79-80 */
  float neg_coef, neg_denom;  /* same for the negative image                                      :82-83 */
  int32_t flags;              /* bit 0: bool_transform[0] (hard-negative mix), bit 1: [1] (low-pass) :71 */
  int32_t hard_index;         /* image of `hard` to mix in */
  int32_t pad;
  uint64_t noise_seed;        /* stream of the device counter RNG when noise_unit == NULL */
};

/* PNG scanline reconstruction (filter types 0-4, PNG spec 9.2) of `batch` inflated 8-bit greyscale images:
 * raw[b] = H lines of (1 filter byte + W bytes) -> out[b] = H*W pixels; *status != 0 afterwards <=> bad filter byte.
 * Replaces the pixel half of `imageio.imread` (image_dataset.py:44,62,77); the zlib inflate stays on the host. */
int nbm_png_unfilter_gray8(const uint8_t* raw, int64_t raw_bs, int batch, int H, int W, uint8_t* out, int64_t out_bs,
                           int32_t* status, void* stream);
/* half_std[b] = std(float32(img_b / 255), unbiased) / 2 -- the noise scale of image_dataset.py:66 */
int nbm_image_half_std_u8(const uint8_t* img, int64_t bs, int batch, int64_t n, float* half_std, void* stream);
/* out[i] = i-th N(0,1) draw of counter stream `seed` (splitmix64 + Box-Muller) -- device stand-in for torch.randn :66 */
int nbm_randn_fill(uint64_t seed, int64_t n, float* out, void* stream);
/* the whole `if self.transform:` block (:65-94) for a batch: img = u8/255 (+gain, +clamp(noise*half_std, +-0.5),
 * hard-negative mix, low-pass curve[b][row]); neg = u8/255 (hard-negative mix).  noise_unit (N(0,1) field, may be
 * NULL => device counter RNG), curve [batch][H]. */
int nbm_augment_batch(const uint8_t* pos, const uint8_t* neg, const uint8_t* hard, int batch, int H, int W,
                      const struct nbm_augment_params* params, const float* half_std, const float* noise_unit,
                      const float* curve, float* img_out, float* neg_out, void* stream);
/* out = float32(in / 255) -- `torch.Tensor(img / 255)` :45,63 (transform=False path) */
int nbm_u8_to_unit(const uint8_t* in, int64_t n, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
