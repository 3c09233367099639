"""ctypes binding of libnbm_hip.so (C ABI: include/nbm_hip.h).

There is NO fallback: if the shared library is missing or a symbol cannot be resolved the import of
the op layer fails loudly (a silent eager/CPU path would void every parity claim).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# NBM_LIB: another build of the same ABI (scripts/wino_fused_probe.py uses the -DNBM_ABLATE build, `make -C csrc ablate`)
LIB_PATH = os.environ.get('NBM_LIB') or os.path.join(_HERE, 'libnbm_hip.so')
CSRC = os.path.join(_HERE, 'csrc')

_lib = None


class GemmDesc(C.Structure):
    """struct nbm_gemm_desc (include/nbm_hip.h)."""
    _fields_ = [('x', C.c_void_p), ('w', C.c_void_p), ('y', C.c_void_p),
                ('scale', C.c_void_p), ('shift', C.c_void_p), ('residual', C.c_void_p),
                ('x_gs', C.c_int64), ('w_gs', C.c_int64), ('y_gs', C.c_int64), ('res_gs', C.c_int64),
                ('groups', C.c_int),
                ('B', C.c_int), ('H', C.c_int), ('W', C.c_int), ('Cin', C.c_int),
                ('N', C.c_int),
                ('kh', C.c_int), ('kw', C.c_int), ('stride', C.c_int), ('pad', C.c_int),
                ('Ho', C.c_int), ('Wo', C.c_int),
                ('x_ld', C.c_int), ('w_ld', C.c_int), ('y_ld', C.c_int), ('res_ld', C.c_int),
                ('alpha', C.c_float), ('act', C.c_int), ('shift_per_row', C.c_int),
                ('up', C.c_void_p), ('up_H', C.c_int), ('up_W', C.c_int),
                ('rows', C.c_void_p), ('rows_blocks', C.c_void_p),
                ('rows_mode', C.c_int), ('rows_count', C.c_int), ('rows_TH', C.c_int), ('rows_TW', C.c_int),
                ('mask', C.c_void_p), ('mask_ld', C.c_int), ('bits_out', C.c_void_p)]


class RoiDesc(C.Structure):
    """struct nbm_roi_desc (include/nbm_hip.h)."""
    _fields_ = [('fmap', C.c_void_p * 5), ('fh', C.c_int * 5), ('fw', C.c_int * 5),
                ('n_levels', C.c_int), ('C', C.c_int),
                ('rois', C.c_void_p), ('n_roi', C.c_void_p), ('B', C.c_int), ('roi_cap', C.c_int),
                ('pe_f', C.c_void_p), ('pe_t', C.c_void_p), ('img_h', C.c_int), ('img_w', C.c_int),
                ('pool', C.c_void_p), ('pe', C.c_void_p), ('level', C.c_void_p), ('n_roi_per_image', C.c_int)]


class AugmentParams(C.Structure):
    """struct nbm_augment_params (include/nbm_hip.h)."""
    _fields_ = [('gain', C.c_float), ('coef', C.c_float), ('denom', C.c_float), ('neg_coef', C.c_float),
                ('neg_denom', C.c_float), ('flags', C.c_int32), ('hard_index', C.c_int32), ('pad', C.c_int32),
                ('noise_seed', C.c_uint64)]


class BwdDesc(C.Structure):
    """struct nbm_bwd_desc (include/nbm_hip.h)."""
    _fields_ = [('g', C.c_void_p), ('w', C.c_void_p), ('x', C.c_void_p), ('out', C.c_void_p),
                ('a_scale', C.c_void_p), ('row_scale', C.c_void_p), ('residual', C.c_void_p), ('mask', C.c_void_p),
                ('g_gs', C.c_int64), ('w_gs', C.c_int64), ('x_gs', C.c_int64), ('out_gs', C.c_int64), ('res_gs', C.c_int64),
                ('groups', C.c_int),
                ('B', C.c_int), ('H', C.c_int), ('W', C.c_int), ('Cin', C.c_int), ('N', C.c_int),
                ('kh', C.c_int), ('kw', C.c_int), ('stride', C.c_int), ('pad', C.c_int), ('Ho', C.c_int), ('Wo', C.c_int),
                ('g_ld', C.c_int), ('w_ld', C.c_int), ('x_ld', C.c_int), ('out_ld', C.c_int), ('res_ld', C.c_int),
                ('mask_ld', C.c_int), ('alpha', C.c_float), ('bias_grad', C.c_void_p), ('residual2', C.c_void_p), ('res2_ld', C.c_int),
                ('mask_bits', C.c_void_p)]


_P, _I, _L, _F, _U64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64

# name -> argtypes (restype is int unless noted); must list every symbol of include/nbm_hip.h
SIGNATURES = {
    'nbm_graph_census': [_P, C.POINTER(C.c_longlong * 6)],
    'nbm_copy_rect': [_P, _P, _L, _L, _I, _L, _L, _I, _I, _P],
    'nbm_gemm_conv': [C.POINTER(GemmDesc), _P],
    'nbm_pcm16_to_wave': [_P, _L, _I, _I, _I, _P, _L, _L, _P, _L, _I, _I, _P],
    'nbm_resample_to_wave': [_P, _L, _I, _L, _I, _I, _P, _I, _L, _L, _P, _L, _I, _I, _I, _P],
    'nbm_minmax_init': [_P, _I, _P],
    'nbm_stft_db': [_P, _L, _I, _I, _I, _I, _P, _I, _I, _I, _F, _P, _L, _I, _P, _P],
    'nbm_spec_windows': [_P, _L, _I, _I, _I, _I, _P, _P, _I, _I, _I, _P, _P],
    'nbm_init_conv': [_P, _L, _P, _P, _I, _P, _P],
    'nbm_stem7x7': [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    'nbm_stem7x7_wgrad': [_P, _P, _I, _I, _I, _P, _P, _P],
    'nbm_maxpool3x3s2': [_P, _I, _I, _I, _I, _P, _I, _I, _P, _P],
    'nbm_space_to_batch2': [_P, _I, _I, _I, _I, _P, _I, _P],
    'nbm_avgpool2x2': [_P, _I, _I, _I, _I, _P, _P],
    'nbm_avgpool2x2_bwd': [_P, _I, _I, _I, _I, _P, _P],
    'nbm_upsample_bilinear_add': [_P, _I, _I, _I, _I, _P, _P, _I, _I, _P],
    'nbm_softmax_rows': [_P, _L, _I, _L, _P],
    'nbm_dwconv3x3': [_P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _L, _P, _I, _I, _P],
    'nbm_silu': [_P, _P, _L, _P],
    'nbm_layernorm': [_P, _L, _I, _P, _P, _F, _P, _P],
    'nbm_mha_small': [_P, _P, _P, _I, _I, _I, _P, _I, _I, _I, _I, _I, _L, _L, _P, _F, _P],
    'nbm_pair_softmax': [_P, _L, _I, _I, _P, _I, _P],
    'nbm_rpn_decode': [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    'nbm_rpn_select': [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P],
    'nbm_nms_batched': [_P, _P, _P, _I, _I, _F, _I, _P, _P, _P, _P, _P, _I, _P],
    'nbm_roi_pool': [C.POINTER(RoiDesc), _P],
    'nbm_rcnn_post': [_P, _P, _I, _I, _P, _P, _I, _I, _I, _F, _F, _I, _P, _P, _I, _P],
    # ---- training path
    'nbm_conv_dgrad': [C.POINTER(BwdDesc), _P],
    'nbm_conv_wgrad': [C.POINTER(BwdDesc), _P],
    'nbm_relu_bwd': [_P, _P, _P, _L, _P],
    'nbm_silu_bwd': [_P, _P, _P, _L, _P],
    'nbm_axpby': [_P, _P, _P, _F, _F, _L, _L, _P],
    'nbm_colsum': [_P, _L, _I, _I, _P, _P],
    'nbm_leaky_relu_bwd': [_P, _P, _P, _F, _L, _P],
    'nbm_layernorm_bwd': [_P, _P, _P, _L, _I, _F, _P, _P, _P, _P],
    'nbm_mha_small_bwd': [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _I, _I, _P, _I, _I, _I, _I, _L, _L, _P, _F, _P],
    'nbm_zero_roi_windows': [_P, _I, _I, _I, _P, _P, _I, _I, _I, _P],
    'nbm_zero_pattern': [_P, _I, _I, _I, _I, _I, _P],
    'nbm_zero_tiles': [_P, _I, _I, _I, _I, _P, _I, _P, _P],
    'nbm_tiles_gather': [_P, _I, _I, _I, _I, _P, _I, _P, _P, _P],
    'nbm_tiles_scatter_add': [_P, _I, _I, _I, _I, _P, _I, _P, _P, _P],
    'nbm_proposal_iou': [_P, _P, _P, _I, _I, _I, _P, _P, _P],
    'nbm_anchor_targets': [_P, _I, _P, _P, _I, _I, _F, _F, _P, _P, _P, _P],
    'nbm_maxpool3x3s2_bwd': [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    'nbm_upsample_bilinear_bwd': [_P, _I, _I, _I, _I, _P, _I, _I, _I, _P],
    'nbm_tiles_upsample_bilinear_bwd_add': [_P, _I, _I, _I, _I, _P, _I, _P, _P, _I, _I, _P],
    'nbm_png_unfilter_gray8': [_P, _L, _I, _I, _I, _P, _L, _P, _P],
    'nbm_image_half_std_u8': [_P, _L, _I, _L, _P, _P],
    'nbm_randn_fill': [_U64, _L, _P, _P],
    'nbm_augment_batch': [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    'nbm_u8_to_unit': [_P, _L, _P, _P],
    'nbm_wino_input': [_P, _I, _I, _I, _I, _P, _I, _P],
    'nbm_wino_output': [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P, _P],
    'nbm_wino_outgrad': [_P, _I, _I, _I, _I, _P, _P, _I, _P],
    'nbm_wino_weight': [_P, _P, _I, _I, _I, _I, _P, _P],
    'nbm_wino_weight_grad': [_P, _P, _I, _I, _I, _P, _P],
    'nbm_wino23_rows': [_P, _I, _I, _I, _I, _P, _P],
    'nbm_wino23_conv_fused': [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P],
    'nbm_wino23_rows_tiles': [_P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _I, _P],
    'nbm_wino23_conv_fused_tiles': [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P],
    'nbm_roi_tiles': [_P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _I, _P],
    'nbm_wino23_input_tiles': [_P, _I, _I, _I, _I, _P, _I, _P, _P, _P],
    'nbm_wino23_outgrad_tiles': [_P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _I, _P],
    'nbm_cell_outgrad': [_P, _I, _I, _I, _I, _I, _P, _P, _P],
    'nbm_cell_weight': [_P, _I, _I, _P, _I, _P, _I, _P],
    'nbm_cell_weight_fold': [_P, _P, _I, _I, _I, _I, _F, _P, _I, _P, _I, _P],
    'nbm_cell_weight_grad': [_P, _I, _I, _I, _P, _P],
    'nbm_cell_input': [_P, _I, _I, _I, _I, _I, _P, _I, _I, _P],
    'nbm_cell_input_up': [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P],
    'nbm_cell_patches': [_P, _I, _I, _I, _I, _I, _P, _I, _I, _P],
    'nbm_cell_patches_up': [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P],
    'nbm_cell_dgrad_output': [_P, _I, _I, _I, _I, _I, _P, _I, _I, _I, _P, _P],
    'nbm_cell_output': [_P, _P, _I, _I, _I, _I, _I, _P, _P],
    'nbm_weighted_sum': [_P, _P, _P, _P, _P, _L, _P],
    'nbm_weighted_sum_bwd': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P],
    'nbm_softmax_rows_bwd': [_P, _P, _P, _L, _I, _F, _P],
    'nbm_pair_softmax_bwd': [_P, _P, _P, _L, _P],
    'nbm_dwconv3x3_bwd': [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _I, _P],
    'nbm_dwconv3x3_bwd_acc': [_P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P],
    'nbm_film_fwd': [_P, _P, _P, _L, _I, _P],
    'nbm_film_bwd': [_P, _P, _P, _P, _P, _L, _I, _P],
    'nbm_bn_train_fwd': [_P, _L, _I, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P],
    'nbm_bn_train_bwd': [_P, _P, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    'nbm_roi_pool_bwd': [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int), _I, _I, _P, _P, _I, _I, _P, _P],
    'nbm_sqnorm_accum': [_P, _L, _P, _P],
    'nbm_adamw_step': [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _P, _F, _P],
}


def build(verbose=False):
    """Compile libnbm_hip.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(['make', '-C', CSRC, '-j4'], capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout[-4000:], r.stderr[-4000:])
    if r.returncode:
        raise RuntimeError('building libnbm_hip.so failed')
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  -- first: libnbm_hip.so must bind to the HIP runtime torch has loaded, not to a second copy
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                           '(the NBM hot path has no CPU / eager fallback)')
    lib = C.CDLL(LIB_PATH)
    lib.nbm_version.restype = C.c_char_p
    lib.nbm_version.argtypes = []
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError => loud failure on a missing symbol
        fn.argtypes = argtypes
        fn.restype = C.c_int
    _lib = lib
    return lib


class NbmHipError(RuntimeError):
    pass


_CODES = {-1: 'NBM_EINVAL (bad argument / shape)', -2: 'NBM_EALIGN (pointer or pitch not 16-byte aligned)',
          -3: 'NBM_EUNSUPPORTED'}


def check(rc, name):
    if rc != 0:
        raise NbmHipError(f'{name} failed: {_CODES.get(rc, f"hipError_t {rc}")}')
