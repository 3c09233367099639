"""Host-side operator layer: thin wrappers that hand raw device pointers of torch tensors to the
C ABI of libnbm_hip.so on torch's current HIP stream.  torch is plumbing here (HBM allocations,
streams); all arithmetic happens in the hand-written kernels.  Activations are NHWC fp32.

Every function raises if the tensor is not a contiguous float32 CUDA tensor -- there is no CPU path.
"""
import ctypes as C
import os
import math

import torch

from . import _lib
from ._lib import GemmDesc, RoiDesc, check

ACT_NONE, ACT_RELU, ACT_SILU, ACT_LEAKY = 0, 1, 2, 3

# bench.py sets this to a list to time every implicit-GEMM launch with HIP events on the launch stream:
# entries are ((Cin, N, kh, H, W, B, groups, stride, label), start_event, end_event).
PROFILE = None
FLOPS = None                   # [float]: executed MFMA FLOPs of every GEMM launch since it was set (bench.py accounting)
FLOPS_DEFERRED = []             # (device counter, FLOPs per count): launches whose size is only known on the device


def flops_total():
    """Executed MFMA FLOPs since `FLOPS = [0.0]` was set (synchronises to read the device-side counts)."""
    tot = FLOPS[0] + sum(float(c.item()) * f for c, f in FLOPS_DEFERRED)
    FLOPS_DEFERRED.clear()
    return tot


_PROFILE_LABEL = None          # set by composite ops (Winograd) so that their GEMM launches can be told apart
PROFILE_ONLY = None             # None: every GEMM-type launch; 'fused': only the fused Winograd kernel; 'deepk': only the deep-K
                                # instantiation of the implicit-GEMM kernel, igemm_kernel<128,128,64,64,A_FAST,EPI_STD,2> (bench.py's
                                # timed loop: event pairs around all ~250 launches of a step cost it 2.5 %)


def _prof_all():
    return PROFILE is not None and PROFILE_ONLY is None


def _prof_fused():
    return PROFILE is not None and PROFILE_ONLY in (None, 'fused')


def is_deepk(Cin, N, kh, kw, rows=None):
    """Does nbm_gemm_conv dispatch this launch to igemm_kernel<128,128,64,64,A_FAST,EPI_STD,STAGES=2> (csrc/igemm.hip: N > 64,
    more than 8 K-steps of 32, channel count a multiple of 32, no row list)?"""
    return rows is None and N > 64 and Cin % 32 == 0 and kh * kw * Cin > 256



def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _chk(t, dtype=torch.float32, name='tensor'):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError(f'{name}: the NBM HIP path needs CUDA (ROCm) tensors; no CPU fallback exists')
    if t.dtype != dtype:
        raise TypeError(f'{name}: expected {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise ValueError(f'{name}: must be contiguous')
    return t


def lib():
    return _lib.load()


# --------------------------------------------------------------------------- implicit GEMM
def gemm_conv(x, w, y, *, B, H, W, Cin, N, kh=1, kw=1, stride=1, pad=0, Ho=None, Wo=None,
              x_ld=None, w_ld=None, y_ld=None, scale=None, shift=None, residual=None, res_ld=None,
              groups=1, x_gs=0, w_gs=0, y_gs=0, res_gs=0, alpha=1.0, act=ACT_NONE, shift_per_row=False, up=None,
              rows=None, rows_mode=0, rows_count=0, rows_blocks=None, rows_thw=(0, 0), mask=None, mask_ld=None, bits_out=None):
    """Raw call of nbm_gemm_conv (see include/nbm_hip.h for the exact semantics).  `bits_out` (int32 [M * N / 32]): one bit per stored
    output element, (y > 0) -- the ReLU mask of the consumer's data gradient (`conv_dgrad(mask_bits=)`)."""
    Ho = (H + 2 * pad - kh) // stride + 1 if Ho is None else Ho
    Wo = (W + 2 * pad - kw) // stride + 1 if Wo is None else Wo
    d = GemmDesc()
    d.x, d.w, d.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
    d.scale = scale.data_ptr() if scale is not None else None
    d.shift = shift.data_ptr() if shift is not None else None
    d.residual = residual.data_ptr() if residual is not None else None
    d.x_gs, d.w_gs, d.y_gs, d.res_gs = x_gs, w_gs, y_gs, res_gs
    d.groups = groups
    d.B, d.H, d.W, d.Cin, d.N = B, H, W, Cin, N
    d.kh, d.kw, d.stride, d.pad, d.Ho, d.Wo = kh, kw, stride, pad, Ho, Wo
    d.x_ld = Cin if x_ld is None else x_ld
    d.w_ld = kh * kw * Cin if w_ld is None else w_ld
    d.y_ld = N if y_ld is None else y_ld
    d.res_ld = (N if res_ld is None else res_ld) if residual is not None else 0
    d.alpha, d.act, d.shift_per_row = float(alpha), int(act), int(bool(shift_per_row))
    if up is not None:                       # [B, up_H, up_W, N] coarse map merged in the epilogue
        d.up, d.up_H, d.up_W = _chk(up, name='up').data_ptr(), up.shape[1], up.shape[2]
    if mask is not None:                     # producer mask: y = 0 where mask <= 0 (a 1x1 data gradient run as a forward GEMM)
        d.mask, d.mask_ld = mask.data_ptr(), int(N if mask_ld is None else mask_ld)
    if bits_out is not None:
        assert bits_out.dtype == torch.int32 and bits_out.is_contiguous() and bits_out.numel() * 32 >= B * Ho * Wo * N and N % 32 == 0
        d.bits_out = bits_out.data_ptr()
    if rows is not None:                     # listed pixels only (the lateral of a demand-driven FPN level)
        d.rows, d.rows_mode, d.rows_count = rows.data_ptr(), int(rows_mode), int(rows_count)
        d.rows_blocks = rows_blocks.data_ptr() if rows_blocks is not None else None
        d.rows_TH, d.rows_TW = rows_thw
    if FLOPS is not None:
        if rows is None:
            FLOPS[0] += 2.0 * B * Ho * Wo * N * kh * kw * Cin * groups
        elif rows_blocks is None:
            FLOPS[0] += 2.0 * rows_count * N * Cin
        else:
            FLOPS_DEFERRED.append((rows_blocks.clone(), 2.0 * 128 * (16 if rows_mode == 2 else 1) * N * Cin))
    if _prof_all() or (PROFILE is not None and PROFILE_ONLY == 'deepk' and is_deepk(Cin, N, kh, kw, rows)):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        check(lib().nbm_gemm_conv(C.byref(d), _stream()), 'nbm_gemm_conv')
        ev1.record()
        if rows is None:
            PROFILE.append(((Cin, N, kh, H, W, B, groups, stride, _PROFILE_LABEL), ev0, ev1))
        elif rows_blocks is None:                  # listed pixels: M = the list
            PROFILE.append(((Cin, N, kh, rows_count, 1, 1, groups, stride, ('rows', H, W)), ev0, ev1))
        else:                                      # device-side count
            PROFILE.append(((Cin, N, kh, 0, 1, 1, groups, stride, ('rows-rois', H, W)), ev0, ev1))
        return y
    check(lib().nbm_gemm_conv(C.byref(d), _stream()), 'nbm_gemm_conv')
    return y


def conv2d(x, w, kh=1, kw=1, stride=1, pad=0, scale=None, shift=None, residual=None, act=ACT_NONE,
           alpha=1.0, out=None, w_ld=None, up=None, bits_out=None):
    """x [B,H,W,Cin] NHWC, w [N, w_ld>=kh*kw*Cin] (KRSC rows) -> [B,Ho,Wo,N]; `up` [B,h,w,N]: + bilinear(up)."""
    _chk(x, name='x'), _chk(w, name='w')
    B, H, W, Cin = x.shape
    N = w.shape[0]
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    y = out if out is not None else torch.empty((B, Ho, Wo, N), device=x.device, dtype=torch.float32)
    if residual is not None:
        _chk(residual, name='residual')
        assert residual.shape == y.shape
    gemm_conv(x, w, y, B=B, H=H, W=W, Cin=Cin, N=N, kh=kh, kw=kw, stride=stride, pad=pad, Ho=Ho, Wo=Wo,
              w_ld=w.shape[1] if w_ld is None else w_ld, scale=scale, shift=shift, residual=residual,
              alpha=alpha, act=act, up=up, bits_out=bits_out)
    return y


WINO_CHUNK_BYTES = 24 << 30          # cap of the transformed-domain scratch (V + M) per batch chunk
_WINO_SCRATCH = {}

# Lanes: detect steps that are IN FLIGHT TOGETHER on one GPU (parallel branches of one captured graph, one stream each:
# bulk.GraphedDetector(lanes=k)) fill each other's kernel tails.  Everything a step allocates through torch is its own; what is shared
# process-wide and WRITTEN by kernels -- the persistent transformed-domain scratch here, the RoI tile-list buffers of ondemand.py -- is
# keyed by the lane that is current while the step is issued (`with ops.lane(k)` around a branch's warm-up and capture).
LANE = 0


class lane:
    def __init__(self, k):
        self.k = int(k)

    def __enter__(self):
        global LANE
        self.prev, LANE = LANE, self.k

    def __exit__(self, *exc):
        global LANE
        LANE = self.prev


def _wino_scratch(device, n_v, n_m):
    """Two views (V: n_v floats, M: n_m floats) of ONE persistent per-device scratch allocation.  Every Winograd call on
    a device runs on the same stream, so reusing the buffer is stream-ordered; a fresh multi-GB torch.empty per call made
    the caching allocator fall back to hipMalloc / hipFree (100 ms host stalls per training step).  The buffer is sized to
    what the largest call so far needed (<= WINO_CHUNK_BYTES by construction of the batch chunks) and grows geometrically:
    a B = 1 detect process holds 0.4 GB, not the 24 GB a B = 64 step uses."""
    need = n_v + n_m
    key = (device, LANE)
    buf = _WINO_SCRATCH.get(key)
    if buf is None or buf.numel() < need:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('the persistent Winograd scratch would have to grow during a graph capture (the warm-up steps of the '
                               'capture must run the same shapes under the same ops.lane)')
        have = 0 if buf is None else buf.numel()
        _WINO_SCRATCH[key] = buf = None          # release before growing
        size = min(max(need, 2 * have), max(need, WINO_CHUNK_BYTES // 4))
        buf = _WINO_SCRATCH[key] = torch.empty((size,), device=device, dtype=torch.float32)
    return buf[:n_v], buf[n_v:need]


GRAPH_NODE_KINDS = ('kernel', 'memcpy', 'memset', 'host', 'empty', 'other')


def graph_census(raw_graph):
    """{node kind: count} of a captured hipGraph_t (`torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph()`), nbm_graph_census."""
    counts = (C.c_longlong * 6)()
    check(_lib.load().nbm_graph_census(C.c_void_p(int(raw_graph)), C.byref(counts)), 'nbm_graph_census')
    return dict(zip(GRAPH_NODE_KINDS, (int(v) for v in counts)))


def lane_buffers(lanes):
    """The persistent scratch / tile-list tensors of the given lanes AS THEY ARE NOW: a captured graph bakes their raw addresses in, so
    its owner keeps these references -- a later growth (`_wino_scratch` drops the dict entry before it allocates the larger buffer) or
    `release_lane_scratch` then cannot hand the memory a live graph writes to to another tensor."""
    from . import ondemand
    held = [buf for key, buf in _WINO_SCRATCH.items() if key[1] in lanes and buf is not None]
    for key, bufs in ondemand._ROI_TILE_BUF.items():
        if key[2] in lanes:
            held.extend(bufs)
    return held


def release_lane_scratch(keep=(0,)):
    """Frees the persistent scratch and tile-list buffers of every lane not in `keep` (tens of GB at B = 64).  Only when no captured
    graph that was issued in those lanes is alive any more: a graph keeps the raw addresses."""
    from . import ondemand
    for key in [k for k in _WINO_SCRATCH if k[1] not in keep]:
        del _WINO_SCRATCH[key]
    for key in [k for k in ondemand._ROI_TILE_BUF if k[2] not in keep]:
        del ondemand._ROI_TILE_BUF[key]


WINO_FUSED_VARIANT = int(os.environ.get('NBM_WINO_FUSED_VARIANT', '0'))     # 0 auto, 128 / 64: channel-tile width


def conv3x3_winograd(x, U, bias=None, m=2, scale=None, relu=False, mask=None, residual=None):
    """3x3 / stride 1 / pad 1 convolution through Winograd F(m x m, 3x3): x [B,H,W,C], U [(m+2)^2,N,C] from
    `_prep.wino23` -> [B,H,W,N].  m = 2 (the forward setting, error ~3e-6): row half of the input transform (2x the input
    in HBM), then ONE kernel that finishes the input transform while staging its operand, runs the 16 transformed-domain
    GEMMs on the fp32 MFMA and applies the output transform + epilogue in its accumulators (csrc/wino_fused.hip; neither
    the transformed input V nor the products M ever reach HBM).  m = 4 (backward only, ~2e-5, see csrc/winograd.hip): input
    transform, 36 grouped GEMMs, output transform.  2.25x / 4x fewer multiplies than the direct kernel.  The batch is cut
    so that the transformed operands stay within WINO_CHUNK_BYTES.  Epilogue: y = relu?((.) * scale + bias), zeroed where
    mask <= 0."""
    _chk(x, name='x'), _chk(U, name='U')
    B, H, W, C_ = x.shape
    N = U.shape[1]
    nxi = (m + 2) ** 2
    assert U.shape == (nxi, N, C_) and C_ % 32 == 0 and N % 4 == 0
    y = torch.empty((B, H, W, N), device=x.device, dtype=torch.float32)
    tiles = (-(-H // m)) * (-(-W // m))
    fused = m == 2 and C_ >= 64                  # the fused pipeline wants >= 2 K-steps per plane (every real layer has)
    assert residual is None or m == 4, 'residual: F(4x4,3x3) output transform only' 
    if fused:                                     # R: four row-combination images, 2 (TW + 1) columns per tile row
        per_img = 4 * (-(-H // 2)) * (2 * (-(-W // 2)) + 2) * C_ * 4
    else:
        per_img = nxi * tiles * (C_ + N) * 4
    chunk = max(1, min(B, WINO_CHUNK_BYTES // per_img))
    if fused:
        V, M = _wino_scratch(x.device, chunk * per_img // 4, 0)
    else:
        V, M = _wino_scratch(x.device, nxi * chunk * tiles * C_, nxi * chunk * tiles * N)
    st = _stream()
    if _prof_all():                       # whole-op bracket (transforms + GEMMs) next to the per-GEMM entries
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    for b0 in range(0, B, chunk):
        nb = min(chunk, B - b0)
        T = nb * tiles
        mk = _ptr(mask[b0:b0 + nb]) if mask is not None else None
        if fused:
            check(lib().nbm_wino23_rows(_ptr(x[b0:b0 + nb]), nb, H, W, C_, _ptr(V), st), 'nbm_wino23_rows')
            if FLOPS is not None:
                FLOPS[0] += 2.0 * nxi * T * C_ * N
            if _prof_fused():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            check(lib().nbm_wino23_conv_fused(_ptr(V), _ptr(U), _ptr(scale), _ptr(bias), mk, int(relu), nb, H, W, C_, N,
                                              _ptr(y[b0:b0 + nb]), WINO_FUSED_VARIANT, st), 'nbm_wino23_conv_fused')
            if _prof_fused():
                e1.record()
                PROFILE.append(((C_, N, 1, T, 1, 1, nxi, 1, ('wino23', H, W)), e0, e1))
            continue
        check(lib().nbm_wino_input(_ptr(x[b0:b0 + nb]), nb, H, W, C_, _ptr(V), m, st), 'nbm_wino_input')
        global _PROFILE_LABEL
        _PROFILE_LABEL = ('wino23', H, W)
        try:
            gemm_conv(V, U, M, B=1, H=T, W=1, Cin=C_, N=N, groups=nxi, x_gs=T * C_, w_gs=N * C_, y_gs=T * N)
        finally:
            _PROFILE_LABEL = None
        check(lib().nbm_wino_output(_ptr(M), _ptr(scale), _ptr(bias), mk, int(relu), nb, H, W, N, _ptr(y[b0:b0 + nb]), m,
                                    _ptr(residual[b0:b0 + nb]) if residual is not None else None, st), 'nbm_wino_output')
    if _prof_all():
        ev[1].record()
        PROFILE.append((('wino23', C_, N, H, W, B), *ev))
    return y


def wino_weight(weight, transposed=False, m=2, scale=None):
    """[N,C,3,3] -> U [(m+2)^2, N', C'] = G g' G^T (float64 arithmetic on the device, rounded once); `transposed`: the
    data-gradient convolution's weights (rotated kernel, swapped channel roles, optional per-output-channel `scale`)."""
    _chk(weight, name='weight')
    N, C_ = weight.shape[:2]
    assert tuple(weight.shape[2:]) == (3, 3)
    No, Co = (C_, N) if transposed else (N, C_)
    U = torch.empty(((m + 2) ** 2, No, Co), device=weight.device, dtype=torch.float32)
    check(lib().nbm_wino_weight(_ptr(weight), _ptr(scale), N, C_, int(bool(transposed)), m, _ptr(U), _stream()),
          'nbm_wino_weight')
    return U


def wino_weight_grad(dU, m=2, row_scale=None):
    """dU [(m+2)^2, N, C] -> dW [N, C, 3, 3] = row_scale[n] * G^T dU G."""
    _chk(dU, name='dU')
    _, N, C_ = dU.shape
    dW = torch.empty((N, C_, 3, 3), device=dU.device, dtype=torch.float32)
    check(lib().nbm_wino_weight_grad(_ptr(dU), _ptr(row_scale), N, C_, m, _ptr(dW), _stream()), 'nbm_wino_weight_grad')
    return dW


def cell_weight(weight, forward=True, lateral=None, alpha=1.0, both=False):
    """Kernel side of the cell transforms (csrc/cellwino.hip, float64 on the device, one rounding): [N,C,3,3] -> U = E w E^T as
    [25][N][K] (`forward`: B operand of the forward plane GEMMs) or [25][K][N] (data-gradient GEMMs); `both`: (forward, transposed).
    `lateral` = KRSC weights [C][Cin] of a deferred lateral 1x1: K = C + Cin, the extra columns / rows hold alpha * U W_lat."""
    _chk(weight, name='weight')
    N, C_ = weight.shape[:2]
    assert tuple(weight.shape[2:]) == (3, 3)
    Cin = 0
    if lateral is not None:
        _chk(lateral, name='lateral')
        assert lateral.shape[0] == C_
        Cin = lateral.shape[1]
    K = C_ + Cin
    Unc = torch.empty((25, N, K), device=weight.device, dtype=torch.float32) if (forward or both) else None
    Ucn = torch.empty((25, K, N), device=weight.device, dtype=torch.float32) if (not forward or both) else None
    check(lib().nbm_cell_weight(_ptr(weight), N, C_, _ptr(Unc), K, _ptr(Ucn), K, _stream()), 'nbm_cell_weight')
    if lateral is not None:
        check(lib().nbm_cell_weight_fold(_ptr(weight), _ptr(lateral), lateral.stride(0), N, C_, Cin, float(alpha), _ptr(Unc), K,
                                         _ptr(Ucn), K, _stream()), 'nbm_cell_weight_fold')
    return (Unc, Ucn) if both else (Unc if forward else Ucn)


def copy_rect(strided, base_off, packed, n_outer, outer_pitch, n_rows, row_pitch, width, to_strided=False, zero=False):
    """nbm_copy_rect on a flat fp32 tensor `strided` (element offset `base_off` of the rectangle's first float) and a contiguous
    `packed` [n_outer * n_rows * width]; counts in floats."""
    assert strided.dtype == torch.float32 and packed.dtype == torch.float32 and packed.is_contiguous()
    assert packed.numel() >= n_outer * n_rows * width
    assert base_off + (n_outer - 1) * outer_pitch + (n_rows - 1) * row_pitch + width <= strided.numel()
    check(lib().nbm_copy_rect(C.c_void_p(strided.data_ptr() + 4 * base_off), _ptr(packed), n_outer, outer_pitch, n_rows, row_pitch, width,
                              int(bool(to_strided)), int(bool(zero)), _stream()), 'nbm_copy_rect')
    return packed


def cell_weight_grad(dU):
    """dU [25, N, K >= C] (columns beyond C ignored when `dU` is a [..., :C] view) -> dW [N, C, 3, 3] = E^T dU E."""
    assert dU.dim() == 3 and dU.shape[0] == 25 and dU.stride(2) == 1 and dU.stride(0) == dU.shape[1] * dU.stride(1)
    _, N, C_ = dU.shape
    dW = torch.empty((N, C_, 3, 3), device=dU.device, dtype=torch.float32)
    check(lib().nbm_cell_weight_grad(_ptr(dU), N, C_, dU.stride(1), _ptr(dW), _stream()), 'nbm_cell_weight_grad')
    return dW


def conv3x3_winograd_wgrad(x, g, want_bias=False, m=2):
    """Weight gradient of a 3x3 / s1 / p1 convolution in the Winograd domain: x [B,H,W,C] (forward input), g [B,H,W,N]
    (gradient wrt the output) -> (dU [(m+2)^2,N,C] with dU[xi] = dM[xi]^T V[xi], bias gradient [N] or None).  The caller
    maps dU back with dW = G^T dU G (`_prep.wino23_weight_grad`)."""
    _chk(x, name='x'), _chk(g, name='g')
    B, H, W, C_ = x.shape
    N = g.shape[-1]
    nxi = (m + 2) ** 2
    assert g.shape[:3] == x.shape[:3] and C_ % 4 == 0 and N % 4 == 0
    tiles = (-(-H // m)) * (-(-W // m))
    per_img = nxi * tiles * (C_ + N) * 4
    chunk = max(1, min(B, WINO_CHUNK_BYTES // per_img))
    V, dM = _wino_scratch(x.device, nxi * chunk * tiles * C_, nxi * chunk * tiles * N)
    dU = torch.zeros((nxi, N, C_), device=x.device, dtype=torch.float32)
    gb = torch.zeros((N,), device=x.device, dtype=torch.float32) if want_bias else None
    st = _stream()
    for b0 in range(0, B, chunk):
        nb = min(chunk, B - b0)
        T = nb * tiles
        check(lib().nbm_wino_input(_ptr(x[b0:b0 + nb]), nb, H, W, C_, _ptr(V), m, st), 'nbm_wino_input')
        check(lib().nbm_wino_outgrad(_ptr(g[b0:b0 + nb]), nb, H, W, N, _ptr(dM), _ptr(gb), m, st), 'nbm_wino_outgrad')
        conv_wgrad(dM, V, dU, B=1, H=T, W=1, Cin=C_, N=N, groups=nxi, g_gs=T * N, x_gs=T * C_, out_gs=N * C_)
    return dU, gb


def linear(x2d, w, bias=None, act=ACT_NONE, out=None, alpha=1.0, residual=None, y_ld=None):
    """x2d [M,K], w [N,K] -> [M,N] (= x @ w.T + bias)."""
    _chk(x2d, name='x'), _chk(w, name='w')
    M, K = x2d.shape
    N = w.shape[0]
    y = out if out is not None else torch.empty((M, N), device=x2d.device, dtype=torch.float32)
    gemm_conv(x2d, w, y, B=1, H=M, W=1, Cin=K, N=N, w_ld=w.shape[1], shift=bias, act=act, alpha=alpha,
              residual=residual, y_ld=y_ld)
    return y


def bgemm_nt(a, b, out=None, alpha=1.0, shift=None, shift_per_row=False, residual=None, act=ACT_NONE):
    """a [G,M,K] (or [M,K] shared by all groups), b [G,N,K] -> [G,M,N] = alpha * a @ b^T (+shift)(+residual)."""
    _chk(a, name='a'), _chk(b, name='b')
    G, N, K = b.shape
    M = a.shape[-2]
    assert a.shape[-1] == K
    a_gs = M * K if a.dim() == 3 else 0
    y = out if out is not None else torch.empty((G, M, N), device=b.device, dtype=torch.float32)
    gemm_conv(a, b, y, B=1, H=M, W=1, Cin=K, N=N, w_ld=K, groups=G, x_gs=a_gs, w_gs=N * K, y_gs=M * N,
              alpha=alpha, shift=shift, shift_per_row=shift_per_row, residual=residual,
              res_gs=M * N if residual is not None else 0, act=act)
    return y


# --------------------------------------------------------------------------- front end
def pcm16_to_wave(pcm, out_ld, lead, upsample, hq=None, reflect=False, first=0, count=None):
    """pcm int16 [batch, n] -> f32 [batch, out_ld]: `lead` padding samples, the piece [first, first + count) of the
    44.1 kHz signal (the input itself, or its 2x up-sampling when `upsample`; default: all of it), `lead` padding
    samples, zeros.  Padding = zeros (librosa >= 0.10 `pad_mode='constant'`) or the mirrored piece (`reflect`,
    librosa <= 0.9)."""
    _chk(pcm, torch.int16, 'pcm')
    batch, n = pcm.shape
    if count is None:
        count = (2 * n if upsample else n) - first
    out = torch.empty((batch, out_ld), device=pcm.device, dtype=torch.float32)
    if upsample:
        _chk(hq, torch.int32, 'hq')
    check(lib().nbm_pcm16_to_wave(_ptr(pcm), n, batch, n, int(bool(upsample)), _ptr(hq), first, count, _ptr(out), out_ld,
                                  lead, int(bool(reflect)), _stream()), 'nbm_pcm16_to_wave')
    return out


def resample_to_wave(x, out_ld, lead, L, M, taps, reflect=False, first=0, count=None, quant16=True):
    """x f32 [batch, n] at any rate -> f32 [batch, out_ld] laid out like `pcm16_to_wave`: the piece [first, first+count)
    of the 44.1 kHz signal = x itself (L = M = 1) or its rational L / M polyphase resampling with the float64 `taps`
    [L, T] (rounded to the 16-bit grid when `quant16`)."""
    _chk(x, torch.float32, 'x')
    batch, n = x.shape
    n_out = n if (L == 1 and M == 1) else -(-n * L // M)
    if count is None:
        count = n_out - first
    if taps is not None:
        _chk(taps, torch.float64, 'taps')
        assert taps.shape[0] == L
    out = torch.empty((batch, out_ld), device=x.device, dtype=torch.float32)
    check(lib().nbm_resample_to_wave(_ptr(x), n, batch, n, L, M, _ptr(taps), taps.shape[1] if taps is not None else 0, first,
                                     count, _ptr(out), out_ld, lead, int(bool(reflect)), int(bool(quant16)), _stream()),
          'nbm_resample_to_wave')
    return out


def stft_db(wave, n_frames, hop, n_fft, basis, n_bins, floor_amp, db_ld=None, out=None, col0=0):
    """wave f32 [batch, wave_ld] (already centre-padded) -> (db [batch, n_bins, db_ld], minmax u32 [batch,2]).
    `basis`: float64 [bin tiles, k steps, 64, 2] in MFMA fragment order (prepare_dataset.dft_basis_f64).
    `out=(db, minmax)` + `col0`: write the frames at column offset col0 of an existing spectrogram and keep accumulating
    its min/max (chunked STFT of a long file, reference prepare_dataset.py:234-237)."""
    _chk(wave, name='wave'), _chk(basis, torch.float64, 'basis')
    batch, wave_ld = wave.shape
    if out is None:
        db_ld = n_frames if db_ld is None else db_ld
        db = torch.empty((batch, n_bins, db_ld), device=wave.device, dtype=torch.float32)
        mm = torch.empty((batch, 2), device=wave.device, dtype=torch.int32)
        check(lib().nbm_minmax_init(_ptr(mm), batch, _stream()), 'nbm_minmax_init')
    else:
        db, mm = out
        db_ld = db.shape[-1]
        assert col0 + n_frames <= db_ld and db.shape[0] == batch
    check(lib().nbm_stft_db(_ptr(wave), wave_ld, batch, n_frames, hop, n_fft, _ptr(basis), basis.shape[0],
                            basis.shape[1], n_bins, float(floor_amp), C.c_void_p(db.data_ptr() + 4 * col0),
                            n_bins * db_ld, db_ld, _ptr(mm), _stream()), 'nbm_stft_db')
    return db, mm


def spec_windows(db, minmax, n_frames, n_img, w_pix, hop_img, last_cols):
    """db [batch, n_bins, db_ld] + min/max -> img [batch, n_img, n_bins, w_pix] normalised to [0,1]; the last window
    reads the columns `last_cols` (int32 [w_pix] on the device)."""
    _chk(db, name='db'), _chk(last_cols, torch.int32, 'last_cols')
    assert last_cols.numel() == w_pix
    batch, n_bins, db_ld = db.shape
    img = torch.empty((batch, n_img, n_bins, w_pix), device=db.device, dtype=torch.float32)
    check(lib().nbm_spec_windows(_ptr(db), n_bins * db_ld, db_ld, batch, n_bins, n_frames, _ptr(minmax), _ptr(img),
                                 n_img, w_pix, hop_img, _ptr(last_cols), _stream()), 'nbm_spec_windows')
    return img


# --------------------------------------------------------------------------- point-wise stages
def init_conv(x, w, b):
    """x [B,H,W,1] -> [B,H,W,C]."""
    _chk(x, name='x')
    C_ = w.numel()
    y = torch.empty(x.shape[:3] + (C_,), device=x.device, dtype=torch.float32)
    check(lib().nbm_init_conv(_ptr(x), x.numel(), _ptr(_chk(w.reshape(-1))), _ptr(_chk(b)), C_, _ptr(y), _stream()),
          'nbm_init_conv')
    return y


def stem7x7(img, weff, wb, wb_full, scale, shift):
    """img [B,H,W,1] (or [B,H,W]) -> relu(scale * conv7x7/s2/p3(init_conv(img)) + shift) [B,Ho,Wo,64]; the folded operands
    come from `_prep.stem_fold`."""
    _chk(img, name='img')
    B, H, W = img.shape[:3]
    assert img.numel() == B * H * W and tuple(weff.shape) == (56, 64)
    y = torch.empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 64), device=img.device, dtype=torch.float32)
    if FLOPS is not None:
        FLOPS[0] += 2.0 * y.numel() * 56
    if _prof_all():
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    check(lib().nbm_stem7x7(_ptr(img), B, H, W, _ptr(_chk(weff)), _ptr(_chk(wb)), _ptr(_chk(wb_full)), _ptr(_chk(scale)),
                            _ptr(_chk(shift)), _ptr(y), _stream()), 'nbm_stem7x7')
    if _prof_all():
        ev[1].record()
        PROFILE.append(((1, 64, 7, H, W, B, 1, 2, None), *ev))
    return y


def stem7x7_wgrad(img, g):
    """img [B,H,W,1], g [B,Ho,Wo,64] (ReLU-masked output gradient) -> (U [64,7,7], V [64,7,7]): sums of g times the image sample /
    the inside-the-image indicator under every tap (csrc/stem.hip)."""
    _chk(img, name='img'), _chk(g, name='g')
    B, H, W = img.shape[:3]
    assert g.shape == (B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 64)
    D = torch.empty((64, 64), device=img.device, dtype=torch.float32)
    Cb = torch.empty((64, 49), device=img.device, dtype=torch.float32)
    check(lib().nbm_stem7x7_wgrad(_ptr(img), _ptr(g), B, H, W, _ptr(D), _ptr(Cb), _stream()), 'nbm_stem7x7_wgrad')
    U = D[:, :56].view(64, 7, 8)[:, :, :7]
    V = D[:, 56].view(64, 1, 1) - Cb.view(64, 7, 7)
    return U, V


def maxpool3x3s2(x, with_index=False):
    """-> y, or (y, idx uint8 [B,Ho,Wo,C]: window position of each maximum, for `maxpool3x3s2_bwd`)."""
    _chk(x, name='x')
    B, H, W, C_ = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((B, Ho, Wo, C_), device=x.device, dtype=torch.float32)
    idx = torch.empty((B, Ho, Wo, C_), device=x.device, dtype=torch.uint8) if with_index else None
    check(lib().nbm_maxpool3x3s2(_ptr(x), B, H, W, C_, _ptr(y), Ho, Wo, _ptr(idx), _stream()), 'nbm_maxpool3x3s2')
    return (y, idx) if with_index else y


def space_to_batch2(x, inverse=False):
    """x [B,H,W,C] (H, W even) -> [4B,H/2,W/2,C]: the four parity classes of the pixels as four images (class 2a+b major); `inverse`:
    [4B,h,w,C] -> [B,2h,2w,C].  A 3x3 / dilation-2 / pad-2 convolution on x is the ordinary 3x3 / pad-1 convolution on this form."""
    _chk(x, name='x')
    if inverse:
        B4, h, w, C_ = x.shape
        assert B4 % 4 == 0
        B, H, W = B4 // 4, 2 * h, 2 * w
        y = torch.empty((B, H, W, C_), device=x.device, dtype=torch.float32)
    else:
        B, H, W, C_ = x.shape
        if (H | W) & 1:
            raise NotImplementedError('space_to_batch2 needs even map sizes (the dilated ResNet stage at 24 x 64)')
        y = torch.empty((4 * B, H // 2, W // 2, C_), device=x.device, dtype=torch.float32)
    check(lib().nbm_space_to_batch2(_ptr(x), B, H, W, C_, _ptr(y), int(bool(inverse)), _stream()), 'nbm_space_to_batch2')
    return y


def avgpool2x2(x):
    """x [B,2h,2w,C] -> [B,h,w,C]: mean of each 2x2 block (nn.AdaptiveAvgPool2d to half the size)."""
    _chk(x, name='x')
    B, H, W, C_ = x.shape
    assert H % 2 == 0 and W % 2 == 0
    y = torch.empty((B, H // 2, W // 2, C_), device=x.device, dtype=torch.float32)
    check(lib().nbm_avgpool2x2(_ptr(x), B, H // 2, W // 2, C_, _ptr(y), _stream()), 'nbm_avgpool2x2')
    return y


def avgpool2x2_bwd(gy):
    _chk(gy, name='gy')
    B, Ho, Wo, C_ = gy.shape
    gx = torch.empty((B, 2 * Ho, 2 * Wo, C_), device=gy.device, dtype=torch.float32)
    check(lib().nbm_avgpool2x2_bwd(_ptr(gy), B, Ho, Wo, C_, _ptr(gx), _stream()), 'nbm_avgpool2x2_bwd')
    return gx


def upsample_bilinear_add(src, Ho, Wo, add=None):
    _chk(src, name='src')
    B, Hi, Wi, C_ = src.shape
    if add is not None:
        _chk(add, name='add')
        assert tuple(add.shape) == (B, Ho, Wo, C_)
    y = torch.empty((B, Ho, Wo, C_), device=src.device, dtype=torch.float32)
    check(lib().nbm_upsample_bilinear_add(_ptr(src), B, Hi, Wi, C_, _ptr(add), _ptr(y), Ho, Wo, _stream()),
          'nbm_upsample_bilinear_add')
    return y


def softmax_rows_(x2d):
    _chk(x2d, name='x')
    rows, cols = x2d.shape
    check(lib().nbm_softmax_rows(_ptr(x2d), rows, cols, cols, _stream()), 'nbm_softmax_rows')
    return x2d


def dwconv3x3(x, w, bias, mult, stride=1, film=None):
    """x [B,H,W,Cin], w [Cin*mult,1,3,3] (reference layout) -> [B,Ho,Wo,Cin*mult]; film [B*Ho*Wo, 2*Cout]."""
    _chk(x, name='x'), _chk(w, name='w')
    B, H, W, Cin = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty((B, Ho, Wo, Cin * mult), device=x.device, dtype=torch.float32)
    film_ld = 0
    if film is not None:
        _chk(film, name='film')
        film_ld = film.shape[-1]
    check(lib().nbm_dwconv3x3(_ptr(x), B, H, W, Cin, mult, stride, _ptr(w), _ptr(bias), _ptr(film), film_ld, _ptr(y),
                              Ho, Wo, _stream()), 'nbm_dwconv3x3')
    return y


def silu(x):
    _chk(x, name='x')
    y = torch.empty_like(x)
    check(lib().nbm_silu(_ptr(x), _ptr(y), x.numel(), _stream()), 'nbm_silu')
    return y


def layernorm(x2d, w, b, eps=1e-5):
    _chk(x2d, name='x')
    rows, E = x2d.shape
    y = torch.empty_like(x2d)
    check(lib().nbm_layernorm(_ptr(x2d), rows, E, _ptr(_chk(w)), _ptr(_chk(b)), float(eps), _ptr(y), _stream()), 'nbm_layernorm')
    return y


def mha_small(q, k, v, S, N, nhead, seq_stride, batch_stride, n_valid=None):
    """q, k, v: 2-D row views [S*N, >= E] (column slices of a fused projection are fine) -> out [S*N, E]."""
    E = q.shape[1]
    hd = E // nhead
    for t in (q, k, v):
        assert t.dim() == 2 and t.stride(1) == 1 and t.is_cuda and t.dtype == torch.float32
    out = torch.zeros((q.shape[0], E), device=q.device, dtype=torch.float32)
    check(lib().nbm_mha_small(_ptr(q), _ptr(k), _ptr(v), q.stride(0), k.stride(0), v.stride(0), _ptr(out), E, S, N, nhead,
                              hd, seq_stride, batch_stride, _ptr(n_valid), 1.0 / math.sqrt(hd), _stream()), 'nbm_mha_small')
    return out


def pair_softmax(x, n_anchor):
    """x [..., 2*n_anchor] -> softmax over each (bg, fg) pair."""
    _chk(x, name='x')
    ld = x.shape[-1]
    y = torch.empty_like(x)
    check(lib().nbm_pair_softmax(_ptr(x), x.numel() // ld, n_anchor, ld, _ptr(y), ld, _stream()), 'nbm_pair_softmax')
    return y


# --------------------------------------------------------------------------- proposal / RoI / detection
def rpn_decode(cls, reg, anchors, n_anchor, img_w, img_h, min_size):
    """cls [B,K,2A] softmaxed, reg [B,K,4A], anchors [K*A,4] -> boxes [B,KA,4], keys u32 [B,KA], keep_count [B]."""
    _chk(cls, name='cls'), _chk(reg, name='reg'), _chk(anchors, name='anchors')
    B = cls.shape[0]
    KA = anchors.shape[0]
    boxes = torch.empty((B, KA, 4), device=cls.device, dtype=torch.float32)
    keys = torch.empty((B, KA), device=cls.device, dtype=torch.int32)
    cnt = torch.empty((B,), device=cls.device, dtype=torch.int32)
    check(lib().nbm_rpn_decode(_ptr(cls), _ptr(reg), _ptr(anchors), B, KA, n_anchor, img_w, img_h, int(min_size),
                               _ptr(boxes), _ptr(keys), _ptr(cnt), _stream()), 'nbm_rpn_decode')
    return boxes, keys, cnt


def proposal_iou(rois, gt, n_gt):
    """rois [B,R,4], gt [B,G,4] (padded), n_gt int32 [B] -> (mx [B,R+G] best IoU of every proposal / ground-truth box with the
    image's ground-truth boxes, asg int32 [B,R+G] index of the first best box): reference layers.py:320-330."""
    _chk(rois, name='rois'), _chk(gt, name='gt')
    B, R, _ = rois.shape
    G = gt.shape[1]
    if n_gt.dtype != torch.int32 or n_gt.numel() != B or gt.shape[0] != B:
        raise ValueError('proposal_iou: n_gt must be int32 [B], gt [B,G,4]')
    mx = torch.empty((B, R + G), device=rois.device, dtype=torch.float32)
    asg = torch.empty((B, R + G), device=rois.device, dtype=torch.int32)
    check(lib().nbm_proposal_iou(_ptr(rois), _ptr(gt), _ptr(n_gt), B, R, G, _ptr(mx), _ptr(asg), _stream()), 'nbm_proposal_iou')
    return mx, asg


def anchor_targets(anchors, gt, n_gt, neg_t, pos_t):
    """anchors [n_in,4] (inside the image), gt [B,G,4] (padded), n_gt int32 [B] -> (lab int8 [B,n_in]: the label of every anchor
    before the random subsampling, amx int16 [B,n_in]: first best box, flag int32 [B]: recompute on the host): reference
    layers.py:150-179."""
    _chk(anchors, name='anchors'), _chk(gt, name='gt')
    B, G = gt.shape[:2]
    n_in = anchors.shape[0]
    if n_gt.dtype != torch.int32 or n_gt.numel() != B:
        raise ValueError('anchor_targets: n_gt must be int32 [B]')
    lab = torch.empty((B, n_in), device=gt.device, dtype=torch.int8)
    amx = torch.empty((B, n_in), device=gt.device, dtype=torch.int16)
    flag = torch.empty((B,), device=gt.device, dtype=torch.int32)
    check(lib().nbm_anchor_targets(_ptr(anchors), n_in, _ptr(gt), _ptr(n_gt), B, G, float(neg_t), float(pos_t), _ptr(lab), _ptr(amx),
                                   _ptr(flag), _stream()), 'nbm_anchor_targets')
    return lab, amx, flag


def per_image_counts(n, B):
    """Count tensors are int32 [1] (one count for the whole batch: the reference's batch-coupled semantics) or int32 [B] (every
    image a batch of its own, `independent` detection) -> the `per_image` flag of the C ABI."""
    if n.numel() == 1:
        return 0
    if n.numel() != B:
        raise ValueError(f'count tensor with {n.numel()} entries for a batch of {B}')
    return 1


def rpn_select(boxes, keys, keep_count, top_n, fail_below, cap, per_image=False):
    B, KA = keys.shape
    sel_boxes = torch.empty((B, cap, 4), device=boxes.device, dtype=torch.float32)
    sel_scores = torch.empty((B, cap), device=boxes.device, dtype=torch.float32)
    n_sel = torch.empty((B if per_image and B > 1 else 1,), device=boxes.device, dtype=torch.int32)
    check(lib().nbm_rpn_select(_ptr(boxes), _ptr(keys), _ptr(keep_count), B, KA, top_n, fail_below, cap,
                               _ptr(sel_boxes), _ptr(sel_scores), _ptr(n_sel), per_image_counts(n_sel, B), _stream()), 'nbm_rpn_select')
    return sel_boxes, sel_scores, n_sel


def nms_batched(boxes, scores, n_in, thresh, post_n):
    """boxes [B,cap,4] in walk order, n_in device int -> rois [B,post_n,4], scores [B,post_n], n_out device int."""
    _chk(boxes, name='boxes'), _chk(scores, name='scores')
    B, cap = scores.shape
    words = cap // 64
    mask_ws = torch.empty((B * cap * words,), device=boxes.device, dtype=torch.int64)
    keep_ws = torch.empty((B * (cap + 1),), device=boxes.device, dtype=torch.int32)
    rois = torch.empty((B, post_n, 4), device=boxes.device, dtype=torch.float32)
    rs = torch.empty((B, post_n), device=boxes.device, dtype=torch.float32)
    per = per_image_counts(n_in, B)
    n_out = torch.empty((B if per else 1,), device=boxes.device, dtype=torch.int32)
    check(lib().nbm_nms_batched(_ptr(boxes), _ptr(scores), _ptr(n_in), B, cap, float(thresh), post_n, _ptr(mask_ws),
                                _ptr(keep_ws), _ptr(rois), _ptr(rs), _ptr(n_out), per, _stream()), 'nbm_nms_batched')
    return rois, rs, n_out


def roi_pool(fmaps, rois, n_roi, pe_f, pe_t, img_h, img_w):
    """fmaps: list of NHWC [B,h,w,C]; rois [B,cap,4]; n_roi device int32[1] -> pool, pe [B*cap,2,2,C], level [B,cap]."""
    B, cap = rois.shape[:2]
    C_ = fmaps[0].shape[-1]
    d = RoiDesc()
    for i, f in enumerate(fmaps):
        _chk(f, name=f'fmap{i}')
        d.fmap[i] = f.data_ptr()
        d.fh[i], d.fw[i] = f.shape[1], f.shape[2]
    d.n_levels, d.C = len(fmaps), C_
    d.rois, d.n_roi, d.B, d.roi_cap = _chk(rois).data_ptr(), n_roi.data_ptr(), B, cap
    d.n_roi_per_image = per_image_counts(n_roi, B)
    d.pe_f, d.pe_t, d.img_h, d.img_w = _chk(pe_f).data_ptr(), _chk(pe_t).data_ptr(), img_h, img_w
    pool = torch.zeros((B * cap, 2, 2, C_), device=rois.device, dtype=torch.float32)
    pe = torch.zeros((B * cap, 2, 2, C_), device=rois.device, dtype=torch.float32)
    level = torch.zeros((B, cap), device=rois.device, dtype=torch.int32)
    d.pool, d.pe, d.level = pool.data_ptr(), pe.data_ptr(), level.data_ptr()
    check(lib().nbm_roi_pool(C.byref(d), _stream()), 'nbm_roi_pool')
    return pool, pe, level


def rcnn_post(rois, n_roi, bbox_reg, bbox_cls, img_w, img_h, nms_thresh, min_score, proposal_number):
    """-> det [B,cap,6] rows {class,x1,y1,x2,y2,score} sorted by (class, score desc), n_det [B]."""
    B, cap = rois.shape[:2]
    n_cls1 = bbox_cls.shape[-1]
    det = torch.zeros((B, cap, 6), device=rois.device, dtype=torch.float32)
    n_det = torch.zeros((B,), device=rois.device, dtype=torch.int32)
    check(lib().nbm_rcnn_post(_ptr(_chk(rois)), _ptr(n_roi), B, cap, _ptr(_chk(bbox_reg)), _ptr(_chk(bbox_cls)),
                              n_cls1, img_w, img_h, float(nms_thresh), float(min_score), int(proposal_number),
                              _ptr(det), _ptr(n_det), per_image_counts(n_roi, B), _stream()), 'nbm_rcnn_post')
    return det, n_det


# =========================================================================== training path
from ._lib import BwdDesc  # noqa: E402


class _timed:
    """HIP-event bracket around one launch, recorded in PROFILE_BWD when that is a list (scripts/trainlayers.py)."""

    def __init__(self, tag):
        self.tag = tag

    def __enter__(self):
        if PROFILE_BWD is not None:
            self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.ev[0].record()

    def __exit__(self, *exc):
        if PROFILE_BWD is not None:
            self.ev[1].record()
            PROFILE_BWD.append((self.tag, *self.ev))


PROFILE_BWD = None


def _bwd_desc(g, *, B, H, W, Cin, N, kh, kw, stride, pad, g_ld, groups=1, alpha=1.0):
    d = BwdDesc()
    d.g = g.data_ptr()
    d.groups = groups
    d.B, d.H, d.W, d.Cin, d.N = B, H, W, Cin, N
    d.kh, d.kw, d.stride, d.pad = kh, kw, stride, pad
    d.Ho = (H + 2 * pad - kh) // stride + 1
    d.Wo = (W + 2 * pad - kw) // stride + 1
    d.g_ld = g_ld
    d.alpha = float(alpha)
    return d


def split_tn():
    """Plain weight-gradient GEMMs through the split-bf16 kernel (opt-in with NBM_SPLIT_BF16=1; NBM_SPLIT_TN=0 keeps them on the fp32
    kernel).  The C side (nbm_conv_wgrad) applies the same switch and the same shape rule."""
    return os.environ.get('NBM_SPLIT_BF16') == '1' and os.environ.get('NBM_SPLIT_TN', '1') != '0'


def split_nn():
    """Deep-K 1x1 data gradients through nbm_gemm_conv's split-bf16 kernel (opt-in with NBM_SPLIT_BF16=1; NBM_SPLIT_NN=0 keeps them on
    the fp32 data-gradient kernel)."""
    return os.environ.get('NBM_SPLIT_BF16') == '1' and os.environ.get('NBM_SPLIT_NN', '1') != '0'


def conv_dgrad(g, w, out, *, B, H, W, Cin, N, kh=1, kw=1, stride=1, pad=0, g_ld=None, w_ld=None, out_ld=None,
               a_scale=None, residual=None, mask=None, alpha=1.0, groups=1, g_gs=0, w_gs=0, out_gs=0, res_gs=0, residual2=None,
               mask_bits=None):
    """Raw nbm_conv_dgrad: out[B*H*W][Cin] = gather(g)[..][N] x W (see include/nbm_hip.h).  `residual2` [B, ceil(H/2), ceil(W/2),
    Cin]: added at the pixels with even row and column (a stride-2 shortcut's data gradient at its own resolution)."""
    if (kh == 1 and kw == 1 and stride == 1 and pad == 0 and residual2 is None and N > 256 and N % 32 == 0 and Cin > 64 and Cin % 4 == 0 and
            (groups == 1 or (mask is None and a_scale is None)) and split_nn()):
        # opt-in (NBM_SPLIT_BF16=1): a deep-K 1x1 data gradient IS the forward GEMM of the incoming gradient with the
        # transposed weights (FrozenBN scale folded in), so it takes nbm_gemm_conv's split-bf16 kernel; shortcut gradient and ReLU mask
        # ride in that kernel's epilogue (`residual`, `mask`).  The transposed copy is rebuilt per call ([N][Cin] -> [Cin][N], a few MB).
        # The rule names the LAYER (K = N, the channel counts, the operands), never the number of rows or groups: the P V product of the
        # attention (one group per image) takes the same kernel whatever the batch.
        wl = kh * kw * Cin if w_ld is None else w_ld
        wv = w.as_strided((groups, N, Cin), (w_gs, wl, 1))         # (`w` may be a row / column slice of a wider matrix: only its pitches count)
        wt = (wv * a_scale[None, :N, None] if a_scale is not None else wv).transpose(1, 2).contiguous()
        with _timed(('dgrad', B, H, W, Cin, N, kh, stride, groups)):
            gemm_conv(g, wt, out, B=B, H=H, W=W, Cin=N, N=Cin, x_ld=N if g_ld is None else g_ld, w_ld=N,
                      y_ld=Cin if out_ld is None else out_ld, residual=residual, res_ld=Cin if residual is not None else None, alpha=alpha,
                      mask=mask, mask_ld=Cin if mask is not None else None, groups=groups, x_gs=g_gs, w_gs=Cin * N, y_gs=out_gs, res_gs=res_gs)
        return out
    d = _bwd_desc(g, B=B, H=H, W=W, Cin=Cin, N=N, kh=kh, kw=kw, stride=stride, pad=pad,
                  g_ld=N if g_ld is None else g_ld, groups=groups, alpha=alpha)
    d.w, d.out = w.data_ptr(), out.data_ptr()
    d.w_ld = kh * kw * Cin if w_ld is None else w_ld
    d.out_ld = Cin if out_ld is None else out_ld
    d.a_scale = a_scale.data_ptr() if a_scale is not None else None
    d.residual = residual.data_ptr() if residual is not None else None
    d.res_ld = Cin if residual is not None else 0
    d.mask = mask.data_ptr() if mask is not None else None
    d.mask_ld = Cin if mask is not None else 0
    if mask_bits is not None and groups == 1 and Cin % 32 == 0:
        # the same mask as one bit per element (written by the producer's epilogue, `gemm_conv(bits_out=)`): 1/32 of the bytes
        assert mask_bits.dtype == torch.int32 and mask_bits.numel() * 32 >= B * H * W * Cin
        d.mask_bits, d.mask, d.mask_ld = mask_bits.data_ptr(), None, 0
    if residual2 is not None:
        if groups != 1 or tuple(residual2.shape) != (B, (H + 1) // 2, (W + 1) // 2, Cin) or kh != 1 or stride != 1:
            raise ValueError('conv_dgrad: residual2 must be [B, ceil(H/2), ceil(W/2), Cin] of a 1x1 / stride-1 data gradient')
        d.residual2, d.res2_ld = _chk(residual2, name='residual2').data_ptr(), Cin
    d.g_gs, d.w_gs, d.out_gs, d.res_gs = g_gs, w_gs, out_gs, res_gs
    if FLOPS is not None:          # executed: the stride-2 kernels visit only the taps that reach each parity class
        FLOPS[0] += 2.0 * B * d.Ho * d.Wo * N * kh * kw * Cin * groups
    with _timed(('dgrad', B, H, W, Cin, N, kh, stride, groups)):
        check(lib().nbm_conv_dgrad(C.byref(d), _stream()), 'nbm_conv_dgrad')
    return out


def conv_wgrad(g, x, out, *, B, H, W, Cin, N, kh=1, kw=1, stride=1, pad=0, g_ld=None, x_ld=None, out_ld=None,
               row_scale=None, alpha=1.0, groups=1, g_gs=0, x_gs=0, out_gs=0, bias_grad=None):
    """Raw nbm_conv_wgrad: out[N][kh*kw*Cin] += g^T x im2col(x); `out` must be zeroed (or hold a partial sum)."""
    deferred_bias = None
    if (bias_grad is not None and groups == 1 and kh == 1 and kw == 1 and stride == 1 and pad == 0 and N >= 192 and Cin > 64 and
            Cin % 4 == 0 and split_tn()):
        # the split-bf16 weight-gradient kernel (opt-in) does not sum the columns of G: the bias gradient comes from nbm_colsum
        deferred_bias, bias_grad = bias_grad, None
    d = _bwd_desc(g, B=B, H=H, W=W, Cin=Cin, N=N, kh=kh, kw=kw, stride=stride, pad=pad,
                  g_ld=N if g_ld is None else g_ld, groups=groups, alpha=alpha)
    d.x, d.out = x.data_ptr(), out.data_ptr()
    d.x_ld = Cin if x_ld is None else x_ld
    d.out_ld = kh * kw * Cin if out_ld is None else out_ld
    d.row_scale = row_scale.data_ptr() if row_scale is not None else None
    d.g_gs, d.x_gs, d.out_gs = g_gs, x_gs, out_gs
    d.bias_grad = bias_grad.data_ptr() if bias_grad is not None else None       # [N] zeros: += column sums of g
    if FLOPS is not None:
        FLOPS[0] += 2.0 * B * d.Ho * d.Wo * N * kh * kw * Cin * groups
    with _timed(('wgrad', B, H, W, Cin, N, kh, stride, groups)):
        check(lib().nbm_conv_wgrad(C.byref(d), _stream()), 'nbm_conv_wgrad')
    if deferred_bias is not None:
        gl = N if g_ld is None else g_ld
        deferred_bias += colsum(g.reshape(-1)[:B * H * W * gl].view(B * H * W, gl), N)
    return out


def relu_bwd(gy, y):
    out = torch.empty_like(gy)
    check(lib().nbm_relu_bwd(_ptr(_chk(gy)), _ptr(_chk(y)), _ptr(out), gy.numel(), _stream()), 'nbm_relu_bwd')
    return out


def leaky_relu_bwd(gy, y, slope=0.01):
    out = torch.empty_like(gy)
    check(lib().nbm_leaky_relu_bwd(_ptr(_chk(gy)), _ptr(_chk(y)), _ptr(out), float(slope), gy.numel(), _stream()),
          'nbm_leaky_relu_bwd')
    return out


def layernorm_bwd(x2d, w, g2d, eps=1e-5):
    """-> (gx [rows,E], gw [E], gb [E])."""
    rows, E = x2d.shape
    gx = torch.empty_like(x2d)
    gwb = torch.zeros((2, E), device=x2d.device, dtype=torch.float32)
    check(lib().nbm_layernorm_bwd(_ptr(_chk(x2d)), _ptr(_chk(w)), _ptr(_chk(g2d)), rows, E, float(eps), _ptr(gx), _ptr(gwb[0]),
                                  _ptr(gwb[1]), _stream()), 'nbm_layernorm_bwd')
    return gx, gwb[0], gwb[1]


def mha_small_bwd(q, k, v, go, S, N, nhead, seq_stride, batch_stride, n_valid=None):
    """Gradients of `mha_small` wrt q, k, v (2-D row views, any row pitch) -> contiguous [S*N, E] each."""
    E = q.shape[1]
    hd = E // nhead
    gq, gk, gv = (torch.zeros((q.shape[0], E), device=q.device, dtype=torch.float32) for _ in range(3))
    ws = torch.empty((N * nhead, 2, S, S), device=q.device, dtype=torch.float32)
    go = _chk(go)
    check(lib().nbm_mha_small_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(go), q.stride(0), k.stride(0), v.stride(0), go.stride(0),
                                  _ptr(gq), _ptr(gk), _ptr(gv), E, E, E, _ptr(ws), S, N, nhead, hd, seq_stride, batch_stride,
                                  _ptr(n_valid), 1.0 / math.sqrt(hd), _stream()), 'nbm_mha_small_bwd')
    return gq, gk, gv


def silu_bwd(gy, x):
    out = torch.empty_like(gy)
    check(lib().nbm_silu_bwd(_ptr(_chk(gy)), _ptr(_chk(x)), _ptr(out), gy.numel(), _stream()), 'nbm_silu_bwd')
    return out


def axpby(a, b=None, alpha=1.0, beta=1.0):
    """alpha*a + beta*b; a smaller `b` is repeated along the leading axes (its size must divide a's)."""
    out = torch.empty_like(a)
    period = 0
    if b is not None and b.numel() != a.numel():
        assert a.numel() % b.numel() == 0
        period = b.numel()
    check(lib().nbm_axpby(_ptr(_chk(a)), _ptr(b), _ptr(out), float(alpha), float(beta), a.numel(), period, _stream()),
          'nbm_axpby')
    return out


def weighted_sum(xs, weights):
    """BiFPN fusion of 2 or 3 same-shaped tensors with ReLU'd learnable weights [len(xs)] (device)."""
    x2 = xs[2] if len(xs) == 3 else None
    out = torch.empty_like(xs[0])
    check(lib().nbm_weighted_sum(_ptr(_chk(xs[0])), _ptr(_chk(xs[1])), _ptr(x2), _ptr(_chk(weights)), _ptr(out), out.numel(),
                                 _stream()), 'nbm_weighted_sum')
    return out


def weighted_sum_bwd(xs, weights, g, need):
    """-> ([gx_i or None], gw [len(xs)])."""
    x2 = xs[2] if len(xs) == 3 else None
    gxs = [torch.empty_like(x) if nd else None for x, nd in zip(xs, need)] + [None] * (3 - len(xs))
    gw = torch.zeros((len(xs),), device=g.device, dtype=torch.float32)
    check(lib().nbm_weighted_sum_bwd(_ptr(_chk(xs[0])), _ptr(_chk(xs[1])), _ptr(x2), _ptr(_chk(weights)), _ptr(_chk(g)),
                                     _ptr(gxs[0]), _ptr(gxs[1]), _ptr(gxs[2]), _ptr(gw), g.numel(), _stream()),
          'nbm_weighted_sum_bwd')
    return gxs[:len(xs)], gw


def colsum(g2d, n=None):
    """g2d [M, ld] -> [n] column sums (n defaults to ld)."""
    _chk(g2d, name='g')
    M, ld = g2d.shape
    n = ld if n is None else n
    out = torch.empty((n,), device=g2d.device, dtype=torch.float32)
    check(lib().nbm_colsum(_ptr(g2d), M, n, ld, _ptr(out), _stream()), 'nbm_colsum')
    return out


def maxpool3x3s2_bwd(idx, gy, H, W, residual=None, mask=None):
    """`residual` [B,H,W,C]: added to the result (the gradient another consumer of the pooled tensor has produced); `mask`
    [B,H,W,C]: result *= (mask > 0) after that addition."""
    B, Ho, Wo, C_ = gy.shape
    gx = torch.empty((B, H, W, C_), device=gy.device, dtype=torch.float32)
    check(lib().nbm_maxpool3x3s2_bwd(_ptr(idx), _ptr(_chk(gy)), _ptr(gx), B, H, W, C_, Ho, Wo, _ptr(residual),
                                     _ptr(_chk(mask, name='mask')) if mask is not None else None, _stream()),
          'nbm_maxpool3x3s2_bwd')
    return gx


def upsample_bilinear_bwd(gy, Hi, Wi, pattern_stride=0, tiles_share=()):
    """d/d(coarse map) of the bilinear top-down merge.  `pattern_stride` = S: gy is zero outside the 5x5 patches around the 3x3 /
    stride-S pattern, only those are read; `tiles_share`: (compact [n][2][2][C], tile list, first image, images) entries whose
    bilinear backward is ADDED on top (the RoI share of a demand-driven level's gradient, kept out of gy -- ondemand.UPBWD_SPLIT)."""
    B, Ho, Wo, C_ = gy.shape
    gs = torch.empty((B, Hi, Wi, C_), device=gy.device, dtype=torch.float32)
    check(lib().nbm_upsample_bilinear_bwd(_ptr(_chk(gy)), B, Hi, Wi, C_, _ptr(gs), Ho, Wo, int(pattern_stride), _stream()),
          'nbm_upsample_bilinear_bwd')
    for compact, tiles, b0, nb in tiles_share:
        if compact.shape != (tiles.numel() * 4, C_) or b0 < 0 or b0 + nb > B:
            raise ValueError('upsample_bilinear_bwd: compact operand / tile list / image range do not fit the map')
        check(lib().nbm_tiles_upsample_bilinear_bwd_add(_ptr(_chk(compact)), nb, Ho, Wo, C_, _ptr(tiles), tiles.numel(), None,
                                                        C.c_void_p(gs.data_ptr() + b0 * Hi * Wi * C_ * 4), Hi, Wi, _stream()),
              'nbm_tiles_upsample_bilinear_bwd_add')
    return gs


def softmax_rows_bwd(p2d, gp2d, alpha=1.0):
    rows, cols = p2d.shape
    out = torch.empty_like(p2d)
    check(lib().nbm_softmax_rows_bwd(_ptr(_chk(p2d)), _ptr(_chk(gp2d)), _ptr(out), rows, cols, float(alpha), _stream()),
          'nbm_softmax_rows_bwd')
    return out


def pair_softmax_bwd(y, gy):
    gx = torch.empty_like(y)
    check(lib().nbm_pair_softmax_bwd(_ptr(_chk(y)), _ptr(_chk(gy)), _ptr(gx), y.numel() // 2, _stream()),
          'nbm_pair_softmax_bwd')
    return gx


def dwconv3x3_bwd(x, g, w, mult, stride, need_gx=True, need_gw=True, has_bias=True):
    B, H, W, Cin = x.shape
    Ho, Wo = g.shape[1], g.shape[2]
    gx = torch.empty_like(x) if need_gx else None
    gw = torch.empty((Cin * mult, 1, 3, 3), device=x.device, dtype=torch.float32) if need_gw else None
    gb = torch.empty((Cin * mult,), device=x.device, dtype=torch.float32) if (need_gw and has_bias) else None
    check(lib().nbm_dwconv3x3_bwd(_ptr(_chk(x)), _ptr(_chk(g)), _ptr(_chk(w)), B, H, W, Cin, mult, stride, _ptr(gx),
                                  _ptr(gw), _ptr(gb), Ho, Wo, _stream()), 'nbm_dwconv3x3_bwd')
    return gx, gw, gb


def dwconv3x3_bwd_acc(g, w, mult, stride, gx):
    """gx [B,H,W,Cin] += data gradient of the depthwise 3x3 (only the pixels a tap reaches are touched)."""
    B, H, W, Cin = gx.shape
    check(lib().nbm_dwconv3x3_bwd_acc(_ptr(_chk(g)), _ptr(_chk(w)), B, H, W, Cin, mult, stride, _ptr(_chk(gx)), g.shape[1], g.shape[2],
                                      _stream()), 'nbm_dwconv3x3_bwd_acc')


def film_fwd(z, film):
    y = torch.empty_like(z)
    C_ = z.shape[-1]
    check(lib().nbm_film_fwd(_ptr(_chk(z)), _ptr(_chk(film)), _ptr(y), z.numel() // C_, C_, _stream()), 'nbm_film_fwd')
    return y


def film_bwd(gy, z, film):
    gz, gf = torch.empty_like(z), torch.empty_like(film)
    C_ = z.shape[-1]
    check(lib().nbm_film_bwd(_ptr(_chk(gy)), _ptr(_chk(z)), _ptr(_chk(film)), _ptr(gz), _ptr(gf), z.numel() // C_, C_,
                             _stream()), 'nbm_film_bwd')
    return gz, gf


def bn_train_fwd(x2d, w, b, eps, momentum, run_mean, run_var):
    """x2d [M,C] -> (y, mean, invstd); running stats updated in place."""
    M, C_ = x2d.shape
    ws = torch.empty((2 * C_,), device=x2d.device, dtype=torch.float64)
    mean = torch.empty((C_,), device=x2d.device, dtype=torch.float32)
    invstd = torch.empty_like(mean)
    y = torch.empty_like(x2d)
    check(lib().nbm_bn_train_fwd(_ptr(_chk(x2d)), M, C_, _ptr(_chk(w)), _ptr(_chk(b)), float(eps), float(momentum),
                                 _ptr(run_mean), _ptr(run_var), _ptr(ws), _ptr(mean), _ptr(invstd), _ptr(y), _stream()),
          'nbm_bn_train_fwd')
    return y, mean, invstd


def bn_train_bwd(g2d, x2d, mean, invstd, w):
    M, C_ = x2d.shape
    ws = torch.empty((2 * C_,), device=x2d.device, dtype=torch.float64)
    gx = torch.empty_like(x2d)
    gw = torch.empty((C_,), device=x2d.device, dtype=torch.float32)
    gb = torch.empty_like(gw)
    check(lib().nbm_bn_train_bwd(_ptr(_chk(g2d)), _ptr(_chk(x2d)), M, C_, _ptr(mean), _ptr(invstd), _ptr(_chk(w)), _ptr(ws),
                                 _ptr(gx), _ptr(gw), _ptr(gb), _stream()), 'nbm_bn_train_bwd')
    return gx, gw, gb


def roi_pool_bwd(gpool, rois, level, fmap_shapes, pooled=None, bases=None):
    """gpool [B*R,2,2,C] -> list of zero-initialised-then-accumulated gradient maps (NHWC) for the FPN levels.
    `pooled` {level index: stride of the RPN's depthwise convolution on that map}: those maps come from the persistent pool of
    `ondemand.zero_acquire` (demand-driven levels: the consumer of the gradient recycles them) instead of a fresh fill."""
    B, R = rois.shape[:2]
    C_ = gpool.shape[-1]
    gf = []
    for i, s in enumerate(fmap_shapes):
        buf = None
        if bases and i in bases:             # another consumer's share of this map's gradient (Fn._PARKED): scatter into it
            base, e = bases[i]
            if e is not None:                # a persistent map: the RoI windows join its footprint
                from . import ondemand
                ondemand.zero_note(e, lambda b_=base, s_=s, i_=i: check(lib().nbm_zero_roi_windows(
                    _ptr(b_), s_[1], s_[2], s_[3], _ptr(rois), _ptr(level), B, R, i_, _stream()), 'nbm_zero_roi_windows'))
                if ondemand.ZERO_POOL_CHECK:
                    e['check'] = _check_all_zero
            gf.append(base)
            continue
        if pooled and i in pooled:
            from . import ondemand
            buf, e = ondemand.zero_acquire(s, gpool.device, ('map-grad', pooled[i]))
        if buf is not None:
            # footprints: the RoI windows of this level, and the 3x3 blocks the strided taps add to (Fn.DwConv.backward)
            ondemand.zero_note(e, lambda b_=buf, s_=s, i_=i: check(lib().nbm_zero_roi_windows(
                _ptr(b_), s_[1], s_[2], s_[3], _ptr(rois), _ptr(level), B, R, i_, _stream()), 'nbm_zero_roi_windows'))
            ondemand.zero_note(e, lambda b_=buf, s_=s, st_=pooled[i]: check(lib().nbm_zero_pattern(
                _ptr(b_), s_[0], s_[1], s_[2], s_[3], st_, _stream()), 'nbm_zero_pattern'))
            if ondemand.ZERO_POOL_CHECK:
                e['check'] = _check_all_zero
            gf.append(buf)
        else:
            gf.append(torch.zeros(s, device=gpool.device, dtype=torch.float32))
    n = len(gf)
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in gf])
    fh = (C.c_int * n)(*[s[1] for s in fmap_shapes])
    fw = (C.c_int * n)(*[s[2] for s in fmap_shapes])
    check(lib().nbm_roi_pool_bwd(ptrs, fh, fw, n, C_, _ptr(_chk(rois)), _ptr(level), B, R, _ptr(_chk(gpool)), _stream()),
          'nbm_roi_pool_bwd')
    return gf


def zero_pattern(buf, stride):
    B, H, W, C_ = buf.shape
    check(lib().nbm_zero_pattern(_ptr(buf), B, H, W, C_, int(stride), _stream()), 'nbm_zero_pattern')


def _check_all_zero(buf):
    bad = float(buf.abs().max())
    if bad != 0.0:
        raise RuntimeError(f'persistent RoI-gradient map not clean after its recycle: {bad}')


def sqnorm_accum(g, out):
    check(lib().nbm_sqnorm_accum(_ptr(g), g.numel(), _ptr(out), _stream()), 'nbm_sqnorm_accum')


def adamw_step(p, g, m, v, lr, beta1, beta2, eps, wd, step, sqnorm=None, max_norm=0.0):
    check(lib().nbm_adamw_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), float(lr), float(beta1), float(beta2),
                               float(eps), float(wd), int(step), _ptr(sqnorm), float(max_norm), _stream()), 'nbm_adamw_step')
