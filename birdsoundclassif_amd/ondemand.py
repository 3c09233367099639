"""Demand-driven evaluation of the finest FPN level: host side of the `*_tiles` / `rows` entry points.

The output map of `fpn.out_convs.4` has two consumers in the reference (layers.py:62-65,81 and :408-417,464-467): the RPN's
depthwise 3x3 with stride anchor_stride / 2 = 8 -- a fixed pixel pattern, 25 % of the 2 x 2 Winograd tiles -- and the RoI pooling
of the RoIs assigned to that level.  `conv3x3_winograd_lazy` computes the pattern tiles at once and remembers the operands;
`lazy_complete` (called by the RoI pooling) computes the tiles under the RoI windows; `conv1x1_lazy` evaluates the lateral 1x1
+ merge in front of it on the pixels those tiles read; `conv3x3_winograd_{w,d}grad_tiles` are the backward passes over the same
lists.  Every other pixel of the two maps is never read by anything and is left unwritten.  NBM_LAZY_FINEST=0 switches the
whole mechanism off (dense maps).
"""
import ctypes as C
import os
import weakref

import numpy as np
import torch

from . import ops
from .ops import check, lib, _ptr, _chk, _stream, gemm_conv, conv_wgrad

LAZY_FINEST = os.environ.get('NBM_LAZY_FINEST', '1') != '0'
LAZY_LATERAL = os.environ.get('NBM_LAZY_LATERAL', '1') != '0'      # also the lateral 1x1 + merge in front of the map
LAZY_POISON = False                 # tests: fill the map with NaN first, so that a read of an unwritten pixel shows
_LAZY = {}                          # data_ptr of the sparse map -> LazyMap
_PATTERNS = {}
_ROI_TILE_BUF = {}


class TilePattern:
    """Tiles of a [B,H,W] map that hold a pixel read by a 3x3 / stride / pad 1 consumer, grouped by WHICH of their 2 x 2
    pixels are read (all four; the second row only; the second column only; the last pixel only -- with stride 8 the pattern
    enters three tiles out of four through one row, one column or one pixel):
      tiles     int32 [n_entries] on the device: per class the linear ids, ascending over the batch, -1 padded to 128
      blk_info  uint32 [n_entries / 128]: planes to compute | pixels to store << 16 (nbm_hip.h) -- 16 / 12 / 12 / 9 planes
      full      uint8 [TH*TW]: 1 where all four pixels of the tile are computed (the RoI phase skips those)
      any       bool  [TH*TW]: 1 where the tile is in the pattern at all (its weight-gradient term comes from this list)
      n         number of listed tiles;  n_eff = sum(planes x tiles) / 16: the executed-FLOP equivalent in 16-plane tiles
      frac      covered fraction of all tiles
      entry_pm  int32 [n_entries]: the plane mask of every list entry;  tile_pm int32 [TH*TW]: the plane mask of the tile's
                class (0 = not in the pattern) -- the weight gradient counts every plane of every tile once
      px_rows   int32 [n_px padded to 128]: the INPUT pixels these tiles read through the planes they compute (b*H*W + y*W + x,
                ascending, -1 padded): with stride 8 the 5 x 5 neighbourhoods of the pattern, 39 % of the map -- the rows of
                the lateral 1x1 convolution that feeds the demand-driven convolution"""
    __slots__ = ('tiles', 'blk_info', 'full', 'any', 'n', 'n_eff', 'frac', 'px_rows', 'n_px', 'entry_pm', 'tile_pm')


def wino23_pattern(B, H, W, stride, device, dilate=0):
    """`dilate` = 1: the tiles / pixels within one pixel of the pattern instead -- where the DATA gradient of the 3x3 convolution
    that produced the pattern pixels is non-zero."""
    key = (B, H, W, stride, str(device), dilate)
    hit = _PATTERNS.get(key)
    if hit is None:
        TH, TW = (H + 1) // 2, (W + 1) // 2

        def need(n, nt):                          # per tile row: which of its two pixel rows are read (bit 0: first, bit 1: second)
            m = np.zeros(nt, dtype=np.int64)
            for o in range((n + 2 - 3) // stride + 1):
                for k in range(3):
                    for r in range(o * stride - 1 + k - dilate, o * stride - 1 + k + dilate + 1):
                        if 0 <= r < n:
                            m[r >> 1] |= 1 << (r & 1)
            return m
        ry, rx = need(H, TH), need(W, TW)
        # planes i (rows of A^T = [1 1 1 0; 0 1 -1 -1]) a tile row needs: first pixel row -> {0,1,2}, second -> {1,2,3}
        planes_of = {0: 0, 1: 0b0111, 2: 0b1110, 3: 0b1111}
        lists, infos, n, n_eff = [], [], 0, 0.0
        for cy in (3, 2, 1):
            for cx in (3, 2, 1):
                m2 = (ry == cy)[:, None] & (rx == cx)[None, :]
                ids = np.flatnonzero(m2.ravel()).astype(np.int64)
                if ids.size == 0:
                    continue
                allt = (np.arange(B, dtype=np.int64)[:, None] * (TH * TW) + ids[None, :]).ravel()
                pm = 0
                for i in range(4):
                    for j in range(4):
                        if (planes_of[cy] >> i) & 1 and (planes_of[cx] >> j) & 1:
                            pm |= 1 << (4 * i + j)
                sm = 0
                for pp in range(2):
                    for qq in range(2):
                        if (cy >> pp) & 1 and (cx >> qq) & 1:
                            sm |= 1 << (2 * pp + qq)
                pad = (-allt.size) % 128
                lists.append(np.concatenate([allt, np.full(pad, -1, dtype=np.int64)]))
                infos.append(np.full((allt.size + pad) // 128, pm | (sm << 16), dtype=np.int64))
                n += allt.size
                n_eff += allt.size * bin(pm).count('1') / 16.0
        anym = (ry > 0)[:, None] & (rx > 0)[None, :]
        full = (ry == 3)[:, None] & (rx == 3)[None, :]
        hit = _PATTERNS[key] = TilePattern()
        # interleave the classes' blocks by relative position: the fused kernel hands contiguous block ranges to the 8 XCDs, and a
        # range made of 16-plane blocks only would finish long after a range of 9-plane blocks (measured: no gain without this)
        blocks = np.concatenate(lists).reshape(-1, 128)
        infos = np.concatenate(infos)
        pos = np.concatenate([(np.arange(len(i)) + 0.5) / len(i) for i in [l.reshape(-1, 128) for l in lists]])
        order = np.argsort(pos, kind='stable')
        # ... and inside each XCD's range the blocks with the most planes first, so that the range ends on short blocks
        nblk = len(order)
        npl = np.array([bin(int(v) & 0xffff).count('1') for v in infos])
        for x8 in range(8):
            lo, hi = nblk * x8 // 8, nblk * (x8 + 1) // 8
            seg = order[lo:hi]
            order[lo:hi] = seg[np.argsort(-npl[seg], kind='stable')]
        infos = infos[order]
        hit.entry_pm = torch.from_numpy(np.repeat((infos & 0xffff) | 0x10000, 128).astype(np.int32)).to(device)   # bit 16: counts in the bias gradient
        tpm = np.zeros(TH * TW, dtype=np.int64)
        for cy in (3, 2, 1):
            for cx in (3, 2, 1):
                pmv = sum(1 << (4 * i + j) for i in range(4) for j in range(4)
                          if (planes_of[cy] >> i) & 1 and (planes_of[cx] >> j) & 1)
                tpm[((ry == cy)[:, None] & (rx == cx)[None, :]).ravel()] = pmv
        hit.tile_pm = torch.from_numpy(tpm.astype(np.int32)).to(device)
        hit.tiles = torch.from_numpy(blocks[order].ravel().astype(np.int32)).to(device)
        hit.blk_info = torch.from_numpy(infos.astype(np.uint32).view(np.int32)).to(device)
        hit.full = torch.from_numpy(full.ravel().astype(np.uint8)).to(device)
        hit.any = torch.from_numpy(anym.ravel()).to(device)
        hit.n, hit.n_eff, hit.frac = n, n_eff, float(anym.mean())
        # input rows a tile row reads: both output rows -> 2ty-1 .. 2ty+2; second only (planes i = 1..3) -> 2ty .. 2ty+2; first
        # only (i = 0..2) -> 2ty-1 .. 2ty+1; same for columns
        def px_need(r, n_):
            m = np.zeros(n_, dtype=bool)
            for t, c in enumerate(r):
                if c:
                    lo, hi = (2 * t - 1 if c & 1 else 2 * t), (2 * t + 2 if c & 2 else 2 * t + 1)
                    m[max(lo, 0):min(hi, n_ - 1) + 1] = True
            return m
        pm = (px_need(ry, H)[:, None] & px_need(rx, W)[None, :]).ravel()
        ids = np.flatnonzero(pm).astype(np.int64)
        allp = (np.arange(B, dtype=np.int64)[:, None] * (H * W) + ids[None, :]).ravel()
        hit.n_px = allp.size
        hit.px_rows = torch.from_numpy(np.concatenate([allp, np.full((-allp.size) % 128, -1, dtype=np.int64)]).astype(np.int32)).to(device)
    return hit


def _wino23_tiles_run(x, U, bias, y_ptr, tiles, n_blocks, n_listed, label, blk_info=None, dense_rows=False, skip_pattern=0,
                      accumulate=False, compact=False):
    """Rows transform + fused kernel for the listed tiles of x [B,H,W,C] -> pixels of the map at device address y_ptr.
    n_listed: executed work in 16-plane tile equivalents (None: device-side count in n_blocks).  skip_pattern = S: the pixels of the
    3x3 / stride-S pattern of x read as zeros; accumulate: the result is added to the map (every tile listed once)."""
    B, H, W, C_ = x.shape
    N = U.shape[1]
    per_img = 4 * (-(-H // 2)) * (2 * (-(-W // 2)) + 2) * C_
    R, _ = ops._wino_scratch(x.device, B * per_img, 0)
    st = _stream()
    nb_ptr = _ptr(n_blocks) if n_blocks is not None else None
    if ops.FLOPS is not None:
        if n_listed is None:                      # device-side count: resolved by flops_total(), no sync here
            ops.FLOPS_DEFERRED.append((n_blocks.clone(), 2.0 * 16 * 128 * C_ * N))
        else:
            ops.FLOPS[0] += 2.0 * 16 * n_listed * C_ * N
    prof = ops._prof_fused()
    if prof:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        if ops._prof_all():
            ev[0].record()
    if dense_rows and not skip_pattern:           # every tile is listed: the plain row transform writes half the bytes
        check(lib().nbm_wino23_rows(_ptr(x), B, H, W, C_, _ptr(R), st), 'nbm_wino23_rows')
    else:
        check(lib().nbm_wino23_rows_tiles(_ptr(x), B, H, W, C_, _ptr(tiles), tiles.numel(), nb_ptr, _ptr(blk_info), _ptr(R), int(skip_pattern),
                                          st), 'nbm_wino23_rows_tiles')
    if prof:
        ev[1].record()
    check(lib().nbm_wino23_conv_fused_tiles(_ptr(R), _ptr(U), None, _ptr(bias), None, (2 if accumulate else 0) | (4 if compact else 0), B, H, W, C_, N, C.c_void_p(y_ptr), _ptr(tiles),
                                            tiles.numel(), nb_ptr, _ptr(blk_info), st), 'nbm_wino23_conv_fused_tiles')
    if prof:
        ev[2].record()
        # listed tiles: known on the host for the pattern, a device counter for the RoI tiles (resolved by the reader after a sync)
        cnt = n_listed if n_listed is not None else n_blocks.clone()
        ops.PROFILE.append(((C_, N, 1, cnt, 1, 1, 16, 1, (label, H, W)), ev[1], ev[2]))
        if ops._prof_all():
            ops.PROFILE.append(((label, C_, N, H, W, B), ev[0], ev[2]))


def lazy_chunk(x):
    """Images per launch of the sparse path: the row-transform scratch keeps its dense layout (holes unwritten)."""
    _, H, W, C_ = x.shape
    per_img = 4 * (-(-H // 2)) * (2 * (-(-W // 2)) + 2) * C_ * 4
    return max(1, min(x.shape[0], ops.WINO_CHUNK_BYTES // per_img))


class LazyMap:
    """Book-keeping of one demand-driven map: operands and, per batch chunk, the tile lists that were computed (pattern
    list; per RoI pooling on the map a RoI list + its block count on the way to the host) -- the weight gradient sums over
    them.  The map itself is NOT referenced (an autograd node owns this object and the map owns the node: a cycle would keep
    12 GB alive until the garbage collector runs); its consumers keep it alive and hand it back to `lazy_complete`.
    The operands stay here for as long as the map lives, so EVERY RoI pooling on the map -- not only the first -- finds the
    tiles under its windows computed (`done` counts them); the state goes when the map does (`_forget`)."""
    __slots__ = ('x', 'U', 'bias', 'skip', 'stride', 'chunks', 'roi', 'keep', 'sparse', 'overlap', 'lateral', 'rois', 'done', 'vg', 'cell_gb', 'cell_gb_done', 'vx', 'raw', 'pending', 'comp', '__weakref__')

    def __init__(self, x, U, bias, stride):
        self.x, self.U, self.bias, self.stride = x, U, bias, stride
        self.skip, self.chunks, self.keep, self.sparse, self.lateral = None, [], False, True, None
        self.overlap = False         # dense-looking level (every tile holds a pattern pixel) whose backward pass still goes through the cells
        self.roi = []           # per RoI pooling: per chunk (tile list, pinned block count, event)
        self.rois = []          # per RoI pooling: (rois, n_roi, n_levels, level, fh, fw)
        self.done = 0
        self.vg = None          # backward pass: {chunk: Vg} of the cell transforms, shared by the data and the weight gradient
        self.cell_gb = None     # ... and the bias gradient of the pattern pixels, summed by the same kernel
        self.cell_gb_done = None    # chunks whose pattern pixels are in cell_gb already (a chunk's Vg may be computed twice)
        self.vx = None          # deferred lateral + a backward pass to come: {chunk: [transform(up + b) | transform(t)]} of the forward pass
        self.raw = None         # (weight, bias) of the convolution as the module holds them (rpn_composite)
        self.pending = None     # the pattern pass was NOT run (Ucell, Ufold) -- rpn_composite (eval) / train_composite_forward, or pattern_materialize
        self.comp = None        # training: state of the RPN reader composed with this convolution in the cell domain (train_composite_*)

    def release(self):
        """Called by the backward pass of the convolution: no RoI pooling can follow on a map whose gradient has been consumed."""
        self.x = self.U = self.bias = self.lateral = self.vg = self.cell_gb = self.cell_gb_done = self.vx = self.raw = self.pending = self.comp = None

    def __del__(self):          # the pinned counters go back to the pool
        try:
            for per_chunk in self.roi:
                for entry in per_chunk:         # (tiles, host, event, dilated tiles, its host counter or None)
                    _PINNED_FREE.extend(h for h in (entry[1], entry[4] if len(entry) > 4 else None) if h is not None)
        except Exception:       # interpreter shutdown
            pass


_PINNED_FREE = []


def _pinned_int():
    """A pinned host int32 of its own for one (LazyMap, RoI pooling, chunk): taken from a pool that grows on demand (during the
    warm-up steps: hipHostMalloc stalls the stream) and refilled when the LazyMap dies -- never shared between two maps that are
    both waiting for their backward pass."""
    if not _PINNED_FREE:
        _PINNED_FREE.extend(torch.empty((1,), dtype=torch.int32, pin_memory=True) for _ in range(16))
    return _PINNED_FREE.pop()


_LAZY_LATERAL = {}                  # data_ptr of a sparse lateral map -> (LateralState, weakref to the map)


class LateralState:
    """Operands of a lateral 1x1 convolution (+ top-down merge) whose output only exists where its one consumer, a demand-driven
    3x3 convolution, reads it; `conv3x3_winograd_lazy` picks the state up and `lazy_complete` finishes the pixels under the RoI
    tiles before it convolves them.  `deferred`: not even the pattern patches were computed -- the consumer takes the lateral's
    OPERANDS into its cell-domain GEMMs instead (`conv3x3_winograd_lazy`, Ufold)."""
    __slots__ = ('t', 'wk', 'bias', 'alpha', 'up', 'deferred', 'stride', 'ufold', 'ufold_t', 'grads')

    def __init__(self, t, wk, bias, alpha, up, deferred=False, stride=0):
        self.t, self.wk, self.bias, self.alpha, self.up, self.deferred, self.stride = t, wk, bias, alpha, up, deferred, stride
        self.ufold = None            # [25][N][C + Cin] of the consumer's forward pass (deferred lateral): kept for the backward pass
        self.ufold_t = None          # ... and the same values as [25][C + Cin][N] (B operand of the data-gradient GEMMs)
        self.grads = None            # the lateral's own gradients, when the consumer's backward pass produced them (LAT_CELL_BWD)


ZERO_FILL = False                   # functional.py sets it when a DENSE backward kernel may read a sparse map (A/B switches off)


def _sparse_map(shape, device):
    """Storage of a map of which only the pixels with a reader are ever written.  torch.empty by default; zeros when a dense
    backward pass will multiply the holes by exact-zero gradients (0 x garbage could be NaN); NaN under LAZY_POISON (tests)."""
    if LAZY_POISON:
        return torch.full(shape, float('nan'), device=device, dtype=torch.float32)
    if ZERO_FILL and torch.is_grad_enabled():
        return torch.zeros(shape, device=device, dtype=torch.float32)
    return torch.empty(shape, device=device, dtype=torch.float32)


def _forget(table, key, ident):
    hit = table.get(key)
    if hit is not None and id(hit[0]) == ident:
        del table[key]


FUSED_FINEST = os.environ.get('NBM_FUSED_FINEST', '1') != '0'     # the lateral's operands go into the consumer's cell-domain GEMMs


def _lateral_pattern_pass(x, ls, stride):
    """The lateral + merge on the pattern patches (`TilePattern.px_rows`) into the sparse map x."""
    B, H, W, N = x.shape
    chunk = lazy_chunk(x)                      # the same batch chunks as the convolution that follows
    for b0 in range(0, B, chunk):
        nb = min(chunk, B - b0)
        pat = wino23_pattern(nb, H, W, stride, x.device)
        gemm_conv(ls.t[b0:b0 + nb], ls.wk, x[b0:b0 + nb], B=nb, H=H, W=W, Cin=ls.t.shape[-1], N=N, w_ld=ls.wk.shape[1], shift=ls.bias,
                  alpha=ls.alpha, up=ls.up[b0:b0 + nb] if ls.up is not None else None, rows=pat.px_rows, rows_mode=1,
                  rows_count=pat.px_rows.numel())


def conv1x1_lazy(t, wk, bias, alpha, up, stride, defer=False):
    """Lateral 1x1 convolution + bilinear top-down merge (fpn.py:143-144) on the pixels that the pattern tiles of the following
    demand-driven 3x3 convolution read (`TilePattern.px_rows`): t [B,H,W,Cin] -> x [B,H,W,N], other pixels unwritten.  Same
    kernel, same arithmetic per pixel as the dense call.
    `defer` (round 3): write NOTHING now.  The consumer's pattern pass works in the cell domain, where the merged map
    x = alpha W t + b + up(x1) need not exist: its patch transform is [transform(up(x1) + b) | transform(t)], and the lateral's
    weights fold into the consumer's ([U | alpha U W], `_prep.cell_weight_folded`).  That removes this GEMM -- with its gathered
    bilinear epilogue the kernel furthest below its roofs in round 2 (3.0 ms at B = 64) -- and the patch transform of its output.
    A consumer that cannot take the operands runs the pattern pass itself (`_lateral_pattern_pass`)."""
    _chk(t, name='t'), _chk(wk, name='w')
    B, H, W, Cin = t.shape
    N = wk.shape[0]
    x = _sparse_map((B, H, W, N), t.device)
    defer = bool(defer and FUSED_FINEST and CELL_FWD and up is not None and stride >= 3 and Cin % 4 == 0 and N % 32 == 0 and
                 (N + Cin) % 32 == 0 and wk.shape[1] == Cin)
    ls = LateralState(t, wk, bias, alpha, up, deferred=defer, stride=stride)
    if not defer:
        _lateral_pattern_pass(x, ls, stride)
    for k in [k for k, v in _LAZY_LATERAL.items() if v[1]() is None]:
        del _LAZY_LATERAL[k]
    _LAZY_LATERAL[x.data_ptr()] = (ls, weakref.ref(x))
    weakref.finalize(x, _forget, _LAZY_LATERAL, x.data_ptr(), id(ls))
    return x


CELL_FWD = os.environ.get('NBM_CELL_FWD', '1') != '0'     # pattern pixels of the FORWARD pass through the cell transforms


def _cell_operand(st, b0, nb, H, W, C_, x, n_out=0, ci=None):
    """A operand of the cell-domain plane GEMMs of batch chunk [b0, b0 + nb): the transformed 5x5 input patches, [25][cells][K].
    From the map x itself (K = C), or -- deferred lateral -- from the lateral's operands: [transform(up(x1) + b) | transform(t)],
    K = C + Cin.  -> (V, M: views of the shared scratch, M with room for [25][cells][n_out]; K; cells).  Deferred lateral with a
    backward pass to come (`st.keep`, chunk index `ci`): V is an allocation of its own, kept in st.vx for the weight gradient (8.8 GB
    at B = 128 for -4 ms: the interpolating transform is not run twice)."""
    T = cell_count(nb, H, W, st.stride)
    stream = _stream()
    lt = st.lateral
    if lt is not None and lt.deferred:
        Cin = lt.t.shape[-1]
        K = C_ + Cin
        if st.vx is not None and ci in st.vx:            # the forward pass left it here
            return st.vx.pop(ci), ops._wino_scratch(lt.t.device, 25 * T * n_out, 0)[0], K, T
        if st.keep and ci is not None and n_out:         # forward pass, a backward pass will follow
            if st.vx is None:
                st.vx = {}
            V = st.vx[ci] = torch.empty((25 * T * K,), device=lt.t.device, dtype=torch.float32)
            M = ops._wino_scratch(lt.t.device, 25 * T * n_out, 0)[0]
        else:
            V, M = ops._wino_scratch(lt.t.device, 25 * T * K, 25 * T * n_out)
        check(lib().nbm_cell_input_up(_ptr(lt.up[b0:b0 + nb]), _ptr(lt.bias), nb, H, W, C_, lt.up.shape[1], lt.up.shape[2], st.stride,
                                      _ptr(V), K, 0, stream), 'nbm_cell_input_up')
        check(lib().nbm_cell_input(_ptr(lt.t[b0:b0 + nb]), nb, H, W, Cin, st.stride, _ptr(V), K, C_, stream), 'nbm_cell_input')
        return V, M, K, T
    if st.vx is not None and ci in st.vx:                # the forward pass left it here
        return st.vx.pop(ci), ops._wino_scratch(x.device, 25 * T * n_out, 0)[0], C_, T
    if st.keep and ci is not None and n_out and (st.sparse or st.overlap) and CELL_BWD:
        # forward pass, the cell-domain weight gradient will want the same operand (7.5 GB at B = 128 for level P2: -2.4 ms)
        if st.vx is None:
            st.vx = {}
        V = st.vx[ci] = torch.empty((25 * T * C_,), device=x.device, dtype=torch.float32)
        M = ops._wino_scratch(x.device, 25 * T * n_out, 0)[0]
    else:
        V, M = ops._wino_scratch(x.device, 25 * T * C_, 25 * T * n_out)
    check(lib().nbm_cell_input(_ptr(x[b0:b0 + nb]), nb, H, W, C_, st.stride, _ptr(V), C_, 0, stream), 'nbm_cell_input')
    return V, M, C_, T


COMPOSITE = os.environ.get('NBM_RPN_COMPOSITE', '1') != '0'     # evaluation mode: the RPN's reader composed with this convolution (rpn_composite)
TRAIN_COMPOSITE = os.environ.get('NBM_RPN_COMPOSITE_TRAIN', '1') != '0'   # training: the same composition in the cell domain (train_composite_*)


def conv3x3_winograd_lazy(x, U, bias, stride, Ucell=None, fold=None, keep=False, raw=None):
    """Finest-level output convolution, pattern tiles only (see above) -> (y [B,H,W,N] with the other pixels unwritten,
    LazyMap).  `Ucell` (_prep.cell_weight(w, forward=True)): the pattern pixels through the cell transforms (F(3x3,3x3) per
    stride x stride cell: 25 plane products per cell instead of the ~49 of the listed F(2x2,3x3) tiles) -- exactly the 9 / 64
    pattern pixels are stored.  `fold` (callable (wk, alpha) -> [25][N][C + Cin], `_prep.cell_weight_folded`): needed when the
    lateral in front was deferred (`conv1x1_lazy(defer=True)`): its operands enter the plane GEMMs directly."""
    _chk(x, name='x'), _chk(U, name='U')
    B, H, W, C_ = x.shape
    N = U.shape[1]
    assert U.shape == (16, N, C_) and C_ % 32 == 0 and C_ >= 64 and N % 4 == 0
    y = _sparse_map((B, H, W, N), x.device)
    st = LazyMap(x, U, bias, stride)
    st.keep = bool(keep)                              # a backward pass will follow: RoI tile lists (and a deferred lateral's operand) are kept
    lat = _LAZY_LATERAL.pop(x.data_ptr(), None)      # x itself only exists where the pattern tiles read it
    if lat is not None and lat[1]() is not None:
        st.lateral = lat[0]
    cell_ok = Ucell is not None and stride >= 3 and C_ % 32 == 0 and N % 4 == 0
    if st.lateral is not None and st.lateral.deferred and not (cell_ok and fold is not None):
        _lateral_pattern_pass(x, st.lateral, stride)             # this consumer cannot take the operands: the patches after all
        st.lateral.deferred = False
    Ufold = fold(st.lateral.wk, st.lateral.alpha) if st.lateral is not None and st.lateral.deferred else None
    if Ufold is not None and st.keep:
        st.lateral.ufold = Ufold
        st.lateral.ufold_t = fold(st.lateral.wk, st.lateral.alpha, transposed=True)      # same cache entry, [25][C + Cin][N]
    img_bytes = H * W * N * 4
    chunk = lazy_chunk(x)
    st.raw = raw
    # evaluation mode: the one reader of the pattern pixels, the RPN's depthwise-separable block, takes this convolution INTO its own
    # (rpn_composite): the pattern pass waits until somebody asks (the block, or pattern_materialize for any other reader)
    defer_pattern = bool(COMPOSITE and cell_ok and raw is not None and not st.keep and not torch.is_grad_enabled())
    # training: the same reader composed with this convolution in the CELL domain (train_composite_forward): the pattern pass waits too
    # (`keep` was decided with grad mode as the caller sees it; inside Function.forward it is off)
    defer_train = bool(TRAIN_COMPOSITE and CELL_BWD and cell_ok and raw is not None and st.keep and N % 32 == 0)
    if defer_pattern or defer_train:
        defer_pattern = True
        st.pending = (Ucell, Ufold)
    for ci, b0 in enumerate(range(0, B, chunk)):
        nb = min(chunk, B - b0)
        pat = wino23_pattern(nb, H, W, stride, x.device)
        st.skip = pat.full
        st.chunks.append((b0, nb, pat))
        st.sparse = pat.frac < 0.6               # the weight gradient over the listed tiles pays off when most are not listed
        # stride 3 / 4: no tile is free of pattern pixels, but the gradient still lives on 9 / S^2 of the pixels (+ the RoI windows):
        # the cell transforms take that share (overlapping 5x5 patches, added class by class), the listed kernel the RoI share
        st.overlap = bool(not st.sparse and cell_ok and CELL_BWD and 3 <= stride < 5 and N % 32 == 0)
        if cell_ok and defer_pattern:
            st.skip = None
            continue
        if cell_ok:                                  # (3x3 blocks of different cells never overlap)
            stream = _stream()
            Vx, M, K, T = _cell_operand(st, b0, nb, H, W, C_, x, n_out=N, ci=ci)
            keep_label, ops._PROFILE_LABEL = ops._PROFILE_LABEL, ('cell-fwd', H, W)
            try:
                gemm_conv(Vx, Ufold if Ufold is not None else Ucell, M, B=1, H=T, W=1, Cin=K, N=N, groups=25, x_gs=T * K, w_gs=N * K,
                          y_gs=T * N)
            finally:
                ops._PROFILE_LABEL = keep_label
            check(lib().nbm_cell_output(_ptr(M), _ptr(bias), nb, H, W, N, stride, C.c_void_p(y.data_ptr() + b0 * img_bytes), stream),
                  'nbm_cell_output')
            # the RoI phase skips nothing: every pixel the RoI pooling reads is then a value of the dense F(2x2,3x3) convolution, bit
            # for bit; only the RPN's strided reader sees the cell values (1 tile in 16 is all pattern pixels: +6 % RoI-phase tiles)
            st.skip = None
            continue
        _wino23_tiles_run(x[b0:b0 + nb], U, bias, y.data_ptr() + b0 * img_bytes, pat.tiles, None, pat.n_eff, 'wino23', pat.blk_info,
                          dense_rows=pat.frac == 1.0)
    for k in [k for k, v in _LAZY.items() if v[1]() is None]:      # maps of earlier forwards that were never completed
        del _LAZY[k]
    _LAZY[y.data_ptr()] = (st, weakref.ref(y))      # valid while the map object itself (or a view of it) is alive
    weakref.finalize(y, _forget, _LAZY, y.data_ptr(), id(st))     # ... and gone with it (operands included)
    return y, st


def pattern_materialize(fm):
    """Run the pattern pass of a demand-driven map whose pass was left pending (evaluation mode, see conv3x3_winograd_lazy): the 9
    pattern pixels of every cell through the cell transforms, into `fm`.  No-op for every other tensor."""
    st = lazy_state(fm)
    if st is None or st.pending is None:
        return
    Ucell, Ufold = st.pending
    st.pending = None
    B, H, W, N = fm.shape
    C_ = st.x.shape[-1]
    img_bytes = H * W * N * 4
    stream = _stream()
    for ci, (b0, nb, pat) in enumerate(st.chunks):
        Vx, M, K, T = _cell_operand(st, b0, nb, H, W, C_, st.x, n_out=N, ci=ci)
        gemm_conv(Vx, Ufold if Ufold is not None else Ucell, M, B=1, H=T, W=1, Cin=K, N=N, groups=25, x_gs=T * K, w_gs=N * K, y_gs=T * N)
        check(lib().nbm_cell_output(_ptr(M), _ptr(st.bias), nb, H, W, N, st.stride, C.c_void_p(fm.data_ptr() + b0 * img_bytes), stream),
              'nbm_cell_output')


_BORDER_IDX = {}


def _border_classes(nb, H, W, S, device):
    """Cells whose depthwise taps do not all lie inside the map, grouped by (row mask, column mask)
    -> [(rmask, smask, taps: patch taps a * 5 + e where the class's weights differ from the interior's AND the patch can hold data,
         tap index tensor [nt, 1], cell index tensor [Tb], (pixel index [nt, Tb] of those taps in a [nb * H * W, C] map (clamped), mask
         [nt, Tb] of the patch pixels that lie in the padding, or None))]."""
    key = (nb, H, W, S, str(device))
    hit = _BORDER_IDX.get(key)
    if hit is None:
        OH, OW = (H - 1) // S + 1, (W - 1) // S + 1
        rm = [sum(1 << r for r in range(3) if 0 <= S * oy - 1 + r < H) for oy in range(OH)]
        sm = [sum(1 << s_ for s_ in range(3) if 0 <= S * ox - 1 + s_ < W) for ox in range(OW)]
        cls = {}
        for oy in range(OH):
            for ox in range(OW):
                if (rm[oy], sm[ox]) != (7, 7):
                    cls.setdefault((rm[oy], sm[ox]), []).append((oy, ox))
        hit = []
        for (r_, s_), cells in sorted(cls.items()):
            cells = sorted(cells)
            live_a = {a for oy, _ in cells for a in range(5) if 0 <= S * oy - 2 + a < H}
            live_e = {e for _, ox in cells for e in range(5) if 0 <= S * ox - 2 + e < W}
            dropped = [(r, c) for r in range(3) for c in range(3) if not ((r_ >> r) & 1 and (s_ >> c) & 1)]
            taps = [a * 5 + e for a in sorted(live_a) for e in sorted(live_e) if any(0 <= a - r <= 2 and 0 <= e - c <= 2 for r, c in dropped)]
            lin = np.array([oy * OW + ox for oy, ox in cells], dtype=np.int64)
            idx = (np.arange(nb, dtype=np.int64)[:, None] * (OH * OW) + lin[None, :]).reshape(-1)
            py = np.array([[S * oy - 2 + t // 5 for oy, _ in cells] for t in taps], dtype=np.int64)          # [nt, cells]
            px = np.array([[S * ox - 2 + t % 5 for _, ox in cells] for t in taps], dtype=np.int64)
            pix = None
            if taps:
                inside = (py >= 0) & (py < H) & (px >= 0) & (px < W)                                         # a patch pixel in the padding reads 0
                one = np.clip(py, 0, H - 1) * W + np.clip(px, 0, W - 1)                                      # [nt, cells]
                lin_px = (np.arange(nb, dtype=np.int64)[None, :, None] * (H * W) + one[:, None, :]).reshape(len(taps), -1)
                bad = None if bool(inside.all()) else torch.from_numpy(np.broadcast_to(~inside[:, None, :], (len(taps), nb, len(cells)))
                                                                       .reshape(len(taps), -1).copy()).to(device)
                pix = (torch.from_numpy(lin_px).to(device), bad)
            hit.append((r_, s_, tuple(taps), torch.tensor(taps, dtype=torch.int64, device=device)[:, None], torch.from_numpy(idx).to(device), pix))
        _BORDER_IDX[key] = hit
    return hit


DIRECT_GATHER = os.environ.get('NBM_RPN_DIRECT', '1') != '0'    # rpn_composite: operands that exist as dense maps are gathered by the GEMM itself


def rpn_composite(fm, block):
    """Evaluation mode: the output of `block` (layers.DepthwiseSepConv2d, the RPN's reader of the map: depthwise 3x3 / stride S ->
    1x1 -> BatchNorm -> SiLU) on the demand-driven map `fm` WITHOUT the map's pattern pixels: the block composed with the map's own
    3x3 convolution is one 5x5 / stride S / pad 2 convolution of the convolution's INPUT (`_prep.rpn_composite`) -- the same 25 K N
    products per cell as the cell transforms' plane GEMMs, but no transform arithmetic, no 25-plane intermediate, no output transform,
    no depthwise pass and no 1x1 behind it.  An operand that exists as a dense map (the merged map of a level whose pattern covers every
    pixel; the lateral's input t of a deferred lateral) is gathered by the implicit GEMM itself, in channel slices; the interpolated
    operand of a deferred lateral (up(x1) + b, position-dependent weights: not a convolution of x1) goes through raw 5x5 patches
    (nbm_cell_patches_up: [25][cells][C], 25 taps) and one launch per patch row.  The launches are chained through the residual input
    (y = scale * acc + [shift + R | y]; SiLU on the last): a single fmaf chain over all 25 K products rounds 25 K times against a running
    sum that has grown to the whole result -- measured 1.5 x the rms error of the route through the pattern pixels; 3-6 shorter chains
    are below it.  Border cells (a depthwise tap in the zero padding of the map) differ from the interior in a handful of taps: their
    difference goes in front (R), through the residual input of the first launch.
    -> [B, OH, OW, N], or None when `fm` is not a map with a pending pattern pass."""
    st = lazy_state(fm)
    if st is None or st.pending is None or st.raw is None:
        return None
    from .nets import _prep as prep                  # (nets imports this module)
    S = st.stride
    if int(max(1, block.stride)) != S or block.stride < 1 or getattr(block, 'pe_proj', None) is not None:
        return None
    B, H, W, _ = fm.shape
    C_ = st.x.shape[-1]
    lt = st.lateral if st.lateral is not None and st.lateral.deferred else None
    Cin = lt.t.shape[-1] if lt is not None else 0
    K = C_ + Cin
    OH, OW = (H - 1) // S + 1, (W - 1) // S + 1
    out_w, out_b = st.raw
    scale, shift = prep.bn_affine(block.norm.weight, block.norm.bias, block.norm.running_mean, block.norm.running_var, block.norm.eps,
                                  conv_bias=block.pt_wise.bias)
    N2 = block.pt_wise.weight.shape[0]
    wargs = (out_w, out_b, block.depth_wise.weight, block.depth_wise.bias, block.pt_wise.weight, scale, shift)
    wkw = dict(lat_wk=lt.wk[:, :Cin] if lt is not None else None, alpha=lt.alpha if lt is not None else 1.0)
    # operands: ('patch', c0, c1) = channels [c0, c1) of K from the patch tensor V; ('map', tensor, c0, c1, k0) = channels [c0, c1) of a
    # dense map that are channels [k0, ...) of K
    if lt is not None:
        direct_t = DIRECT_GATHER and Cin % 32 == 0
        operands = [('patch', 0, C_)] + ([('map', lt.t, 0, Cin, C_)] if direct_t else [])
        KV = C_ if direct_t else K                  # channels held by the patch tensor
    elif DIRECT_GATHER and C_ % 32 == 0:
        q32 = C_ // 32                               # channel slices (multiples of 32): 3 or 4 links where the channel count allows
        cc = C_ // (3 if q32 % 3 == 0 else 4 if q32 % 4 == 0 else 2 if q32 % 2 == 0 else 1)
        operands = [('map', st.x, c0, c0 + cc, c0) for c0 in range(0, C_, cc)]
        KV = 0
    else:
        operands, KV = [('patch', 0, C_)], C_
    parts = tuple((0, KV) if op[0] == 'patch' else (op[4], op[4] + op[3] - op[2]) for op in operands)
    ws, sc, sh = prep.rpn_composite(*wargs, **wkw, parts=parts)
    f = torch.empty((B, OH, OW, N2), device=fm.device, dtype=torch.float32)
    stream = _stream()
    keep_label, ops._PROFILE_LABEL = ops._PROFILE_LABEL, ('rpn-composite', H, W)
    try:
        for (b0, nb, pat) in st.chunks:
            T = nb * OH * OW
            V = None
            if KV:
                V = ops._wino_scratch(fm.device, 25 * T * KV, 0)[0]
                if lt is not None:
                    check(lib().nbm_cell_patches_up(_ptr(lt.up[b0:b0 + nb]), _ptr(lt.bias), nb, H, W, C_, lt.up.shape[1], lt.up.shape[2], S,
                                                    _ptr(V), KV, 0, stream), 'nbm_cell_patches_up')
                    if KV > C_:
                        check(lib().nbm_cell_patches(_ptr(lt.t[b0:b0 + nb]), nb, H, W, Cin, S, _ptr(V), KV, C_, stream), 'nbm_cell_patches')
                else:
                    check(lib().nbm_cell_patches(_ptr(st.x[b0:b0 + nb]), nb, H, W, C_, S, _ptr(V), KV, 0, stream), 'nbm_cell_patches')
            fc = f[b0:b0 + nb]
            # border classes first: scale * (W_class - W_interior) . patch + scale * (const_class - const_interior) of their cells, a few
            # taps each, scattered into an otherwise zero [T, N] tensor that the first launch below adds
            border = _border_classes(nb, H, W, S, fm.device)
            R = None
            if border:
                R = torch.zeros((T, N2), device=fm.device, dtype=torch.float32)
                Vv = V[:25 * T * KV].view(25, T, KV) if KV else None
                for rmask, smask, taps, tap_idx, idx, pix in border:
                    if not taps:
                        continue
                    dwt, dsh = prep.rpn_composite_delta(*wargs, rmask, smask, taps, **wkw)
                    Tb, nt = idx.numel(), len(taps)
                    cols = []                                   # the class's operand rows [taps, cells, K], gathered piece by piece
                    for op in operands:
                        if op[0] == 'patch':
                            cols.append(Vv[tap_idx, idx[None, :]])
                        else:
                            m_ = op[1][b0:b0 + nb]
                            g_ = m_.reshape(-1, m_.shape[-1]).index_select(0, pix[0].reshape(-1))[:, op[2]:op[3]].reshape(nt, Tb, -1)
                            if pix[1] is not None:
                                g_ = g_.masked_fill(pix[1][..., None], 0.0)
                            cols.append(g_)
                    Vb = cols[0] if len(cols) == 1 else torch.cat(cols, -1)
                    fb = torch.empty((Tb, N2), device=fm.device, dtype=torch.float32)
                    # few rows, a long K: 32-column slices as groups (128 x 32 tiles: a quarter of the serial MFMA work per K-step of
                    # the 128 x 128 tile; 150 -> ~35 us per class); the shift comes in through the residual input, which has a group stride
                    if N2 % 32 == 0:
                        gemm_conv(Vb, dwt, fb, B=1, H=nt, W=Tb, Cin=K, N=32, kh=nt, kw=1, Ho=1, Wo=Tb, x_ld=K, w_ld=nt * K, y_ld=N2,
                                  groups=N2 // 32, x_gs=0, w_gs=32 * nt * K, y_gs=32, residual=dsh.expand(Tb, N2).contiguous(), res_ld=N2,
                                  res_gs=32)
                    else:
                        gemm_conv(Vb, dwt, fb, B=1, H=nt, W=Tb, Cin=K, N=N2, kh=nt, kw=1, Ho=1, Wo=Tb, x_ld=K, w_ld=nt * K, shift=dsh)
                    R.index_copy_(0, idx, fb)
            # the chain of launches
            links = []
            for op, w in zip(operands, ws):
                if op[0] == 'patch':
                    # one launch per patch ROW (5 taps x KV); a tap's plane lies (tap index) x T x KV floats behind the first -- groups of
                    # planes whose offsets stay inside the 2 GB window of a buffer resource
                    ppl = max(1, min(5, ((1 << 31) - (1 << 24)) // (T * KV * 4)))
                    for p0 in range(0, 25, ppl):
                        npl = min(ppl, 25 - p0)
                        links.append(dict(x=V[p0 * T * KV:], w=w[:, p0 * KV:], B=1, H=npl, W=T, Cin=KV, kh=npl, kw=1, Ho=1, Wo=T, x_ld=KV,
                                          w_ld=25 * KV))
                else:
                    m_ = op[1][b0:b0 + nb]
                    cw = op[3] - op[2]
                    links.append(dict(x=m_[..., op[2]:], w=w, B=nb, H=H, W=W, Cin=cw, kh=5, kw=5, stride=S, pad=2, Ho=OH, Wo=OW,
                                      x_ld=m_.shape[-1], w_ld=25 * cw))
            for i, ln in enumerate(links):
                res = R if i == 0 else fc
                ops._PROFILE_LABEL = ('rpn-composite' if ln['kw'] == 1 else 'rpn-composite-map', H, W)
                gemm_conv(ln.pop('x'), ln.pop('w'), fc, N=N2, scale=sc, shift=sh if i == 0 else None, residual=res,
                          res_ld=N2 if res is not None else None, act=ops.ACT_SILU if i == len(links) - 1 else ops.ACT_NONE, **ln)
    finally:
        ops._PROFILE_LABEL = keep_label
    return f


# ---- training mode: the RPN's reader composed with the demand-driven convolution in the CELL domain
# Reference chain (fpn.py:137,145 out_conv 3x3 + bias -> layers.py:22-29 depthwise 3x3 / stride S / pad 1, channel multiplier `mult`, + bias ->
# 1x1 + bias -> BatchNorm): linear in front of the BatchNorm also with batch statistics.  The cell transforms (4c) give the 3x3 block of
# the map that the depthwise taps read as  Y_n' = E^T M_n' E + out_b[n'],  M_xi,n' = Vx_xi . U_xi[n']  (xi = the 25 planes), so
#     f[n2] = sum_m pt[n2][m] (<dw_m, Y_{m // mult}> + dw_b[m]) + pt_b[n2]
#           = sum_xi Vx_xi . (A_xi U_xi)[n2] + const[n2],      A_xi[n2][n'] = sum_{m // mult = n'} pt[n2][m] (E dw_m E^T)[xi]
# -- ONE GEMM over the transformed patches Vx the forward pass computes and keeps anyway ([25][cells][K] read as 25 taps), instead of 25
# plane GEMMs into a [25][cells][N] intermediate, the block transform (`cell_output`), the depthwise pass and the 1x1; the pattern
# pixels of the map are never formed.  Backward: dW'_xi = g_f^T Vx_xi (one grouped weight-gradient GEMM; no `cell_outgrad`, no depthwise /
# 1x1 backward kernels), dA_xi = dW'_xi U_xi^T -> d(pt), d(dw) (autograd over the tiny weight-side function `_compose_rpn_cell`),
# dU_xi = A_xi^T dW'_xi goes where the cell path's `dUc` goes (weight gradient of out_conv, lateral fold-back), M'_xi = g_f W'_xi goes
# where the plane data-gradient GEMMs' output goes (`cell_dgrad_output`, deferred-lateral variants included).  The RoI share of the
# map's gradient then includes the pattern pixels (skip_pattern = 0) and is ADDED to the cell share.
# Border cells (a depthwise tap in the zero padding of the map: top row / left column / corner at this geometry): the padding zeroes
# those taps, i.e. their class has its own dw (masked) and therefore its own A and W' -- the same GEMMs on the class's gathered rows.
_CE = ((1.0, 0.0, 0.0), (1.0, 1.0, 1.0), (1.0, -1.0, 1.0), (1.0, 2.0, 4.0), (1.0, -2.0, 4.0))      # csrc/cellwino.hip CE


_DEV_CONST = {}


def _dev_const(device, key, values):
    """Small constant on the device, uploaded ONCE per device: a per-call torch.tensor(..., device=...) is a pageable H2D copy, i.e. a
    stream synchronisation -- 36 of them per training step cost 36 ms (the host lost its run-ahead every time)."""
    k = (str(device), key)
    t = _DEV_CONST.get(k)
    if t is None:
        t = _DEV_CONST[k] = torch.tensor(values, dtype=torch.float32).to(device)
    return t


def _compose_rpn_cell(dw_w, dw_b, pt_w, pt_b, out_b, masks, mult):
    """Weight side of the composition, differentiable (plain torch elementwise / reduce ops on weight-sized tensors, no GEMM):
    -> (A [classes][25][N2][N1], const [classes][N2]) for the cell classes `masks` = ((row mask, column mask), ...)."""
    m_ = dw_w.shape[0]
    n1, n2 = m_ // mult, pt_w.shape[0]
    dev = dw_w.device
    E = _dev_const(dev, 'E', _CE)                                                    # [5][3]
    mr = _dev_const(dev, ('rmask', masks), [[float((rm >> r) & 1) for r in range(3)] for rm, _ in masks])      # [classes][3]
    ms = _dev_const(dev, ('smask', masks), [[float((sm >> c) & 1) for c in range(3)] for _, sm in masks])
    dw = dw_w.reshape(1, m_, 3, 3)
    pt = pt_w.reshape(n2, n1, mult)
    dwm = dw * mr[:, None, :, None] * ms[:, None, None, :]                           # [classes][M][3][3]: taps in the padding zeroed
    t = (E[None, None, :, :, None] * dwm[:, :, None, :, :]).sum(3)                   # [classes][M][5 (a)][3 (s)]
    cw = (t[:, :, :, None, :] * E[None, None, None, :, :]).sum(4)                    # (E dw E^T)[a][b], plane xi = 5 a + b
    cw = cw.reshape(len(masks), n1, mult, 25)
    A = (pt[None, None] * cw.permute(0, 3, 1, 2)[:, :, None]).sum(4)                 # [classes][25][N2][N1]
    cb = dwm.sum((2, 3))                                                              # [classes][M]
    if out_b is not None:
        cb = cb * out_b.repeat_interleave(mult)[None, :]
    else:
        cb = cb * 0.0
    if dw_b is not None:
        cb = cb + dw_b[None, :]
    const = (pt.reshape(1, n2, m_) * cb[:, None, :]).sum(2)                           # [classes][N2]
    if pt_b is not None:
        const = const + pt_b[None, :]
    return A, const


_CELL_CLASSES = {}


def _cell_classes(H, W, S):
    """Classes of cells by which of their depthwise taps lie inside the map: [(row mask, column mask, (oy0, oy1), (ox0, ox1))], every
    class a rectangle of the OH x OW cell grid (interior: all nine taps inside; at the reference geometries the top row, the left column
    and the corner have taps in the padding).  The interior class comes first.  None when a class is not a rectangle."""
    key = (H, W, S)
    if key not in _CELL_CLASSES:
        OH, OW = (H - 1) // S + 1, (W - 1) // S + 1
        rm = [sum(1 << r for r in range(3) if 0 <= S * oy - 1 + r < H) for oy in range(OH)]
        sm = [sum(1 << c for c in range(3) if 0 <= S * ox - 1 + c < W) for ox in range(OW)]

        def ranges(masks):
            out = {}
            for i, m in enumerate(masks):
                if m in out:
                    lo, hi = out[m]
                    if i != hi:
                        return None                  # the same mask in two separate runs: not a rectangle
                    out[m] = (lo, i + 1)
                else:
                    out[m] = (i, i + 1)
            return out
        rr, cc = ranges(rm), ranges(sm)
        cls = None
        if rr is not None and cc is not None and 7 in rr and 7 in cc:
            cls = [(7, 7, rr[7], cc[7])] + [(r_, c_, rr[r_], cc[c_]) for r_ in sorted(rr) for c_ in sorted(cc) if (r_, c_) != (7, 7)]
        _CELL_CLASSES[key] = cls
    return _CELL_CLASSES[key]


def _chain_planes(V, Wf, out, T, K, N2, shift):
    """out [T][N2] = sum over the 25 planes of V ([25][T][K] flat) x Wf [N2][25 K] (tap-major) + shift: launches of up to five planes
    each, chained through the residual input (see rpn_composite: shorter fmaf chains, and a plane group stays inside the 2 GB window of
    a buffer resource)."""
    ppl = max(1, min(5, ((1 << 31) - (1 << 24)) // (T * K * 4)))
    for p0 in range(0, 25, ppl):
        npl = min(ppl, 25 - p0)
        gemm_conv(V[p0 * T * K:], Wf[:, p0 * K:], out, B=1, H=npl, W=T, Cin=K, N=N2, kh=npl, kw=1, Ho=1, Wo=T, x_ld=K, w_ld=25 * K,
                  shift=shift if p0 == 0 else None, residual=out if p0 else None, res_ld=N2 if p0 else None)


def train_composite_ready(fm, block):
    """The LazyMap of `fm` if `block` (layers.DepthwiseSepConv2d in training mode) can take the map's pending pattern pass into itself."""
    st = lazy_state(fm)
    if (st is None or st.pending is None or st.raw is None or not st.keep or not TRAIN_COMPOSITE or not torch.is_grad_enabled() or
            not (st.sparse or st.overlap) or getattr(block, 'pe_proj', None) is not None or block.stride < 1 or
            int(max(1, block.stride)) != st.stride or block.depth_wise.weight.shape[0] % st.U.shape[1] or
            _cell_classes(fm.shape[1], fm.shape[2], st.stride) is None):
        return None
    return st


TRAIN_COMPOSITE_CALLS = [0]      # tests: how many blocks went through the composed form
_SIDE = {}


def _side_stream(device):
    key = str(device)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


def train_composite_forward(st, fm, dw_w, dw_b, pt_w, pt_b):
    """-> f [B, OH, OW, N2]: the block's output in front of its BatchNorm.  Leaves in st.comp what the backward pass needs."""
    from .nets import _prep as prep
    TRAIN_COMPOSITE_CALLS[0] += 1
    B, H, W, _ = fm.shape
    S = st.stride
    C_ = st.x.shape[-1]
    N1 = st.U.shape[1]
    N2 = pt_w.shape[0]
    mult = dw_w.shape[0] // N1
    OH, OW = (H - 1) // S + 1, (W - 1) // S + 1
    Ucell, Ufold = st.pending
    st.pending = None
    out_w, out_b = st.raw
    lt = st.lateral if st.lateral is not None and st.lateral.deferred else None
    Ufwd = Ufold if Ufold is not None else Ucell                                      # [25][N1][K]
    Ut = lt.ufold_t if lt is not None else prep.cell_weight(out_w)                     # [25][K][N1]
    K = Ufwd.shape[2]
    assert tuple(Ut.shape) == (25, K, N1) and tuple(Ufwd.shape) == (25, N1, K)
    classes = _cell_classes(H, W, S)
    masks = tuple((r_, s_) for r_, s_, _, _ in classes)
    with torch.no_grad():
        A, const = _compose_rpn_cell(dw_w.detach(), None if dw_b is None else dw_b.detach(), pt_w.detach(),
                                     None if pt_b is None else pt_b.detach(), None if out_b is None else out_b.detach(), masks, mult)
    A = A.contiguous()
    Wf, Wb = [], []
    for c in range(len(masks)):
        wf = torch.empty((N2, 25 * K), device=fm.device, dtype=torch.float32)         # W'[n2][xi K + k] = (A_xi U_xi)[n2][k]
        gemm_conv(A[c], Ut, wf, B=1, H=N2, W=1, Cin=N1, N=K, groups=25, x_gs=N2 * N1, w_gs=K * N1, y_gs=K, y_ld=25 * K)
        wb = torch.empty((25, K, N2), device=fm.device, dtype=torch.float32)          # the same values as [xi][k][n2]
        gemm_conv(Ut, A[c], wb, B=1, H=K, W=1, Cin=N1, N=N2, groups=25, x_gs=K * N1, w_gs=N2 * N1, y_gs=K * N2)
        Wf.append(wf), Wb.append(wb)
    f = torch.empty((B, OH, OW, N2), device=fm.device, dtype=torch.float32)
    kept_vb = {}
    keep_label, ops._PROFILE_LABEL = ops._PROFILE_LABEL, ('rpn-composite-train', H, W)
    try:
        for ci, (b0, nb, pat) in enumerate(st.chunks):
            V, _, Kc, T = _cell_operand(st, b0, nb, H, W, C_, st.x, n_out=N1, ci=ci)   # transformed patches; kept in st.vx for the backward pass
            assert Kc == K and T == nb * OH * OW
            fc = f[b0:b0 + nb].view(T, N2)
            # border classes on a side stream, beside the interior launches (few rows: they would leave most of the chip idle): the
            # class's rows -- a rectangle of the cell grid -- copied out, ONE grouped GEMM over the 25 planes with the class's weights
            # (a single launch over K = 25 x 448 is a serial loop of 350 K-steps on a quarter of the CUs: 0.74 ms per class and chunk),
            # the 25 partial sums added; the rows then replace what the interior weights wrote there
            main = torch.cuda.current_stream()
            side = _side_stream(fm.device)
            side.wait_stream(main)
            border = []
            fflat = f[b0:b0 + nb].view(-1)
            with torch.cuda.stream(side):
                for c, (_, _, (r0, r1), (c0, c1)) in enumerate(classes):
                    if c == 0:
                        continue
                    nr, nc = r1 - r0, c1 - c0
                    Tb = nb * nr * nc
                    Vb = torch.empty((25 * Tb * K,), device=fm.device, dtype=torch.float32)
                    ops.copy_rect(V, (r0 * OW + c0) * K, Vb, 25 * nb, OH * OW * K, nr, OW * K, nc * K)
                    part = torch.empty((25, Tb, N2), device=fm.device, dtype=torch.float32)
                    gemm_conv(Vb, Wf[c], part, B=1, H=Tb, W=1, Cin=K, N=N2, groups=25, x_gs=Tb * K, w_ld=25 * K, w_gs=K, y_gs=Tb * N2)
                    fb = part.sum(0)
                    fb += const[c]
                    fb.record_stream(main)
                    Vb.record_stream(main)
                    kept_vb[(ci, c)] = Vb             # the weight gradient of the class reads the same rows again
                    border.append(((r0, r1), (c0, c1), fb))
            # interior weights on every cell: five groups of five planes (K = 5 x 448 each) into partial sums, added afterwards -- one
            # launch of 5 x (cells / 128) x 2 workgroups fills whole rounds of the chip where a chain of five 2-tile-wide launches
            # ended each in a partly filled third round (+4.8 ms per step, measured), and five short fmaf chains round less
            part5 = torch.empty((5, T, N2), device=fm.device, dtype=torch.float32)
            gemm_conv(V, Wf[0], part5, B=1, H=5, W=T, Cin=K, N=N2, kh=5, kw=1, Ho=1, Wo=T, x_ld=K, w_ld=25 * K, groups=5, x_gs=5 * T * K,
                      w_gs=5 * K, y_gs=T * N2)
            torch.sum(part5, 0, out=fc)
            fc += const[0]
            main.wait_stream(side)
            for (r0, r1), (c0, c1), fb in border:
                ops.copy_rect(fflat, (r0 * OW + c0) * N2, fb, nb, OH * OW * N2, r1 - r0, OW * N2, (c1 - c0) * N2, to_strided=True)
    finally:
        ops._PROFILE_LABEL = keep_label
    st.comp = dict(A=A, Wf=Wf, Wb=Wb, Ufwd=Ufwd, Ut=Ut, K=K, N2=N2, mult=mult, masks=masks, classes=classes, vb=kept_vb, g=None, dUc=None, gb=None, gf=None,
                   dw_w=dw_w.detach(), pt_w=pt_w.detach())
    return f


def train_composite_backward(st, gf, dw_w, dw_b, pt_w, pt_b):
    """gf [B, OH, OW, N2] = d/df.  -> (d dw_w, d dw_b, d pt_w, d pt_b); leaves in st.comp: `g` (per chunk: d/df with the border rows
    zeroed + the border classes' rows) for the data gradient, `dUc` [25][N1][K] (gradient wrt the transformed kernel) and `gb` (the
    share of out_conv's bias gradient) for the convolution's own backward pass."""
    cp = st.comp
    if cp is None or st.vx is None:
        raise RuntimeError('the composed RPN reader has no forward state left (backward called twice, or the demand-driven map was released)')
    B, OH, OW, N2 = gf.shape
    K, A = cp['K'], cp['A']
    N1 = A.shape[3]
    ncls = A.shape[0]
    dev = gf.device
    dWf = [torch.zeros((N2, 25 * K), device=dev, dtype=torch.float32) for _ in range(ncls)]
    G = torch.zeros((ncls, N2), device=dev, dtype=torch.float32)
    per_chunk = []
    for ci, (b0, nb, pat) in enumerate(st.chunks):
        T = nb * OH * OW
        V = st.vx[ci]
        g_int = gf[b0:b0 + nb].reshape(T, N2).clone()
        gflat = g_int.view(-1)
        parts = []
        for c, (_, _, (r0, r1), (c0, c1)) in enumerate(cp['classes']):
            if c == 0:
                continue
            nr, nc = r1 - r0, c1 - c0
            Tb = nb * nr * nc
            g_c = torch.empty((Tb, N2), device=dev, dtype=torch.float32)
            ops.copy_rect(gflat, (r0 * OW + c0) * N2, g_c, nb, OH * OW * N2, nr, OW * N2, nc * N2, zero=True)     # ... and zeroed in g_int
            Vb = cp['vb'].pop((ci, c))
            conv_wgrad(g_c, Vb, dWf[c], B=1, H=Tb, W=1, Cin=K, N=N2, groups=25, g_gs=0, x_gs=Tb * K, out_gs=K, out_ld=25 * K)
            G[c] += ops.colsum(g_c, N2)
            parts.append((c, (r0, r1), (c0, c1), g_c))
        conv_wgrad(g_int, V, dWf[0], B=1, H=T, W=1, Cin=K, N=N2, groups=25, g_gs=0, x_gs=T * K, out_gs=K, out_ld=25 * K)
        G[0] += ops.colsum(g_int, N2)
        per_chunk.append((g_int, parts, (nb, OH, OW)))
    # d/dA_xi = dW'_xi U_xi^T; d/dU_xi = sum over the classes of A_xi^T dW'_xi
    dA = torch.empty((ncls, 25, N2, N1), device=dev, dtype=torch.float32)
    dUc = torch.zeros((25, N1, K), device=dev, dtype=torch.float32)
    for c in range(ncls):
        gemm_conv(dWf[c], cp['Ufwd'], dA[c], B=1, H=N2, W=1, Cin=K, N=N1, groups=25, x_ld=25 * K, x_gs=K, w_gs=N1 * K, y_gs=N2 * N1)
        conv_wgrad(A[c], dWf[c], dUc, B=1, H=N2, W=1, Cin=K, N=N1, groups=25, g_gs=N2 * N1, x_ld=25 * K, x_gs=K, out_gs=N1 * K)
    # the weight side by autograd over the small differentiable function (elementwise / reduce kernels on weight-sized tensors)
    out_b = st.raw[1]
    leaves = [t.detach().requires_grad_(True) if t is not None else None for t in (dw_w, dw_b, pt_w, pt_b, out_b)]
    with torch.enable_grad():
        A2, c2 = _compose_rpn_cell(*leaves, cp['masks'], cp['mult'])
        wanted = [t for t in leaves if t is not None]
        grads = torch.autograd.grad([A2, c2], wanted, [dA, G])
    it = iter(grads)
    g_dw, g_dwb, g_pt, g_ptb, g_ob = [next(it) if t is not None else None for t in leaves]
    cp['g'], cp['dUc'], cp['gb'], cp['gf'] = per_chunk, dUc, g_ob, gf
    st.vx = None                                      # the transformed patches have served (forward GEMM, dW')
    return g_dw, g_dwb, g_pt, g_ptb


def train_composite_fallback(st, gy):
    """The convolution's backward node cannot take the composed reader's shares in the cell domain (two RoI poolings on an overlap
    level: the dense kernels run): the reader's share of d/d(map) is formed in the pixel domain after all -- d/d(depthwise output) =
    d/df pt (1x1 data gradient), scattered through the depthwise taps onto the pattern pixels of `gy` -- and the cell-domain shares
    are dropped; the backward pass then proceeds exactly as without the composition.  -> gy (in place)."""
    cp = st.comp
    st.comp = None
    if cp is None or cp['gf'] is None:
        raise RuntimeError('the composed RPN reader of this demand-driven map has not been through its backward pass')
    gf, dw_w, pt_w = cp['gf'], cp['dw_w'], cp['pt_w']
    B, OH, OW, N2 = gf.shape
    M_ = dw_w.shape[0]
    gd = torch.empty((B * OH * OW, M_), device=gf.device, dtype=torch.float32)
    ops.conv_dgrad(gf.reshape(-1, N2), pt_w.reshape(N2, M_), gd, B=1, H=B * OH * OW, W=1, Cin=M_, N=N2, g_ld=N2, w_ld=M_)
    gy = gy.contiguous()
    ops.dwconv3x3_bwd_acc(gd.view(B, OH, OW, M_), dw_w, cp['mult'], st.stride, gy)
    for e in _ZERO_POOL.values():                # a persistent gradient map: the pattern pixels are part of its footprint now
        if e['busy'] and e['buf'].data_ptr() == gy.data_ptr():
            zero_note(e, lambda b_=e['buf'], s_=st.stride: ops.zero_pattern(b_, s_))
    return gy


def lazy_state(fm):
    """The LazyMap of a demand-driven map (or None)."""
    hit = _LAZY.get(fm.data_ptr())
    return hit[0] if hit is not None and hit[1]() is not None else None


def lazy_pending(fm):
    hit = _LAZY.get(fm.data_ptr())
    return hit is not None and hit[1]() is not None


def lazy_complete(fm, rois, n_roi, fmap_hw, level=0):
    """Compute the tiles of the deferred map `fm` under the windows of the RoIs assigned to `level` (the windows
    `roi_pool` reads).  rois [B,cap,4], n_roi device int32[1] (or [B]: per-image counts), fmap_hw: (h, w) of every pyramid
    level.  No-op for a map that is not deferred.  May be called any number of times on the same map (each RoI pooling calls it
    with its own RoIs): the operands live as long as the map."""
    hit = _LAZY.get(fm.data_ptr())
    if hit is None or hit[1]() is None:
        return
    st = hit[0]
    x, U, bias = st.x, st.U, st.bias
    if x is None:
        raise RuntimeError('this demand-driven FPN map has been through its backward pass: its operands are gone and the tiles '
                           'under new RoIs cannot be computed any more (pool before calling backward, or ask '
                           'forward_first_stage for dense maps: lazy=False)')
    B, H, W, C_ = x.shape
    if tuple(fm.shape[0:1]) != (B,) or rois.shape[0] != B:
        raise ValueError('lazy_complete: the RoIs do not belong to this map (batch size differs)')
    img_bytes = H * W * U.shape[1] * 4
    cap = rois.shape[1]
    _chk(rois, name='rois')
    per = ops.per_image_counts(n_roi, B)
    nl = len(fmap_hw)
    fh = (C.c_int * nl)(*[int(h) for h, _ in fmap_hw])
    fw = (C.c_int * nl)(*[int(w) for _, w in fmap_hw])
    blocks_per_img = -(-((H + 1) // 2 * ((W + 1) // 2)) // 128)
    keep = st.keep and (st.sparse or st.overlap)  # a backward pass will want the lists
    per_chunk = []
    for b0, nb, _ in st.chunks:
        key = (str(x.device), nb * blocks_per_img * 128, ops.LANE)
        nr = n_roi[b0:b0 + nb] if per else n_roi
        if keep:                                  # the backward pass reads the list again: a buffer of its own
            tiles = torch.empty((key[1],), device=x.device, dtype=torch.int32)
            n_blocks = torch.zeros((1,), device=x.device, dtype=torch.int32)
        else:
            buf = _ROI_TILE_BUF.get(key)
            if buf is None:
                buf = _ROI_TILE_BUF[key] = (torch.empty((key[1],), device=x.device, dtype=torch.int32),
                                            torch.zeros((1,), device=x.device, dtype=torch.int32))
            tiles, n_blocks = buf
        check(lib().nbm_roi_tiles(_ptr(rois[b0:b0 + nb]), _ptr(nr), nb, cap, nl, level, fh, fw, _ptr(st.skip), 0, _ptr(tiles),
                                  _ptr(n_blocks), per, _stream()), 'nbm_roi_tiles')
        if st.lateral is not None:                # the input patches of these tiles first (16 pixels per listed tile)
            lt = st.lateral
            gemm_conv(lt.t[b0:b0 + nb], lt.wk, x[b0:b0 + nb], B=nb, H=H, W=W, Cin=lt.t.shape[-1], N=C_, w_ld=lt.wk.shape[1],
                      shift=lt.bias, alpha=lt.alpha, up=lt.up[b0:b0 + nb] if lt.up is not None else None, rows=tiles, rows_mode=2,
                      rows_count=tiles.numel() * 16, rows_blocks=n_blocks, rows_thw=((H + 1) // 2, (W + 1) // 2))
        _wino23_tiles_run(x[b0:b0 + nb], U, bias, fm.data_ptr() + b0 * img_bytes, tiles, n_blocks, None, 'wino23-rois')
        if keep:
            host = _pinned_int()
            host.copy_(n_blocks, non_blocking=True)
            tiles_d = host_d = None
            if st.lateral is not None and st.lateral.ufold is not None and LAT_CELL_BWD and st.stride >= 5:
                # the data gradient's list (tiles within a pixel of the windows) now, so that its length is known on the host when
                # the backward pass sizes the compact operands of the lateral's RoI share
                tiles_d = torch.empty((key[1],), device=x.device, dtype=torch.int32)
                nbd = torch.zeros((1,), device=x.device, dtype=torch.int32)
                check(lib().nbm_roi_tiles(_ptr(rois[b0:b0 + nb]), _ptr(nr), nb, cap, nl, level, fh, fw, None, 1, _ptr(tiles_d), _ptr(nbd),
                                          per, _stream()), 'nbm_roi_tiles')
                host_d = _pinned_int()
                host_d.copy_(nbd, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            per_chunk.append((tiles, host, ev, tiles_d, host_d))
    if keep:                                      # the data gradient lists the tiles around these windows again
        st.roi.append(per_chunk)
        st.rois.append((rois, n_roi, nl, level, fh, fw))
    st.done += 1


CELL_BWD = os.environ.get('NBM_CELL_BWD', '1') != '0'     # pattern share of both gradients through the cell transforms (cellwino.hip)
# The deferred lateral's OWN gradients from the consumer's backward pass: its pattern share rides in the cell-domain GEMMs (the
# folded weights [U | alpha U W] give d/dt in the transform domain, the weight-gradient GEMMs already hold d/d(alpha U W)), its RoI
# share runs on compact [tiles x 4 pixels] operands -- instead of three dense passes over the 18.9 GB gradient of the merged map
LAT_CELL_BWD = os.environ.get('NBM_LAT_CELL_BWD', '1') != '0'
# ... and then the only reader left of d/d(merged map) is the bilinear backward of the top-down merge: the RoI share stays in its
# compact form (nbm_tiles_upsample_bilinear_bwd_add) and the map holds the pattern patches only, which lets that reader skip the rows
# and columns between the patches (nbm_upsample_bilinear_bwd(pattern_stride): 61 % of 18.9 GB at B = 128)
UPBWD_SPLIT = os.environ.get('NBM_UPBWD_SPLIT', '1') != '0'


# ---- persistent gradient maps of a demand-driven level (training)
# The two big gradient maps of the finest level -- d/d(output map) [B,188,512,256], filled by the RoI pooling's scatter and the
# RPN's strided taps, and d/d(merged map) [B,188,512,384], written on the cell patches and the RoI tiles -- are zero almost
# everywhere, yet a fresh `torch.zeros` per step fills 31.5 GB (4.8 ms at B = 128).  They are kept across steps instead: every
# writer notes how to undo its footprint (`zero_note`), the LAST reader (Conv.backward of the node that consumes the map) calls
# `zero_recycle`, which runs those targeted kernels (nbm_zero_*) and marks the buffer clean.  A buffer that was not recycled (other
# consumer, exception, autograd accumulated something into it in place: its version counter moved) is filled completely at its next
# `zero_acquire`.  Because the buffer is handed to autograd as a gradient, code that KEEPS that gradient beyond the backward pass
# (`retain_grad()` on the FPN map) sees it change in the next step: set NBM_ZERO_POOL=0 for that.
ZERO_POOL = os.environ.get('NBM_ZERO_POOL', '1') != '0'
ZERO_POOL_CHECK = os.environ.get('NBM_ZERO_POOL_CHECK', '0') == '1'      # tests: verify the buffer after every recycle (synchronises)
_ZERO_POOL = {}


def zero_pool_new_pass():
    """Start of a forward pass: a buffer still marked busy was never handed back by a reader -- its content is unknown."""
    for e in _ZERO_POOL.values():
        if e['busy']:
            e['busy'], e['stale'] = False, True


def zero_acquire(shape, device, tag):
    """-> (zero-filled [shape] buffer, entry) from the pool (see above), or (None, None) when the buffer is in flight in THIS
    backward pass (a second RoI pooling on the same map: autograd will sum two maps, they cannot be one buffer); `tag` separates
    users whose footprints differ."""
    key = (str(device), tuple(shape), tag)
    e = _ZERO_POOL.get(key)
    if e is not None and e['busy']:
        return None, None
    if e is None:
        for k in [k for k in _ZERO_POOL if k[0] == key[0] and k[2] == tag]:       # another batch size: the old buffer goes
            del _ZERO_POOL[k]
        e = _ZERO_POOL[key] = dict(buf=torch.zeros(shape, device=device, dtype=torch.float32), busy=False, stale=False, zeroers=[],
                                   check=None)
    elif e['stale']:                     # never recycled: anything may have been written anywhere
        e['buf'].zero_()
    e['busy'], e['stale'], e['zeroers'], e['version'] = True, False, [], e['buf']._version
    return e['buf'], e


def zero_note(e, fn):
    e['zeroers'].append(fn)


def zero_recycle(t):
    """Called by the last reader of a gradient map: if `t` is a pool buffer that autograd has not touched, undo the writers'
    footprints.  -> True if it was."""
    if not _ZERO_POOL or t is None:
        return False
    for e in _ZERO_POOL.values():
        if e['busy'] and e['buf'].data_ptr() == t.data_ptr() and e['buf'].shape == t.shape:
            break
    else:
        return False
    if t._version != e['version']:       # accumulated into in place by autograd: unknown footprint, fill next time
        return False
    for fn in e['zeroers']:
        fn()
    if ZERO_POOL_CHECK and e['check'] is not None:
        e['check'](e['buf'])
    e['busy'], e['zeroers'] = False, []
    return True


def zero_pool_clear():
    _ZERO_POOL.clear()


def cell_count(B, H, W, stride):
    return B * ((H + 2 - 3) // stride + 1) * ((W + 2 - 3) // stride + 1)


def _cell_outgrad(st, g, ci, b0, nb):
    """Vg [25][cells][N] of batch chunk `ci` (E blk E^T of the pattern blocks of g): the A operand of the data-gradient GEMMs and
    the G operand of the weight-gradient GEMMs.  Computed once per backward pass (the sum of the block pixels = the pattern share
    of the bias gradient rides along into st.cell_gb) and kept on the LazyMap until the weight gradient has used it."""
    if st.vg is None:
        st.vg = {}
    hit = st.vg.get(ci)
    if hit is not None:
        return hit
    B, H, W, N = g.shape
    T = cell_count(nb, H, W, st.stride)
    if st.cell_gb is None:
        st.cell_gb, st.cell_gb_done = torch.zeros((N,), device=g.device, dtype=torch.float32), set()
    vg = st.vg[ci] = torch.empty((25, T, N), device=g.device, dtype=torch.float32)
    check(lib().nbm_cell_outgrad(_ptr(g[b0:b0 + nb]), nb, H, W, N, st.stride, _ptr(vg), None if ci in st.cell_gb_done else _ptr(st.cell_gb),
                                 _stream()), 'nbm_cell_outgrad')
    st.cell_gb_done.add(ci)
    return vg


def cell_usable(st, H, W, C_, N):
    return CELL_BWD and (st.stride >= 5 or st.overlap) and C_ % 32 == 0 and N % 32 == 0 and H >= 3 and W >= 3


def listed_backward(st):
    """The backward pass of this demand-driven convolution can run over its pattern / RoI footprint (Fn.Conv.backward): always for
    a sparse level; for an `overlap` level while at most one RoI pooling read the map (its RoI share is ADDED to the cell share, so a
    tile must not be listed twice -- two poolings fall back to the dense kernels)."""
    return st.sparse or (st.overlap and len(st.rois) <= 1)


def conv3x3_winograd_dgrad_tiles(st, g, Ut, Ucell=None, base=None, lateral_grads=False):
    """Data gradient of a demand-driven convolution (LazyMap `st`): g [B,H,W,N] is zero except on the pattern pixels and inside
    the RoI windows, so the gradient wrt the input is zero except within two pixels of the pattern blocks and one pixel of the
    windows.  Pattern share (`Ucell` = _prep.cell_weight: [25][C][N]): per stride x stride cell the full convolution of the 3x3
    gradient block with the kernel, Toom-Cook in 25 multiplications (cellwino.hip: block transform, 25 grouped GEMMs, patch
    transform into a zero-filled map).  RoI share: the same convolution operator (Ut = weights rotated / channel-swapped,
    F(2x2,3x3)) through the listed fused kernel on the tiles within a pixel of the RoI windows, which reads ALL of g there and
    so overwrites those tiles with their complete values.  Without `Ucell` (or NBM_CELL_BWD=0) the pattern share also goes
    through the listed fused kernel: the static list of the tiles around the pattern (56 % of the tiles, 16 / 12 / 9 planes).
    `st.overlap` (stride 3 / 4: the 5x5 patches of neighbouring cells overlap): the patches are ADDED, one parity class of cells per
    launch, into `base` (the gradient another consumer of the input left, taken over in place) or zeros, and the RoI share -- g
    with its pattern pixels read as zeros -- is added on top by the listed kernel."""
    _chk(g, name='g'), _chk(Ut, name='Ut')
    B, H, W, N = g.shape
    C_ = Ut.shape[1]
    assert Ut.shape == (16, C_, N)
    img_bytes = H * W * C_ * 4
    blocks_per_img = -(-((H + 1) // 2 * ((W + 1) // 2)) // 128)
    cell = Ucell is not None and cell_usable(st, H, W, C_, N)
    overlap = cell and st.stride < 5
    if base is not None and not overlap:
        raise ValueError('conv3x3_winograd_dgrad_tiles: `base` is only taken over by the accumulating (overlap) form')
    pool = None
    if overlap:
        assert len(st.rois) <= 1
        gx = base if base is not None else torch.zeros((B, H, W, C_), device=g.device, dtype=torch.float32)
    elif cell and ZERO_POOL:
        # persistent map (see zero_acquire): the cell patches are rewritten by every pass, only the RoI tiles have to be undone
        gx, pool = zero_acquire((B, H, W, C_), g.device, ('cell-dgrad', st.stride))
        if pool is not None and ZERO_POOL_CHECK:
            pool['check'] = lambda buf, s_=st.stride: _check_zero_outside_patches(buf, s_)
    if pool is None and not overlap:
        gx = torch.zeros((B, H, W, C_), device=g.device, dtype=torch.float32)
    if cell:
        assert Ucell.shape == (25, C_, N)
    # `lateral_grads`: the deferred lateral's data / bias gradient and the RoI share of its weight gradient are produced here too
    # (LAT_CELL_BWD above) and left in st.lateral.grads for the lateral's own backward node
    lt = st.lateral
    do_lat = bool(lateral_grads and LAT_CELL_BWD and cell and not overlap and lt is not None and lt.ufold is not None and
                  all(pc[3] is not None for per_chunk in st.roi for pc in per_chunk) and len(st.roi) == len(st.rois) and
                  len(st.rois) <= 1)          # RoI shares are ADDED: a tile listed by two poolings would count twice
    if do_lat:
        Cin = lt.t.shape[-1]
        K = C_ + Cin
        Ud = lt.ufold_t                                                   # [25][C + Cin][N]: rows C.. = alpha W^T U, d/dt in the transform domain
        assert Ud is not None and tuple(Ud.shape) == (25, K, N)
        dt = dt_pool = None
        if ZERO_POOL:
            dt, dt_pool = zero_acquire((B, H, W, Cin), g.device, ('lat-dt', st.stride))
        if dt is None:
            dt = torch.zeros((B, H, W, Cin), device=g.device, dtype=torch.float32)
        gb_lat = torch.zeros((C_,), device=g.device, dtype=torch.float32)
        gw_roi = torch.zeros((C_, Cin), device=g.device, dtype=torch.float32)
        dt_img_bytes = H * W * Cin * 4
        split = bool(UPBWD_SPLIT and st.stride >= 5)
        shares = []                  # split: (compact RoI share of d/d(merged map), its tile list, first image, images) per chunk
    comp = st.comp if cell else None
    if st.comp is not None and (not cell or st.comp['g'] is None):
        raise RuntimeError('the RPN reader of this demand-driven map was composed with its convolution (train_composite_forward) but '
                           'its backward pass has not run / the cell-domain data gradient is switched off: the pattern share of the '
                           'gradient would be dropped (NBM_RPN_COMPOSITE_TRAIN=0 restores the uncomposed chain)')
    for ci, (b0, nb, _) in enumerate(st.chunks):
        if cell:
            mp = C_                                    # row pitch of the plane products M [25][T][mp]
            global_label = ops._PROFILE_LABEL
            ops._PROFILE_LABEL = ('cell-dgrad', H, W)
            try:
                if comp is not None:
                    # composed reader: M'_xi = g_f W'_xi straight from d/df [T][N2] (shared by the 25 groups); the border classes' rows
                    # with their own weights, written over the (zero) rows the interior launch left for them
                    g_int, parts, (nb_, OHc, OWc) = comp['g'][ci]
                    T, N2, mp = g_int.shape[0], comp['N2'], comp['K']
                    assert mp == (K if do_lat else mp) and mp >= C_
                    M, _ = ops._wino_scratch(g.device, 25 * T * mp, 0)
                    gemm_conv(g_int, comp['Wb'][0], M, B=1, H=T, W=1, Cin=N2, N=mp, groups=25, x_gs=0, w_gs=mp * N2, y_gs=T * mp)
                    for c, (r0, r1), (c0, c1), g_c in parts:
                        Tb = g_c.shape[0]
                        Mc = torch.empty((25 * Tb * mp,), device=g.device, dtype=torch.float32)
                        gemm_conv(g_c, comp['Wb'][c], Mc, B=1, H=Tb, W=1, Cin=N2, N=mp, groups=25, x_gs=0, w_gs=mp * N2, y_gs=Tb * mp)
                        ops.copy_rect(M, (r0 * OWc + c0) * mp, Mc, 25 * nb_, OHc * OWc * mp, r1 - r0, OWc * mp, (c1 - c0) * mp, to_strided=True)
                else:
                    vg = _cell_outgrad(st, g, ci, b0, nb)
                    T = vg.shape[1]
                    if do_lat:
                        mp = K
                        M, _ = ops._wino_scratch(g.device, 25 * T * K, 0)
                        gemm_conv(vg, Ud, M, B=1, H=T, W=1, Cin=N, N=K, groups=25, x_gs=T * N, w_gs=K * N, y_gs=T * K)
                    else:
                        M, _ = ops._wino_scratch(g.device, 25 * T * C_, 0)
                        gemm_conv(vg, Ucell, M, B=1, H=T, W=1, Cin=N, N=C_, groups=25, x_gs=T * N, w_gs=C_ * N, y_gs=T * C_)
            finally:
                ops._PROFILE_LABEL = global_label
            if do_lat:            # d/d(merged map) patches (+ their sum = pattern share of the lateral's bias gradient), d/dt patches
                check(lib().nbm_cell_dgrad_output(_ptr(M), nb, H, W, C_, st.stride, C.c_void_p(gx.data_ptr() + b0 * img_bytes), -1, K, 0,
                                                  _ptr(gb_lat), _stream()), 'nbm_cell_dgrad_output')
                check(lib().nbm_cell_dgrad_output(_ptr(M), nb, H, W, Cin, st.stride, C.c_void_p(dt.data_ptr() + b0 * dt_img_bytes), -1, K, C_,
                                                  None, _stream()), 'nbm_cell_dgrad_output')
            else:
                for cls in (range(4) if overlap else (-1,)):
                    check(lib().nbm_cell_dgrad_output(_ptr(M), nb, H, W, C_, st.stride, C.c_void_p(gx.data_ptr() + b0 * img_bytes), cls, mp, 0,
                                                      None, _stream()), 'nbm_cell_dgrad_output')
        else:
            pat = wino23_pattern(nb, H, W, st.stride, g.device, dilate=1)
            _wino23_tiles_run(g[b0:b0 + nb], Ut, None, gx.data_ptr() + b0 * img_bytes, pat.tiles, None, pat.n_eff, 'wino23-dgrad',
                              pat.blk_info)
        for si, (rois, n_roi, nl, level, fh, fw) in enumerate(st.rois):          # one entry per RoI pooling that read the map
            if do_lat:
                # RoI share, kept apart from the pattern share: g with its pattern pixels read as zeros through the listed kernel into a
                # COMPACT [tiles][2][2][C] operand; added to the map (the bilinear backward reads it there), and the lateral's three
                # gradients of these pixels from compact operands (a few % of the map)
                _, _, ev, tl, host_d = st.roi[si][ci]
                ev.synchronize()                          # recorded during the forward pass: long done
                n = int(host_d.item()) * 128
                if n:
                    tl = tl[:n]
                    gx_p, dt_p = C.c_void_p(gx.data_ptr() + b0 * img_bytes), C.c_void_p(dt.data_ptr() + b0 * dt_img_bytes)
                    Gc = torch.zeros((n * 4, C_), device=g.device, dtype=torch.float32)
                    # (composed reader: the pattern pixels of g hold the RoI pooling's share only -- nothing is masked)
                    _wino23_tiles_run(g[b0:b0 + nb], Ut, None, Gc.data_ptr(), tl, None, n, 'wino23-dgrad-rois',
                                      skip_pattern=0 if comp is not None else st.stride, compact=True)
                    if split:
                        shares.append((Gc, tl, b0, nb))
                    else:
                        check(lib().nbm_tiles_scatter_add(gx_p, nb, H, W, C_, _ptr(tl), n, None, _ptr(Gc), _stream()), 'nbm_tiles_scatter_add')
                    tc = torch.zeros((n * 4, Cin), device=g.device, dtype=torch.float32)
                    check(lib().nbm_tiles_gather(_ptr(lt.t[b0:b0 + nb]), nb, H, W, Cin, _ptr(tl), n, None, _ptr(tc), _stream()), 'nbm_tiles_gather')
                    # the lateral's three gradients on the compact rows, through the library's own NN / TN GEMMs (was: torch.matmul /
                    # addmm_ / sum = vendor GEMMs on the training path): d/dt = alpha Gc W_lat, dW_lat += alpha Gc^T t, db += colsum(Gc)
                    dtc = torch.empty((n * 4, Cin), device=g.device, dtype=torch.float32)
                    ops.conv_dgrad(Gc, lt.wk, dtc, B=1, H=n * 4, W=1, Cin=Cin, N=C_, g_ld=C_, w_ld=lt.wk.shape[1], alpha=float(lt.alpha))
                    check(lib().nbm_tiles_scatter_add(dt_p, nb, H, W, Cin, _ptr(tl), n, None, _ptr(dtc), _stream()), 'nbm_tiles_scatter_add')
                    conv_wgrad(Gc, tc, gw_roi, B=1, H=n * 4, W=1, Cin=Cin, N=C_, alpha=float(lt.alpha), bias_grad=gb_lat)
                    for pl, (p_, c_) in ((None if split else pool, (gx_p, C_)), (dt_pool, (dt_p, Cin))):
                        if pl is not None:
                            zero_note(pl, lambda p_=p_, c_=c_, nb_=nb, tl_=tl, n_=n: check(
                                lib().nbm_zero_tiles(p_, nb_, H, W, c_, _ptr(tl_), n_, None, _stream()), 'nbm_zero_tiles'))
                continue
            per = ops.per_image_counts(n_roi, B)
            key = (str(g.device), nb * blocks_per_img * 128, ops.LANE)
            buf = _ROI_TILE_BUF.get(key)
            if buf is None:
                buf = _ROI_TILE_BUF[key] = (torch.empty((key[1],), device=g.device, dtype=torch.int32),
                                            torch.zeros((1,), device=g.device, dtype=torch.int32))
            tiles, n_blocks = buf
            check(lib().nbm_roi_tiles(_ptr(rois[b0:b0 + nb]), _ptr(n_roi[b0:b0 + nb] if per else n_roi), nb, rois.shape[1], nl, level,
                                      fh, fw, None if cell else _ptr(pat.full), 1, _ptr(tiles), _ptr(n_blocks), per, _stream()), 'nbm_roi_tiles')
            # composed reader: g holds the RoI pooling's share only (pattern pixels included), ADDED to the cell patches
            _wino23_tiles_run(g[b0:b0 + nb], Ut, None, gx.data_ptr() + b0 * img_bytes, tiles, n_blocks, None, 'wino23-dgrad-rois',
                              skip_pattern=st.stride if (overlap and comp is None) else 0, accumulate=overlap or comp is not None)
            if pool is not None:
                tl, nbk = tiles.clone(), n_blocks.clone()          # the list buffer is shared by all chunks / levels
                zero_note(pool, lambda p_=gx.data_ptr() + b0 * img_bytes, nb_=nb, tl_=tl, nbk_=nbk: check(
                    lib().nbm_zero_tiles(C.c_void_p(p_), nb_, H, W, C_, _ptr(tl_), tl_.numel(), _ptr(nbk_), _stream()), 'nbm_zero_tiles'))
    if do_lat:
        if dt_pool is not None and ZERO_POOL_CHECK:
            dt_pool['check'] = lambda buf, s_=st.stride: _check_zero_outside_patches(buf, s_)
        lt.grads = dict(dt=dt, gb=gb_lat, gw_roi=gw_roi, gw_cell=None, up_share=None)
        if split:
            def complete(gy, shares=shares, pool=pool):
                """A dense kernel is going to read d/d(merged map) after all: the RoI share goes into the map."""
                for Gc_, tl_, b0_, nb_ in shares:
                    p_ = C.c_void_p(gy.data_ptr() + b0_ * img_bytes)
                    check(lib().nbm_tiles_scatter_add(p_, nb_, H, W, C_, _ptr(tl_), tl_.numel(), None, _ptr(Gc_), _stream()),
                          'nbm_tiles_scatter_add')
                    if pool is not None and pool['buf'].data_ptr() == gy.data_ptr():
                        zero_note(pool, lambda p_=p_, nb_=nb_, tl_=tl_: check(
                            lib().nbm_zero_tiles(p_, nb_, H, W, C_, _ptr(tl_), tl_.numel(), None, _stream()), 'nbm_zero_tiles'))
            lt.grads.update(up_share=shares, pat_stride=st.stride, gx_ptr=gx.data_ptr(), complete=complete)
    return gx


def _check_zero_outside_patches(buf, stride):
    """NBM_ZERO_POOL_CHECK: after a recycle the persistent data-gradient map may hold values on the 5x5 cell patches only."""
    B, H, W, _ = buf.shape
    def axis(n):
        i = torch.arange(n, device=buf.device)
        cells = (n + 2 - 3) // stride + 1
        return ((i + 2) % stride < 5) & ((i + 2) // stride < cells)
    outside = ~(axis(H)[:, None] & axis(W)[None, :])
    bad = float(buf[:, outside].abs().max()) if bool(outside.any()) else 0.0
    if bad != 0.0:
        raise RuntimeError(f'persistent gradient map not clean after its recycle: {bad}')


def conv3x3_winograd_wgrad_tiles(st, x, g, want_bias=False, cell=None):
    """Weight gradient of a demand-driven convolution (LazyMap `st`): g [B,H,W,N] is zero outside the pixels that were read.
    `cell` (default: when usable): the PATTERN pixels of g through the cell transforms (cellwino.hip: per stride x stride cell the
    correlation of the 5x5 input patch with the 3x3 gradient block, 25 TN GEMMs dUc[xi] = Vg[xi]^T Vx[xi] over the cells), the
    other pixels of the tiles under the RoI windows through F(2x2,3x3) over the RoI lists (compact operands, 16 TN GEMMs, pattern
    pixels read as zeros): every pixel of g counts once.  -> (dU [16,N,C], bias gradient [N] | None, dUc [25,N,C] | None); the
    weight gradient is G^T dU G + E^T dUc E.  cell=False: everything through F(2x2,3x3) over the pattern + RoI tile lists with
    their plane masks (round 2)."""
    _chk(x, name='x'), _chk(g, name='g')
    B, H, W, C_ = x.shape
    N = g.shape[-1]
    if cell is None:
        cell = cell_usable(st, H, W, C_, N)
    if st.lateral is not None and st.lateral.deferred and not cell:
        raise RuntimeError('the lateral in front of this demand-driven convolution was deferred (its pattern patches were never '
                           'written): only the cell-domain weight gradient can run')
    dU = torch.zeros((16, N, C_), device=x.device, dtype=torch.float32)
    dUc = torch.zeros((25, N, C_), device=x.device, dtype=torch.float32) if cell else None
    gb = torch.zeros((N,), device=x.device, dtype=torch.float32) if want_bias else None
    lmax = max(128, (ops.WINO_CHUNK_BYTES // (16 * (C_ + N) * 4)) // 128 * 128)
    stream = _stream()
    thw = ((H + 1) // 2) * ((W + 1) // 2)
    comp = st.comp if cell else None
    if st.comp is not None and (not cell or st.comp['dUc'] is None):
        raise RuntimeError('the RPN reader of this demand-driven map was composed with its convolution but its backward pass has not run '
                           '/ the cell-domain weight gradient is switched off (NBM_RPN_COMPOSITE_TRAIN=0 restores the uncomposed chain)')
    for ci, (b0, nb, pat) in enumerate(st.chunks):
        lists, infos = ([], []) if cell else ([pat.tiles], [pat.entry_pm])
        if cell and comp is not None:
            dUc = comp['dUc']                                  # composed reader: A^T dW', summed over chunks and classes by its backward pass
        elif cell:
            vg = _cell_outgrad(st, g, ci, b0, nb)              # from the data gradient's pass when that ran first
            Vx, _, K, T = _cell_operand(st, b0, nb, H, W, C_, x, ci=ci)   # deferred lateral: K = C + Cin, dUc is the gradient of [U | alpha U W]
            if dUc.shape[2] != K:
                dUc = torch.zeros((25, N, K), device=x.device, dtype=torch.float32)
            conv_wgrad(vg, Vx, dUc, B=1, H=T, W=1, Cin=K, N=N, groups=25, g_gs=T * N, x_gs=T * K, out_gs=N * K)
            st.vg.pop(ci, None)
        parts = []
        for per_chunk in st.roi:                    # one entry per RoI pooling that read the map
            tiles, host, ev = per_chunk[ci][:3]
            ev.synchronize()                        # recorded during the forward pass: long done
            n = int(host.item()) * 128
            if n:
                parts.append(tiles[:n])
        if parts:
            if len(parts) == 1:
                roi = parts[0]
            else:                                   # several RoI sets: every tile once (ascending ids, -1 padding at the end)
                u = torch.unique(torch.cat(parts))
                u = u[u >= 0]
                roi = torch.full((-(-u.numel() // 128) * 128,), -1, device=u.device, dtype=torch.int32)
                roi[:u.numel()] = u
            lists.append(roi)
            if cell:        # all planes of every RoI tile; the pattern pixels inside them are masked out of g instead
                infos.append(torch.full_like(roi, 0x1ffff))
            else:
                # the RoI phase recomputed pattern tiles of which only some pixels had been stored: the planes of their class
                # come from the pattern list, this entry contributes the others
                tpm = pat.tile_pm[roi.clamp(min=0) % thw]
                infos.append((0xffff & ~tpm) | ((tpm == 0).int() << 16))
        if not lists:
            continue
        full = torch.cat(lists) if len(lists) > 1 else lists[0]
        info = torch.cat(infos) if len(infos) > 1 else infos[0]
        xs, gs = x[b0:b0 + nb], g[b0:b0 + nb]
        for l0 in range(0, full.numel(), lmax):
            lst = full[l0:l0 + lmax]
            L = lst.numel()
            V, dM = ops._wino_scratch(x.device, 16 * L * C_, 16 * L * N)
            check(lib().nbm_wino23_input_tiles(_ptr(xs), nb, H, W, C_, _ptr(lst), L, _ptr(info[l0:]), _ptr(V), stream),
                  'nbm_wino23_input_tiles')
            check(lib().nbm_wino23_outgrad_tiles(_ptr(gs), nb, H, W, N, _ptr(lst), L, _ptr(info[l0:]), _ptr(dM), _ptr(gb),
                                                 st.stride if (cell and comp is None) else 0, stream), 'nbm_wino23_outgrad_tiles')
            conv_wgrad(dM, V, dU, B=1, H=L, W=1, Cin=C_, N=N, groups=16, g_gs=L * N, x_gs=L * C_, out_gs=N * C_)
    if cell and gb is not None:
        share = comp['gb'] if comp is not None else st.cell_gb
        if share is not None:
            gb += share
    if cell:
        st.vg = st.cell_gb = st.cell_gb_done = None
        if dUc.shape[2] != C_:          # deferred lateral: d/dU = d/d(first block) + alpha * d/d(second block) W^T  (second block = alpha U W)
            lt = st.lateral
            K, Cin = dUc.shape[2], dUc.shape[2] - C_
            d2 = dUc[:, :, C_:]                                           # [25][N][Cin] view, row pitch K
            if lt.grads is not None and lt.ufold is not None:        # ... and d/dW_lat = alpha * sum_xi U_xi^T d/d(second block)_xi
                # one TN GEMM over the 25 N rows: out [C][Cin] = alpha * U2d^T d2 (was: torch.einsum -> vendor GEMM)
                gwc = torch.zeros((C_, Cin), device=x.device, dtype=torch.float32)
                conv_wgrad(lt.ufold, d2, gwc, B=1, H=25 * N, W=1, Cin=Cin, N=C_, g_ld=K, x_ld=K, alpha=float(lt.alpha))
                lt.grads['gw_cell'] = gwc
            # dU = d/d(first block) + alpha * d/d(second block) W_lat^T: an NT GEMM with the first block as its residual
            dU1 = torch.empty((25, N, C_), device=x.device, dtype=torch.float32)
            gemm_conv(d2, lt.wk, dU1, B=1, H=25 * N, W=1, Cin=Cin, N=C_, x_ld=K, w_ld=lt.wk.shape[1], residual=dUc, res_ld=K,
                      alpha=float(lt.alpha))
            dUc = dU1
    return dU, gb, dUc
