"""Weight-layout preparation for the HIP kernels.

Checkpoints keep the reference layout (conv weight [Cout, Cin, kh, kw]); the implicit-GEMM kernel
wants KRSC rows [Cout, kh*kw*Cin] (k contiguous, padded to a multiple of 32 floats when Cin is not).
Prepared copies are cached per parameter and invalidated by torch's version counter.
"""
import weakref

import torch
import torch.nn.functional as F

_cache = {}         # (id(tensor), tag) -> (weakref to the tensor, version tuple, prepared value)
_epoch = 0          # bumped whenever a kernel rewrote parameters / buffers through raw pointers


def bump():
    """Invalidate every prepared copy: the fused optimiser and the train-mode BatchNorm kernels update tensors in
    place through raw pointers, which torch's version counters cannot see."""
    global _epoch
    _epoch += 1


_DEBUG = __import__('os').environ.get('NBM_PREP_DEBUG') == '1'


def _evict(obj_id):
    for k in [k for k in _cache if k[0] == obj_id]:
        hit = _cache.pop(k, None)
        if _DEBUG and hit is not None:
            v = hit[2]
            ts = [t for t in (v if isinstance(v, (tuple, list)) else [v]) if torch.is_tensor(t)]
            print(f'_prep: evicted {k[1]} (key tensor died), values at {[hex(t.data_ptr()) for t in ts]}', flush=True)


def _cached(key_t, tag, fn, extra=(), epoch=True):
    """Prepared copy of `key_t`.  An entry belongs to ONE tensor object: it keeps a weak reference to it and is only
    valid while that referent is alive and is `key_t` itself -- CPython reuses ids and the caching allocator reuses
    device addresses, so (id, data_ptr, version) alone can match a different model's parameter after the first model was
    freed.  Entries are dropped when their tensor dies (the prepared copies would otherwise leak on the device)."""
    key = (id(key_t), tag)
    # `epoch=False`: the inputs are buffers that no kernel ever writes through a raw pointer (FrozenBatchNorm): torch's own version
    # counters see every change they can undergo (load_state_dict, .to()), so the entry outlives the optimiser steps
    ver = (key_t.data_ptr(), key_t._version, key_t.device, _epoch if epoch else -1) + tuple(extra)
    hit = _cache.get(key)
    if hit is not None and hit[0]() is key_t and hit[1] == ver:
        return hit[2]
    with torch.no_grad():
        val = fn()
    if _DEBUG:
        cap = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        print(f'_prep: made {tag} for a {tuple(key_t.shape)} key (replaces an entry: {hit is not None}; during a capture: {cap})', flush=True)
    obj_id = id(key_t)
    _cache[key] = (weakref.ref(key_t, lambda _r, obj_id=obj_id: _evict(obj_id)), ver, val)
    return val


def krsc(weight):
    """[Cout, Cin, kh, kw] -> contiguous [Cout, K'] with K' = kh*kw*Cin rounded up to 32 when Cin % 32 != 0."""
    def make():
        co, ci, kh, kw = weight.shape
        w = weight.detach().permute(0, 2, 3, 1).reshape(co, kh * kw * ci)
        if ci % 32:
            w = F.pad(w, (0, (-w.shape[1]) % 32))
        return w.contiguous()
    return _cached(weight, 'krsc', make)


def wino23(weight, transposed=False, m=2, scale=None):
    """Winograd F(m x m, 3x3) weights U[(m+2)^2][N][C] = (G g G^T)[i][j] of a [Cout, Cin, 3, 3] convolution (`nbm_wino_weight`:
    float64 arithmetic on the device, rounded once).  `transposed`: the weights of the DATA-GRADIENT convolution (kernel
    rotated by 180 degrees, channel roles swapped): U[..][Cin][Cout]; `scale` [Cout] (transposed only): the FrozenBN scale
    that multiplies the incoming gradient per output channel, folded into the weights."""
    from .. import ops

    def make():
        return ops.wino_weight(weight.detach().contiguous(), transposed, m, None if scale is None else scale.detach().contiguous())
    # ONE entry per (weight, m, transposed, scaled?): the scale's identity is part of the entry's VERSION, not of its key -- the
    # FrozenBN affine is a fresh tensor every forward pass, so a key holding its data_ptr added a 36-plane copy per layer and step
    # that nothing evicted (60-130 MB per training step: scripts/soak.py)
    extra = () if scale is None else (scale.data_ptr(), scale._version)
    return _cached(weight, ('wino', m, transposed, scale is not None), make, extra=extra)


def wino23_weight_grad(dU, m=2, row_scale=None):
    """dU [(m+2)^2, N, C] (gradient wrt the transformed weights) -> dW [N, C, 3, 3] = row_scale[n] G^T dU G."""
    from .. import ops
    return ops.wino_weight_grad(dU, m, row_scale)


def cell_weight(weight, forward=False):
    """Kernel side of the cell transforms (csrc/cellwino.hip, `nbm_cell_weight`): [Cout, Cin, 3, 3] -> U [25][Cin][Cout] =
    (E w E^T)[a][b] -- the B operand of the 25 data-gradient GEMMs M_xi = Vg_xi U_xi^T; `forward`: [25][Cout][Cin], the B operand of
    the forward GEMMs M_xi = Vx_xi U_xi^T.  float64 on the device, rounded once; cached per weight version."""
    from .. import ops
    return _cached(weight, ('cell', forward), lambda: ops.cell_weight(weight.detach().contiguous(), forward=forward))


def cell_weight_folded(w_out, wk_lat, alpha, transposed=False):
    """B operand of the cell-domain plane GEMMs of a demand-driven 3x3 convolution whose lateral was deferred
    (ondemand.conv1x1_lazy(defer=True)): [25][N][C + Cin] = [ U | alpha * U W_lat ] with U = E w_out E^T ([25][N][C]) and
    W_lat = wk_lat the lateral's KRSC weights [C][Cin] -- the merged map x = alpha * W_lat t + b + up(x1) never exists
    on the pattern patches: its transform is [transform(up(x1) + b) | transform(t)].  `transposed`: the same values as
    [25][C + Cin][N], the B operand of the data-gradient GEMMs (rows C.. give d/dt in the transform domain).  Both layouts come out
    of one pair of launches (`nbm_cell_weight` + `nbm_cell_weight_fold`) and share the cache entry."""
    from .. import ops
    pair = _cached(w_out, ('cellfold', float(alpha)),
                   lambda: ops.cell_weight(w_out.detach().contiguous(), lateral=wk_lat.detach(), alpha=alpha, both=True),
                   extra=(wk_lat.data_ptr(), wk_lat._version, tuple(wk_lat.shape)))
    return pair[1] if transposed else pair[0]


def cell_weight_grad(dU):
    """dU [25][N][C] (gradient wrt the transformed kernel) -> dW [N, C, 3, 3] = E^T dU E (`nbm_cell_weight_grad`)."""
    from .. import ops
    return ops.cell_weight_grad(dU)


def stem_fold(w1, w_init, b_init):
    """Operands of `nbm_stem7x7`: init_conv (1 -> 3 channels, weight a_c, bias b_c) folded into conv1 [64,3,7,7], in float64:
    weff [56,64] (k = 8 r + s, the s = 7 column zero) = sum_c W1 a_c; wb [64,49] = sum_c W1 b_c; wb_full [64] = wb.sum(1)."""
    def make():
        w = w1.detach().double()
        a, b = w_init.detach().double().view(3), b_init.detach().double().view(3)
        we = (w * a.view(1, 3, 1, 1)).sum(1)                       # [64,7,7]
        wb = (w * b.view(1, 3, 1, 1)).sum(1)
        weff = torch.zeros((7, 8, 64), dtype=torch.float64, device=w.device)
        weff[:, :7] = we.permute(1, 2, 0)
        return (weff.view(56, 64).float().contiguous(), wb.reshape(64, 49).float().contiguous(), wb.sum((1, 2)).float().contiguous())
    return _cached(w1, 'stem', make, extra=(w_init.data_ptr(), w_init._version, b_init.data_ptr(), b_init._version))


def bn_affine(weight, bias, mean, var, eps, conv_bias=None, frozen=False):
    """Fold (conv bias +) BatchNorm running statistics into per-channel (scale, shift):
    y = (z + conv_bias - mean) * weight / sqrt(var + eps) + bias  =  z * scale + shift."""
    def make():
        scale = weight.detach() * (var + eps).rsqrt()
        shift = bias.detach() - mean * scale
        if conv_bias is not None:
            shift = shift + conv_bias.detach() * scale
        return scale.contiguous(), shift.contiguous()
    # any of the five tensors changing must invalidate: their versions are part of the entry's version (`extra`), NOT of its key -- a
    # key that changes with every update would leave one stale entry per update behind
    extra = (weight._version, bias._version, mean._version, var._version, None if conv_bias is None else conv_bias._version,
             mean.data_ptr(), var.data_ptr(), bias.data_ptr())
    # `frozen` (backbone.FrozenBatchNorm2d): the 53 affines of the ResNet body used to be recomputed after EVERY optimiser step (five
    # pointwise launches each: 265 of the ~650 small torch launches of a training step, scripts/small_launches.py) because the step
    # invalidates every prepared copy; their four tensors are buffers nothing writes behind torch's back
    return _cached(weight, ('bn', conv_bias is not None), make, extra=extra, epoch=not frozen)


def _rpn_composite64(out_w, out_b, dw_w, dw_b, pt_w, rmask, smask, lat_wk, alpha):
    """float64: (W_eff [N][5][5][K], const [N]) of `rpn_composite` for one class of cells."""
    n1, c = out_w.shape[0], out_w.shape[1]
    m_ = dw_w.shape[0]
    mult = m_ // n1
    wo = out_w.detach().double().repeat_interleave(mult, 0)              # [M, C, 3, 3]: depthwise channel m reads out_conv channel m // mult
    wd = dw_w.detach().double().reshape(m_, 3, 3)
    bo = (out_b.detach().double() if out_b is not None else torch.zeros(n1, dtype=torch.float64, device=out_w.device)).repeat_interleave(mult)
    d = torch.zeros((m_, c, 5, 5), dtype=torch.float64, device=out_w.device)
    cb = torch.zeros((m_,), dtype=torch.float64, device=out_w.device)
    for r in range(3):
        for s_ in range(3):
            if (rmask >> r) & 1 and (smask >> s_) & 1:
                d[:, :, r:r + 3, s_:s_ + 3] += wd[:, r, s_, None, None, None] * wo
                cb += wd[:, r, s_] * bo
    if dw_b is not None:
        cb += dw_b.detach().double()
    wp = pt_w.detach().double().reshape(pt_w.shape[0], m_)
    we = torch.einsum('nm,mcae->naec', wp, d)                            # [N, 5, 5, C]
    const = wp @ cb
    if lat_wk is not None:
        wl = lat_wk.detach().double()                                    # [C, Cin]
        we = torch.cat([we, float(alpha) * torch.einsum('naec,ci->naei', we, wl)], -1)
    return we, const


def _rpn_extra(out_b, dw_w, dw_b, pt_w, scale, shift, lat_wk, alpha):
    others = [t for t in (out_b, dw_w, dw_b, pt_w, scale, shift, lat_wk) if t is not None]
    return tuple(v for t in others for v in (t.data_ptr(), t._version)) + (float(alpha),)


def rpn_composite(out_w, out_b, dw_w, dw_b, pt_w, scale, shift, lat_wk=None, alpha=1.0, parts=None):
    """The RPN's strided reader composed with the output convolution in front of it (evaluation mode): fpn.py:137,145 `out_conv`
    (3x3 / pad 1, linear) -> layers.py:22-29 depthwise 3x3 / stride S / pad 1 (channel multiplier m) -> 1x1 -> BatchNorm (running
    statistics) -> SiLU is ONE 5x5 / stride S / pad 2 convolution of the merged map followed by scale / shift / SiLU:
        W_eff[n][a, e][c] = sum_m pt[n][m] sum_{r + u = a, s + v = e} dw[m][r, s] out_w[m // mult][c][u, v]
        const[n]          = sum_m pt[n][m] (sum_{r, s} dw[m][r, s] out_b[m // mult] + dw_b[m])
    `lat_wk` [C][Cin] (+ alpha): the lateral 1x1 in front of out_conv was deferred, the operands are [up(x1) + b | t] and the weights
    [W_eff | alpha W_eff W_lat].  `scale`, `shift`: `bn_affine` of the block (1x1 bias folded in).  `parts`: channel ranges
    ((c0, c1), ...) of K -- one weight tensor [N][25 * (c1 - c0)] (tap-major) per range, for launches that each take a slice of the
    channels; default: all of K in one.
    -> (tuple of weight tensors, scale [N], shift' [N] = scale * const + shift) for the cells whose nine depthwise taps all lie inside
    the map; float64 arithmetic, once per weight version.  Border cells: `rpn_composite_delta`."""
    def make():
        we, const = _rpn_composite64(out_w, out_b, dw_w, dw_b, pt_w, 7, 7, lat_wk, alpha)
        rng = parts if parts is not None else ((0, we.shape[-1]),)
        ws = tuple(we[..., c0:c1].reshape(we.shape[0], -1).float().contiguous() for c0, c1 in rng)
        return ws, scale, (scale.double() * const + shift.double()).float().contiguous()
    return _cached(out_w, ('rpnc', lat_wk is not None, parts), make, extra=_rpn_extra(out_b, dw_w, dw_b, pt_w, scale, shift, lat_wk, alpha))


def rpn_composite_delta(out_w, out_b, dw_w, dw_b, pt_w, scale, shift, rmask, smask, taps, lat_wk=None, alpha=1.0):
    """Border cells of `rpn_composite`: the depthwise convolution pads the OUTPUT of out_conv with zeros, which a padded 5x5 convolution
    of the input does not reproduce -- a cell whose depthwise tap rows / columns are not all inside the map (`rmask` / `smask`: bit set =
    inside) has weights of its own.  -> (dw [N][len(taps) * K], dshift [N]): scale * (W_class - W_interior) on the listed patch taps
    (a * 5 + e; the caller lists those where the difference is not zero and the patch can hold data) and scale * (const_class -
    const_interior) -- the BatchNorm scale is folded in, so the launch that applies them needs no per-channel epilogue operand."""
    def make():
        we, const = _rpn_composite64(out_w, out_b, dw_w, dw_b, pt_w, rmask, smask, lat_wk, alpha)
        wi, ci = _rpn_composite64(out_w, out_b, dw_w, dw_b, pt_w, 7, 7, lat_wk, alpha)
        d = (we - wi).reshape(we.shape[0], 25, -1)[:, list(taps)] * scale.double()[:, None, None]
        return d.reshape(d.shape[0], -1).float().contiguous(), (scale.double() * (const - ci)).float().contiguous()
    return _cached(out_w, ('rpnd', int(rmask), int(smask), tuple(taps), lat_wk is not None), make,
                   extra=_rpn_extra(out_b, dw_w, dw_b, pt_w, scale, shift, lat_wk, alpha))


def lateral_of_projection(lat_w, lat_b, proj_w, proj_b):
    """FPN lateral 1x1 (fpn.py:143) applied to `fm + final_projection(ctx)` (self_attention.py:55,76) without forming that sum:
    lateral(fm + W_o ctx + b_o) = W_l fm + (W_l W_o) ctx + (W_l b_o + b_l).  -> (W_l W_o as KRSC rows [N][d], shift [N]); float64
    arithmetic, once per weight version (evaluation mode: fpn.FPN.forward on a `Projected` level)."""
    def make():
        wl = lat_w.detach().double().reshape(lat_w.shape[0], -1)
        wc = wl @ proj_w.detach().double()
        sh = wl @ proj_b.detach().double() + (lat_b.detach().double() if lat_b is not None else 0.0)
        return wc.float().contiguous(), sh.float().contiguous()
    others = [t for t in (lat_b, proj_w, proj_b) if t is not None]
    return _cached(lat_w, 'latproj', make, extra=tuple(v for t in others for v in (t.data_ptr(), t._version)))


def value_of_lateral(lat_w, proj_w, val_w, val_b):
    """The attention's value projection composed with its final projection AND the FPN lateral that reads the level (evaluation mode):
    ctx W_o^T W_l^T = P (x W_v^T + b_v) W_o^T W_l^T = P (x W_v'^T + b_v') with W_v' = W_l W_o W_v [p][C], b_v' = W_l W_o b_v -- the value
    rows are p = 384 wide instead of d = C / 2, and P V' IS the lateral's image of the attention branch (rows of P sum to one).
    -> (W_v' [p][C], b_v' [p]); float64 arithmetic, once per weight version."""
    def make():
        wlo = lat_w.detach().double().reshape(lat_w.shape[0], -1) @ proj_w.detach().double()      # [p][d]
        return (wlo @ val_w.detach().double()).float().contiguous(), (wlo @ val_b.detach().double()).float().contiguous()
    others = [proj_w, val_w, val_b]
    return _cached(lat_w, 'vallat', make, extra=tuple(v for t in others for v in (t.data_ptr(), t._version)))


def cat_rows(tag, *tensors):
    """Concatenate several [Ni, K] weight matrices (or [Ni] biases) along dim 0, cached on the first."""
    def make():
        return torch.cat([t.detach().reshape(t.shape[0], -1) if t.dim() > 1 else t.detach() for t in tensors], 0).contiguous()
    extra = tuple(v for t in tensors for v in (t.data_ptr(), t._version))
    return _cached(tensors[0], ('cat', tag), make, extra=extra)


def clear():
    _cache.clear()
