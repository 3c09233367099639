"""Differentiable operator layer of the training path.

torch.autograd is used as the tape (plumbing: it sequences the backward calls and owns the buffers); every
forward AND backward computation below is a HIP kernel of libnbm_hip.so.  Replaces what autograd derives for the
reference's `losses.backward()` (reference train.py:212).  Activations NHWC fp32; weights arrive in the checkpoint
layout and their gradients are returned in the same layout.
"""
import os
import weakref

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import ops, ondemand
from . import _prep

ACT_NONE, ACT_RELU, ACT_LEAKY = ops.ACT_NONE, ops.ACT_RELU, ops.ACT_LEAKY


def _pad32_rows(g2d, n):
    """[M, n] -> zero-padded contiguous [M, ceil32(n)] (the dgrad kernel reads whole 32-float K steps)."""
    n32 = (n + 31) // 32 * 32
    if n32 == n:
        return g2d
    out = torch.zeros((g2d.shape[0], n32), device=g2d.device, dtype=torch.float32)
    out[:, :n] = g2d
    return out


# Weight / bias gradients straight into the optimiser's flat gradient buffer: `train.FusedAdamW` gives every
# parameter a pre-assigned gradient view that ONE memset zeroes per step; a weight-gradient kernel whose output layout IS the
# parameter's layout (1x1 convolutions, nn.Linear: KRSC rows == [Cout][Cin]) accumulates into that view and the node returns None --
# instead of a zero fill of a temporary, the kernel, and autograd's AccumulateGrad add into the same view (~100 fills + ~100 adds of a
# B = 128 step).  The optimiser's post-accumulate hook is called by hand (`mark`), since autograd never sees the gradient.
GRAD_SINK = {}          # data_ptr of a parameter -> (weakref to it, mark)
DIRECT_WGRAD = os.environ.get('NBM_DIRECT_WGRAD', '1') != '0'


def grad_sink(param, rows, cols):
    """-> ([rows][cols] view of the parameter's own zero-initialised gradient, mark) or None."""
    if not DIRECT_WGRAD or param is None or not torch.is_tensor(param):
        return None
    e = GRAD_SINK.get(param.data_ptr())
    if e is None:
        return None
    q = e[0]()
    if q is None or q.data_ptr() != param.data_ptr() or q.shape != param.shape:
        return None
    g = q.grad
    if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.numel() != rows * cols or (g.data_ptr() & 15):
        return None
    return g.view(rows, cols), (lambda q=q, mark=e[1]: mark(q))


# The ReLU mask of a bottleneck's output as BITS (round 5): the epilogue that stores y = relu(...) also stores (y > 0), 32 channels per word
# (`nbm_gemm_desc.bits_out`); the next block's first data-gradient kernel, whose epilogue zeroes d/dx where its input was <= 0, then reads
# 1/32 of the bytes it read from y itself (`nbm_bwd_desc.mask_bits`) -- those kernels run at the HBM roof through exactly these bytes
# (profiles/r05_dgrad_attribution.txt).  The bits belong to the tensor OBJECT that the forward pass produced (same storage, same version).
RELU_BITS = os.environ.get('NBM_RELU_BITS', '1') != '0'
_RELU_BITS = {}         # data_ptr of a ReLU output -> (weakref to it, its version, int32 bits [numel / 32])


def _relu_bits_forget(ptr, ident):
    e = _RELU_BITS.get(ptr)
    if e is not None and id(e[2]) == ident:
        del _RELU_BITS[ptr]


def relu_bits_note(y, bits):
    _RELU_BITS[y.data_ptr()] = (weakref.ref(y), y._version, bits)
    weakref.finalize(y, _relu_bits_forget, y.data_ptr(), id(bits))


def relu_bits_of(t):
    """The (t > 0) bits written together with `t`, or None."""
    e = _RELU_BITS.get(t.data_ptr()) if RELU_BITS else None
    if e is None:
        return None
    q = e[0]()
    if q is None or q.data_ptr() != t.data_ptr() or q.shape != t.shape or t._version != e[1]:
        return None
    return e[2]


def _w_to_ref_layout(gw, weight):
    """KRSC gradient rows [N, >=K] -> the parameter's own layout."""
    if weight.dim() == 2:
        return gw[:, :weight.shape[1]]
    n, c, kh, kw = weight.shape
    return gw[:, :kh * kw * c].view(n, kh, kw, c).permute(0, 3, 1, 2)


STEM_FOLDED = True       # init_conv folded into conv1 (csrc/stem.hip); False: init_conv kernel + generic implicit GEMM
WINO_MIN_CIN = int(__import__('os').environ.get('NBM_WINO_MIN_CIN', '128'))   # 64-channel 3x3 layers: measured, see DESIGN 5
WINOGRAD = True          # module switch for A/B tests (tests/test_gpu_e2e.py compares both convolution paths)
GRAD_SHARE = True        # the two consumers of an FPN map accumulate their gradients into one buffer (RoiPool / DwConv)
_GRAD_ACC = {}           # data_ptr of an FPN map -> weak reference to the gradient map the RoI pooling's backward pass filled
LAZY_DGRAD = True        # data gradient of the demand-driven finest FPN map through the listed fused kernel
LAZY_WGRAD = True        # weight gradient of the demand-driven finest FPN map over its computed tiles only
WINO_BWD_TILE = 4        # F(4x4,3x3) for the two backward convolutions (gradients tolerate its 2e-5 error); 2 = F(2x2,3x3)

# Gradient hand-over between the consumers of one tensor.  A tensor with two consumers (a merged FPN map: its output convolution and
# the next finer level's top-down merge; a backbone tap: the next ResNet stage and the FPN lateral) gets its gradient from autograd
# as a SUM of two dense tensors -- a pass of its own that reads both and writes a third (9 ms of a B = 128 step).  Instead the
# consumer whose backward node runs FIRST (the one created later: autograd runs ready nodes by descending sequence number, and the
# FPN nodes never wait for a backbone node) leaves its gradient here and returns None; the other consumer, which registered itself in
# the forward pass (`stash_accept`), adds it in the epilogue of the kernel that produces its own gradient (`residual`).
GRAD_STASH = True
_STASH = {}              # data_ptr -> gradient left by the first consumer's backward pass
_STASH_OK = {}           # data_ptr -> shape of the tensors whose OTHER consumer will pick a stashed gradient up (this forward pass)


# A consumer that ends up with the COMPLETE gradient of a ReLU output in its own kernel (its share + the stashed one) multiplies it
# by (x > 0) there and says so here: activation data_ptr -> (data_ptr, version counter) of the gradient tensor it returned.  The
# producer skips its own masking pass (read gy, read y, write: 1.9 ms for the stem output at B = 128) only if the gradient autograd
# hands it is that very tensor, untouched -- a sum with a third consumer's share is another tensor or (accumulated in place) another
# version, and is then masked again, which changes nothing.
_PREMASKED = {}
PREMASK = True           # module switch for A/B measurements (scripts/trainloop.py)
HALF_RES_SHORTCUT = True  # Bottleneck.backward: a stride-2 shortcut's data gradient stays at half resolution (A/B switch)
UPBWD_SPLIT_READ = True   # the finest lateral's node reads a split gradient (ondemand.UPBWD_SPLIT) as it is; False: it puts the RoI share
                          # into the map first and reads that densely (A/B switch, and the fallback's test)


def _premasked(y, gy):
    tag = _PREMASKED.pop(y.data_ptr(), None)
    return tag is not None and tag == (gy.data_ptr(), gy._version)


# Early backward pass of the RPN branch (train.step, SPLIT_BACKWARD): the first-stage loss does not depend on the second stage, so its
# gradient is propagated through the RPN head -- down to the FPN output maps, no further -- BEFORE the host waits for the RoI count:
# those ~15 ms of kernels fill the window in which the GPU used to idle behind the host-side proposal targets.  The RPN's share of
# d/d(FPN map) is PARKED here (data_ptr of the map -> (gradient map, the map itself)); the RoI pooling's backward pass, the map's
# other consumer, scatters into it and hands the sum to the producer -- the mirror image of _GRAD_ACC.
EARLY = False            # True while train.step runs that early pass
_PARKED = {}
_FPN_OUT = {}            # data_ptr -> FPN output map (NHWC) of this forward pass
_FPN_UP = {}             # data_ptr of an up-sampled copy of an FPN map (the RPN's first operator on level P5) -> that map


# ---- one training pass in flight per process.  The registries above are keyed by `data_ptr()` of tensors that are alive for the whole
# pass (saved tensors / parked maps), so keys of two passes never collide; what CAN go wrong is ORDER: a forward pass that starts while
# an earlier pass still has gradients parked (its RPN branch was back-propagated early) or stashed (its backward pass is half-way) used
# to wipe them silently -- the earlier pass then lost a share of d/d(FPN map).  Now:
#   * a forward pass under `no_grad` (validation between two training steps, reference train.py:362-377; `detect`) registers nothing and
#     touches nothing here;
#   * a grad-enabled forward pass gets a token owned by its model; starting one while `_PARKED` / `_STASH` still hold gradients of the
#     previous token raises instead of dropping them (gradient accumulation over two forwards, two models interleaved inside a step);
#   * the early backward pass and `parked_flush` must belong to the token of the forward pass that registered the FPN maps;
#   * every backward pass validates itself: the first node that stashes / consumes a hand-over queues an engine callback that runs
#     `stash_check_empty` when the engine finishes -- also for callers that use `step()` + their own `backward()`.
_PASS = {'token': 0, 'owner': None}


def pass_token():
    return _PASS['token']


def _owner_name(ref):
    o = ref() if ref is not None else None
    return type(o).__name__ + f'@{id(o):#x}' if o is not None else 'a model that no longer exists'


def fpn_out_register(maps, token=None):
    if token is not None and token != _PASS['token']:
        raise RuntimeError('FPN maps registered by a forward pass that is no longer the current one (functional._PASS): another '
                           'grad-enabled forward pass started in between')
    if not torch.is_grad_enabled():        # inference: nothing will be parked, and the registry must not keep 8 GB of maps alive
        return
    _FPN_OUT.clear()
    _FPN_UP.clear()
    _PARKED.clear()
    for m in maps:
        _FPN_OUT[m.data_ptr()] = m


def parked_flush():
    """No RoI pooling consumed the parked gradients (the step ended after the first stage): propagate them into the FPN now."""
    items = [(m, g) for (g, m, _) in _PARKED.values()]
    _PARKED.clear()
    _FPN_OUT.clear()                       # end of the step: the registry lets go of the maps
    _FPN_UP.clear()
    if items:
        torch.autograd.backward([m for m, _ in items], [g for _, g in items])


def stash_reset(owner=None):
    """Start of a forward pass of `owner` (NbmModel._fpn_nhwc) -> the pass token (0 under no_grad: such a pass registers nothing --
    `ctx.needs_input_grad` is all False -- and leaves a training pass in flight alone)."""
    if not torch.is_grad_enabled():
        return 0
    if _PARKED or _STASH:
        what = ', '.join(f'{len(d)} {n}' for d, n in ((_PARKED, 'parked RPN share(s) of d/d(FPN map)'), (_STASH, 'stashed gradient(s)')) if d)
        raise RuntimeError(f'a grad-enabled forward pass started while the previous one (token {_PASS["token"]}, {_owner_name(_PASS["owner"])}) '
                           f'still holds {what}: its backward pass has not finished.  One forward / backward pair in flight per process '
                           '(gradient accumulation: finish each micro-batch with backward(); two models: finish one step before the '
                           'other starts; to abandon the pending pass call functional.pass_abandon())')
    _STASH_OK.clear()
    _PREMASKED.clear()
    _FPN_OUT.clear()
    _FPN_UP.clear()
    ondemand.zero_pool_new_pass()
    _PASS['token'] += 1
    _PASS['owner'] = weakref.ref(owner) if owner is not None else None
    return _PASS['token']


def pass_abandon():
    """Drop whatever the current pass parked / stashed (an exception ended its step; a caller decided not to back-propagate it)."""
    _STASH.clear()
    _PARKED.clear()
    _STASH_OK.clear()
    _PREMASKED.clear()
    _FPN_OUT.clear()
    _FPN_UP.clear()
    _GRAD_ACC.clear()
    ondemand.zero_pool_new_pass()


def pass_check_owner(model, what):
    o = _PASS['owner']() if _PASS['owner'] is not None else None
    if o is not None and model is not None and o is not model:
        raise RuntimeError(f'{what}: the pass in flight (token {_PASS["token"]}) belongs to {_owner_name(_PASS["owner"])}, not to '
                           f'{type(model).__name__}@{id(model):#x} -- two models interleaved inside one step')


def stash_accept(t, needs_grad):
    """Called from the forward pass of the consumer that will pick the stash up; `needs_grad` = ctx.needs_input_grad of `t`
    (inside Function.forward grad mode is off, so torch.is_grad_enabled() says nothing)."""
    if GRAD_STASH and needs_grad:
        _STASH_OK[t.data_ptr()] = tuple(t.shape)
        return True
    return False


def _stash_wanted(t):
    return bool(GRAD_STASH and t is not None and _STASH_OK.get(t.data_ptr()) == tuple(t.shape))


def stash_check_empty():
    """After a backward pass: a gradient that nobody picked up would have been dropped silently."""
    if _STASH:
        n = len(_STASH)
        _STASH.clear()
        raise RuntimeError(f'{n} stashed gradient(s) were never picked up by their second consumer (functional._STASH)')


_END_CHECK = [False]


def _end_of_backward():
    _END_CHECK[0] = False
    stash_check_empty()


def arm_end_of_backward_check():
    """Called from inside a backward node that leaves a gradient in `_STASH`: when the engine finishes THIS backward run, nothing may
    be left there (queued once per run; an exception raised in the callback reaches the caller of `backward()`)."""
    if not _END_CHECK[0]:
        _END_CHECK[0] = True
        torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)


def _winograd_ok(x, weight, kh, kw, stride, pad):
    """3x3 / stride 1 / pad 1 with MFMA-friendly channel counts (odd map sizes cost one zero-padded tile row / column)."""
    if not (kh == 3 and kw == 3 and stride == 1 and pad == 1 and weight.dim() == 4 and x.dim() == 4) or not WINOGRAD:
        return False
    _, H, W, Cin = x.shape
    return H >= 8 and W >= 8 and Cin % 32 == 0 and weight.shape[0] % 4 == 0 and Cin >= WINO_MIN_CIN


def lazy3x3_ok(H, W, Cin, weight):
    """Would `conv(x [.,H,W,Cin], weight, kh=3, kw=3, pad=1, lazy_stride=...)` take the demand-driven path?  (The lateral in
    front of it may only go sparse if it does.)"""
    return bool(ondemand.LAZY_FINEST and WINOGRAD and weight.dim() == 4 and tuple(weight.shape[2:]) == (3, 3) and H >= 8 and W >= 8 and
                Cin % 32 == 0 and Cin >= 128 and weight.shape[0] % 4 == 0)


class Conv(Function):
    """y = act(alpha * conv(x, W) * scale + (bias | shift) + residual); also nn.Linear (x [1,M,1,K])."""

    @staticmethod
    def forward(ctx, x, weight, bias, scale, shift, residual, kh, kw, stride, pad, act, alpha, up=None, lazy_stride=None,
                accept_stash=False):
        sh = bias.detach() if bias is not None else shift
        ctx.wino = _winograd_ok(x, weight, kh, kw, stride, pad) and scale is None and residual is None and \
            act == ACT_NONE and alpha == 1.0 and up is None
        ctx.lazy = ctx.lat_state = None
        if ctx.wino and lazy_stride:
            # demand-driven map (ondemand.conv3x3_winograd_lazy): the tiles a 3x3 / lazy_stride consumer reads now, the tiles under
            # the RoIs when the RoI pooling asks for them; the backward pass is the dense one (the incoming gradient is
            # zero wherever nothing was read)
            y, ctx.lazy = ondemand.conv3x3_winograd_lazy(x, _prep.wino23(weight), sh, lazy_stride[0],
                                                         _prep.cell_weight(weight, forward=True) if ondemand.CELL_FWD else None,
                                                         fold=lambda wk, alpha, transposed=False: _prep.cell_weight_folded(weight, wk, alpha, transposed),
                                                         keep=lazy_stride[1], raw=(weight, bias))
        elif lazy_stride and kh == 1:
            # the lateral 1x1 (+ top-down merge) in front of a demand-driven 3x3: only the pixels that convolution reads
            # deferred: write nothing, the consumer takes t / up / the weights into its cell-domain GEMMs -- provided its backward
            # pass (if one can follow) is the cell-domain one, the only one that does not read the pattern patches of this map
            defer = not any(ctx.needs_input_grad) or (ondemand.CELL_BWD and LAZY_WGRAD and LAZY_DGRAD)
            y = ondemand.conv1x1_lazy(x, _prep.krsc(weight), sh, alpha, up, lazy_stride[0], defer=defer)
            hit = ondemand._LAZY_LATERAL.get(y.data_ptr())
            ctx.lat_state = hit[0] if hit is not None else None      # its consumer's backward pass may leave this node's gradients there
        elif ctx.wino:        # large 3x3 (FPN output convolutions): Winograd F(2x2,3x3), 2.25x fewer multiplies
            y = ops.conv3x3_winograd(x, _prep.wino23(weight), sh)
        else:
            wk = _prep.krsc(weight) if weight.dim() == 4 else weight.detach()
            y = ops.conv2d(x, wk, kh, kw, stride, pad, scale=scale, shift=sh, residual=residual, act=act, alpha=alpha, up=up)
        ctx.geom = (kh, kw, stride, pad, act, alpha)
        ctx.has_bias, ctx.has_res = bias is not None, residual is not None
        ctx.bias_param = bias                 # (for grad_sink: the bias gradient may go straight into the parameter's own gradient view)
        ctx.up_hw = tuple(up.shape[1:3]) if up is not None else None
        # gradient hand-over (see _STASH): leave d/dx / d/d(up) for the tensor's other consumer, or pick up what it left
        ctx.stash_x = kh == 1 and _stash_wanted(x)
        ctx.stash_up, ctx.up_ptr = _stash_wanted(up), (up.data_ptr() if up is not None else 0)
        ctx.take_x = bool(accept_stash) and (ctx.lazy is None or not ctx.lazy.sparse) and stash_accept(x, ctx.needs_input_grad[0])
        ctx.save_for_backward(x, weight, scale, y if act in (ACT_RELU, ACT_LEAKY) else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight, scale, y = ctx.saved_tensors
        kh, kw, stride, pad, act, alpha = ctx.geom
        gy = gy.contiguous()
        g = (gy if _premasked(y, gy) else ops.relu_bwd(gy, y)) if act == ACT_RELU else ops.leaky_relu_bwd(gy, y) if act == ACT_LEAKY else gy
        B, H, W, Cin = x.shape
        N = weight.shape[0]
        wk = _prep.krsc(weight) if weight.dim() == 4 else weight.detach()
        gp = _pad32_rows(g.view(-1, N), N)
        gx = gw = gb = None
        if ctx.lazy is not None and (ctx.lazy.sparse or ctx.lazy.overlap) and ctx.lazy.done != len(ctx.lazy.rois):
            raise RuntimeError('demand-driven FPN map: a RoI pooling ran on it without recording its tile lists (the map was '
                               'produced under no_grad?) -- its gradient would be dropped')
        listed = ctx.lazy is not None and ondemand.listed_backward(ctx.lazy) and (ctx.lazy.sparse or (LAZY_DGRAD and LAZY_WGRAD))
        if ctx.lazy is not None and ctx.lazy.comp is not None and ctx.lazy.comp['g'] is None:
            ctx.lazy.comp = None          # the composed RPN block received no gradient (its output did not reach the loss): no share to add
        if ctx.lazy is not None and ctx.lazy.comp is not None and not (listed and LAZY_DGRAD and LAZY_WGRAD and N % 32 == 0 and N >= 64 and
                                                                      len(ctx.lazy.rois) <= 1):
            # the map was read by the composed RPN block (Fn.RpnComposite) but the listed cell-domain passes will not run (a second RoI
            # pooling on the map: the RoI shares of two tile lists are ADDED to the cell share in the composed form, a tile in both lists
            # would count twice): its share goes into gy in the pixel domain, and the pass proceeds as without the composition
            g = gy = ondemand.train_composite_fallback(ctx.lazy, g.view(B, H, W, N))
            gp = _pad32_rows(g.view(-1, N), N)
        # a deferred lateral whose consumer's backward pass (which ran before this node's) already produced this node's gradients
        pre = ctx.lat_state.grads if ctx.lat_state is not None else None
        raw = pre
        if pre is not None and (pre['gw_cell'] is None or scale is not None or kh != 1):
            pre = None
        share = None
        if raw is not None and raw.get('up_share') is not None:
            # the consumer kept the RoI share of this node's incoming gradient out of the map (ondemand.UPBWD_SPLIT).  Good when all this
            # node computes from gy is the bilinear backward below; if a dense kernel is going to read gy, the share goes in first
            if UPBWD_SPLIT_READ and pre is not None and gy.data_ptr() == raw['gx_ptr'] and not (ctx.take_x and x.data_ptr() in _STASH):
                share = raw['up_share']
            else:
                raw['complete'](gy)
            raw['up_share'] = None
        if ctx.needs_input_grad[0] and listed and LAZY_DGRAD and N % 32 == 0 and N >= 64:
            # demand-driven map: the incoming gradient lives on the pattern pixels and in the RoI windows, the outgoing one
            # within a pixel of them -> the listed fused kernel on the tiles around them (F(2x2,3x3), no transforms through HBM)
            other = _STASH.pop(x.data_ptr(), None) if ctx.take_x else None       # overlap level: taken over as the accumulation base
            gx = ondemand.conv3x3_winograd_dgrad_tiles(ctx.lazy, g.view(B, H, W, N), _prep.wino23(weight, transposed=True, m=2),
                                                       _prep.cell_weight(weight) if ondemand.CELL_BWD else None, base=other,
                                                       lateral_grads=LAZY_WGRAD and ctx.needs_input_grad[1])
        elif ctx.needs_input_grad[0] and ctx.wino and N % 32 == 0:
            # data gradient of a 3x3 / stride 1 / pad 1 convolution = the same convolution with the kernel rotated by 180
            # degrees and the channel roles swapped: Winograd again
            other = _STASH.pop(x.data_ptr(), None) if ctx.take_x else None       # the other consumer's share of d/dx
            fuse = other is not None and WINO_BWD_TILE == 4
            gx = ops.conv3x3_winograd(g.view(B, H, W, N), _prep.wino23(weight, transposed=True, m=WINO_BWD_TILE), None,
                                      m=WINO_BWD_TILE, residual=other if fuse else None)
            if other is not None and not fuse:
                gx = ops.axpby(gx, other)
        elif ctx.needs_input_grad[0]:
            other = _STASH.pop(x.data_ptr(), None) if ctx.take_x else None
            if pre is not None and other is None:     # deferred lateral: d/dt came out of the consumer's cell-domain pass
                gx = pre['dt']
            else:
                gx = torch.empty_like(x)
                ops.conv_dgrad(gp, wk, gx, B=B, H=H, W=W, Cin=Cin, N=N, kh=kh, kw=kw, stride=stride, pad=pad,
                               g_ld=gp.shape[1], w_ld=wk.shape[1], a_scale=scale, alpha=alpha, residual=other)
            if ctx.stash_x:                           # the tensor's other consumer adds this in its own kernel
                _STASH[x.data_ptr()] = gx
                arm_end_of_backward_check()
                gx = None
        want_gb = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] and listed and LAZY_WGRAD:
            # demand-driven map: the gradient is zero outside the tiles that were computed -> F(2x2,3x3) over those tiles only
            dU, gb, dUc = ondemand.conv3x3_winograd_wgrad_tiles(ctx.lazy, x, g.view(B, H, W, N), want_bias=want_gb)
            gw = _prep.wino23_weight_grad(dU, 2)
            if dUc is not None:                       # pattern share: cell transforms (csrc/cellwino.hip)
                gw = gw + _prep.cell_weight_grad(dUc)
        elif ctx.needs_input_grad[1] and ctx.wino and N % 32 == 0:
            # weight gradient in the Winograd domain: 16 TN GEMMs dU = dM^T V, mapped back with dW = G^T dU G
            dU, gb = ops.conv3x3_winograd_wgrad(x, g.view(B, H, W, N), want_bias=want_gb, m=WINO_BWD_TILE)
            gw = _prep.wino23_weight_grad(dU, WINO_BWD_TILE)
        elif ctx.needs_input_grad[1] and pre is not None:
            gwk = torch.zeros_like(wk)
            gwk[:, :Cin] = pre['gw_roi'] + pre['gw_cell']
            gw = _w_to_ref_layout(gwk, weight)
            gb = pre['gb'] if want_gb else None
        elif ctx.needs_input_grad[1]:
            same_layout = (weight.dim() == 2 or (kh == 1 and kw == 1)) and wk.shape[1] == weight.shape[1]
            sink = grad_sink(weight, N, wk.shape[1]) if same_layout else None
            gwk = sink[0] if sink is not None else torch.zeros_like(wk)
            bsink = grad_sink(ctx.bias_param, 1, N) if want_gb else None
            if want_gb:                                   # the bias gradient rides along in the weight-gradient kernel
                gb = bsink[0].view(N) if bsink is not None else torch.zeros((N,), device=x.device, dtype=torch.float32)
            ops.conv_wgrad(gp, x, gwk, B=B, H=H, W=W, Cin=Cin, N=N, kh=kh, kw=kw, stride=stride, pad=pad,
                           g_ld=gp.shape[1], out_ld=wk.shape[1], row_scale=scale, alpha=alpha, bias_grad=gb)
            if sink is not None:                          # accumulated in place: autograd gets no gradient to add
                sink[1]()
                gw = None
            else:
                gw = _w_to_ref_layout(gwk, weight)
            if bsink is not None:
                bsink[1]()
                gb = None
        elif want_gb:
            gb = pre['gb'] if pre is not None else ops.colsum(gp, N)
        if ctx.lazy is not None:
            # the map's gradient has been consumed: drop the operands the LazyMap kept for further RoI poolings, as the tape
            # drops its saved tensors (19 GB of merged map + the lateral's inputs at B = 128 would otherwise stay allocated
            # through the rest of the backward pass)
            ctx.lazy.release()
        gres = g if (ctx.has_res and ctx.needs_input_grad[5]) else None
        gup = None
        if ctx.up_hw is not None and ctx.needs_input_grad[12]:        # fused top-down merge: d/d(coarse map)
            gup = (ops.upsample_bilinear_bwd(g, *ctx.up_hw) if share is None else
                   ops.upsample_bilinear_bwd(g, *ctx.up_hw, pattern_stride=raw['pat_stride'], tiles_share=share))
            if ctx.stash_up:                          # the coarse map's output convolution adds it in its data-gradient epilogue
                _STASH[ctx.up_ptr] = gup
                arm_end_of_backward_check()
                gup = None
        if gres is None:
            ondemand.zero_recycle(gy)   # a persistent gradient map (ondemand.zero_acquire): this node was its last reader
        return gx, gw, gb, None, None, gres, None, None, None, None, None, None, gup, None, None


class Bottleneck(Function):
    """One ResNet v1.5 bottleneck (conv1x1-BN-ReLU, conv3x3/s-BN-ReLU, conv1x1-BN, + shortcut, ReLU) with FrozenBN
    affines, as ONE tape node.  The backward pass knows the block's structure, so the ReLU masks and the shortcut
    addition ride in the data-gradient GEMM epilogues (`mask`, `residual`) instead of separate read-modify-write passes:
      * `mask_input`: the block input is itself a ReLU output, so d/dx is returned already multiplied by (x > 0);
      * `mask_gy`: the incoming gradient still has to be masked by (y > 0) -- true for blocks whose output has other
        consumers (the layer taps); interior blocks receive it pre-masked from their successor's `mask_input`.
    Masking is a 0/1 multiply, so a gradient that was masked early and is masked again after autograd's summation is
    unchanged."""

    @staticmethod
    def forward(ctx, x, w1, w2, w3, wd, s1, b1, s2, b2, s3, b3, sd, bd, stride, mask_input, mask_gy):
        a1 = ops.conv2d(x, _prep.krsc(w1), scale=s1, shift=b1, act=ACT_RELU)
        wino = stride == 1 and _winograd_ok(a1, w2, 3, 3, 1, 1)
        if wino:              # 3x3 / stride 1 with >= 128 channels: Winograd, FrozenBN + ReLU in the output transform
            a2 = ops.conv3x3_winograd(a1, _prep.wino23(w2), b2, scale=s2, relu=True)
        else:
            # (direct 3x3: its epilogue can write the ReLU bits of a2 too -- the mask of conv3's data gradient, 1/6 of that launch's bytes)
            P_ = w2.shape[0]
            ctx.a2_bits = None
            if RELU_BITS and any(ctx.needs_input_grad) and P_ % 32 == 0:
                Ho_, Wo_ = (a1.shape[1] + 2 - 3) // stride + 1, (a1.shape[2] + 2 - 3) // stride + 1
                ctx.a2_bits = torch.empty((a1.shape[0] * Ho_ * Wo_ * (P_ // 32),), device=x.device, dtype=torch.int32)
            a2 = ops.conv2d(a1, _prep.krsc(w2), 3, 3, stride, 1, scale=s2, shift=b2, act=ACT_RELU, bits_out=ctx.a2_bits)
        idt = x if wd is None else ops.conv2d(x, _prep.krsc(wd), 1, 1, stride, 0, scale=sd, shift=bd)
        # a backward pass may follow: the epilogue also writes (y > 0) as bits for the NEXT block's data gradient (relu_bits_note)
        N3 = w3.shape[0]
        bits = None
        if RELU_BITS and any(ctx.needs_input_grad) and N3 % 32 == 0:
            bits = torch.empty((a2.shape[0] * a2.shape[1] * a2.shape[2] * (N3 // 32),), device=x.device, dtype=torch.int32)
        y = ops.conv2d(a2, _prep.krsc(w3), scale=s3, shift=b3, residual=idt, act=ACT_RELU, bits_out=bits)
        if bits is not None:
            relu_bits_note(y, bits)
        ctx.x_bits = relu_bits_of(x)             # this block's input as bits, if its producer wrote them
        ctx.save_for_backward(x, a1, a2, y, w1, w2, w3, wd, s1, s2, s3, sd)
        ctx.cfg = (stride, mask_input, mask_gy, wino)
        # first block of a stage: its input is a backbone tap that the FPN lateral reads too -- the lateral's backward pass runs
        # first and leaves its share of d/dx in _STASH; the shortcut's data gradient below adds it in its epilogue
        ctx.take_x = wd is not None and stash_accept(x, ctx.needs_input_grad[0])
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, a1, a2, y, w1, w2, w3, wd, s1, s2, s3, sd = ctx.saved_tensors
        stride, mask_input, mask_gy, wino = ctx.cfg
        need = ctx.needs_input_grad
        B, H, W, Cin = x.shape
        Ho, Wo = y.shape[1:3]
        P, N3 = w1.shape[0], w3.shape[0]
        g3 = ops.relu_bwd(gy.contiguous(), y) if mask_gy and not _premasked(y, gy) else gy.contiguous()
        g3r = g3.view(-1, N3)
        k1, k2, k3 = _prep.krsc(w1), _prep.krsc(w2), _prep.krsc(w3)

        def wgrad(g2d, inp, wk, weight, scale, **geom):
            sink = grad_sink(weight, wk.shape[0], wk.shape[1]) if (weight.shape[2] == 1 and weight.shape[3] == 1 and
                                                                    wk.shape[1] == weight.shape[1]) else None
            out = sink[0] if sink is not None else torch.zeros_like(wk)
            ops.conv_wgrad(g2d, inp, out, row_scale=scale, g_ld=g2d.shape[1], out_ld=wk.shape[1], **geom)
            if sink is not None:
                sink[1]()
                return None
            return _w_to_ref_layout(out, weight)

        gw3 = wgrad(g3r, a2, k3, w3, s3, B=B, H=Ho, W=Wo, Cin=P, N=N3) if need[3] else None
        g2 = torch.empty_like(a2)
        ops.conv_dgrad(g3r, k3, g2, B=B, H=Ho, W=Wo, Cin=P, N=N3, g_ld=N3, w_ld=k3.shape[1], a_scale=s3, mask=a2,
                       mask_bits=getattr(ctx, 'a2_bits', None))
        g2r = g2.view(-1, P)
        geom2 = dict(B=B, H=H, W=W, Cin=P, N=P, kh=3, kw=3, stride=stride, pad=1)
        if wino:              # both gradients of the 3x3 in the Winograd domain (F(4x4,3x3)); BN scale folded into the weights
            m = WINO_BWD_TILE
            gw2 = None
            if need[2]:
                dU, _ = ops.conv3x3_winograd_wgrad(a1, g2, m=m)
                gw2 = _prep.wino23_weight_grad(dU, m, row_scale=s2)
            g1 = ops.conv3x3_winograd(g2, _prep.wino23(w2, transposed=True, m=m, scale=s2), None, m=m, mask=a1)
        else:
            gw2 = wgrad(g2r, a1, k2, w2, s2, **geom2) if need[2] else None
            g1 = torch.empty_like(a1)
            ops.conv_dgrad(g2r, k2, g1, g_ld=P, w_ld=k2.shape[1], a_scale=s2, mask=a1, **geom2)
        g1r = g1.view(-1, P)
        gwd, premask, gid2 = None, False, None
        if wd is None:
            gid = g3                                                     # identity shortcut
        else:
            kd = _prep.krsc(wd)
            if need[4]:
                gwd = wgrad(g3r, x, kd, wd, sd, B=B, H=H, W=W, Cin=Cin, N=N3, stride=stride)
            other = _STASH.pop(x.data_ptr(), None) if ctx.take_x else None     # the FPN lateral's share of d/dx (see _STASH)
            premask = PREMASK and other is not None                      # d/dx is complete below: mask it here (see _PREMASKED)
            if stride == 2 and HALF_RES_SHORTCUT and Cin % 4 == 0:
                # strided 1x1: only the (even, even) pixels have a tap -> its data gradient stays at ITS resolution (a plain GEMM
                # over the Ho x Wo grid) and the block's last data-gradient kernel adds it there (`residual2`), next to the
                # lateral's share: no full-resolution map that is 3/4 a copy (4.5 -> 1.7 ms for layer2.0 at B = 128)
                gid2 = torch.empty((B, Ho, Wo, Cin), device=x.device, dtype=torch.float32)
                ops.conv_dgrad(g3r, kd, gid2, B=B, H=Ho, W=Wo, Cin=Cin, N=N3, g_ld=N3, w_ld=kd.shape[1], a_scale=sd)
                gid = other
            else:
                gid = torch.empty_like(x)                                # strided 1x1: only the (even, even) class has a tap
                ops.conv_dgrad(g3r, kd, gid, B=B, H=H, W=W, Cin=Cin, N=N3, stride=stride, g_ld=N3, w_ld=kd.shape[1], a_scale=sd,
                               residual=other)
        gw1 = wgrad(g1r, x, k1, w1, s1, B=B, H=H, W=W, Cin=Cin, N=P) if need[1] else None
        gx = None
        if need[0]:
            gx = torch.empty_like(x)
            masked = mask_input or premask
            ops.conv_dgrad(g1r, k1, gx, B=B, H=H, W=W, Cin=Cin, N=P, g_ld=P, w_ld=k1.shape[1], a_scale=s1, residual=gid,
                           residual2=gid2, mask=x if masked else None, mask_bits=ctx.x_bits if masked else None)
            if premask:
                _PREMASKED[x.data_ptr()] = (gx.data_ptr(), gx._version)
        return (gx, gw1, gw2, gw3, gwd) + (None,) * 11


def conv(x, weight, bias=None, scale=None, shift=None, residual=None, kh=1, kw=1, stride=1, pad=0, act=ACT_NONE, alpha=1.0,
         up=None, lazy_stride=None, accept_stash=False):
    """`up` [B,h,w,N]: + bilinear_align_corners(up) in the GEMM epilogue (FPN top-down merge, act must be NONE).
    `lazy_stride`: the output has exactly two consumers, a 3x3 / lazy_stride / pad 1 convolution and the RoI pooling: only the
    pixels they read are computed (3x3 Winograd layers only; ignored elsewhere)."""
    if lazy_stride and kh == 1:       # lateral in front of the demand-driven convolution (ondemand.conv1x1_lazy)
        Cin, N = x.shape[-1], weight.shape[0]
        if not (ondemand.LAZY_FINEST and ondemand.LAZY_LATERAL and kw == 1 and stride == 1 and pad == 0 and weight.dim() == 4 and scale is None
                and residual is None and act == ACT_NONE and Cin % 32 == 0 and Cin <= 256 and N > 64 and N % 4 == 0):
            lazy_stride = None
    elif lazy_stride and not (ondemand.LAZY_FINEST and _winograd_ok(x, weight, kh, kw, stride, pad) and x.shape[-1] >= 64):
        lazy_stride = None
    if lazy_stride:
        # `keep`: a backward pass can follow (gradient wrt the weight OR the input): the RoI poolings record their tile lists
        lazy_stride = (int(lazy_stride), bool(torch.is_grad_enabled() and (weight.requires_grad or x.requires_grad)))
        # with one of the listed backward passes switched off (A/B switches) a DENSE backward kernel multiplies the holes of the
        # sparse maps by exact-zero gradients: the holes must then be zeros, not uninitialised memory
        ondemand.ZERO_FILL = not (LAZY_WGRAD and LAZY_DGRAD)
    return Conv.apply(x, weight, bias, scale, shift, residual, kh, kw, stride, pad, act, alpha, up, lazy_stride, accept_stash)


def linear(x2d, weight, bias=None, act=ACT_NONE, residual=None):
    M, K = x2d.shape
    res = residual.view(1, M, 1, -1) if residual is not None else None
    return Conv.apply(x2d.view(1, M, 1, K), weight, bias, None, None, res, 1, 1, 1, 0, act, 1.0, None, None, False).view(M, -1)


class Add(Function):
    """a + b (same shape) on the axpby kernel."""

    @staticmethod
    def forward(ctx, a, b):
        return ops.axpby(a.contiguous(), b.contiguous())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return g, g


class Scale(Function):
    """alpha * x (materialises the `fm + Identity(fm)` levels of SAPyramid when no convolution follows to absorb it)."""

    @staticmethod
    def forward(ctx, x, alpha):
        ctx.alpha = alpha
        return ops.axpby(x.contiguous(), None, alpha=alpha)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return ops.axpby(g.contiguous(), None, alpha=ctx.alpha), None


class AddConst(Function):
    """x + c with a constant c broadcast over the batch axis (sine position encoding of --add_posenc)."""

    @staticmethod
    def forward(ctx, x, c):
        return ops.axpby(x.contiguous(), c)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return g, None


class WeightedSum(Function):
    """BiFPN FusionModule arithmetic (reference fpn.py:20-30) for 2 or 3 inputs; x2 may be None."""

    @staticmethod
    def forward(ctx, x0, x1, x2, weights):
        xs = [x0.contiguous(), x1.contiguous()] + ([x2.contiguous()] if x2 is not None else [])
        ctx.save_for_backward(*xs, weights)
        return ops.weighted_sum(xs, weights.detach())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        *xs, weights = ctx.saved_tensors
        gxs, gw = ops.weighted_sum_bwd(list(xs), weights.detach(), g.contiguous(), ctx.needs_input_grad[:len(xs)])
        gxs = list(gxs) + [None] * (3 - len(gxs))
        return gxs[0], gxs[1], gxs[2], gw


class LayerNorm(Function):
    """nn.LayerNorm over the last axis of [rows, E] (Transformer_RCNN encoder, reference layers.py:618-621)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        ctx.save_for_backward(x, weight)
        ctx.eps = eps
        return ops.layernorm(x, weight.detach(), bias.detach(), eps)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        gx, gw, gb = ops.layernorm_bwd(x, weight.detach(), g.contiguous(), ctx.eps)
        return gx, gw, gb, None


class MhaSmall(Function):
    """softmax(q k^T / sqrt(hd)) v per (batch entry, head) over short sequences (nn.MultiheadAttention core); the
    probabilities are recomputed in the backward kernels."""

    @staticmethod
    def forward(ctx, q, k, v, S, N, nhead, seq_stride, batch_stride, n_valid):
        ctx.save_for_backward(q, k, v, n_valid)
        ctx.geom = (S, N, nhead, seq_stride, batch_stride)
        return ops.mha_small(q, k, v, S, N, nhead, seq_stride, batch_stride, n_valid)

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        q, k, v, n_valid = ctx.saved_tensors
        gq, gk, gv = ops.mha_small_bwd(q, k, v, go.contiguous(), *ctx.geom, n_valid)
        return gq, gk, gv, None, None, None, None, None, None


class Stem(Function):
    """init_conv (1 -> 3, 1x1 + bias) followed by conv1 7x7/s2 + FrozenBN + ReLU (reference backbone.py:104-113 and
    torchvision's stem).  Backward never forms the 3-channel data gradient: with U[n,t] = sum_m g[m,n] x[pix(m)+t] and
    V[n,t] = sum_m g[m,n] [pix(m)+t inside], dW1[n,c,t] = w_c U + b_c V, dw_c = sum W1[n,c,t] U, db_c = sum W1[n,c,t] V;
    U and V are ONE weight-gradient launch over the 2-channel image (x, 1)."""

    @staticmethod
    def forward(ctx, x, w_init, b_init, w1, scale, shift):
        if STEM_FOLDED and w1.shape[0] == 64 and x.shape[-1] == 1:
            y = ops.stem7x7(x, *_prep.stem_fold(w1, w_init, b_init), scale, shift)      # one kernel on the 1-channel image
        else:
            y0 = ops.init_conv(x, w_init.detach(), b_init.detach())
            y = ops.conv2d(y0, _prep.krsc(w1), 7, 7, 2, 3, scale=scale, shift=shift, act=ACT_RELU)
        ctx.save_for_backward(x, w_init, b_init, w1, scale, y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, w_init, b_init, w1, scale, y = ctx.saved_tensors
        g = gy.contiguous() if _premasked(y, gy) else ops.relu_bwd(gy.contiguous(), y)
        B, H, W, _ = x.shape
        if STEM_FOLDED and w1.shape[0] == 64 and x.shape[-1] == 1:
            U, V = ops.stem7x7_wgrad(x, g)                                # one MFMA kernel over the 1-channel image (stem.hip)
            U, V = U * scale.view(64, 1, 1), V * scale.view(64, 1, 1)
        else:
            x2 = torch.ones((B, H, W, 2), device=x.device, dtype=torch.float32)
            x2[..., 0] = x[..., 0]
            uv = torch.zeros((64, 7 * 7 * 2), device=x.device, dtype=torch.float32)
            ops.conv_wgrad(g.view(-1, 64), x2, uv, B=B, H=H, W=W, Cin=2, N=64, kh=7, kw=7, stride=2, pad=3, row_scale=scale)
            uv = uv.view(64, 7, 7, 2)
            U, V = uv[..., 0], uv[..., 1]                                 # [64,7,7]
        wi, bi = w_init.detach().view(3), b_init.detach().view(3)
        gw1 = wi.view(1, 3, 1, 1) * U[:, None] + bi.view(1, 3, 1, 1) * V[:, None]
        w1d = w1.detach()
        gwi = (w1d * U[:, None]).sum(dim=(0, 2, 3)).view_as(w_init)
        gbi = (w1d * V[:, None]).sum(dim=(0, 2, 3))
        return None, gwi, gbi, gw1, None, None


class MaxPool(Function):
    """`relu_input`: x is a ReLU output (the stem's): when the complete gradient is formed here it is handed on masked."""

    @staticmethod
    def forward(ctx, x, relu_input=False):
        y, idx = ops.maxpool3x3s2(x, with_index=True)
        ctx.hw = x.shape[1:3]
        ctx.x_ptr, ctx.take_x = x.data_ptr(), stash_accept(x, ctx.needs_input_grad[0])   # the stem output is also read by the finest FPN lateral
        ctx.premask = bool(relu_input and ctx.take_x)
        ctx.save_for_backward(idx, x if ctx.premask else None)       # x is alive anyway (its producer keeps it for its own mask)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        idx, x = ctx.saved_tensors
        other = _STASH.pop(ctx.x_ptr, None) if ctx.take_x else None     # the lateral's share of d/dx, added in the kernel
        premask = PREMASK and ctx.premask and other is not None
        gx = ops.maxpool3x3s2_bwd(idx, gy.contiguous(), *ctx.hw, residual=other, mask=x if premask else None)
        ondemand.zero_recycle(other)                  # the lateral's share may be a persistent map (ondemand.zero_acquire)
        if premask:
            _PREMASKED[ctx.x_ptr] = (gx.data_ptr(), gx._version)
        return gx, None


class SpaceToBatch2(Function):
    """[B,H,W,C] -> [4B,H/2,W/2,C] (`inverse`: back): the pixel permutation around the dilated ResNet blocks (`--dilation`); its
    gradient is the inverse permutation of the incoming gradient."""

    @staticmethod
    def forward(ctx, x, inverse):
        ctx.inverse = bool(inverse)
        return ops.space_to_batch2(x, inverse=ctx.inverse)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        return ops.space_to_batch2(gy.contiguous(), inverse=not ctx.inverse), None


class AvgPool2x2(Function):
    """nn.AdaptiveAvgPool2d to exactly half the size (reference layers.py:84,94 on the RPN map of the dilated level)."""

    @staticmethod
    def forward(ctx, x):
        return ops.avgpool2x2(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        return ops.avgpool2x2_bwd(gy.contiguous())


class UpsampleAdd(Function):
    @staticmethod
    def forward(ctx, src, add, Ho, Wo):
        ctx.hw = src.shape[1:3]
        ctx.has_add = add is not None
        y = ops.upsample_bilinear_add(src, Ho, Wo, add=add)
        if add is None and src.data_ptr() in _FPN_OUT:
            _FPN_UP[y.data_ptr()] = _FPN_OUT[src.data_ptr()]
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        gy = gy.contiguous()
        gs = ops.upsample_bilinear_bwd(gy, ctx.hw[0], ctx.hw[1]) if ctx.needs_input_grad[0] else None
        return gs, (gy if ctx.has_add and ctx.needs_input_grad[1] else None), None, None


class DwConv(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, mult, stride):
        y = ops.dwconv3x3(x, weight.detach(), bias.detach() if bias is not None else None, mult, stride)
        ctx.cfg = (mult, stride, bias is not None)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        mult, stride, has_bias = ctx.cfg
        gy = gy.contiguous()
        # The RPN's first convolution of a level and the RoI pooling read the same FPN map.  The RoI pooling's backward pass runs
        # first (later node) and leaves its scatter map in _GRAD_ACC: add this gradient into it on the few pixels a tap reaches
        # and return nothing -- instead of a dense write here, and a dense add by autograd (12.6 GB maps at level 0).  The
        # producer's backward pass waits for both consumers either way.
        if EARLY and ctx.needs_input_grad[0] and x.data_ptr() in _FPN_UP and GRAD_SHARE:
            # level P5: the RPN up-samples the map first (layers.py:35-37); the early pass does not reach that node (it lies on no
            # path to an RPN parameter), so its transpose is applied here and the result parked under the FPN map
            fm = _FPN_UP[x.data_ptr()]
            gx, gw, gb = ops.dwconv3x3_bwd(x, gy, weight.detach(), mult, stride, need_gx=True, need_gw=True, has_bias=has_bias)
            _PARKED[fm.data_ptr()] = (ops.upsample_bilinear_bwd(gx, fm.shape[1], fm.shape[2]), fm, None)
            return None, gw, gb, None, None
        if EARLY and ctx.needs_input_grad[0] and x.data_ptr() in _FPN_OUT and GRAD_SHARE:
            # early pass of the RPN branch: this gradient is the FIRST share of d/d(FPN map); park it for the RoI pooling's backward
            # pass (a persistent map for the demand-driven level, see ondemand.zero_acquire; the taps' 3x3 blocks are its footprint)
            st = ondemand.lazy_state(x)
            acc = e = None
            if (ondemand.ZERO_POOL and LAZY_DGRAD and LAZY_WGRAD and ondemand.CELL_BWD and st is not None and st.sparse and st.keep and
                    st.stride >= 5 and stride >= 3):
                acc, e = ondemand.zero_acquire(tuple(x.shape), x.device, ('map-grad', st.stride))
                if acc is not None:
                    ondemand.zero_note(e, lambda b_=acc, s_=tuple(x.shape), st_=st.stride: ops.zero_pattern(b_, st_))
            if acc is None and stride >= 3:
                acc = torch.zeros_like(x)
            if acc is not None:
                ops.dwconv3x3_bwd_acc(gy, weight.detach(), mult, stride, acc)
                _, gw, gb = ops.dwconv3x3_bwd(x, gy, weight.detach(), mult, stride, need_gx=False, need_gw=True, has_bias=has_bias)
            else:
                acc, gw, gb = ops.dwconv3x3_bwd(x, gy, weight.detach(), mult, stride, need_gx=True, need_gw=True, has_bias=has_bias)
            _PARKED[x.data_ptr()] = (acc, _FPN_OUT[x.data_ptr()], e)
            return None, gw, gb, None, None
        ref = _GRAD_ACC.pop(x.data_ptr(), None) if ctx.needs_input_grad[0] else None
        acc = ref() if ref is not None else None          # weak: alive only while autograd still holds the RoI pooling's map
        if acc is not None and acc.shape == x.shape:
            ops.dwconv3x3_bwd_acc(gy, weight.detach(), mult, stride, acc)
            _, gw, gb = ops.dwconv3x3_bwd(x, gy, weight.detach(), mult, stride, need_gx=False, need_gw=True, has_bias=has_bias)
            return None, gw, gb, None, None
        gx, gw, gb = ops.dwconv3x3_bwd(x, gy, weight.detach(), mult, stride,
                                       need_gx=ctx.needs_input_grad[0], need_gw=True, has_bias=has_bias)
        return gx, gw, gb, None, None


class RpnComposite(Function):
    """Training mode: the RPN's first block on a demand-driven FPN level -- depthwise 3x3 / stride S -> 1x1, in front of its BatchNorm
    (reference layers.py:22-29, 62-65) -- composed with the level's output convolution (fpn.py:137,145) in the cell domain
    (ondemand.train_composite_*): ONE GEMM over the transformed patches the forward pass keeps anyway; the map's pattern
    pixels, the 25-plane intermediate, `cell_output` / `cell_outgrad`, the depthwise and 1x1 passes do not exist.  `fm` is an input for
    the graph's sake only: its share of d/d(map) never exists in the pixel domain -- the data- and weight-gradient shares are left on
    the LazyMap for the convolution's own backward node, which runs after this one."""

    @staticmethod
    def forward(ctx, fm, dw_w, dw_b, pt_w, pt_b):
        st = ondemand.lazy_state(fm)
        f = ondemand.train_composite_forward(st, fm, dw_w, dw_b, pt_w, pt_b)
        ctx.st, ctx.fm_ptr, ctx.fm_shape = st, fm.data_ptr(), tuple(fm.shape)
        ctx.save_for_backward(dw_w, dw_b, pt_w, pt_b)
        return f

    @staticmethod
    @once_differentiable
    def backward(ctx, gf):
        dw_w, dw_b, pt_w, pt_b = ctx.saved_tensors
        st = ctx.st
        g_dw, g_dwb, g_pt, g_ptb = ondemand.train_composite_backward(st, gf.contiguous(), dw_w, dw_b, pt_w, pt_b)
        gfm = None
        if ctx.needs_input_grad[0]:
            # the convolution's backward node must run even if no RoI pooling sends it a gradient: hand it an all-zero share
            ptr = ctx.fm_ptr
            if EARLY and ptr in _FPN_OUT and GRAD_SHARE:
                acc = e = None
                if (ondemand.ZERO_POOL and LAZY_DGRAD and LAZY_WGRAD and ondemand.CELL_BWD and st.keep and
                        ((st.sparse and st.stride >= 5) or st.overlap)):
                    acc, e = ondemand.zero_acquire(ctx.fm_shape, gf.device, ('map-grad', st.stride))      # persistent zeros: no fill
                if acc is None:
                    acc = torch.zeros(ctx.fm_shape, device=gf.device, dtype=torch.float32)
                _PARKED[ptr] = (acc, _FPN_OUT[ptr], e)       # the RoI pooling's backward pass scatters into it; parked_flush otherwise
            else:
                ref = _GRAD_ACC.pop(ptr, None)
                if ref is None or ref() is None:             # no RoI pooling left a map for this level
                    gfm = torch.zeros(ctx.fm_shape, device=gf.device, dtype=torch.float32)
        return gfm, g_dw, g_dwb, g_pt, g_ptb


class Film(Function):
    """y = z * gamma + beta with film[..., :C] = gamma, film[..., C:] = beta (reference layers.py:42)."""

    @staticmethod
    def forward(ctx, z, film):
        ctx.save_for_backward(z, film)
        return ops.film_fwd(z, film)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        z, film = ctx.saved_tensors
        return ops.film_bwd(gy.contiguous(), z, film)


class Silu(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.silu(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        return ops.silu_bwd(gy.contiguous(), x)


class BatchNormTrain(Function):
    """nn.BatchNorm2d in training mode over NHWC rows; running statistics updated in place."""

    @staticmethod
    def forward(ctx, x, weight, bias, run_mean, run_var, eps, momentum):
        C_ = x.shape[-1]
        y, mean, invstd = ops.bn_train_fwd(x.view(-1, C_), weight.detach(), bias.detach(), eps, momentum, run_mean, run_var)
        _prep.bump()                       # running statistics changed behind torch's back
        ctx.save_for_backward(x, weight, mean, invstd)
        return y.view(x.shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight, mean, invstd = ctx.saved_tensors
        C_ = x.shape[-1]
        gx, gw, gb = ops.bn_train_bwd(gy.contiguous().view(-1, C_), x.view(-1, C_), mean, invstd, weight.detach())
        return gx.view(x.shape), gw, gb, None, None, None, None


class PairSoftmax(Function):
    @staticmethod
    def forward(ctx, x, n_anchor):
        y = ops.pair_softmax(x, n_anchor)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        return ops.pair_softmax_bwd(y, gy.contiguous()), None


class SoftmaxRows(Function):
    @staticmethod
    def forward(ctx, x2d):
        y = ops.softmax_rows_(x2d.clone())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        return ops.softmax_rows_bwd(y, gy.contiguous())


class RoiPool(Function):
    """ROIPooling (reference layers.py:406-497): gradient flows to the five FPN maps only."""

    @staticmethod
    def forward(ctx, rois, n_roi, pe_f, pe_t, img_h, img_w, *fmaps):
        pool, pe, level = ops.roi_pool(list(fmaps), rois, n_roi, pe_f, pe_t, img_h, img_w)
        ctx.shapes = [tuple(f.shape) for f in fmaps]
        ctx.fm_ptrs = [f.data_ptr() for f in fmaps]
        # demand-driven levels whose backward pass goes through the cell transforms: their gradient maps are persistent
        # (ondemand.zero_acquire); the map's producer (Conv.backward) recycles them
        ctx.pooled = {}
        if ondemand.ZERO_POOL and GRAD_SHARE and LAZY_DGRAD and LAZY_WGRAD and ondemand.CELL_BWD:
            for i, f in enumerate(fmaps):
                st = ondemand.lazy_state(f)
                # (an overlap level read by the composed RPN block, Fn.RpnComposite: nothing but the RoI windows is ever written to its
                # gradient map either -- 3.15 GB at B = 128 that a fresh torch.zeros filled every step)
                if st is not None and st.keep and ((st.sparse and st.stride >= 5) or (st.overlap and st.comp is not None)):
                    ctx.pooled[i] = st.stride
        _GRAD_ACC.clear()                      # nothing of an earlier step may survive into this one's backward pass
        ctx.save_for_backward(rois, level)
        ctx.mark_non_differentiable(pe, level)
        return pool, pe, level

    @staticmethod
    @once_differentiable
    def backward(ctx, gpool, _gpe, _glvl):
        rois, level = ctx.saved_tensors
        # the RPN branch's share, if its early backward pass parked one (train.step): scatter into it
        bases = {i: _PARKED.pop(ptr) for i, ptr in enumerate(ctx.fm_ptrs) if ptr in _PARKED}
        gf = ops.roi_pool_bwd(gpool.contiguous(), rois, level, ctx.shapes, pooled=ctx.pooled,
                              bases={i: (g, e) for i, (g, _, e) in bases.items()})
        _GRAD_ACC.clear()
        if GRAD_SHARE:                         # the other consumer of each map may add its gradient here (Fn.DwConv.backward)
            for i, (ptr, g) in enumerate(zip(ctx.fm_ptrs, gf)):
                if i not in bases:
                    _GRAD_ACC[ptr] = weakref.ref(g)
        return (None, None, None, None, None, None, *gf)


def attention_context(x, wq, bq, wk, bk, wv, bv, inv):
    """x [B, L, C] -> (qkv [B*L, 2d + dv], p [B, L, L] softmaxed, ctx [B*L, dv], wqkv): self_attention.py:40-50 up to (not including)
    the final projection.  dv = rows of wv (= d in the module; the evaluation path passes value weights composed with what follows)."""
    B, L, Cc = x.shape
    d, dv = wq.shape[0], wv.shape[0]
    ld = 2 * d + dv
    x2d = x.view(B * L, Cc)
    wqkv = torch.cat([wq.detach(), wk.detach(), wv.detach()], 0)
    bqkv = torch.cat([bq.detach(), bk.detach(), bv.detach()], 0)
    qkv = ops.linear(x2d, wqkv, bqkv)                                             # [B*L, 2d + dv]
    p = torch.empty((B, L, L), device=x.device, dtype=torch.float32)
    ops.gemm_conv(qkv, qkv[:, d:], p, B=1, H=L, W=1, Cin=d, N=L, x_ld=ld, w_ld=ld, groups=B,
                  x_gs=L * ld, w_gs=L * ld, y_gs=L * L, alpha=inv)
    ops.softmax_rows_(p.view(B * L, L))
    cx = torch.empty((B * L, dv), device=x.device, dtype=torch.float32)
    ops.conv_dgrad(p, qkv[:, 2 * d:], cx, B=1, H=L, W=1, Cin=dv, N=L, g_ld=L, w_ld=ld, out_ld=dv, groups=B,
                   g_gs=L * L, w_gs=L * ld, out_gs=L * dv)                         # ctx = P @ V
    return qkv, p, cx, wqkv


def _mat(a, b_t):
    """a [m, k] @ b_t[n, k]^T -> [m, n] through the library's own GEMM (small weight-sized products of the composed attention branch)."""
    return ops.linear(a.contiguous(), b_t.contiguous())


class AttnLateral(Function):
    """lateral(fm + SelfAttention(fm)) [+ bilinear(up)] as ONE tape node, composed: the module's final projection W_o and the
    FPN lateral W_l that reads the level are both linear, so the value projection is taken to W_v' = W_l W_o W_v (p = 384 rows instead
    of d = C / 2), P V' is already the lateral's image of the attention branch, and the level fm + ctx W_o^T + b_o (C wide) is never
    formed: y = W_l fm + (W_l b_o + b_l) + P V' + up.  Reference: self_attention.py:24-56,76 + fpn.py:143-144; the gradients of all
    eleven tensors follow by the chain rule through the two small weight products (hand-written: 3 + 6 small GEMMs)."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, wo, bo, wl, bl, up, inv):
        B, h, w, Cc = x.shape
        L, M = h * w, B * h * w
        d, p_ = wq.shape[0], wl.shape[0]
        wl2 = wl.detach().reshape(p_, Cc)
        wlo = _mat(wl2, wo.detach().t())                              # [p, d] = W_l W_o
        wvp = _mat(wlo, wv.detach().t())                              # [p, C] = W_l W_o W_v
        bvp = _mat(wlo, bv.detach()[None, :]).view(p_)                # W_l W_o b_v
        sh = torch.empty((p_, 1), device=x.device, dtype=torch.float32)
        ops.gemm_conv(wl2, bo.detach()[None, :].contiguous(), sh, B=1, H=p_, W=1, Cin=Cc, N=1, shift=bl.detach(), shift_per_row=True)
        sh = sh.view(p_)                                              # W_l b_o + b_l
        qkv, p, cx, wqkv = attention_context(x.view(B, L, Cc), wq, bq, wk, bk, wvp, bvp, inv)
        y = ops.conv2d(x, _prep.krsc(wl), shift=sh, residual=cx.view(B, h, w, p_), up=up)
        ctx.save_for_backward(x, qkv, p, wqkv, wlo, wl, wo, wv, bv, bo)
        ctx.inv, ctx.d = inv, d
        ctx.up_hw = tuple(up.shape[1:3]) if up is not None else None
        # gradient hand-over (see _STASH): d/dx to the backbone stage that reads the tap too, d/d(up) to the coarse map's output convolution
        ctx.stash_x = _stash_wanted(x)
        ctx.stash_up, ctx.up_ptr = _stash_wanted(up), (up.data_ptr() if up is not None else 0)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, qkv, p, wqkv, wlo, wl, wo, wv, bv, bo = ctx.saved_tensors
        inv, d = ctx.inv, ctx.d
        B, h, w, Cc = x.shape
        L, M = h * w, B * h * w
        p_ = wl.shape[0]
        ld = 2 * d + p_
        dev = x.device
        need = ctx.needs_input_grad
        g = gy.contiguous()
        g2 = g.view(M, p_)
        x2d = x.view(M, Cc)
        wl2 = wl.detach().reshape(p_, Cc)
        gup = None
        if ctx.up_hw is not None and need[11]:
            gup = ops.upsample_bilinear_bwd(g, *ctx.up_hw)
            if ctx.stash_up:
                _STASH[ctx.up_ptr] = gup
                arm_end_of_backward_check()
                gup = None
        # ---- the lateral's own term W_l fm + shift
        gsh = ops.colsum(g2)                                          # d/d(W_l b_o + b_l)
        gwl = torch.zeros((p_, Cc), device=dev, dtype=torch.float32)
        ops.conv_wgrad(g2, x2d, gwl, B=1, H=M, W=1, Cin=Cc, N=p_)      # g^T fm
        # ---- the attention branch: ctx' = P V' enters y through the residual input, so d/d(ctx') = g
        gp = torch.empty((B, L, L), device=dev, dtype=torch.float32)
        ops.gemm_conv(g2, qkv[:, 2 * d:], gp, B=1, H=L, W=1, Cin=p_, N=L, x_ld=p_, w_ld=ld, groups=B,
                      x_gs=L * p_, w_gs=L * ld, y_gs=L * L)            # dP = g V'^T
        gqkv = torch.zeros((M, ld), device=dev, dtype=torch.float32)
        ops.conv_wgrad(p, g2, gqkv[:, 2 * d:], B=1, H=L, W=1, Cin=p_, N=L, g_ld=L, x_ld=p_, out_ld=ld, groups=B,
                       g_gs=L * L, x_gs=L * p_, out_gs=L * ld)          # dV' = P^T g
        gs = ops.softmax_rows_bwd(p.view(M, L), gp.view(M, L), alpha=inv)
        ops.conv_dgrad(gs, qkv[:, d:], gqkv, B=1, H=L, W=1, Cin=d, N=L, g_ld=L, w_ld=ld, out_ld=ld, groups=B,
                       g_gs=L * L, w_gs=L * ld, out_gs=L * ld)          # dQ = dS K
        ops.conv_wgrad(gs, qkv, gqkv[:, d:], B=1, H=L, W=1, Cin=d, N=L, g_ld=L, x_ld=ld, out_ld=ld, groups=B,
                       g_gs=L * L, x_gs=L * ld, out_gs=L * ld)          # dK = dS^T Q
        gbqkv = ops.colsum(gqkv)
        gwqkv = torch.zeros((ld, Cc), device=dev, dtype=torch.float32)
        ops.conv_wgrad(gqkv, x2d, gwqkv, B=1, H=M, W=1, Cin=Cc, N=ld)
        gx = None
        if need[0]:
            gx_lat = torch.empty((M, Cc), device=dev, dtype=torch.float32)
            ops.conv_dgrad(g2, wl2, gx_lat, B=1, H=M, W=1, Cin=Cc, N=p_)                 # g W_l
            gx = torch.empty((M, Cc), device=dev, dtype=torch.float32)
            ops.conv_dgrad(gqkv, wqkv, gx, B=1, H=M, W=1, Cin=Cc, N=ld, residual=gx_lat)  # + d[Q | K | V'] W
            gx = gx.view(B, h, w, Cc)
            if ctx.stash_x:                           # the backbone stage that reads the tap adds this in its own kernel
                _STASH[x.data_ptr()] = gx
                arm_end_of_backward_check()
                gx = None
        # ---- chain rule through W_v' = W_lo W_v, b_v' = W_lo b_v, W_lo = W_l W_o, shift = W_l b_o + b_l
        gwvp, gbvp = gwqkv[2 * d:], gbqkv[2 * d:]
        gwlo = _mat(gwvp, wv.detach()) + gbvp[:, None] * bv.detach()[None, :]            # [p, d]
        gwv = _mat(wlo.t(), gwvp.t())                                                    # W_lo^T dW_v'  [d, C]
        gbv = _mat(wlo.t(), gbvp[None, :]).view(d)
        gbo = _mat(wl2.t(), gsh[None, :]).view(Cc)
        gwl = gwl + gsh[:, None] * bo.detach()[None, :] + _mat(gwlo, wo.detach())        # direct + through the shift + through W_lo
        gwo = _mat(wl2.t(), gwlo.t())                                                    # W_l^T dW_lo  [C, d]
        return (gx, gwqkv[:d], gbqkv[:d], gwqkv[d:2 * d], gbqkv[d:2 * d], gwv, gbv, gwo, gbo, gwl.view_as(wl), gsh, gup, None)


class Attention(Function):
    """fm + SelfAttention(fm) (reference self_attention.py:24-56,76) with a hand-written backward: 5 forward and
    9 backward fp32-MFMA GEMM launches (NT / NN / TN forms), row softmax and its gradient."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, wo, bo, inv):
        B, L, Cc = x.shape
        x2d = x.view(B * L, Cc)
        qkv, p, cx, wqkv = attention_context(x, wq, bq, wk, bk, wv, bv, inv)
        out = ops.linear(cx, wo.detach(), bo.detach(), residual=x2d)
        ctx.save_for_backward(x, qkv, p, cx, wqkv, wo)
        ctx.inv = inv
        return out.view(B, L, Cc)

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        x, qkv, p, cx, wqkv, wo = ctx.saved_tensors
        inv = ctx.inv
        B, L, Cc = x.shape
        d = wo.shape[1]
        M = B * L
        go = gout.contiguous().view(M, Cc)
        x2d = x.view(M, Cc)
        dev = x.device
        gbo = ops.colsum(go)
        gwo = torch.zeros((Cc, d), device=dev, dtype=torch.float32)
        ops.conv_wgrad(go, cx, gwo, B=1, H=M, W=1, Cin=d, N=Cc)                        # dWo = go^T cx
        gcx = torch.empty((M, d), device=dev, dtype=torch.float32)
        ops.conv_dgrad(go, wo, gcx, B=1, H=M, W=1, Cin=d, N=Cc)                        # dctx = go Wo
        gp = torch.empty((B, L, L), device=dev, dtype=torch.float32)
        ops.gemm_conv(gcx, qkv[:, 2 * d:], gp, B=1, H=L, W=1, Cin=d, N=L, x_ld=d, w_ld=3 * d, groups=B,
                      x_gs=L * d, w_gs=L * 3 * d, y_gs=L * L)                          # dP = dctx V^T
        gqkv = torch.zeros((M, 3 * d), device=dev, dtype=torch.float32)
        ops.conv_wgrad(p, gcx, gqkv[:, 2 * d:], B=1, H=L, W=1, Cin=d, N=L, g_ld=L, x_ld=d, out_ld=3 * d, groups=B,
                       g_gs=L * L, x_gs=L * d, out_gs=L * 3 * d)                        # dV = P^T dctx
        gs = ops.softmax_rows_bwd(p.view(M, L), gp.view(M, L), alpha=inv)              # dS (incl. 1/denominator)
        ops.conv_dgrad(gs, qkv[:, d:], gqkv, B=1, H=L, W=1, Cin=d, N=L, g_ld=L, w_ld=3 * d, out_ld=3 * d, groups=B,
                       g_gs=L * L, w_gs=L * 3 * d, out_gs=L * 3 * d)                    # dQ = dS K
        ops.conv_wgrad(gs, qkv, gqkv[:, d:], B=1, H=L, W=1, Cin=d, N=L, g_ld=L, x_ld=3 * d, out_ld=3 * d, groups=B,
                       g_gs=L * L, x_gs=L * 3 * d, out_gs=L * 3 * d)                    # dK = dS^T Q
        gbqkv = ops.colsum(gqkv)
        gwqkv = torch.zeros((3 * d, Cc), device=dev, dtype=torch.float32)
        ops.conv_wgrad(gqkv, x2d, gwqkv, B=1, H=M, W=1, Cin=Cc, N=3 * d)
        gx = torch.empty((M, Cc), device=dev, dtype=torch.float32)
        ops.conv_dgrad(gqkv, wqkv, gx, B=1, H=M, W=1, Cin=Cc, N=3 * d, residual=go)    # + residual branch
        return (gx.view(B, L, Cc), gwqkv[:d], gbqkv[:d], gwqkv[d:2 * d], gbqkv[d:2 * d], gwqkv[2 * d:], gbqkv[2 * d:],
                gwo, gbo, None)
