"""CPU: the algebra of the composed RPN reader -- `_prep.rpn_composite` / `rpn_composite_delta` / `ondemand._border_classes` --
against the reference's chain of layers in torch (fpn.py:137,145 out_conv 3x3 / pad 1 -> layers.py:22-29 depthwise 3x3 / stride S / pad 1,
channel multiplier 2 -> 1x1 -> BatchNorm with running statistics -> SiLU), float64, on every cell incl. the border classes; the deferred
lateral's fold ([up + b | t] x [W_eff | alpha W_eff W_lat]) too.  The GPU test (test_gpu_lazy.py) checks the launches; this one the weights."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from birdsoundclassif_amd import ondemand
from birdsoundclassif_amd.nets import _prep


def _rnd(seed, *shape, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g, dtype=torch.float64) * scale).float()


@pytest.mark.parametrize('geom', [(2, 21, 27, 8, 12, 16, 8, 4), (1, 17, 33, 0, 16, 8, 4, 2), (2, 14, 10, 4, 8, 8, 3, 2), (1, 24, 40, 0, 8, 16, 8, 2)])
def test_composed_weights_reproduce_the_chain_of_layers(geom):
    B, H, W, Cin, C, N, S, mult = geom
    out_w, out_b = _rnd(1, N, C, 3, 3, scale=0.2), _rnd(2, N)
    dw_w, dw_b = _rnd(3, mult * N, 1, 3, 3, scale=0.4), _rnd(4, mult * N)
    pt_w, pt_b = _rnd(5, N, mult * N, 1, 1, scale=0.3), _rnd(6, N)
    bn_w, bn_b, bn_m, bn_v = 1 + _rnd(7, N, scale=0.1), _rnd(8, N, scale=0.1), _rnd(9, N, scale=0.2), _rnd(10, N).abs() + 0.5
    eps = 1e-5
    if Cin:
        t, wl, bl, alpha = _rnd(11, B, Cin, H, W), _rnd(12, C, Cin, scale=0.3), _rnd(13, C), 2.0
        up = _rnd(14, B, C, H, W)                                   # (already interpolated: the composite sees up + b as data)
        x = alpha * torch.einsum('ck,bkhw->bchw', wl.double(), t.double()) + bl.double()[None, :, None, None] + up.double()
        operand = torch.cat([up.double() + bl.double()[None, :, None, None], t.double()], 1)      # [up + b | t]
    else:
        x = _rnd(15, B, C, H, W).double()
        operand, wl, alpha = x, None, 1.0
    o = F.conv2d(x, out_w.double(), out_b.double(), padding=1)
    d = F.conv2d(o, dw_w.double(), dw_b.double(), stride=S, padding=1, groups=N)
    p = F.conv2d(d, pt_w.double(), pt_b.double())
    scale, shift = _prep.bn_affine(bn_w, bn_b, bn_m, bn_v, eps, conv_bias=pt_b)
    z = p - pt_b.double()[None, :, None, None]                      # the affine folds the 1x1 bias in
    ref = z * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]
    ref = ref * torch.sigmoid(ref)
    K = operand.shape[1]
    wargs = (out_w, out_b, dw_w, dw_b, pt_w, scale, shift)
    wkw = dict(lat_wk=wl, alpha=alpha) if Cin else {}
    (w,), sc, sh = _prep.rpn_composite(*wargs, **wkw)
    assert tuple(w.shape) == (N, 25 * K)
    w5 = w.double().reshape(N, 5, 5, K).permute(0, 3, 1, 2)          # [N, K, 5, 5]
    pre = F.conv2d(operand, w5, stride=S, padding=2) * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]
    OH, OW = (H - 1) // S + 1, (W - 1) // S + 1
    assert tuple(pre.shape[2:]) == (OH, OW) == tuple(ref.shape[2:])
    # border classes: scale * (W_class - W_int) on the listed taps + the shift difference, added in front of the activation
    patches = F.unfold(operand, 5, padding=2, stride=S).reshape(B, K, 25, OH * OW)              # [B, K, tap, cell]
    flat = pre.reshape(B, N, OH * OW).clone()
    seen = 0
    for rmask, smask, taps, _, idx, pix in ondemand._border_classes(B, H, W, S, 'cpu'):
        dwt, dsh = _prep.rpn_composite_delta(*wargs, rmask, smask, taps, **wkw)
        cells = idx[: idx.numel() // B]                             # image 0's cells of the class (idx lists image after image)
        seen += cells.numel()
        if not taps:
            continue
        pv = patches[:, :, list(taps)][..., cells]                  # [B, K, nt, cells]
        dwt = dwt.double().reshape(N, len(taps), K)
        flat[:, :, cells] += torch.einsum('ntk,bktc->bnc', dwt, pv) + dsh.double()[None, :, None]
        # the gathered pixel indices are those of the patch taps (clamped where the tap lies in the padding, and masked)
        py, px = (pix[0][:, : cells.numel()] % (H * W)) // W, pix[0][:, : cells.numel()] % W
        for k_, tp in enumerate(taps):
            oy, ox = cells // OW, cells % OW
            ey, ex = S * oy - 2 + tp // 5, S * ox - 2 + tp % 5
            inside = (ey >= 0) & (ey < H) & (ex >= 0) & (ex < W)
            assert torch.equal(py[k_][inside], ey[inside]) and torch.equal(px[k_][inside], ex[inside])
            assert pix[1] is None or torch.equal(pix[1][k_, : cells.numel()], ~inside)
    got = flat.reshape(B, N, OH, OW)
    got = got * torch.sigmoid(got)
    border_cells = sum(1 for oy in range(OH) for ox in range(OW)
                       if any(not (0 <= S * oy - 1 + r < H) for r in range(3)) or any(not (0 <= S * ox - 1 + c < W) for c in range(3)))
    assert seen == border_cells > 0
    err = float((got - ref).abs().max())
    assert err < 5e-6 * max(1.0, float(ref.abs().max())), err      # (the composed weights are rounded to fp32 once)
    # without the border terms the interior matches and the border does not: the classes are needed and complete
    raw = pre * torch.sigmoid(pre)
    bad = (raw - ref).abs() > 1e-4 * max(1.0, float(ref.abs().max()))
    assert bool(bad.any()) and int(bad.any(1).any(0).sum()) <= border_cells
