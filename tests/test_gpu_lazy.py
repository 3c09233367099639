"""GPU tests of the demand-driven finest FPN map (ondemand.conv3x3_winograd_lazy / lazy_complete, csrc/wino_fused.hip *_tiles,
csrc/detect.hip roi_tiles): the listed tiles are bit-identical to the dense convolution, the tile lists are exactly the
tiles the consumers read, nothing reads an unwritten pixel (NaN poison), and detections / losses / gradients do not change."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from birdsoundclassif_amd import ondemand, ops, synth                                  # noqa: E402
from birdsoundclassif_amd.nets import _prep, functional as Fn                # noqa: E402
from helpers import filler_state_dict                                        # noqa: E402


def rnd(key, *shape, scale=1.0):
    return torch.from_numpy((synth.normal(key, int(np.prod(shape))) * scale).astype(np.float32).reshape(shape))


def window(roi, fh, fw, n_levels=5):
    """numpy restatement of the RoI window (oracle nets_ref.roi_pooling / reference layers.py:408-462)."""
    size = np.float32(np.sqrt(np.float32((roi[2] - roi[0]) * (roi[3] - roi[1]))))
    with np.errstate(divide='ignore', invalid='ignore'):
        lf = np.float32(np.log(np.float32(size * np.float32(0.1)))) / np.float32(0.6931471805599453)
    lvl = int(np.clip(int(lf) if np.isfinite(lf) else -2 ** 31, 0, n_levels - 1))
    s = np.float32(2 << lvl)
    x1, y1, x2, y2 = (int(np.rint(np.float32(v) / s)) for v in roi)
    H, W = fh[lvl], fw[lvl]
    y2 = min(y2, H - 1)
    while y2 - y1 + 1 < 2:
        y1, y2 = max(0, y1 - 1), min(H - 1, y2 + 1)
    while x2 - x1 + 1 < 2:
        x1, x2 = max(0, x1 - 1), min(W - 1, x2 + 1)
    return lvl, x1, y1, min(x2, W - 1), y2


@pytest.mark.parametrize('shape', [(2, 24, 40, 128, 64, 8), (3, 47, 66, 128, 128, 8), (1, 188, 512, 384, 256, 8),
                                   (2, 47, 66, 128, 128, 4), (1, 94, 256, 384, 256, 4)])
def test_listed_tiles_equal_the_dense_convolution(shape):
    B, H, W, C, N, S = shape
    x = rnd(('lx', shape), B, H, W, C).cuda()
    w = rnd(('lw', shape), N, C, 3, 3, scale=0.05).cuda()
    b = rnd(('lb', shape), N).cuda()
    U = _prep.wino23(w)
    dense = ops.conv3x3_winograd(x, U, b)
    ondemand.LAZY_POISON = True
    try:
        y, st = ondemand.conv3x3_winograd_lazy(x, U, b, S)
    finally:
        ondemand.LAZY_POISON = False
    pat = ondemand.wino23_pattern(B, H, W, S, x.device)
    assert (0.15 < pat.frac < 0.45 if S == 8 else pat.frac > 0.9) and pat.n == int(pat.any.sum()) * B and pat.n_eff <= 0.8 * pat.n
    TH, TW = (H + 1) // 2, (W + 1) // 2
    # the pixels a 3x3 / stride S / pad 1 convolution reads: rows {So-1, So, So+1} x the same columns -- exactly these are
    # stored (a tile entered through one row / column / pixel stores just that), bit-identical to the dense convolution
    rows = torch.zeros(H, dtype=torch.bool)
    cols = torch.zeros(W, dtype=torch.bool)
    for n_, v in ((H, rows), (W, cols)):
        for o in range((n_ - 1) // S + 1):
            for k in range(3):
                if 0 <= S * o - 1 + k < n_:
                    v[S * o - 1 + k] = True
    m = (rows[:, None] & cols[None, :]).cuda()
    tile_any = pat.any.view(TH, TW).repeat_interleave(2, 0).repeat_interleave(2, 1)[:H, :W]
    assert bool((m <= tile_any).all())
    assert torch.equal(y[:, m], dense[:, m]), 'stored pixels differ from the dense convolution'
    assert bool(torch.isnan(y[:, ~m]).all()), 'a pixel outside the pattern was written'
    # RoI phase: random boxes, most of them small enough for level 0
    fh = [H, (H + 1) // 2, (H + 3) // 4, (H + 7) // 8, (H + 15) // 16]
    fw = [W, (W + 1) // 2, (W + 3) // 4, (W + 7) // 8, (W + 15) // 16]
    rng = np.random.default_rng(3)
    cap, n_roi = 16, 11
    rois = np.zeros((B, cap, 4), np.float32)
    for bi in range(B):
        for r in range(cap):
            x1, y1 = rng.integers(0, 2 * W - 3), rng.integers(0, 2 * H - 3)
            bw, bh = rng.integers(1, 30), rng.integers(1, 30)
            rois[bi, r] = (x1, y1, min(x1 + bw, 2 * W - 1), min(y1 + bh, 2 * H - 1))
    rois[0, 0] = (2 * W - 3, 0, 2 * W - 1, 5)                         # touches the right border
    rois_d = torch.from_numpy(rois).cuda()
    n_d = torch.tensor([n_roi], dtype=torch.int32, device='cuda')
    assert ondemand.lazy_pending(y)
    ondemand.lazy_complete(y, rois_d, n_d, list(zip(fh, fw)), level=0)
    assert ondemand.lazy_pending(y) and st.done == 1       # the state lives as long as the map: a later RoI set can be completed too
    want = m[None].repeat(B, 1, 1).clone().cpu()
    n_lvl0 = 0
    for bi in range(B):
        for r in range(n_roi):
            lvl, x1, y1, x2, y2 = window(rois[bi, r], fh, fw)
            if lvl == 0:
                n_lvl0 += 1
                want[bi, (y1 >> 1) * 2:(y2 >> 1) * 2 + 2, (x1 >> 1) * 2:(x2 >> 1) * 2 + 2] = True
    assert n_lvl0 > 0
    want = want.cuda()
    assert torch.equal(y[want], dense[want]), 'RoI tiles differ from the dense convolution'
    assert bool(torch.isnan(y[~want]).all()), 'a pixel outside pattern pixels + RoI tiles was written'


@pytest.mark.parametrize('shape', [(2, 24, 40, 128, 64, 8), (3, 47, 66, 128, 128, 8), (1, 188, 512, 384, 256, 8), (2, 45, 61, 64, 32, 6)])
def test_cell_forward_stores_exactly_the_pattern_pixels(shape):
    """Default forward of the pattern pixels: per stride x stride cell F(3x3,3x3) through the cell transforms (cellwino.hip,
    25 plane products per cell).  Exactly the pattern pixels are written (NaN poison survives everywhere else), within fp32
    rounding of the dense convolution (another summation order: not bit-identical, unlike the listed F(2x2,3x3) tiles above);
    the RoI phase afterwards stores its tiles bit-identical to the dense F(2x2,3x3) convolution."""
    B, H, W, C, N, S = shape
    x = rnd(('cx', shape), B, H, W, C).cuda()
    w = rnd(('cw', shape), N, C, 3, 3, scale=0.05).cuda()
    b = rnd(('cb', shape), N).cuda()
    U = _prep.wino23(w)
    dense = ops.conv3x3_winograd(x, U, b) if C >= 64 else ops.conv2d(x, _prep.krsc(w), 3, 3, 1, 1, shift=b)
    ondemand.LAZY_POISON = True
    try:
        y, st = ondemand.conv3x3_winograd_lazy(x, U, b, S, _prep.cell_weight(w, forward=True))
    finally:
        ondemand.LAZY_POISON = False
    rows = torch.zeros(H, dtype=torch.bool)
    cols = torch.zeros(W, dtype=torch.bool)
    for n_, v in ((H, rows), (W, cols)):
        for o in range((n_ - 1) // S + 1):
            for k in range(3):
                if 0 <= S * o - 1 + k < n_:
                    v[S * o - 1 + k] = True
    m = (rows[:, None] & cols[None, :]).cuda()
    assert bool(torch.isnan(y[:, ~m]).all()), 'a pixel outside the pattern was written'
    err = float((y[:, m] - dense[:, m]).abs().max())
    assert err < 1e-5 * max(1.0, float(dense.abs().max())), (err, float(dense.abs().max()))
    if C < 64:
        return
    fh = [H, (H + 1) // 2, (H + 3) // 4, (H + 7) // 8, (H + 15) // 16]
    fw = [W, (W + 1) // 2, (W + 3) // 4, (W + 7) // 8, (W + 15) // 16]
    rois = torch.tensor([[[10., 12., 25., 20.], [60., 40., 70., 66.], [0., 0., 8., 9.]]] * B).cuda()
    before = y.clone()
    ondemand.lazy_complete(y, rois, torch.tensor([3], dtype=torch.int32, device='cuda'), list(zip(fh, fw)))
    new = ~torch.isnan(y[..., 0]) & torch.isnan(before[..., 0])
    assert int(new.sum()) > 0 and torch.equal(y[new], dense[new])
    changed = (y != before).any(-1) & ~torch.isnan(before[..., 0])            # pattern pixels inside RoI tiles: stored again, whole
    assert torch.equal(y[changed], dense[changed])


@pytest.mark.parametrize('shape', [(2, 47, 66, 64, 384, 128, 8), (1, 188, 512, 64, 384, 256, 8), (2, 45, 61, 32, 96, 64, 6)])
def test_finest_level_without_its_merged_map(shape):
    """Deferred lateral (conv1x1_lazy(defer=True)) + its consumer: lateral 1x1 + bilinear merge + output 3x3 on the pattern pixels
    through [transform(up(x1) + b) | transform(t)] x [U | alpha U W_lat]^T -- the merged map never exists on the pattern patches.  The
    stored pixels are the dense chain's (lateral GEMM with merge epilogue, then the dense 3x3) within fp32 rounding, nothing else is
    written, and the RoI phase afterwards (its own lateral patches + listed F(2x2,3x3) tiles) is bit-identical to the dense chain."""
    B, H, W, Cin, C, N, S = shape
    t = rnd(('ft', shape), B, H, W, Cin).cuda()
    wl = rnd(('fwl', shape), C, Cin, 1, 1, scale=0.1).cuda()
    bl = rnd(('fbl', shape), C).cuda()
    up = rnd(('fup', shape), B, (H + 1) // 2, (W + 1) // 2, C).cuda()
    wo = rnd(('fwo', shape), N, C, 3, 3, scale=0.05).cuda()
    bo = rnd(('fbo', shape), N).cuda()
    alpha = 2.0
    merged = ops.conv2d(t, _prep.krsc(wl), shift=bl, alpha=alpha, up=up)
    dense = ops.conv3x3_winograd(merged, _prep.wino23(wo), bo)
    ondemand.LAZY_POISON = True
    try:
        x = ondemand.conv1x1_lazy(t, _prep.krsc(wl), bl, alpha, up, S, defer=True)
        y, st = ondemand.conv3x3_winograd_lazy(x, _prep.wino23(wo), bo, S, _prep.cell_weight(wo, forward=True),
                                               fold=lambda wk, a, transposed=False: _prep.cell_weight_folded(wo, wk, a, transposed))
    finally:
        ondemand.LAZY_POISON = False
    assert st.lateral is not None and st.lateral.deferred and bool(torch.isnan(x).all())
    ondemand.LAZY_POISON = True
    try:
        # a consumer that cannot take the operands (no cell weights) runs the lateral's pattern pass itself
        x2 = ondemand.conv1x1_lazy(t, _prep.krsc(wl), bl, alpha, up, S, defer=True)
        y2, st2 = ondemand.conv3x3_winograd_lazy(x2, _prep.wino23(wo), bo, S)
        assert not st2.lateral.deferred and not bool(torch.isnan(x2).all())
        w2 = ~torch.isnan(y2[..., 0])
        assert torch.equal(y2[w2], dense[w2])
    finally:
        ondemand.LAZY_POISON = False
    rows = torch.zeros(H, dtype=torch.bool)
    cols = torch.zeros(W, dtype=torch.bool)
    for n_, v in ((H, rows), (W, cols)):
        for o in range((n_ - 1) // S + 1):
            for k in range(3):
                if 0 <= S * o - 1 + k < n_:
                    v[S * o - 1 + k] = True
    m = (rows[:, None] & cols[None, :]).cuda()
    assert bool(torch.isnan(y[:, ~m]).all()), 'a pixel outside the pattern was written'
    err = float((y[:, m] - dense[:, m]).abs().max())
    assert err < 2e-5 * max(1.0, float(dense.abs().max())), (err, float(dense.abs().max()))
    assert bool(torch.isnan(st.x).all())                                     # the merged map: still untouched
    fh = [H, (H + 1) // 2, (H + 3) // 4, (H + 7) // 8, (H + 15) // 16]
    fw = [W, (W + 1) // 2, (W + 3) // 4, (W + 7) // 8, (W + 15) // 16]
    rois = torch.tensor([[[10., 12., 25., 20.], [60., 40., 70., 66.], [0., 0., 8., 9.]]] * B).cuda()
    before = y.clone()
    ondemand.lazy_complete(y, rois, torch.tensor([3], dtype=torch.int32, device='cuda'), list(zip(fh, fw)))
    new = ~torch.isnan(y[..., 0]) & torch.isnan(before[..., 0])
    assert int(new.sum()) > 0 and torch.equal(y[new], dense[new])
    xm = ~torch.isnan(st.x[..., 0])
    assert int(xm.sum()) > 0 and torch.equal(st.x[xm], merged[xm])


def test_many_rois_at_the_real_geometry():
    """1000 RoIs per image (the negative training step's load) on the 188x512 map: every pixel of every level-0 window equals the
    dense convolution, nothing outside pattern pixels + window tiles is written, and the dilated list covers the data gradient."""
    B, H, W, C, N = 2, 188, 512, 128, 64
    x = rnd('mx', B, H, W, C).cuda()
    w = rnd('mw', N, C, 3, 3, scale=0.05).cuda()
    b = rnd('mb', N).cuda()
    U = _prep.wino23(w)
    dense = ops.conv3x3_winograd(x, U, b)
    ondemand.LAZY_POISON = True
    try:
        y, st = ondemand.conv3x3_winograd_lazy(x, U, b, 8)
    finally:
        ondemand.LAZY_POISON = False
    st.keep = True
    fh = [188, 94, 47, 24, 12]
    fw = [512, 256, 128, 64, 32]
    rng = np.random.default_rng(5)
    cap = 1000
    x1 = rng.integers(0, 1000, (B, cap)); y1 = rng.integers(0, 360, (B, cap))
    bw = rng.integers(1, 60, (B, cap)); bh = rng.integers(1, 40, (B, cap))
    rois = np.stack([x1, y1, np.minimum(x1 + bw, 1023), np.minimum(y1 + bh, 374)], -1).astype(np.float32)
    rois_d = torch.from_numpy(rois).cuda()
    ondemand.lazy_complete(y, rois_d, torch.tensor([cap], dtype=torch.int32, device='cuda'), list(zip(fh, fw)))
    pat = ondemand.wino23_pattern(B, H, W, 8, x.device)
    rows = torch.zeros(H, dtype=torch.bool); cols = torch.zeros(W, dtype=torch.bool)
    for n_, v in ((H, rows), (W, cols)):
        for o in range((n_ - 1) // 8 + 1):
            for k in range(3):
                if 0 <= 8 * o - 1 + k < n_:
                    v[8 * o - 1 + k] = True
    want = (rows[:, None] & cols[None, :])[None].repeat(B, 1, 1).clone()
    win = torch.zeros(B, H, W, dtype=torch.bool)
    n0 = 0
    for bi in range(B):
        for r in range(cap):
            lvl, a1, b1, a2, b2 = window(rois[bi, r], fh, fw)
            if lvl == 0:
                n0 += 1
                want[bi, (b1 >> 1) * 2:(b2 >> 1) * 2 + 2, (a1 >> 1) * 2:(a2 >> 1) * 2 + 2] = True
                win[bi, b1:b2 + 1, a1:a2 + 1] = True
    assert n0 > 100
    want = want[:, :H, :W].cuda()
    assert torch.equal(y[want], dense[want]) and bool(torch.isnan(y[~want]).all())
    # data gradient: g lives on the pattern pixels and inside the windows
    m = ((rows[:, None] & cols[None, :])[None] | win).cuda()
    g = (rnd('mg', B, H, W, N).cuda() * m[..., None]).contiguous()
    got = ondemand.conv3x3_winograd_dgrad_tiles(st, g, _prep.wino23(w, transposed=True, m=2))      # all through the listed F(2x2,3x3)
    ref = ops.conv3x3_winograd(g, _prep.wino23(w, transposed=True, m=2), None)        # the dense operator, same F(2x2,3x3)
    assert torch.equal(got, ref)
    # pattern share through the cell transforms (Toom-Cook, other rounding): same support, values within fp32 noise
    got2 = ondemand.conv3x3_winograd_dgrad_tiles(st, g, _prep.wino23(w, transposed=True, m=2), _prep.cell_weight(w))
    assert torch.equal(got2 == 0, ref == 0) or bool(((got2 != 0) <= (ref != 0)).all())
    assert float((got2 - ref).abs().max()) < 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize('cell', [True, False])
def test_weight_gradient_over_the_listed_tiles_equals_the_dense_one(cell):
    """g zero outside the computed tiles: the weight gradient over the lists == torch's conv weight gradient.  cell: the pattern
    pixels through the cell transforms (cellwino.hip), the rest of the RoI tiles through F(2x2,3x3) with the pattern pixels masked;
    otherwise everything through the listed F(2x2,3x3) with plane masks."""
    import torch.nn.functional as F
    B, H, W, C, N = 2, 47, 66, 128, 64
    x = rnd('wx', B, H, W, C).cuda()
    w = rnd('ww', N, C, 3, 3, scale=0.05).cuda()
    b = rnd('wb', N).cuda()
    ondemand.LAZY_POISON = True
    try:
        y, st = ondemand.conv3x3_winograd_lazy(x, _prep.wino23(w), b, 8)
    finally:
        ondemand.LAZY_POISON = False
    st.keep = True
    fh = [H, (H + 1) // 2, (H + 3) // 4, (H + 7) // 8, (H + 15) // 16]
    fw = [W, (W + 1) // 2, (W + 3) // 4, (W + 7) // 8, (W + 15) // 16]
    rois = torch.tensor([[[10., 12., 25., 20.], [60., 40., 70., 66.], [0., 0., 8., 9.]]] * B).cuda()
    ondemand.lazy_complete(y, rois, torch.tensor([3], dtype=torch.int32, device='cuda'), list(zip(fh, fw)))
    # pixels with a reader: the pattern pixels + every pixel of the tiles under the RoIs (re-derived from the kept RoI list)
    TH, TW = (H + 1) // 2, (W + 1) // 2
    m = ~torch.isnan(y[..., 0])                      # the map was NaN-poisoned: written == has a reader
    tiles, host, ev = st.roi[0][0][:3]
    ev.synchronize()
    ids = tiles[:int(host.item()) * 128]
    assert int((ids >= 0).sum()) > 0
    g = rnd('wg', B, H, W, N).cuda() * m[..., None]
    dU, gb, dUc = ondemand.conv3x3_winograd_wgrad_tiles(st, x, g.contiguous(), want_bias=True, cell=cell)
    assert (dUc is not None) == cell
    gw = _prep.wino23_weight_grad(dU, 2) + (_prep.cell_weight_grad(dUc) if cell else 0)
    xr = x.permute(0, 3, 1, 2).double().cpu().requires_grad_(False)
    wr = w.double().cpu().requires_grad_(True)
    F.conv2d(xr, wr, None, 1, 1).backward(g.permute(0, 3, 1, 2).double().cpu())
    ref = wr.grad.float()
    err = (gw.cpu() - ref).abs().max().item()
    assert err < 2e-4 * ref.abs().max().item() + 1e-5, (err, ref.abs().max().item())
    assert torch.allclose(gb.cpu(), g.sum((0, 1, 2)).cpu(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('cell', [True, False])
def test_data_gradient_over_the_listed_tiles_equals_the_dense_one(cell):
    """g non-zero on the pattern pixels and inside the level-0 RoI windows only: the data gradient (cell: pattern share through
    the cell transforms + listed F(2x2,3x3) around the RoI windows; otherwise all listed F(2x2,3x3)) == torch's conv2d input
    gradient everywhere (zeros included)."""
    import torch.nn.functional as F
    B, H, W, C, N = 2, 47, 66, 128, 64
    x = rnd('gx', B, H, W, C).cuda()
    w = rnd('gw', N, C, 3, 3, scale=0.05).cuda()
    y, st = ondemand.conv3x3_winograd_lazy(x, _prep.wino23(w), None, 8)
    st.keep = True
    fh = [H, (H + 1) // 2, (H + 3) // 4, (H + 7) // 8, (H + 15) // 16]
    fw = [W, (W + 1) // 2, (W + 3) // 4, (W + 7) // 8, (W + 15) // 16]
    rois_np = np.array([[[10., 12., 25., 20.], [60., 40., 70., 66.], [0., 0., 8., 9.], [120., 86., 131., 93.], [40., 40., 90., 90.]]] * B,
                       dtype=np.float32)
    rois = torch.from_numpy(rois_np).cuda()
    ondemand.lazy_complete(y, rois, torch.tensor([5], dtype=torch.int32, device='cuda'), list(zip(fh, fw)))
    m = torch.zeros(B, H, W, dtype=torch.bool)
    rows = torch.zeros(H, dtype=torch.bool)
    cols = torch.zeros(W, dtype=torch.bool)
    for n_, v in ((H, rows), (W, cols)):
        for o in range((n_ - 1) // 8 + 1):
            for k in range(3):
                if 0 <= 8 * o - 1 + k < n_:
                    v[8 * o - 1 + k] = True
    m |= (rows[:, None] & cols[None, :])[None]
    n0 = 0
    for bi in range(B):
        for r in range(5):
            lvl, x1, y1, x2, y2 = window(rois_np[bi, r], fh, fw)
            if lvl == 0:
                n0 += 1
                m[bi, y1:y2 + 1, x1:x2 + 1] = True
    assert n0 >= 4 * B
    g = (rnd('gg', B, H, W, N) * m[..., None]).cuda().contiguous()
    got = ondemand.conv3x3_winograd_dgrad_tiles(st, g, _prep.wino23(w, transposed=True, m=2), _prep.cell_weight(w) if cell else None)
    xr = x.permute(0, 3, 1, 2).double().cpu().requires_grad_(True)
    F.conv2d(xr, w.double().cpu(), None, 1, 1).backward(g.permute(0, 3, 1, 2).double().cpu())
    ref = xr.grad.permute(0, 2, 3, 1).float()
    err = (got.cpu() - ref).abs().max().item()
    assert err < 2e-5 * ref.abs().max().item() + 1e-6, (err, ref.abs().max().item())
    assert bool((got.cpu()[(ref == 0).all(-1)] == 0).all())          # exact zeros where no gradient can arrive


@pytest.mark.parametrize('with_base', [False, True])
def test_overlap_level_backward_equals_the_dense_one(with_base):
    """Stride 4 (FPN level P2): every tile holds a pattern pixel, so the level is not `sparse`, but its gradient still lives on the
    3x3 blocks of the 4x4 cells and in the RoI windows.  `overlap` backward: data gradient = the 5x5 cell patches ADDED class by
    class (they overlap) onto `base` (another consumer's gradient, taken over in place) + the RoI share from the listed kernel
    with the pattern pixels masked, accumulated; weight gradient = cells + masked RoI tiles.  Both == torch."""
    import torch.nn.functional as F
    B, H, W, C, N, S = 2, 47, 66, 128, 64, 4
    x = rnd('ox', B, H, W, C).cuda()
    w = rnd('ow', N, C, 3, 3, scale=0.05).cuda()
    b = rnd('ob', N).cuda()
    y, st = ondemand.conv3x3_winograd_lazy(x, _prep.wino23(w), b, S, Ucell=_prep.cell_weight(w, forward=True), keep=True)
    assert st.overlap and not st.sparse and ondemand.listed_backward(st)
    fh = [2 * H, H, (H + 1) // 2, (H + 3) // 4, (H + 7) // 8]
    fw = [2 * W, W, (W + 1) // 2, (W + 3) // 4, (W + 7) // 8]
    rois_np = np.array([[[10., 12., 40., 36.], [100., 60., 131., 90.], [0., 0., 30., 22.], [200., 150., 236., 180.], [40., 40., 52., 50.],
                         [60., 20., 160., 140.]]] * B, dtype=np.float32)
    rois_np[1, :, [0, 2]] += 7
    rois = torch.from_numpy(rois_np).cuda()
    ondemand.lazy_complete(y, rois, torch.tensor([6], dtype=torch.int32, device='cuda'), list(zip(fh, fw)), level=1)
    assert len(st.rois) == 1
    m = torch.zeros(B, H, W, dtype=torch.bool)
    rows, cols = torch.zeros(H, dtype=torch.bool), torch.zeros(W, dtype=torch.bool)
    for n_, v in ((H, rows), (W, cols)):
        for o in range((n_ - 1) // S + 1):
            for k in range(3):
                if 0 <= S * o - 1 + k < n_:
                    v[S * o - 1 + k] = True
    m |= (rows[:, None] & cols[None, :])[None]
    n1 = 0
    for bi in range(B):
        for r in range(rois_np.shape[1]):
            lvl, x1, y1, x2, y2 = window(rois_np[bi, r], fh, fw)
            if lvl == 1:
                n1 += 1
                m[bi, y1:y2 + 1, x1:x2 + 1] = True
    assert n1 >= 3 * B and not bool(m.all())
    g = (rnd('og', B, H, W, N) * m[..., None]).cuda().contiguous()
    base = rnd('obase', B, H, W, C).cuda() if with_base else None
    base0 = base.clone() if with_base else None
    got = ondemand.conv3x3_winograd_dgrad_tiles(st, g, _prep.wino23(w, transposed=True, m=2), _prep.cell_weight(w), base=base)
    if with_base:
        assert got.data_ptr() == base.data_ptr()                          # taken over in place
    xr = x.permute(0, 3, 1, 2).double().cpu().requires_grad_(True)
    wr = w.double().cpu().requires_grad_(True)
    F.conv2d(xr, wr, None, 1, 1).backward(g.permute(0, 3, 1, 2).double().cpu())
    ref = xr.grad.permute(0, 2, 3, 1).float() + (base0.cpu() if with_base else 0)
    err = (got.cpu() - ref).abs().max().item()
    assert err < 2e-5 * ref.abs().max().item() + 1e-6, (err, ref.abs().max().item())
    dU, gb, dUc = ondemand.conv3x3_winograd_wgrad_tiles(st, x, g, want_bias=True)
    assert dUc is not None
    gw = _prep.wino23_weight_grad(dU, 2) + _prep.cell_weight_grad(dUc)
    errw = (gw.cpu() - wr.grad.float()).abs().max().item()
    assert errw < 2e-4 * wr.grad.abs().max().item() + 1e-5, (errw, wr.grad.abs().max().item())
    assert torch.allclose(gb.cpu(), g.sum((0, 1, 2)).cpu(), rtol=1e-4, atol=1e-4)


def test_lateral_on_listed_pixels_equals_the_dense_lateral():
    """ondemand.conv1x1_lazy (igemm ROWS variant, pixel list) and the RoI-phase patches (tile list x 16): the written pixels equal
    the dense lateral + merge bit for bit, nothing else is written."""
    B, H, W, Cin, N = 2, 47, 66, 64, 384
    t = rnd('lt', B, H, W, Cin).cuda()
    w = rnd('lw1', N, Cin, 1, 1, scale=0.1).cuda()
    b = rnd('lb1', N).cuda()
    up = rnd('lu', B, 24, 33, N).cuda()
    wk = _prep.krsc(w)
    dense = ops.conv2d(t, wk, shift=b, alpha=2.0, up=up)
    ondemand.LAZY_POISON = True
    try:
        x = ondemand.conv1x1_lazy(t, wk, b, 2.0, up, 8)
    finally:
        ondemand.LAZY_POISON = False
    pat = ondemand.wino23_pattern(B, H, W, 8, t.device)
    m = torch.zeros(B * H * W, dtype=torch.bool, device='cuda')
    m[pat.px_rows[pat.px_rows >= 0].long()] = True
    m = m.view(B, H, W)
    assert 0.3 < float(m.float().mean()) < 0.5
    assert torch.equal(x[m], dense[m]) and bool(torch.isnan(x[~m]).all())
    # the consumer: pattern tiles of the 3x3 convolution read only pixels that now exist
    w3 = rnd('lw3', 64, N, 3, 3, scale=0.05).cuda()
    b3 = rnd('lb3', 64).cuda()
    U = _prep.wino23(w3)
    ref = ops.conv3x3_winograd(dense, U, b3)
    ondemand.LAZY_POISON = True
    try:
        y, st = ondemand.conv3x3_winograd_lazy(x, U, b3, 8)
    finally:
        ondemand.LAZY_POISON = False
    assert st.lateral is not None
    written = ~torch.isnan(y[..., 0])
    assert int(written.sum()) > 0 and torch.equal(y[written], ref[written])
    # RoI phase: lateral patches of the RoI tiles, then the tiles
    fh = [H, (H + 1) // 2, (H + 3) // 4, (H + 7) // 8, (H + 15) // 16]
    fw = [W, (W + 1) // 2, (W + 3) // 4, (W + 7) // 8, (W + 15) // 16]
    rois = torch.tensor([[[10., 12., 25., 20.], [60., 40., 70., 66.], [0., 0., 8., 9.], [120., 80., 131., 93.]]] * B).cuda()
    ondemand.lazy_complete(y, rois, torch.tensor([4], dtype=torch.int32, device='cuda'), list(zip(fh, fw)))
    written2 = ~torch.isnan(y[..., 0])
    assert int(written2.sum()) > int(written.sum()) and torch.equal(y[written2], ref[written2])
    xm = ~torch.isnan(x[..., 0])
    assert torch.equal(x[xm], dense[xm]) and int(xm.sum()) > int(m.sum())


@pytest.fixture(scope='module')
def model():
    from birdsoundclassif_amd.nets import build_model
    from birdsoundclassif_amd.train import default_args
    m, _ = build_model(default_args(device='cuda'))
    m.load_state_dict(filler_state_dict())
    return m.cuda().eval()


@pytest.mark.parametrize('B', [2, 5])
def test_detections_do_not_change_and_no_unwritten_pixel_is_read(model, B):
    x = torch.from_numpy(synth.image_batch(0, B))[:, None].cuda()
    with torch.no_grad():
        ondemand.LAZY_FINEST = False
        try:
            det0, n0 = model.detect(x, min_score=0.1)
        finally:
            ondemand.LAZY_FINEST = True
        ondemand.LAZY_POISON = True
        try:
            det1, n1 = model.detect(x, min_score=0.1)
        finally:
            ondemand.LAZY_POISON = False
    assert int(n0.sum()) > 0
    # NaN anywhere in the consumed pixels would break this.  Classes and boxes identical; the scores see the pattern pixels of the
    # finest map through another summation order (cell transforms vs the dense F(2x2,3x3) tiles)
    assert torch.equal(n0, n1) and torch.equal(det0[..., :5], det1[..., :5])
    assert float((det0[..., 5] - det1[..., 5]).abs().max()) < 1e-5


@pytest.mark.parametrize('B,seed', [(2, 0), (5, 70), (3, 31)])
def test_rois_of_the_on_demand_path_equal_the_dense_path_bit_for_bit(model, B, seed):
    """The RPN reads FPN levels 0 / 1 of the on-demand path through the cell transforms (F(3x3,3x3): within 1e-5 of the dense map,
    DESIGN 4c).  Every RoI -- also the ones that never become a detection -- must nevertheless be the dense path's: same boxes, same
    order, same count; RPN outputs within 1e-5."""
    x = torch.from_numpy(synth.image_batch(seed, B))[:, None].cuda()
    with torch.no_grad():
        a = model.forward_first_stage(x)
        b = model.forward_first_stage(x, lazy=True)
    assert a['rois'].shape == b['rois'].shape and a['rois'].shape[1] > 0
    assert torch.equal(a['rois'], b['rois'])
    assert float((a['rpn_cls_scores'] - b['rpn_cls_scores']).abs().max()) < 1e-5
    assert float((a['rpn_bbox_reg'] - b['rpn_bbox_reg']).abs().max()) < 2e-5


def test_independent_detection_equals_one_image_per_call(model):
    """detect(..., independent=True) on a batch == detect() on each image alone, bit for bit (bulk inference: the reference CLI
    runs one file per model call).  Image 2 is blank: its proposal counts differ from its launch-mates'."""
    x = torch.from_numpy(synth.image_batch(0, 5))[:, None].cuda()
    x[2] = 0.0
    with torch.no_grad():
        det, n = model.detect(x, min_score=0.05, independent=True)
        for b in range(5):
            d1, n1 = model.detect(x[b:b + 1], min_score=0.05)
            assert int(n[b]) == int(n1[0]), (b, int(n[b]), int(n1[0]))
            assert torch.equal(det[b, :int(n1[0])], d1[0, :int(n1[0])]), b
    assert int(n.sum()) > 0


def _train_model():
    from birdsoundclassif_amd import train as T
    from birdsoundclassif_amd.nets import build_model
    args = T.default_args(device='cuda')
    m, _ = build_model(args)
    m.load_state_dict(filler_state_dict())
    return m.cuda().train()


def test_a_second_roi_pooling_on_a_held_map_sees_its_tiles():
    """The on-demand map stays completable for as long as it lives: two forward_second_stage calls with DIFFERENT RoIs on one
    train-mode lazy `fpn_out` both equal the dense path (values and the gradients of both calls summed), and
    forward_first_stage() hands out dense maps unless lazy=True is asked for (like the reference, nbm_model.py:39-54)."""
    m = _train_model()
    B = 2
    x = torch.from_numpy(synth.image_batch(0, B))[:, None].cuda()
    u = synth.uniform('rois2', 2 * B * 12 * 4).reshape(2, B, 12, 4)

    def boxes(v, big):
        x1, y1 = np.floor(v[..., 0] * 900), np.floor(v[..., 1] * 330)
        w, h = np.floor(4 + v[..., 2] * (200 if big else 18)), np.floor(4 + v[..., 3] * (40 if big else 14))
        return torch.tensor(np.stack([x1, y1, np.minimum(x1 + w, 1023), np.minimum(y1 + h, 374)], -1), dtype=torch.float32).cuda()
    r1, r2 = boxes(u[0], False), boxes(u[1], True)
    r2[:, :6] = boxes(u[1][:, :6] * 0.5 + 0.3, False)        # both sets hold level-0 RoIs, at different places
    res = {}
    for lazy in (False, True):
        m.zero_grad(set_to_none=True)
        ondemand.LAZY_POISON = lazy
        try:
            o = m.forward_first_stage(x, lazy=lazy)
            if lazy:
                assert bool(torch.isnan(o['fpn_out'][0]).any())              # holes: the map really is sparse here
            s1 = m.forward_second_stage(o['fpn_out'], r1, training=True)
            s2 = m.forward_second_stage(o['fpn_out'], r2, training=True)
        finally:
            ondemand.LAZY_POISON = False
        loss = (s1['bbox_classes'][:, 1:8].sum() + s1['bbox_reg'][:, :40].sum() * 0.1 +
                s2['bbox_classes'][:, 3:9].sum() * 2.0 + s2['bbox_reg'][:, 40:90].sum() * 0.05 + o['rpn_cls_scores'][:, :3].sum() * 0.01)
        loss.backward()
        torch.cuda.synchronize()
        res[lazy] = ([v.detach().clone() for v in (s1['bbox_reg'], s1['bbox_classes'], s2['bbox_reg'], s2['bbox_classes'])],
                     {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
    dflt = m.forward_first_stage(x)['fpn_out'][0]
    assert bool(torch.isfinite(dflt).all())                                   # default: dense
    for a, b in zip(res[False][0], res[True][0]):
        assert torch.isfinite(b).all() and torch.equal(a, b)
    assert set(res[False][1]) == set(res[True][1])
    for k, g in res[False][1].items():
        g2 = res[True][1][k]
        assert torch.isfinite(g2).all(), k
        tol = 2e-3 * float(g.abs().max()) + 1e-6          # dense path: F(4x4,3x3) gradients of out_convs.4; on-demand path: cell transforms
        assert float((g - g2).abs().max()) <= tol, (k, float((g - g2).abs().max()), tol)


@pytest.mark.parametrize('case', ['frozen_weight', 'wgrad_switch_off', 'dgrad_switch_off'])
def test_listed_backward_with_a_frozen_weight_or_a_switch_off(case):
    """the RoI tile lists must be recorded whenever ANY gradient can follow -- with fpn.out_convs.4.weight frozen the
    data gradient under the RoI windows used to be dropped -- and with one of the listed backward passes switched off the dense
    kernel that replaces it must not meet uninitialised memory in the holes of the sparse maps (0 x NaN)."""
    m = _train_model()
    B = 2
    x = torch.from_numpy(synth.image_batch(0, B))[:, None].cuda()
    u = synth.uniform('rois3', B * 12 * 4).reshape(B, 12, 4)
    x1, y1 = np.floor(u[..., 0] * 900), np.floor(u[..., 1] * 330)
    rois = torch.tensor(np.stack([x1, y1, np.minimum(x1 + np.floor(4 + u[..., 2] * 18), 1023),
                                  np.minimum(y1 + np.floor(4 + u[..., 3] * 14), 374)], -1), dtype=torch.float32).cuda()
    if case == 'frozen_weight':
        m.fpn.out_convs['4'].weight.requires_grad_(False)
    res = {}
    for lazy in (False, True):
        m.zero_grad(set_to_none=True)
        if lazy and case == 'wgrad_switch_off':
            Fn.LAZY_WGRAD = False
        if lazy and case == 'dgrad_switch_off':
            Fn.LAZY_DGRAD = False
        try:
            o = m.forward_first_stage(x, lazy=lazy)
            s = m.forward_second_stage(o['fpn_out'], rois, training=True)
            (s['bbox_classes'][:, 1:8].sum() + s['bbox_reg'][:, :40].sum() * 0.1).backward()
        finally:
            Fn.LAZY_WGRAD = Fn.LAZY_DGRAD = True
        torch.cuda.synchronize()
        res[lazy] = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    assert set(res[False]) == set(res[True]) and ('fpn.out_convs.4.weight' in res[True]) == (case != 'frozen_weight')
    assert 'fpn.pt_wise.0.weight' in res[True] and float(res[False]['fpn.pt_wise.0.weight'].abs().max()) > 0
    for k, g in res[False].items():
        g2 = res[True][k]
        assert torch.isfinite(g2).all(), k
        tol = 2e-3 * float(g.abs().max()) + 1e-6
        assert float((g - g2).abs().max()) <= tol, (k, float((g - g2).abs().max()), tol)


@pytest.mark.parametrize('negative', [False, True])
def test_train_step_losses_and_gradients_do_not_change(negative):
    """positive step (16 sampled RoIs per image) and negative step (all 1000 proposals through the head): on-demand maps
    NaN-poisoned vs the dense path."""
    from birdsoundclassif_amd import train as T
    from birdsoundclassif_amd.nets import build_model
    args = T.default_args(device='cuda')
    B = 2
    img = torch.from_numpy(synth.image_batch(0, B))
    bb, ids, lens = synth.label_batch(0, B)
    batch = [img, img, bb, ids, lens]
    res = {}
    for lazy in (False, True):
        model, crit = build_model(args)
        model.load_state_dict(filler_state_dict())
        model = model.cuda().train()
        crit.train()
        opt, _ = T.build_optimizer(model, args)
        np.random.seed(5)
        # the baseline also runs without the shared gradient buffer of the two consumers of an FPN map (Fn.DwConv.backward)
        ondemand.LAZY_FINEST, ondemand.LAZY_POISON, Fn.GRAD_SHARE = lazy, lazy, lazy
        try:
            loss = T.train_one_step(model, crit, opt, batch, args.clip_max_norm, 'cuda', negative_sample=negative)
        finally:
            ondemand.LAZY_FINEST, ondemand.LAZY_POISON, Fn.GRAD_SHARE = True, False, True
        torch.cuda.synchronize()
        res[lazy] = ({k: float(v) for k, v in loss.items()}, float(opt.grad_norm()),
                     {k: v.detach().clone() for k, v in model.state_dict().items()})
    for k, v in res[False][0].items():          # the pattern pixels go through the cell transforms: same value up to fp32 rounding
        assert abs(v - res[True][0][k]) <= 2e-6 * max(1.0, abs(v)), (k, v, res[True][0][k])
    assert np.isfinite(res[True][1]) and abs(res[False][1] - res[True][1]) <= 1e-5 * res[False][1]
    # post-AdamW weights: the weight gradients are summed with float atomics (order varies from run to run), and the first
    # AdamW step turns a gradient into lr * g / (|g| + eps), which amplifies that noise where g is tiny
    for k, v in res[False][2].items():
        assert torch.allclose(v, res[True][2][k], rtol=2e-4, atol=2e-6), k


def test_persistent_gradient_maps_are_clean_after_every_step():
    """Training keeps the two large, almost-empty gradient maps of the finest level across steps (ondemand.zero_acquire) and
    undoes the writers' footprints instead of refilling them.  Four steps (positive, positive with other boxes, negative,
    positive) with the check switch on -- after every recycle the maps are verified to be zero (outside the cell patches that
    every pass rewrites) -- and the gradient norms / losses of every step equal those of a run with fresh maps."""
    from birdsoundclassif_amd import train as T
    from birdsoundclassif_amd.nets import build_model
    args = T.default_args(device='cuda')
    B = 2
    batches = []
    for s in (0, 3):
        bb, ids, lens = synth.label_batch(s, B)
        img = torch.from_numpy(synth.image_batch(s, B))
        batches.append([img, img, bb, ids, lens])
    plan = [(0, False), (1, False), (0, True), (1, False)]
    res = {}
    for pool in (False, True):
        model, crit = build_model(args)
        model.load_state_dict(filler_state_dict())
        model = model.cuda().train()
        crit.train()
        opt, _ = T.build_optimizer(model, args)
        np.random.seed(11)
        ondemand.zero_pool_clear()
        ondemand.ZERO_POOL, ondemand.ZERO_POOL_CHECK = pool, pool
        out = []
        try:
            for bi, neg in plan:
                loss = T.train_one_step(model, crit, opt, batches[bi], args.clip_max_norm, 'cuda', negative_sample=neg)
                out.append(({k: float(v) for k, v in loss.items()}, float(opt.grad_norm())))
            if pool:
                tags = sorted(k[2][0] for k in ondemand._ZERO_POOL)
                # the three maps of level 0 came from the pool, and (round 5: the composed RPN block leaves nothing but the RoI windows in it)
                # the gradient of level 1's output map ...
                assert tags == ['cell-dgrad', 'lat-dt', 'map-grad', 'map-grad'], tags
                assert not any(e['busy'] for e in ondemand._ZERO_POOL.values())       # ... and were handed back by their readers
        finally:
            ondemand.ZERO_POOL, ondemand.ZERO_POOL_CHECK = True, False
            ondemand.zero_pool_clear()
        res[pool] = out
    # the switch's own guarantee is the check above (a dirty map raises); the numbers: step 0 strictly, the later steps loosely --
    # float atomics order the weight-gradient sums differently from run to run and AdamW's g / (|g| + eps) amplifies that
    # noise where g is tiny, so two runs drift apart by ~1e-5 per step with or without the pool
    for si, ((l0, n0), (l1, n1)) in enumerate(zip(res[False], res[True])):
        assert set(l0) == set(l1)
        tol = 5e-6 if si == 0 else 2e-3
        for k, v in l0.items():
            assert abs(v - l1[k]) <= tol * max(1.0, abs(v)), (si, k, v, l1[k])
        assert abs(n0 - n1) <= 4 * tol * n0, (si, n0, n1)


def test_lateral_gradients_from_the_cell_domain_equal_the_dense_passes():
    """ondemand.LAT_CELL_BWD: the finest lateral's own gradients (data, weight, bias) come out of its consumer's backward pass --
    pattern share in the cell-domain GEMMs (folded weights), RoI share on compact operands -- instead of three dense passes over the
    gradient of the merged map.  Every parameter gradient of a positive step equals the one of the dense passes.  Also with the
    variants of what the bilinear backward of the top-down merge reads (ondemand.UPBWD_SPLIT): the RoI share from its compact form
    + the pattern patches only (default); the share scattered into the map by the producer (split off); by the lateral's node
    (the fallback a dense reader of that gradient takes)."""
    from birdsoundclassif_amd import train as T
    from birdsoundclassif_amd.nets import build_model
    args = T.default_args(device='cuda')
    B = 2
    img = torch.from_numpy(synth.image_batch(0, B))
    bb, ids, lens = synth.label_batch(0, B)
    batch = [img, img, bb, ids, lens]
    res = {}
    for mode in ('dense', 'split', 'split-off', 'split-completed-by-the-reader'):
        model, crit = build_model(args)
        model.load_state_dict(filler_state_dict())
        model = model.cuda().train()
        crit.train()
        np.random.seed(5)
        ondemand.LAT_CELL_BWD = mode != 'dense'
        ondemand.UPBWD_SPLIT = mode != 'split-off'
        Fn.UPBWD_SPLIT_READ = mode != 'split-completed-by-the-reader'
        ondemand.ZERO_POOL_CHECK = True
        try:
            loss = T.step(model, crit, batch, 'cuda', False)
            sum(loss[k] * crit.weight_dict[k] for k in loss if k in crit.weight_dict).backward()
            Fn.stash_check_empty()
        finally:
            ondemand.LAT_CELL_BWD = ondemand.UPBWD_SPLIT = Fn.UPBWD_SPLIT_READ = True
            ondemand.ZERO_POOL_CHECK = False
            ondemand.zero_pool_clear()
        torch.cuda.synchronize()
        res[mode] = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    for mode in res:
        if mode == 'dense':
            continue
        assert set(res['dense']) == set(res[mode])
        for k, g in res['dense'].items():
            g2 = res[mode][k]
            assert torch.isfinite(g2).all(), (mode, k)
            tol = 1e-4 * float(g.abs().max()) + 1e-7
            err = float((g - g2).abs().max())
            assert err <= tol, (mode, k, err, tol)
    assert any('fpn' in k for k in res['split'])


@pytest.mark.parametrize('mode', ['early-backward', 'one-backward', 'chunked', 'negative-step', 'rpn-failed'])
def test_rpn_block_composed_with_the_output_convolution_in_training(mode, monkeypatch):
    """DESIGN 4h (round 5): in TRAINING mode the RPN's first block on the demand-driven levels 0 / 1 -- depthwise 3x3 / stride S -> 1x1,
    reference layers.py:22-29,62-65 -- composed with the level's output convolution (fpn.py:137,145) in the cell domain
    (Fn.RpnComposite): losses, BatchNorm buffers and EVERY parameter gradient of a step equal those of the uncomposed chain of tape
    nodes (cell transforms -> pattern pixels -> DwConv -> 1x1), border cells (top row / left column / corner: depthwise taps in the
    map's zero padding) included.  Variants: the RPN branch back-propagated early (train.step's split backward: the composed node
    parks an empty share for the RoI pooling) or with everything else; the batch cut into chunks of one image; a negative step
    (1000 RoIs per image, no early pass); a step that ends after the first stage (the parked share is flushed)."""
    from birdsoundclassif_amd import train as T
    from birdsoundclassif_amd.nets import build_model
    args = T.default_args(device='cuda', **(dict(min_threshold=5000) if mode == 'rpn-failed' else {}))    # no box survives: "RPN failed"
    B = 3
    img = torch.from_numpy(synth.image_batch(0, B))
    neg = torch.from_numpy(synth.image_batch(50, B))
    bb, ids, lens = synth.label_batch(0, B)
    batch = [img, neg, bb, ids, lens]
    if mode == 'chunked':
        monkeypatch.setattr(ops, 'WINO_CHUNK_BYTES', 700 << 20)         # level 0: one image per chunk
    if mode == 'one-backward':
        monkeypatch.setattr(T, 'SPLIT_BACKWARD', False)
    res = {}
    for composed in (False, True):
        model, crit = build_model(args)
        model.load_state_dict(filler_state_dict())
        model = model.cuda().train()
        crit.train()
        np.random.seed(5)
        ondemand.TRAIN_COMPOSITE = composed
        ondemand.ZERO_POOL_CHECK = True
        ondemand.TRAIN_COMPOSITE_CALLS[0] = 0
        try:
            model.zero_grad()
            loss = T.step(model, crit, batch, 'cuda', mode == 'negative-step', early_backward=True)
            tot = sum(loss[k] * crit.weight_dict[k] for k in loss if k in crit.weight_dict)
            if tot.requires_grad:
                tot.backward()
            Fn.parked_flush()
            Fn.stash_check_empty()
        finally:
            ondemand.TRAIN_COMPOSITE = True
            ondemand.ZERO_POOL_CHECK = False
            ondemand.zero_pool_clear()
        torch.cuda.synchronize()
        assert ondemand.TRAIN_COMPOSITE_CALLS[0] == (2 if composed else 0)
        res[composed] = dict(loss={k: float(v) for k, v in loss.items()},
                             grads={k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None},
                             bufs={k: v.detach().clone() for k, v in model.named_buffers() if 'rpn.convs' in k and 'running' in k})
    a, b = res[False], res[True]
    if mode == 'rpn-failed':
        assert not any(k.startswith('sec') for k in a['loss'])
    assert set(a['loss']) == set(b['loss'])
    for k in a['loss']:
        assert abs(a['loss'][k] - b['loss'][k]) <= 2e-5 * max(1.0, abs(a['loss'][k])), (k, a['loss'][k], b['loss'][k])
    assert set(a['grads']) == set(b['grads']) and any('rpn.convs.0.depth_wise' in k for k in a['grads'])
    for k, g in a['grads'].items():
        g2 = b['grads'][k]
        assert torch.isfinite(g2).all(), k
        # (the biases in front of the BatchNorm have a zero gradient: rounding noise of ~1e-7 on both sides)
        tol = 2e-4 * float(g.abs().max()) + 1e-6
        err = float((g - g2).abs().max())
        assert err <= tol, (mode, k, err, tol)
    for k, v in a['bufs'].items():
        assert float((v - b['bufs'][k]).abs().max()) <= 1e-5 * max(1.0, float(v.abs().max())), k


def test_cell_weight_kernels_equal_the_float64_einsum():
    """`nbm_cell_weight` / `nbm_cell_weight_fold` / `nbm_cell_weight_grad` (csrc/cellwino.hip: the kernel side of the cell transforms,
    float64 arithmetic, one rounding) against the float64 `torch.einsum` restatement they replaced on the training path: U = E w E^T in
    both operand layouts, the folded lateral [U | alpha U W_lat], and dW = E^T dU E -- equal after the one rounding to fp32 (up to an
    ulp where the float64 sums associate differently)."""
    from birdsoundclassif_amd import ops
    E = torch.tensor([[1.0, 0.0, 0.0], [1.0, 1.0, 1.0], [1.0, -1.0, 1.0], [1.0, 2.0, 4.0], [1.0, -2.0, 4.0]], dtype=torch.float64).cuda()
    N, C_, Cin, alpha = 96, 160, 64, 0.75
    w = (torch.from_numpy(synth.normal('cw', N * C_ * 9).astype(np.float32)).view(N, C_, 3, 3) * 0.05).cuda()
    wl = (torch.from_numpy(synth.normal('cwl', C_ * Cin).astype(np.float32)).view(C_, Cin) * 0.1).cuda()
    u = torch.einsum('ar,bs,ncrs->abnc', E, E, w.double()).reshape(25, N, C_)
    got_f, got_t = ops.cell_weight(w, forward=True), ops.cell_weight(w, forward=False)
    ulp = 1.2e-7 * float(u.abs().max())
    assert float((got_f - u.float()).abs().max()) <= ulp and torch.equal(got_t, got_f.transpose(1, 2).contiguous())
    uf = torch.cat([u, alpha * torch.einsum('knc,ci->kni', u, wl.double())], dim=-1)          # [25][N][C + Cin]
    f_nc, f_cn = ops.cell_weight(w, lateral=wl, alpha=alpha, both=True)
    assert tuple(f_nc.shape) == (25, N, C_ + Cin) and tuple(f_cn.shape) == (25, C_ + Cin, N)
    assert torch.equal(f_nc[:, :, :C_], got_f) and torch.equal(f_cn, f_nc.transpose(1, 2).contiguous())
    ref = uf[:, :, C_:].float()
    assert float((f_nc[:, :, C_:] - ref).abs().max()) <= 2e-7 * float(ref.abs().max())
    dU = torch.from_numpy(synth.normal('cdu', 25 * N * (C_ + Cin)).astype(np.float32)).view(25, N, C_ + Cin).cuda()
    ref_w = torch.einsum('at,bs,abnc->ncts', E, E, dU[:, :, :C_].double().reshape(5, 5, N, C_)).float()
    got_w = ops.cell_weight_grad(dU[:, :, :C_])                   # a column view of the wider operand (row pitch C + Cin)
    assert got_w.shape == ref_w.shape and float((got_w - ref_w).abs().max()) <= 2e-7 * float(ref_w.abs().max())
    got_w2 = ops.cell_weight_grad(dU[:, :, :C_].contiguous())
    assert torch.equal(got_w, got_w2)


@pytest.mark.parametrize('shape', [(2, 47, 66, 64, 384, 128, 8, True), (1, 188, 512, 64, 384, 256, 8, True), (2, 94, 128, 0, 384, 256, 4, False),
                                   (3, 45, 61, 32, 96, 64, 6, True), (2, 33, 41, 0, 64, 64, 4, False), (2, 24, 32, 0, 128, 128, 3, False)])
def test_rpn_block_composed_with_the_output_convolution(shape):
    """Evaluation mode (ondemand.rpn_composite): the RPN's reader of a demand-driven map -- depthwise 3x3 / stride S (multiplier 2) -> 1x1
    -> BatchNorm -> SiLU, reference layers.py:13-46 -- composed with the map's own 3x3 convolution (and the deferred lateral + merge in
    front of it) is one 5x5 / stride S convolution of the inputs.  Against float64 of the reference's chain of layers on EVERY cell (the
    top / left border classes, where a depthwise tap falls into the zero padding, included), and against the route through the map's
    pattern pixels (pattern_materialize + the block's own kernels): the composite must not be the less accurate one.  The map itself
    stays unwritten (NaN poison)."""
    import torch.nn.functional as F
    from birdsoundclassif_amd.nets.layers import DepthwiseSepConv2d
    B, H, W, Cin, C, N, S, deferred = shape
    wo = rnd(('cwo', shape), N, C, 3, 3, scale=0.05).cuda()
    bo = rnd(('cbo', shape), N).cuda()
    blk = DepthwiseSepConv2d(N, N, stride=S, expansion_fact=2).cuda().eval()
    with torch.no_grad():
        for k_, p_ in blk.named_parameters():
            p_.copy_(rnd(('cblk', k_, shape), *p_.shape, scale=0.3 if p_.dim() > 1 else 0.5).cuda())
        blk.norm.weight.add_(1.0)
        blk.norm.running_mean.copy_(rnd(('cbm', shape), N, scale=0.2).cuda())
        blk.norm.running_var.copy_(rnd(('cbv', shape), N).abs().cuda() + 0.5)
    if deferred:
        t = rnd(('ct', shape), B, H, W, Cin).cuda()
        wl = rnd(('cwl', shape), C, Cin, 1, 1, scale=0.1).cuda()
        bl = rnd(('cbl', shape), C).cuda()
        up = rnd(('cup', shape), B, (H + 1) // 2, (W + 1) // 2, C).cuda()
        alpha = 2.0
        merged = ops.conv2d(t, _prep.krsc(wl), shift=bl, alpha=alpha, up=up)
    else:
        merged = rnd(('cx', shape), B, H, W, C).cuda()

    def lazy_map(raw):
        ondemand.LAZY_POISON = True
        try:
            x = ondemand.conv1x1_lazy(t, _prep.krsc(wl), bl, alpha, up, S, defer=True) if deferred else merged
            with torch.no_grad():
                return ondemand.conv3x3_winograd_lazy(x, _prep.wino23(wo), bo, S, _prep.cell_weight(wo, forward=True),
                                                      fold=lambda wk, a, transposed=False: _prep.cell_weight_folded(wo, wk, a, transposed),
                                                      raw=raw)
        finally:
            ondemand.LAZY_POISON = False
    # float64 reference of the chain of layers
    md = merged.double().permute(0, 3, 1, 2)
    o = F.conv2d(md, wo.double(), bo.double(), padding=1)
    d = F.conv2d(o, blk.depth_wise.weight.double(), blk.depth_wise.bias.double(), stride=S, padding=1, groups=N)
    p = F.conv2d(d, blk.pt_wise.weight.double(), blk.pt_wise.bias.double())
    p = (p - blk.norm.running_mean.double()[None, :, None, None]) / torch.sqrt(blk.norm.running_var.double() + blk.norm.eps)[None, :, None, None] \
        * blk.norm.weight.double()[None, :, None, None] + blk.norm.bias.double()[None, :, None, None]
    ref = (p * torch.sigmoid(p)).permute(0, 2, 3, 1)
    with torch.no_grad():
        y, st = lazy_map((wo, bo))
        assert st.pending is not None and bool(torch.isnan(y).all())             # nothing computed yet
        f = blk(y)
        assert st.pending is not None and bool(torch.isnan(y).all()), 'the composite route wrote pattern pixels'
        y2, st2 = lazy_map(None)                                                # the route through the pattern pixels
        assert st2.pending is None
        f2 = blk(y2)
        y3, st3 = lazy_map((wo, bo))                                            # pending, then asked for by another reader
        ondemand.pattern_materialize(y3)
        assert st3.pending is None and torch.equal(torch.nan_to_num(y3), torch.nan_to_num(y2))
    assert tuple(f.shape) == tuple(ref.shape) == tuple(f2.shape)
    e1, e2 = (f.double() - ref).abs(), (f2.double() - ref).abs()
    scale_ = max(1.0, float(ref.abs().max()))
    assert float(e1.max()) <= 2e-5 * scale_, (float(e1.max()), scale_)
    assert float(e1[:, 0].max()) <= 2e-5 * scale_ and float(e1[:, :, 0].max()) <= 2e-5 * scale_          # border classes
    assert float((e1 ** 2).mean().sqrt()) <= 1.05 * float((e2 ** 2).mean().sqrt()) + 1e-9, \
        f'composite rms error {float((e1 ** 2).mean().sqrt()):.3e} vs pattern route {float((e2 ** 2).mean().sqrt()):.3e}'


def test_composed_reader_under_batch_chunking(model):
    """ondemand.rpn_composite walks the batch chunks of the map (ops.WINO_CHUNK_BYTES): several chunks give the one-chunk result bit for
    bit, and the route through the pattern pixels (NBM_RPN_COMPOSITE=0) the same RoIs."""
    B = 16
    x = torch.from_numpy(np.tile(synth.image_batch(0, 8), (B // 8, 1, 1))).cuda()[:, None]
    keep = (ops.WINO_CHUNK_BYTES, ondemand.COMPOSITE)
    res = {}
    try:
        for chunk_gb, comp in ((24, True), (2, True), (2, False)):
            ops.WINO_CHUNK_BYTES, ondemand.COMPOSITE = chunk_gb << 30, comp
            with torch.no_grad():
                o = model.forward_first_stage(x, lazy=True)
            res[(chunk_gb, comp)] = (o['rpn_cls_scores'].float().clone(), o['rois'].clone())
        assert ondemand.lazy_chunk(torch.empty(B, 188, 512, 384, device='meta')) < B          # the 2 GB setting did chunk level 0
    finally:
        ops.WINO_CHUNK_BYTES, ondemand.COMPOSITE = keep
    one, many, pattern = res[(24, True)], res[(2, True)], res[(2, False)]
    assert torch.equal(one[0], many[0]) and torch.equal(one[1], many[1])
    assert float((many[0] - pattern[0]).abs().max()) < 2e-5 and torch.equal(many[1], pattern[1])
