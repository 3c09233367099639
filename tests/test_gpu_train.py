"""GPU parity tests of the training path: every backward kernel against torch autograd of the same op on the CPU
(fp32), then the whole optimisation step against the golden fixtures generated from the REAL reference
(two steps of reference train.py:205-217: one positive, one negative)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from birdsoundclassif_amd import ops, synth                     # noqa: E402
from birdsoundclassif_amd.nets import functional as Fn          # noqa: E402
from helpers import check_packed, filler_state_dict, load_golden  # noqa: E402
from oracle import nets_ref as O                                # noqa: E402


def rnd(key, *shape, scale=1.0):
    return torch.from_numpy((synth.normal(key, int(np.prod(shape))) * scale).astype(np.float32).reshape(shape))


def nhwc(t):
    return t.detach().permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.detach().cpu().permute(0, 3, 1, 2)


def close(got, ref, tol, name):
    got, ref = got.detach().cpu().float(), ref.detach().cpu().float()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    scale = float(ref.abs().max()) + 1e-12
    err = float((got - ref).abs().max())
    assert err <= tol * scale, f'{name}: max err {err:.3e} vs scale {scale:.3e}'


@pytest.mark.parametrize('cfg', [
    # B, H, W, Cin, Cout, k, stride, pad, relu, residual, alpha, bias
    (2, 12, 14, 64, 64, 1, 1, 0, True, True, 1.0, False),
    (1, 14, 18, 128, 128, 3, 1, 1, True, False, 1.0, False),
    (2, 15, 17, 128, 128, 3, 2, 1, True, False, 1.0, False),
    (2, 12, 14, 256, 512, 1, 2, 0, False, False, 1.0, False),
    (2, 16, 18, 64, 64, 3, 2, 1, True, False, 1.0, False),            # stride 2, even dims, 64-wide data-gradient tile
    (1, 13, 16, 64, 128, 1, 2, 0, False, False, 1.0, True),
    (3, 9, 150, 96, 64, 3, 2, 1, False, False, 1.0, True),            # parity classes spanning several M tiles
    (1, 11, 13, 384, 256, 3, 1, 1, False, False, 1.0, True),
    (2, 9, 10, 64, 384, 1, 1, 0, False, False, 2.0, True),
    (1, 9, 8, 256, 6, 1, 1, 0, False, False, 1.0, True),
    (1, 9, 8, 256, 12, 1, 1, 0, False, False, 1.0, True),
])
def test_conv_backward(cfg):
    B, H, W, Ci, Co, k, st, pad, relu, res, alpha, bias = cfg
    x = rnd(('x', cfg), B, Ci, H, W).requires_grad_(True)
    w = rnd(('w', cfg), Co, Ci, k, k, scale=(2.0 / (Ci * k * k)) ** 0.5).requires_grad_(True)
    b = rnd(('b', cfg), Co, scale=0.1).requires_grad_(True) if bias else None
    scale = None if bias else (1 + 0.1 * rnd(('s', cfg), Co))
    shift = None if bias else 0.1 * rnd(('sh', cfg), Co)
    y = F.conv2d(alpha * x, w, stride=st, padding=pad)
    y = y + b.view(1, -1, 1, 1) if bias else y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    r = rnd(('r', cfg), *y.shape).requires_grad_(True) if res else None
    if res:
        y = y + r
    if relu:
        y = F.relu(y)
    gy = rnd(('g', cfg), *y.shape)
    y.backward(gy)

    xd = nhwc(x).requires_grad_(True)
    wd = w.detach().cuda().requires_grad_(True)
    bd = b.detach().cuda().requires_grad_(True) if bias else None
    rd = nhwc(r).requires_grad_(True) if res else None
    yd = Fn.conv(xd, wd, bias=bd, scale=None if bias else scale.cuda(), shift=None if bias else shift.cuda(), residual=rd,
                 kh=k, kw=k, stride=st, pad=pad, act=ops.ACT_RELU if relu else ops.ACT_NONE, alpha=alpha)
    close(nchw(yd), y, 3e-5, f'fwd {cfg}')
    yd.backward(nhwc(gy))
    close(nchw(xd.grad), x.grad, 5e-5, f'dgrad {cfg}')
    close(wd.grad, w.grad, 5e-5, f'wgrad {cfg}')
    if bias:
        close(bd.grad, b.grad, 5e-5, f'bias grad {cfg}')
    if res:
        close(nchw(rd.grad), r.grad, 1e-6, f'residual grad {cfg}')


def test_linear_backward_odd_sizes():
    x = rnd('lx', 70, 1024).requires_grad_(True)
    w = rnd('lw', 151, 1024, scale=0.05).requires_grad_(True)
    b = rnd('lb', 151).requires_grad_(True)
    y = F.linear(x, w, b)
    gy = rnd('lg', 70, 151)
    y.backward(gy)
    xd, wd, bd = (t.detach().cuda().requires_grad_(True) for t in (x, w, b))
    yd = Fn.linear(xd, wd, bd)
    close(yd, y, 3e-5, 'linear fwd')
    yd.backward(gy.cuda())
    close(xd.grad, x.grad, 5e-5, 'linear dx')
    close(wd.grad, w.grad, 5e-5, 'linear dw')
    close(bd.grad, b.grad, 5e-5, 'linear db')


def test_stem_backward():
    x = rnd('stx', 2, 1, 37, 41)
    wi = (1 + 0.2 * rnd('stwi', 3, 1, 1, 1)).requires_grad_(True)
    bi = rnd('stbi', 3, scale=0.1).requires_grad_(True)
    w1 = rnd('stw1', 64, 3, 7, 7, scale=0.1).requires_grad_(True)
    scale, shift = 1 + 0.1 * rnd('sts', 64), 0.1 * rnd('stsh', 64)
    y = F.relu(F.conv2d(F.conv2d(x, wi, bi), w1, stride=2, padding=3) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    gy = rnd('stg', *y.shape)
    y.backward(gy)
    wid, bid, w1d = (t.detach().cuda().requires_grad_(True) for t in (wi, bi, w1))
    yd = Fn.Stem.apply(nhwc(x), wid, bid, w1d, scale.cuda(), shift.cuda())
    close(nchw(yd), y, 3e-5, 'stem fwd')
    yd.backward(nhwc(gy))
    close(w1d.grad, w1.grad, 5e-5, 'stem dW1')
    close(wid.grad, wi.grad, 5e-5, 'stem dw_init')
    close(bid.grad, bi.grad, 5e-5, 'stem db_init')


def test_attention_backward():
    from birdsoundclassif_amd.nets.self_attention import SelfAttention
    torch.manual_seed(1)
    m = SelfAttention(128, 64)
    sd = {'a.' + k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    x = rnd('attx', 2, 128, 8, 12).requires_grad_(True)
    y = x + O.self_attention(sd, 'a', x)
    gy = rnd('attg', *y.shape)
    y.backward(gy)
    m = m.cuda()
    xd = nhwc(x).requires_grad_(True)
    yd = m(xd)
    close(nchw(yd), y, 3e-5, 'attention fwd')
    yd.backward(nhwc(gy))
    close(nchw(xd.grad), x.grad, 1e-4, 'attention dx')
    for k, p in m.named_parameters():
        if 'key.bias' in k:
            continue                      # mathematically zero (softmax is shift invariant): rounding noise only
        close(p.grad, sd['a.' + k].grad, 1e-4, f'attention d{k}')


def test_pointwise_backward():
    # max pool (post-ReLU input: many exact ties at 0)
    x = F.relu(rnd('mpx', 2, 64, 19, 23)).requires_grad_(True)
    y = F.max_pool2d(x, 3, 2, 1)
    gy = rnd('mpg', *y.shape)
    y.backward(gy)
    xd = nhwc(x).requires_grad_(True)
    Fn.MaxPool.apply(xd).backward(nhwc(gy))
    close(nchw(xd.grad), x.grad, 1e-6, 'maxpool bwd')
    # bilinear up-sampling + add
    for (hi, wi, ho, wo) in ((6, 8, 12, 15), (12, 32, 24, 64), (5, 7, 10, 14)):
        s = rnd(('uss', hi), 2, 32, hi, wi).requires_grad_(True)
        a = rnd(('usa', hi), 2, 32, ho, wo).requires_grad_(True)
        y = F.interpolate(s, size=(ho, wo), mode='bilinear', align_corners=True) + a
        gy = rnd(('usg', hi), *y.shape)
        y.backward(gy)
        sd_, ad = nhwc(s).requires_grad_(True), nhwc(a).requires_grad_(True)
        Fn.UpsampleAdd.apply(sd_, ad, ho, wo).backward(nhwc(gy))
        close(nchw(sd_.grad), s.grad, 5e-6, f'upsample bwd {hi}x{wi}')
        close(nchw(ad.grad), a.grad, 0, 'upsample add grad')
    # silu, softmax rows, pair softmax
    v = rnd('siv', 3000, scale=3.0).requires_grad_(True)
    gv = rnd('sig', 3000)
    F.silu(v).backward(gv)
    vd = v.detach().cuda().requires_grad_(True)
    Fn.Silu.apply(vd).backward(gv.cuda())
    close(vd.grad, v.grad, 2e-6, 'silu bwd')
    s = rnd('smx', 37, 151, scale=3.0).requires_grad_(True)
    gs = rnd('smg', 37, 151)
    s.softmax(-1).backward(gs)
    sd_ = s.detach().cuda().requires_grad_(True)
    Fn.SoftmaxRows.apply(sd_).backward(gs.cuda())
    close(sd_.grad, s.grad, 5e-6, 'softmax bwd')
    p = rnd('psx', 5, 7, 30, scale=2.0).requires_grad_(True)
    gp = rnd('psg', 5, 7, 30)
    p.view(5, 7, 15, 2).softmax(-1).view(5, 7, 30).backward(gp)
    pd = p.detach().cuda().requires_grad_(True)
    Fn.PairSoftmax.apply(pd, 15).backward(gp.cuda())
    close(pd.grad, p.grad, 5e-6, 'pair softmax bwd')


@pytest.mark.parametrize('cfg', [(2, 16, 20, 64, 2, 1), (2, 33, 41, 32, 2, 4), (1, 24, 32, 32, 2, 8), (3, 2, 2, 64, 4, 1)])
def test_dwconv_film_backward(cfg):
    B, H, W, C, mult, st = cfg
    x = rnd(('dx', cfg), B, C, H, W).requires_grad_(True)
    w = rnd(('dw', cfg), C * mult, 1, 3, 3, scale=0.3).requires_grad_(True)
    b = rnd(('db', cfg), C * mult, scale=0.1).requires_grad_(True)
    z = F.conv2d(x, w, b, stride=st, padding=1, groups=C)
    film = rnd(('df', cfg), B, 2 * C * mult, *z.shape[-2:]).requires_grad_(True)
    y = z * film[:, :C * mult] + film[:, C * mult:]
    gy = rnd(('dg', cfg), *y.shape)
    y.backward(gy)
    xd, fd = nhwc(x).requires_grad_(True), nhwc(film).requires_grad_(True)
    wd, bd = w.detach().cuda().requires_grad_(True), b.detach().cuda().requires_grad_(True)
    yd = Fn.Film.apply(Fn.DwConv.apply(xd, wd, bd, mult, st), fd)
    close(nchw(yd), y, 5e-6, f'dw+film fwd {cfg}')
    yd.backward(nhwc(gy))
    close(nchw(xd.grad), x.grad, 1e-5, f'dw dx {cfg}')
    close(wd.grad, w.grad, 2e-5, f'dw dw {cfg}')
    close(bd.grad, b.grad, 2e-5, f'dw db {cfg}')
    close(nchw(fd.grad), film.grad, 1e-5, f'film grad {cfg}')


@pytest.mark.parametrize('cfg', [(2, 33, 41, 32, 2, 4), (1, 24, 32, 32, 2, 8), (2, 47, 61, 64, 2, 8), (2, 19, 23, 16, 4, 3), (2, 17, 18, 8, 1, 5),
                                 (2, 16, 20, 64, 2, 2)])
def test_dwconv_data_gradient_added_into_a_map(cfg):
    """nbm_dwconv3x3_bwd_acc (the RPN's strided taps add their gradient into the map the RoI pooling's backward pass filled): the
    scatter form for strides >= 3 (disjoint 3x3 blocks) and the gather form for stride 2, both == content + torch's data gradient,
    and every pixel no tap reaches keeps its bits."""
    B, H, W, C, mult, st = cfg
    x = rnd(('ax', cfg), B, C, H, W).requires_grad_(True)
    w = rnd(('aw', cfg), C * mult, 1, 3, 3, scale=0.3)
    z = F.conv2d(x, w, None, stride=st, padding=1, groups=C)
    gy = rnd(('ag', cfg), *z.shape)
    z.backward(gy)
    base = rnd(('ab', cfg), B, C, H, W)
    acc = nhwc(base).clone()
    ops.dwconv3x3_bwd_acc(nhwc(gy), w.cuda(), mult, st, acc)
    close(nchw(acc), base + x.grad, 2e-6, f'dw acc {cfg}')
    untouched = (x.grad == 0) & (F.conv_transpose2d(torch.ones_like(gy), torch.ones_like(w), stride=st, padding=1, groups=C,
                                                    output_padding=(H - ((z.shape[2] - 1) * st + 1), W - ((z.shape[3] - 1) * st + 1))) == 0)
    assert torch.equal(nchw(acc)[untouched], base[untouched])


def test_batchnorm_train():
    x = rnd('bnx', 3, 256, 6, 10, scale=2.0).requires_grad_(True)
    w = (1 + 0.1 * rnd('bnw', 256)).requires_grad_(True)
    b = (0.1 * rnd('bnb', 256)).requires_grad_(True)
    rm, rv = 0.05 * rnd('bnrm', 256), 1 + 0.1 * rnd('bnrv', 256).abs()
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = F.batch_norm(x, rm_ref, rv_ref, w, b, True, 0.1, 1e-5)
    gy = rnd('bng', *y.shape)
    y.backward(gy)
    xd = nhwc(x).requires_grad_(True)
    wd, bd = w.detach().cuda().requires_grad_(True), b.detach().cuda().requires_grad_(True)
    rmd, rvd = rm.cuda(), rv.cuda()
    yd = Fn.BatchNormTrain.apply(xd, wd, bd, rmd, rvd, 1e-5, 0.1)
    close(nchw(yd), y, 1e-5, 'bn fwd')
    close(rmd, rm_ref, 1e-6, 'running mean')
    close(rvd, rv_ref, 1e-6, 'running var')
    yd.backward(nhwc(gy))
    close(nchw(xd.grad), x.grad, 2e-5, 'bn dx')
    close(wd.grad, w.grad, 2e-5, 'bn dw')
    close(bd.grad, b.grad, 2e-5, 'bn db')


def test_roi_pool_backward():
    cfg = O.make_cfg()
    B = 2
    fm = [rnd(('rfm', i), B, 256, h, w).requires_grad_(True) for i, (h, w) in
          enumerate([(188, 512), (94, 256), (47, 128), (24, 64), (12, 32)])]
    u = synth.uniform('rb_rois', B * 16 * 4).reshape(B, 16, 4)
    x1, y1 = np.floor(u[..., 0] * 1000), np.floor(u[..., 1] * 360)
    w, h = np.floor(2 + u[..., 2] ** 3 * 1000), np.floor(2 + u[..., 3] ** 3 * 370)
    rois = torch.tensor(np.stack([x1, y1, np.minimum(x1 + w, 1023), np.minimum(y1 + h, 374)], -1), dtype=torch.float32)
    pool, _, _ = O.roi_pooling(cfg, rois, fm)
    gp = rnd('rb_g', *pool.shape)
    pool.backward(gp)
    from birdsoundclassif_amd.nets.layers import ROIPooling
    from birdsoundclassif_amd.train import default_args
    rp = ROIPooling(default_args())
    fmd = [nhwc(f).requires_grad_(True) for f in fm]
    n = torch.tensor([16], dtype=torch.int32).cuda()
    pd, _, _ = rp.forward_device(rois.cuda(), n, fmd)
    close(pd.view(B, 16, 2, 2, 256).permute(0, 1, 4, 2, 3), pool, 5e-6, 'roi pool fwd')
    pd.backward(gp.permute(0, 1, 3, 4, 2).reshape(B * 16, 2, 2, 256).contiguous().cuda())
    for i in range(5):
        if fm[i].grad is None:
            assert float(fmd[i].grad.abs().max()) == 0
        else:
            close(nchw(fmd[i].grad), fm[i].grad, 1e-5, f'roi pool grad level {i}')


def test_fused_adamw_matches_torch():
    from birdsoundclassif_amd.train import FusedAdamW
    ps = [rnd(('p', i), *s) for i, s in enumerate([(64, 32, 3, 3), (100,), (7, 13)])]
    ref = [p.clone().requires_grad_(True) for p in ps]
    got = [p.clone().cuda().requires_grad_(True) for p in ps]
    o_ref = torch.optim.AdamW([{'params': ref[:2]}, {'params': ref[2:], 'lr': 1e-2}], lr=1e-3, weight_decay=1e-2)
    o_got = FusedAdamW([{'params': got[:2]}, {'params': got[2:], 'lr': 1e-2}], lr=1e-3, weight_decay=1e-2)
    for it in range(3):
        for i, (r, g) in enumerate(zip(ref, got)):
            gr = rnd(('g', it, i), *r.shape, scale=3.0)
            r.grad = gr.clone()
            g.grad.copy_(gr)                     # gradients live in the optimiser's flat buffer
        o_got.mark_all()
        gn = torch.nn.utils.clip_grad_norm_(ref, 0.5)
        o_ref.step()
        o_got.step(max_norm=0.5)
        assert abs(o_got.grad_norm() - float(gn)) < 1e-4 * float(gn)
        for r, g in zip(ref, got):
            close(g, r, 2e-6, f'adamw step {it}')


# --------------------------------------------------------------------------- whole optimisation steps vs the real reference
def test_two_train_steps_vs_reference_golden():
    from birdsoundclassif_amd.nets import build_model
    from birdsoundclassif_amd.train import default_args, build_optimizer, train_one_step
    g = load_golden('train_b2.npz')
    args = default_args(device='cuda')
    model, crit = build_model(args)
    model.load_state_dict(filler_state_dict())
    model = model.cuda().train()
    crit.train()
    opt, _ = build_optimizer(model, args)
    img = torch.from_numpy(synth.image_batch(0, 2))
    neg_img = torch.from_numpy(synth.image_batch(100, 2))
    bb, ids, lengths = synth.label_batch(0, 2)
    batch = [img, neg_img, bb, ids, lengths]
    params = dict(model.named_parameters())
    np.random.seed(1234)
    for si, neg in enumerate((False, True)):
        loss = train_one_step(model, crit, opt, batch, args.clip_max_norm, 'cuda', negative_sample=neg)
        for k, v in loss.items():
            ref = float(g[f's{si}.loss.{k}'])
            assert abs(float(v) - ref) <= 2e-4 * max(1.0, abs(ref)), (si, k, float(v), ref)
        gn_ref = float(g[f's{si}.grad_norm'])
        assert abs(opt.grad_norm() - gn_ref) <= 2e-3 * gn_ref, (si, opt.grad_norm(), gn_ref)
        coef = min(1.0, args.clip_max_norm / (gn_ref + 1e-6))
        for name in [k[len(f's{si}.grad.'):-len('.shape')] for k in g if k.startswith(f's{si}.grad.') and k.endswith('.shape')]:
            # stored unclipped; ours are unclipped too (the clip coefficient is applied inside the AdamW kernel)
            check_packed(g, f's{si}.grad.{name}', params[name].grad, atol=2e-4 * gn_ref / 50, rtol=2e-3)
        for name in [k[len(f's{si}.param.'):-len('.shape')] for k in g if k.startswith(f's{si}.param.') and k.endswith('.shape')]:
            check_packed(g, f's{si}.param.{name}', params[name], atol=2e-6, rtol=1e-5)
        msd = model.state_dict()
        for name in [k[len(f's{si}.buffer.'):-len('.shape')] for k in g if k.startswith(f's{si}.buffer.') and k.endswith('.shape')]:
            check_packed(g, f's{si}.buffer.{name}', msd[name], atol=2e-5, rtol=2e-5)


# --------------------------------------------------------------------------- Transformer_RCNN head
def test_layernorm_and_leaky_linear_backward():
    M, E = 300, 512
    x = rnd('lnx', M, E).requires_grad_(True)
    w = (1 + 0.1 * rnd('lnw', E)).requires_grad_(True)
    b = (0.1 * rnd('lnb', E)).requires_grad_(True)
    go = rnd('lng', M, E)
    F.layer_norm(x, (E,), w, b, 1e-5).backward(go)
    xd, wd, bd = (t.detach().cuda().requires_grad_(True) for t in (x, w, b))
    y = Fn.LayerNorm.apply(xd, wd, bd, 1e-5)
    close(y, F.layer_norm(x, (E,), w, b, 1e-5), 2e-6, 'layernorm fwd')
    y.backward(go.cuda())
    close(xd.grad, x.grad, 2e-5, 'layernorm dx'); close(wd.grad, w.grad, 2e-5, 'layernorm dw'); close(bd.grad, b.grad, 2e-5, 'layernorm db')
    # linear + LeakyReLU + residual
    x2 = rnd('llx', 200, 256).requires_grad_(True)
    w2 = rnd('llw', 96, 256, scale=0.1).requires_grad_(True)
    b2 = rnd('llb', 96, scale=0.5).requires_grad_(True)
    g2 = rnd('llg', 200, 96)
    F.leaky_relu(F.linear(x2, w2, b2)).backward(g2)
    xd, wd, bd = (t.detach().cuda().requires_grad_(True) for t in (x2, w2, b2))
    y = Fn.linear(xd, wd, bd, act=Fn.ACT_LEAKY)
    close(y, F.leaky_relu(F.linear(x2, w2, b2)), 2e-6, 'leaky linear fwd')
    y.backward(g2.cuda())
    close(xd.grad, x2.grad, 2e-5, 'leaky dx'); close(wd.grad, w2.grad, 2e-5, 'leaky dw'); close(bd.grad, b2.grad, 2e-5, 'leaky db')


@pytest.mark.parametrize('seq_major,S,N,nv', [(True, 128, 5, None), (False, 16, 7, 16), (False, 50, 3, 37), (True, 3, 16, None)])
def test_mha_small_backward(seq_major, S, N, nv):
    nh, E = 8, 512
    hd = E // nh
    qkv = rnd(('mhab', seq_major, S), S * N, 3 * E).requires_grad_(True)
    go = rnd(('mhag', seq_major, S), S * N, E)
    n_use = S if nv is None else nv
    t = qkv.view(S, N, 3 * E) if seq_major else qkv.view(N, S, 3 * E).transpose(0, 1)
    q, k, v = (t[..., i * E:(i + 1) * E].reshape(S, N * nh, hd).transpose(0, 1) for i in range(3))
    att = torch.softmax((q[:, :n_use] @ k[:, :n_use].transpose(1, 2)) / hd ** 0.5, -1) @ v[:, :n_use]      # [N*nh, n_use, hd]
    ref = att.transpose(0, 1).reshape(n_use, N, E)
    gref = go.view(S, N, E) if seq_major else go.view(N, S, E).transpose(0, 1)
    ref.backward(gref[:n_use])
    qd = qkv.detach().cuda().requires_grad_(True)
    cnt = None if nv is None else torch.tensor([nv], dtype=torch.int32, device='cuda')
    ss, bs = (N, 1) if seq_major else (1, S)
    out = Fn.MhaSmall.apply(qd[:, :E], qd[:, E:2 * E], qd[:, 2 * E:], S, N, nh, ss, bs, cnt)
    got = out.view(S, N, E) if seq_major else out.view(N, S, E).transpose(0, 1)
    close(got[:n_use], ref, 3e-6, 'mha fwd')
    valid = torch.zeros(S, N, 1)
    valid[:n_use] = 1
    valid = (valid if seq_major else valid.transpose(0, 1)).reshape(S * N, 1)
    out.backward((go * valid).cuda())
    close(qd.grad, qkv.grad, 3e-5, 'mha dqkv')
    assert float(qd.grad.cpu()[valid[:, 0] == 0].abs().max() if (valid == 0).any() else 0.0) == 0.0


@pytest.mark.parametrize('pe_qk', [False, True])
def test_transformer_rcnn_head_gradients_vs_oracle(pe_qk):
    """Training-mode head (forward + every parameter / input gradient) against torch autograd through the oracle
    restatement, which is itself pinned on the real reference (tests/golden/tf_rcnn_b3.npz)."""
    from birdsoundclassif_amd.nets.layers import Transformer_RCNN
    from birdsoundclassif_amd.train import default_args
    args = default_args(device='cuda', tf_rcnn=True, tf_pe_qk=pe_qk, tf_num_encoder_layers=2)
    head = Transformer_RCNN(args)
    pre = 'head.fast_rcnn.rcnn.'
    sd = synth.fill_state_dict({pre + k: tuple(v.shape) for k, v in head.state_dict().items()})
    head.load_state_dict({k[len(pre):]: v for k, v in sd.items()})
    head = head.cuda().train()
    B, R, C = 5, 16, 256
    pool = rnd(('tfpool', pe_qk), B, R, C, 2, 2).abs().requires_grad_(True)
    pe = rnd(('tfpe', pe_qk), B, R, C, 2, 2)
    r1, r2 = rnd('tfr1', B * R, 604), rnd('tfr2', B * R, 151)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    cfg = O.make_cfg(tf_rcnn=True, tf_pe_qk=pe_qk, tf_num_encoder_layers=2)
    reg, cls = O.transformer_rcnn_forward(sdg, cfg, pool, pe)
    ((reg * r1).sum() + (cls * r2).sum() * 50).backward()
    pd = pool.detach().flatten(end_dim=1).permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
    ped = pe.flatten(end_dim=1).permute(0, 2, 3, 1).contiguous().cuda()
    n = torch.full((1,), R, dtype=torch.int32, device='cuda')
    greg, gcls = head.forward_nhwc(pd, ped, B, R, n)
    close(greg, reg, 2e-5, 'tf reg'); close(gcls, cls, 2e-5, 'tf cls')
    ((greg * r1.cuda()).sum() + (gcls * r2.cuda()).sum() * 50).backward()
    close(pd.grad.view(B, R, 2, 2, C).permute(0, 1, 4, 2, 3), pool.grad, 2e-4, 'tf dpool')
    for name, p in head.named_parameters():
        close(p.grad, sdg[pre + name].grad, 3e-4, 'tf d' + name)


@pytest.mark.parametrize('tag,pe_qk', [('std', False), ('peqk', True)])
def test_tf_rcnn_train_step_vs_reference_golden(tag, pe_qk):
    """One positive `train_one_step` with `--tf_rcnn` against the REAL reference (oracle/make_golden.py
    tf_rcnn_train_golden): losses, clip-norm and sampled gradients from the head down to the backbone."""
    from birdsoundclassif_amd.nets import build_model
    from birdsoundclassif_amd.train import default_args, build_optimizer, train_one_step
    g = load_golden('train_tf_b2.npz')
    args = default_args(device='cuda', tf_rcnn=True, tf_pe_qk=pe_qk)
    model, crit = build_model(args)
    model.load_state_dict(filler_state_dict(tf_rcnn=True, tf_pe_qk=pe_qk))
    model = model.cuda().train()
    crit.train()
    opt, _ = build_optimizer(model, args)
    bb, ids, lengths = synth.label_batch(0, 2)
    img = torch.from_numpy(synth.image_batch(0, 2))
    params = dict(model.named_parameters())
    np.random.seed(4321)
    loss = train_one_step(model, crit, opt, [img, img, bb, ids, lengths], args.clip_max_norm, 'cuda', negative_sample=False)
    for k, v in loss.items():
        ref = float(g[f'{tag}.loss.{k}'])
        assert abs(float(v) - ref) <= 2e-4 * max(1.0, abs(ref)), (k, float(v), ref)
    gn_ref = float(g[f'{tag}.grad_norm'])
    assert abs(opt.grad_norm() - gn_ref) <= 2e-3 * gn_ref, (opt.grad_norm(), gn_ref)
    for name in [k[len(f'{tag}.grad.'):-len('.shape')] for k in g if k.startswith(f'{tag}.grad.') and k.endswith('.shape')]:
        check_packed(g, f'{tag}.grad.{name}', params[name].grad, atol=2e-4 * gn_ref / 50, rtol=2e-3)


@pytest.mark.parametrize('P', [32, 128])
def test_fused_bottleneck_chain_backward(P):
    """Fn.Bottleneck (one tape node per block, ReLU masks and shortcut add inside the dgrad epilogues) on a chain
    downsample-block -> identity-block -> identity-block, the way _ResNetBody wires the mask flags, vs torch autograd.
    P = 128: the stride-1 3x3 convolutions take the Winograd path (F(2x2,3x3) forward with FrozenBN + ReLU in the output
    transform, F(4x4,3x3) for both gradients)."""
    B, H, W, Cin, st = 2, 18, 22, 64, 2
    x = rnd('bnx', B, Cin, H, W).abs().requires_grad_(True)          # a ReLU output, like every block input but layer1.0
    def mk(tag, co, ci, k):
        return rnd(('bnw', tag), co, ci, k, k, scale=(2.0 / (ci * k * k)) ** 0.5).requires_grad_(True)
    blocks = []
    cin = Cin
    for bi in range(3):
        ws = dict(w1=mk((bi, 1), P, cin, 1), w2=mk((bi, 2), P, P, 3), w3=mk((bi, 3), 4 * P, P, 1),
                  wd=mk((bi, 4), 4 * P, cin, 1) if bi == 0 else None)
        aff = {k: (1 + 0.1 * rnd(('bns', bi, k), n), 0.1 * rnd(('bnb', bi, k), n))
               for k, n in (('1', P), ('2', P), ('3', 4 * P), ('d', 4 * P))}
        blocks.append((ws, aff, st if bi == 0 else 1))
        cin = 4 * P
    def bn(t, a):
        return t * a[0].view(1, -1, 1, 1) + a[1].view(1, -1, 1, 1)
    h = x
    for ws, aff, s in blocks:
        o = F.relu(bn(F.conv2d(h, ws['w1']), aff['1']))
        o = F.relu(bn(F.conv2d(o, ws['w2'], stride=s, padding=1), aff['2']))
        idt = h if ws['wd'] is None else bn(F.conv2d(h, ws['wd'], stride=s), aff['d'])
        h = F.relu(bn(F.conv2d(o, ws['w3']), aff['3']) + idt)
    go = rnd('bng', *h.shape)
    h.backward(go)
    xd = nhwc(x).requires_grad_(True)
    dev = [{k: (v.detach().cuda().requires_grad_(True) if v is not None else None) for k, v in ws.items()} for ws, _, _ in blocks]
    hd = xd
    for bi, ((ws, aff, s), wdv) in enumerate(zip(blocks, dev)):
        a = {k: (v[0].cuda(), v[1].cuda()) for k, v in aff.items()}
        sd, bd = a['d'] if wdv['wd'] is not None else (None, None)
        hd = Fn.Bottleneck.apply(hd, wdv['w1'], wdv['w2'], wdv['w3'], wdv['wd'], *a['1'], *a['2'], *a['3'], sd, bd, s,
                                 True, bi == 2)
    close(nchw(hd), h, 1e-5, 'bottleneck chain fwd')
    hd.backward(nhwc(go))
    # the chain input is declared a ReLU output (mask_input): d/dx carries the (x > 0) mask of the producer's ReLU
    close(nchw(xd.grad), x.grad * (x > 0), 3e-5, 'bottleneck dx')
    for bi, (ws, wdv) in enumerate(zip([b[0] for b in blocks], dev)):
        for k in ('w1', 'w2', 'w3', 'wd'):
            if ws[k] is not None:
                close(wdv[k].grad, ws[k].grad, 5e-5, f'bottleneck {bi} d{k}')
    # the ReLU masks as bits (functional.RELU_BITS: written by the blocks' forward epilogues, read by the next block's data gradient) or
    # read from the activations themselves: the same d/dx, bit for bit
    assert Fn.RELU_BITS
    Fn.RELU_BITS = False
    try:
        xd2 = nhwc(x).requires_grad_(True)
        hd2 = xd2
        for bi, ((ws, aff, s), wdv) in enumerate(zip(blocks, dev)):
            a = {k: (v[0].cuda(), v[1].cuda()) for k, v in aff.items()}
            sd, bd = a['d'] if wdv['wd'] is not None else (None, None)
            hd2 = Fn.Bottleneck.apply(hd2, wdv['w1'], wdv['w2'], wdv['w3'], wdv['wd'], *a['1'], *a['2'], *a['3'], sd, bd, s,
                                      True, bi == 2)
        hd2.backward(nhwc(go))
    finally:
        Fn.RELU_BITS = True
    assert torch.equal(hd2, hd) and torch.equal(xd2.grad, xd.grad)


@pytest.mark.parametrize('override,expect', [
    (dict(min_threshold=5000), {'first_class_loss', 'first_regression_loss'}),        # no box survives: "RPN failed"
    (dict(post_nms_topN=4), {'first_class_loss', 'first_regression_loss'}),           # < 16 RoIs: ProposalTargetLayer -> None
])
def test_train_step_soft_failures_keep_first_stage_loss(override, expect, capsys):
    """reference train.py:236-243: when the RPN yields no RoIs, or the proposal target layer cannot fill its batch, the
    step optimises the first-stage losses only (the sentinels of layers.py:297-299,359-364)."""
    from birdsoundclassif_amd.nets import build_model
    from birdsoundclassif_amd.train import default_args, build_optimizer, train_one_step
    args = default_args(device='cuda', **override)
    model, crit = build_model(args)
    model.load_state_dict(filler_state_dict())
    model = model.cuda().train()
    crit.train()
    opt, _ = build_optimizer(model, args)
    bb, ids, lengths = synth.label_batch(0, 2)
    img = torch.from_numpy(synth.image_batch(0, 2))
    before = model.fpn.out_convs['0'].weight.detach().clone()
    head_before = model.head.fast_rcnn.rcnn.bbox_reg_layer.weight.detach().clone()
    np.random.seed(7)
    loss = train_one_step(model, crit, opt, [img, img, bb, ids, lengths], args.clip_max_norm, 'cuda', negative_sample=False)
    assert set(loss) == expect and all(torch.isfinite(v) for v in loss.values())
    assert not torch.equal(model.fpn.out_convs['0'].weight.detach(), before)            # first stage was optimised
    assert torch.equal(model.head.fast_rcnn.rcnn.bbox_reg_layer.weight.detach(), head_before)   # no gradient: untouched
    out = capsys.readouterr().out
    assert ('RPN failed' in out) or ('IMPOSSIBLE TO FILL' in out)


def test_winograd_conv_forward_and_gradients():
    """Conv on the Winograd F(2x2,3x3) path (3x3 / s1 / p1, even map, Cin >= 128: the FPN output convolutions): forward,
    data gradient (Winograd with the rotated kernel), weight gradient (16 transformed-domain TN GEMMs + G^T dU G) and
    bias gradient vs torch autograd, incl. a batch that is cut into several scratch chunks."""
    from birdsoundclassif_amd import ops as _ops
    for (B, H, W, Ci, Co), chunk_bytes in (((3, 12, 20, 128, 64), None), ((5, 10, 16, 160, 96), 16 * 40 * (160 + 96) * 4 * 2),
                                           ((2, 11, 13, 128, 32), None)):                     # odd sizes
        x = rnd(('wx', B, H), B, Ci, H, W).requires_grad_(True)
        w = rnd(('ww', B, H), Co, Ci, 3, 3, scale=(2.0 / (Ci * 9)) ** 0.5).requires_grad_(True)
        b = rnd(('wb', B, H), Co, scale=0.1).requires_grad_(True)
        gy = rnd(('wg', B, H), B, Co, H, W)
        y = F.conv2d(x, w, b, padding=1)
        y.backward(gy)
        xd = nhwc(x).requires_grad_(True)
        wd, bd = w.detach().cuda().requires_grad_(True), b.detach().cuda().requires_grad_(True)
        old = _ops.WINO_CHUNK_BYTES
        if chunk_bytes:
            _ops.WINO_CHUNK_BYTES = chunk_bytes                      # 2 images per chunk -> 3 chunks, the last one short
        try:
            assert Fn._winograd_ok(xd, wd, 3, 3, 1, 1)
            yd = Fn.conv(xd, wd, bias=bd, kh=3, kw=3, pad=1)
            close(nchw(yd), y, 5e-6, 'winograd fwd')
            yd.backward(nhwc(gy))
        finally:
            _ops.WINO_CHUNK_BYTES = old
        close(nchw(xd.grad), x.grad, 2e-5, 'winograd dx')          # backward runs F(4x4,3x3): ~2e-5 of the largest value
        close(wd.grad, w.grad, 3e-5, 'winograd dw')
        close(bd.grad, b.grad, 2e-5, 'winograd db')
        # the F(2x2,3x3) backward setting (tighter) stays available
        Fn.WINO_BWD_TILE = 2
        try:
            xd2 = nhwc(x).requires_grad_(True)
            wd2, bd2 = w.detach().cuda().requires_grad_(True), b.detach().cuda().requires_grad_(True)
            Fn.conv(xd2, wd2, bias=bd2, kh=3, kw=3, pad=1).backward(nhwc(gy))
        finally:
            Fn.WINO_BWD_TILE = 4
        close(nchw(xd2.grad), x.grad, 1e-5, 'winograd F(2,3) dx')
        close(wd2.grad, w.grad, 2e-5, 'winograd F(2,3) dw')


def test_weighted_sum_fusion_backward():
    """BiFPN FusionModule arithmetic (2 and 3 inputs, one weight clipped by the ReLU) vs torch autograd."""
    for n_in, wv in ((2, [0.7, 1.3]), (3, [0.9, -0.2, 0.4])):
        xs = [rnd(('ws', n_in, i), 2, 5, 7, 16).requires_grad_(True) for i in range(n_in)]
        w = torch.tensor(wv).requires_grad_(True)
        g = rnd(('wsg', n_in), 2, 5, 7, 16)
        r = F.relu(w)
        (sum(ri * xi for ri, xi in zip(r, xs)) / (r.sum() + 1e-4)).backward(g)
        xd = [x.detach().cuda().requires_grad_(True) for x in xs]
        wd = w.detach().cuda().requires_grad_(True)
        y = Fn.WeightedSum.apply(xd[0], xd[1], xd[2] if n_in == 3 else None, wd)
        ref = sum(ri * xi for ri, xi in zip(r, xs)) / (r.sum() + 1e-4)
        close(y, ref, 2e-6, 'weighted sum fwd')
        y.backward(g.cuda())
        for a, b in zip(xd, xs):
            close(a.grad, b.grad, 2e-6, 'weighted sum dx')
        close(wd.grad, w.grad, 2e-5, 'weighted sum dw')


@pytest.mark.parametrize('third_consumer', [False, True])
def test_premasked_gradient_handover(third_consumer, monkeypatch):
    """A ReLU output with two consumers (stem output: max pooling + finest FPN lateral; layer tap: next stage's first block + lateral):
    the consumer that forms the complete gradient (its own share + the stashed one) masks it by (y > 0) in its own kernel, and the
    producer skips its masking pass -- unless autograd touched the gradient in between (a third consumer), in which case the producer
    masks as before.  Either way the gradients equal torch autograd's."""
    B, H, W, C0, P = 2, 18, 22, 64, 32
    x0 = rnd('pmx', B, C0, H, W).requires_grad_(True)
    w0 = rnd('pmw0', C0, C0, 1, 1, scale=(2.0 / C0) ** 0.5).requires_grad_(True)
    wl0 = rnd('pmwl0', 96, C0, 1, 1, scale=(1.0 / C0) ** 0.5).requires_grad_(True)
    wl1 = rnd('pmwl1', 96, C0, 1, 1, scale=(1.0 / C0) ** 0.5).requires_grad_(True)
    ws = dict(w1=rnd('pmb1', P, C0, 1, 1, scale=(2.0 / C0) ** 0.5), w2=rnd('pmb2', P, P, 3, 3, scale=(2.0 / (9 * P)) ** 0.5),
              w3=rnd('pmb3', 4 * P, P, 1, 1, scale=(2.0 / P) ** 0.5), wd=rnd('pmbd', 4 * P, C0, 1, 1, scale=(2.0 / C0) ** 0.5))
    for v in ws.values():
        v.requires_grad_(True)
    # torch reference
    y = F.relu(F.conv2d(x0, w0))                                  # "stem": ReLU output, read by the pooling and a lateral
    lat0 = F.conv2d(y, wl0)
    p = F.max_pool2d(y, 3, 2, 1)                                  # p is not a ReLU output of a block here, but >= 0 all the same
    lat1 = F.conv2d(p, wl1)
    o = F.relu(F.conv2d(p, ws['w1']))
    o = F.relu(F.conv2d(o, ws['w2'], stride=2, padding=1))
    h = F.relu(F.conv2d(o, ws['w3']) + F.conv2d(p, ws['wd'], stride=2))
    extra = (y * 0.5).sum() + (p * 0.25).sum() if third_consumer else 0.0
    g0, g1, gh = rnd('pmg0', *lat0.shape), rnd('pmg1', *lat1.shape), rnd('pmgh', *h.shape)
    ((lat0 * g0).sum() + (lat1 * g1).sum() + (h * gh).sum() + extra).backward()
    # device
    calls = []
    real = ops.relu_bwd
    monkeypatch.setattr(ops, 'relu_bwd', lambda gy, yy: (calls.append(tuple(gy.shape)), real(gy, yy))[1])
    Fn.stash_reset()
    dv = {k: v.detach().cuda().requires_grad_(True) for k, v in dict(ws, w0=w0, wl0=wl0, wl1=wl1).items()}
    xd = nhwc(x0).requires_grad_(True)
    one = torch.ones(4 * P, device='cuda'); zero = torch.zeros(4 * P, device='cuda')
    yd = Fn.conv(xd, dv['w0'], act=ops.ACT_RELU)
    pd = Fn.MaxPool.apply(yd, True)                               # registers itself as the consumer that picks the stash up
    hd = Fn.Bottleneck.apply(pd, dv['w1'], dv['w2'], dv['w3'], dv['wd'], one[:P], zero[:P], one[:P], zero[:P], one, zero, one, zero,
                             2, False, True)
    l0 = Fn.conv(yd, dv['wl0'])                                   # created last -> their backward nodes run first and stash
    l1 = Fn.conv(pd, dv['wl1'])
    extra_d = (yd * 0.5).sum() + (pd * 0.25).sum() if third_consumer else 0.0
    ((l0 * nhwc(g0)).sum() + (l1 * nhwc(g1)).sum() + (hd * nhwc(gh)).sum() + extra_d).backward()
    Fn.stash_check_empty()
    close(nchw(xd.grad), x0.grad, 3e-5, 'premask dx')
    for k, ref in dict(ws, w0=w0, wl0=wl0, wl1=wl1).items():
        close(dv[k].grad, ref.grad, 5e-5, f'premask d{k}')
    stem_masks = [c for c in calls if c == tuple(yd.shape)]
    if third_consumer:
        assert len(stem_masks) == 1, calls                        # autograd summed a third share in: masked by the producer
    else:
        assert not stem_masks, calls                              # complete and masked inside maxpool_bwd: no extra pass


def test_prepared_weight_cache_does_not_grow_with_training_steps():
    """Regression (scripts/soak.py): cache entries keyed by a per-step tensor identity (the FrozenBN scale handed to the Winograd
    data-gradient weights) or by version counters were never evicted -- one 36-plane weight copy per ResNet 3x3 layer and step.
    The number of prepared copies and the device memory held between steps must not depend on the number of steps."""
    from birdsoundclassif_amd import train as T
    from birdsoundclassif_amd.nets import build_model, _prep
    args = T.default_args(device='cuda')
    model, crit = build_model(args)
    model.load_state_dict(filler_state_dict())
    model = model.cuda().train()
    crit.train()
    opt, _ = T.build_optimizer(model, args)
    B = 2
    img = torch.from_numpy(synth.image_batch(0, B))
    bb, ids, lens = synth.label_batch(0, B)
    batch = [img, img, bb, ids, lens]
    np.random.seed(3)
    seen = []
    for it in range(6):
        T.train_one_step(model, crit, opt, batch, args.clip_max_norm, 'cuda', negative_sample=(it == 3))
        torch.cuda.synchronize()
        seen.append((len(_prep._cache), torch.cuda.memory_allocated()))
    assert seen[5][0] == seen[2][0], seen
    assert seen[5][1] <= seen[2][1] + (8 << 20), seen


@pytest.mark.parametrize('geom', [(2, 188, 512, 94, 256, 32, 8), (3, 47, 66, 24, 33, 8, 8), (1, 25, 33, 13, 17, 16, 5), (2, 24, 64, 12, 32, 4, 8)])
def test_bilinear_backward_reads_the_pattern_patches_only_and_takes_a_tile_share(geom):
    """nbm_upsample_bilinear_bwd(pattern_stride): on a gradient that is zero outside the 5x5 patches of the stride-S pattern the
    pruned gather equals the full one bit for bit (the rows / columns it leaves out contribute exact zeros);
    nbm_tiles_upsample_bilinear_bwd_add: the scatter form on a compact tile operand equals scattering the tiles into a map and
    gathering from that (different summation order: 1e-6)."""
    from birdsoundclassif_amd.ops import _ptr, _stream, check, lib
    B, Ho, Wo, Hi, Wi, C_, S = geom
    gen = torch.Generator().manual_seed(11)

    def band(n):
        i = torch.arange(n)
        return ((i + 2) % S < 5) & ((i + 2) // S < (n - 1) // S + 1)
    inside = (band(Ho)[:, None] & band(Wo)[None, :]).float()
    g = (torch.randn((B, Ho, Wo, C_), generator=gen) * inside[None, :, :, None]).cuda().contiguous()
    full = ops.upsample_bilinear_bwd(g, Hi, Wi)
    pruned = ops.upsample_bilinear_bwd(g, Hi, Wi, pattern_stride=S)
    assert torch.equal(full, pruned)
    # a pixel outside the patches is NOT read by the pruned form (that is the point of it)
    g2 = g.clone()
    for axis, n in ((1, Ho), (2, Wo)):
        out = torch.nonzero(~band(n)).flatten()
        if out.numel():                           # (S = 5: the bands are contiguous, only the rows past the last cell are outside)
            g2.index_fill_(axis, out.cuda(), float('nan'))
    assert torch.equal(ops.upsample_bilinear_bwd(g2, Hi, Wi, pattern_stride=S), full)
    # tile share
    TH, TW = (Ho + 1) // 2, (Wo + 1) // 2
    ids = torch.randperm(B * TH * TW, generator=gen)[:max(1, B * TH * TW // 7)].sort().values.int()
    n = -(-ids.numel() // 128) * 128 + 128                        # one block of padding only
    tiles = torch.full((n,), -1, dtype=torch.int32)
    tiles[:ids.numel()] = ids
    tiles = tiles.cuda()
    compact = torch.randn((n * 4, C_), generator=gen).cuda()
    dense = torch.zeros_like(g)
    check(lib().nbm_tiles_scatter_add(_ptr(dense), B, Ho, Wo, C_, _ptr(tiles), n, None, _ptr(compact), _stream()), 'scatter')
    ref = ops.upsample_bilinear_bwd(dense + g, Hi, Wi)
    got = ops.upsample_bilinear_bwd(g, Hi, Wi, pattern_stride=S, tiles_share=[(compact, tiles, 0, B)])
    err = float((ref - got).abs().max())
    assert err <= 1e-5 * max(1.0, float(ref.abs().max())), err
    with pytest.raises(ValueError):
        ops.upsample_bilinear_bwd(g, Hi, Wi, tiles_share=[(compact[:-4], tiles, 0, B)])


def _train_setup(seed=0):
    from birdsoundclassif_amd.nets import build_model
    from birdsoundclassif_amd.train import default_args, build_optimizer
    args = default_args(device='cuda')
    model, crit = build_model(args)
    model.load_state_dict(filler_state_dict(seed))
    model = model.cuda().train()
    crit.train()
    opt, _ = build_optimizer(model, args)
    x = torch.from_numpy(synth.image_batch(0, 2)).cuda()
    bb, ids, lengths = synth.label_batch(0, 2)
    return args, model, crit, opt, [x, x, bb, ids, lengths]


def test_one_training_pass_in_flight_two_models_interleaved_raise():
    """the gradient hand-over registries (functional._PARKED / _STASH) serve ONE forward / backward pair at a time.  A
    second grad-enabled forward pass -- of another model, or of the same one (gradient accumulation over two forwards) -- that starts
    while the first pass still has its RPN share parked raises instead of silently dropping that share."""
    from birdsoundclassif_amd import train as T
    args, a, crit_a, opt_a, batch = _train_setup()
    _, b, crit_b, opt_b, _ = _train_setup(1)
    np.random.seed(3)
    opt_a.zero_grad()
    loss = T.step(a, crit_a, batch, 'cuda', negative_sample=False, early_backward=True)     # forward of A: the RPN branch is back-propagated
    if T.SPLIT_BACKWARD and Fn.GRAD_SHARE:
        # the early pass parked the RPN's share of d/d(FPN map) on every level; the RoI pooling's backward pass has not run yet:
        # nothing may start a new grad-enabled pass now
        assert len(Fn._PARKED) == 5
        with pytest.raises(RuntimeError, match='still holds'):
            b.forward_first_stage(batch[0][:, None], lazy=True)
        with pytest.raises(RuntimeError, match='still holds'):
            a.forward_first_stage(batch[0][:, None], lazy=True)
        assert len(Fn._PARKED) == 5              # the refused passes dropped nothing
    # ... while a forward pass under no_grad (validation between two training steps, reference train.py:362-377) is harmless:
    with torch.no_grad():
        b.eval()
        b.detect(batch[0][:, None], min_score=0.2)
        b.train()
    wd = crit_a.weight_dict
    sum(loss[k] * wd[k] for k in loss if k in wd).backward()
    Fn.parked_flush()
    Fn.stash_check_empty()
    got = {n: p.grad.clone() for n, p in a.named_parameters() if p.grad is not None}
    # reference: the same step of the same model, nothing in between
    _, a2, crit_a2, opt_a2, _ = _train_setup()
    np.random.seed(3)
    opt_a2.zero_grad()
    loss2 = T.step(a2, crit_a2, batch, 'cuda', negative_sample=False, early_backward=True)
    sum(loss2[k] * wd[k] for k in loss2 if k in wd).backward()
    Fn.parked_flush()
    Fn.stash_check_empty()
    gmax = max(float(p.grad.abs().max()) for p in a2.parameters() if p.grad is not None)
    for n, p in a2.named_parameters():
        if p.grad is None:
            assert n not in got or float(got[n].abs().max()) == 0.0, n
            continue
        # per-parameter scale, with a floor for gradients that are mathematically zero (the attention's key bias: softmax does not
        # see it) and hold the atomic-order noise of two runs
        scale = float(p.grad.abs().max())
        assert float((got[n] - p.grad).abs().max()) <= 2e-4 * scale + 1e-6 * gmax, n
    # the pass is closed: both models may start a new one
    b.forward_first_stage(batch[0][:, None], lazy=True)
    Fn.pass_abandon()


def test_early_backward_of_another_models_pass_raises():
    from birdsoundclassif_amd import train as T
    args, a, crit_a, opt_a, batch = _train_setup()
    _, b, crit_b, opt_b, _ = _train_setup(1)
    if not T.SPLIT_BACKWARD:
        pytest.skip('NBM_SPLIT_BACKWARD=0')
    np.random.seed(3)
    # b's forward pass is the one in flight; an early RPN backward issued for `a` must not park its gradients under b's maps
    o = b.forward_first_stage(batch[0][:, None], lazy=True)
    crit_a._pre_loss = {'first_class_loss': o['rpn_cls_scores'].sum() * 0 + 1.0}
    with pytest.raises(RuntimeError, match='two models interleaved'):
        T._early_rpn_backward(a, crit_a)
    Fn.pass_abandon()


def test_a_backward_pass_that_leaves_a_stashed_gradient_raises_by_itself():
    """every backward pass validates itself (engine callback queued by the node that stashes): a caller that runs its own
    `backward()` on a sub-graph -- so that the second consumer never runs -- gets a RuntimeError from `backward()`, not a dropped share."""
    x = rnd('sx', 2, 8, 8, 64).cuda().requires_grad_(True)
    w1 = rnd('sw1', 64, 64, 1, 1, scale=0.1).cuda().requires_grad_(True)
    h = Fn.conv(x, w1, None)                     # producer of the shared tensor
    Fn.stash_reset()
    Fn.stash_accept(h, True)                     # a second consumer announced itself in the forward pass ...
    y = Fn.Conv.apply(h, w1, None, None, None, None, 1, 1, 1, 0, Fn.ACT_NONE, 1.0)
    with pytest.raises(RuntimeError, match='never picked up'):
        y.sum().backward()                       # ... but only the first consumer's branch is back-propagated
    assert not Fn._STASH


@pytest.mark.parametrize('shape', [(2, 8, 16, 128, 96, True), (1, 12, 32, 256, 64, False)])
def test_attention_and_lateral_as_one_composed_node(shape):
    """functional.AttnLateral (training form): lateral(fm + SelfAttention(fm)) + bilinear(up) with the value projection taken to
    W_l W_o W_v -- output and the gradients of all twelve tensors against the uncomposed chain of tape nodes (Attention, then the lateral
    convolution with its fused merge), which the golden training fixtures pin against the reference."""
    B, h, w, Cc, p_, with_up = shape
    d = Cc // 2
    torch.manual_seed(3)

    def mk(*s, scale=1.0):
        return (torch.randn(*s, device='cuda') * scale).requires_grad_(True)
    x = mk(B, h, w, Cc)
    wq, wk, wv = mk(d, Cc, scale=Cc ** -0.5), mk(d, Cc, scale=Cc ** -0.5), mk(d, Cc, scale=Cc ** -0.5)
    bq, bk, bv = mk(d, scale=0.1), mk(d, scale=0.1), mk(d, scale=0.1)
    wo, bo = mk(Cc, d, scale=d ** -0.5), mk(Cc, scale=0.1)
    wl, bl = mk(p_, Cc, 1, 1, scale=Cc ** -0.5), mk(p_, scale=0.1)
    up = mk(B, (h + 1) // 2, (w + 1) // 2, p_) if with_up else None
    inv = float(np.float32(1.0) / np.float32(np.round(np.sqrt(d), 2)))
    gy = torch.randn(B, h, w, p_, device='cuda')
    leaves = [x, wq, bq, wk, bk, wv, bv, wo, bo, wl, bl] + ([up] if with_up else [])
    names = ['x', 'wq', 'bq', 'wk', 'bk', 'wv', 'bv', 'wo', 'bo', 'wl', 'bl', 'up']

    def run(composed):
        for t in leaves:
            t.grad = None
        Fn.stash_reset(None)
        if composed:
            y = Fn.AttnLateral.apply(x, wq, bq, wk, bk, wv, bv, wo, bo, wl, bl, up, inv)
        else:
            out = Fn.Attention.apply(x.view(B, h * w, Cc), wq, bq, wk, bk, wv, bv, wo, bo, inv).view(B, h, w, Cc)
            y = Fn.conv(out, wl, bias=bl, up=up)
        y.backward(gy)
        return y.detach().clone(), [t.grad.detach().clone() for t in leaves]
    y0, g0 = run(False)
    y1, g1 = run(True)
    assert float((y1 - y0).abs().max()) <= 2e-5 * max(1.0, float(y0.abs().max()))
    gmax = max(float(b.abs().max()) for b in g0)
    for n, a, b in zip(names, g1, g0):
        # (the key bias shifts every score of a row alike: its gradient is zero up to rounding noise, ~1e-6 here -- hence the absolute term)
        scale_ = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-4 * scale_ + 1e-6 * gmax, (n, float((a - b).abs().max()), scale_, gmax)
