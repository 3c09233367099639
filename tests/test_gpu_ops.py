"""GPU parity tests, op level: every HIP kernel (called through the C ABI via birdsoundclassif_amd.ops)
against the CPU oracle / a plain torch fp32 reference of the same op on seeded inputs.

Tolerances: fp32 GEMM-like ops compare at 2e-5 * (1 + |ref|) scaled by sqrt(K)-ish accumulation noise (stated
per test); integer / index / box outputs must be bit-exact.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from birdsoundclassif_amd import ops, synth          # noqa: E402
from oracle import nets_ref as O                      # noqa: E402


def dev(t):
    return t.cuda().contiguous()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.cpu().permute(0, 3, 1, 2)


def rnd(key, *shape, scale=1.0):
    return torch.from_numpy((synth.normal(key, int(np.prod(shape))) * scale).astype(np.float32).reshape(shape))


def assert_close(got, ref, atol, rtol, name=''):
    got, ref = got.detach().cpu().float(), ref.detach().cpu().float()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    assert bool((err <= tol).all()), f'{name}: max err {float(err.max()):.3e}, ref max {float(ref.abs().max()):.3e}'


def krsc(w):
    co, ci, kh, kw = w.shape
    w2 = w.permute(0, 2, 3, 1).reshape(co, -1)
    if ci % 32:
        w2 = F.pad(w2, (0, (-w2.shape[1]) % 32))
    return w2.contiguous().cuda()


@pytest.mark.parametrize('cfg', [
    # B, H, W, Cin, Cout, k, stride, pad
    (2, 13, 17, 64, 64, 1, 1, 0),
    (2, 13, 17, 64, 256, 1, 1, 0),
    (1, 20, 24, 128, 128, 3, 1, 1),
    (2, 21, 19, 128, 128, 3, 2, 1),
    (2, 12, 14, 256, 512, 1, 2, 0),
    (1, 25, 33, 384, 256, 3, 1, 1),
    (2, 37, 41, 3, 64, 7, 2, 3),
    (1, 9, 8, 256, 6, 1, 1, 0),
    (1, 9, 8, 512, 40, 1, 1, 0),
])
def test_conv_igemm(cfg):
    B, H, W, Ci, Co, k, st, pad = cfg
    x = rnd(('x', cfg), B, Ci, H, W)
    w = rnd(('w', cfg), Co, Ci, k, k, scale=(2.0 / (Ci * k * k)) ** 0.5)
    scale = 1 + 0.1 * rnd(('s', cfg), Co)
    shift = 0.1 * rnd(('b', cfg), Co)
    ref = F.conv2d(x, w, stride=st, padding=pad)
    res = rnd(('r', cfg), *ref.shape)
    ref = F.relu(ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res)
    got = ops.conv2d(nhwc(x), krsc(w), k, k, st, pad, scale=dev(scale), shift=dev(shift), residual=nhwc(res),
                     act=ops.ACT_RELU)
    torch.cuda.synchronize()
    assert_close(nchw(got), ref, 2e-5, 2e-5, f'conv {cfg}')


@pytest.mark.parametrize('cfg', [(64, 256, True, 'relu'), (64, 64, False, 'relu'), (256, 64, True, 'none'), (256, 64, False, 'silu'),
                                 (64, 64, True, 'leaky'), (128, 512, True, 'relu'), (256, 1024, True, 'relu'), (256, 128, True, 'none')])
def test_streaming_1x1_equals_the_tiled_kernel_bit_for_bit(cfg, monkeypatch):
    """The 1x1 layers whose weights fit LDS (ResNet layer1; wider ones in slices of N: 128 -> 512, 256 -> 1024) go through
    stream1x1_kernel (csrc/igemm.hip) once the map is large enough: same K order, same epilogue arithmetic -> the SAME BITS as the tiled kernel (NBM_STREAM1X1=0), incl. a ragged last
    32-row tile, and both within fp32 tolerance of torch."""
    K, N, with_res, act = cfg
    B, H, W = 2, 67, 63                                    # M = 8442: not a multiple of 32, above the streaming threshold
    x = rnd(('sx', cfg), B, K, H, W)
    w = rnd(('sw', cfg), N, K, 1, 1, scale=(2.0 / K) ** 0.5)
    scale, shift = 1 + 0.1 * rnd(('ss', cfg), N), 0.1 * rnd(('sb', cfg), N)
    res = rnd(('sr', cfg), B, N, H, W) if with_res else None
    ref = F.conv2d(x, w) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + (res if with_res else 0)
    ref = {'relu': F.relu, 'silu': F.silu, 'leaky': lambda t: F.leaky_relu(t, 0.01), 'none': lambda t: t}[act](ref)
    code = {'relu': ops.ACT_RELU, 'silu': ops.ACT_SILU, 'leaky': ops.ACT_LEAKY, 'none': ops.ACT_NONE}[act]
    out = {}
    for flag in ('0', '1'):
        monkeypatch.setenv('NBM_STREAM1X1', flag)
        out[flag] = ops.conv2d(nhwc(x), krsc(w), scale=dev(scale), shift=dev(shift), residual=nhwc(res) if with_res else None, act=code)
        torch.cuda.synchronize()
    assert torch.equal(out['0'], out['1'])
    assert_close(nchw(out['1']), ref, 2e-5, 2e-5, f'streaming 1x1 {cfg}')


def test_conv_alpha_silu_bias():
    x = rnd('xa', 2, 64, 10, 12)
    w = rnd('wa', 96, 64, 1, 1, scale=0.2)
    b = rnd('ba', 96, scale=0.1)
    ref = F.silu(F.conv2d(x + x, w, b))
    got = ops.conv2d(nhwc(x), krsc(w), shift=dev(b), alpha=2.0, act=ops.ACT_SILU)
    assert_close(nchw(got), ref, 2e-5, 2e-5, 'alpha/silu')


def test_bgemm_and_row_shift():
    a = rnd('ga', 3, 70, 96)
    b = rnd('gb', 3, 150, 96)
    got = ops.bgemm_nt(dev(a), dev(b), alpha=0.5)
    assert_close(got, 0.5 * torch.matmul(a, b.transpose(1, 2)), 3e-5, 3e-5, 'bgemm')
    wv = rnd('gw', 40, 96)
    bias = rnd('gbias', 40)
    got = ops.bgemm_nt(dev(wv), dev(b), shift=dev(bias), shift_per_row=True)
    ref = torch.matmul(wv[None], b.transpose(1, 2)) + bias[None, :, None]
    assert_close(got, ref, 3e-5, 3e-5, 'bgemm row shift')


def test_self_attention_level():
    from birdsoundclassif_amd.nets.self_attention import SelfAttention
    torch.manual_seed(0)
    m = SelfAttention(128, 64)
    sd = {'a.' + k: v.detach().clone() for k, v in m.state_dict().items()}
    x = rnd('sa', 2, 128, 8, 12)            # L = 96: the P.V GEMM needs L % 32 == 0 (true for 24x64 and 12x32 maps)
    ref = x + O.self_attention(sd, 'a', x)
    got = m.cuda()(nhwc(x), residual=True)
    assert_close(nchw(got), ref, 3e-5, 3e-5, 'self attention')


def test_maxpool_upsample_softmax_silu():
    x = rnd('mp', 2, 64, 19, 23)
    assert_close(nchw(ops.maxpool3x3s2(nhwc(x))), F.max_pool2d(x, 3, 2, 1), 0, 0, 'maxpool')
    src = rnd('us', 2, 32, 6, 8)
    add = rnd('ua', 2, 32, 12, 15)
    ref = F.interpolate(src, size=(12, 15), mode='bilinear', align_corners=True) + add
    assert_close(nchw(ops.upsample_bilinear_add(nhwc(src), 12, 15, add=nhwc(add))), ref, 2e-6, 2e-6, 'upsample+add')
    ref = F.interpolate(src, size=(12, 16), mode='bilinear', align_corners=True)
    assert_close(nchw(ops.upsample_bilinear_add(nhwc(src), 12, 16)), ref, 2e-6, 2e-6, 'upsample x2')
    s = rnd('sm', 37, 151, scale=3.0)
    assert_close(ops.softmax_rows_(dev(s)), s.softmax(-1), 1e-6, 1e-5, 'softmax 151')
    s = rnd('sm2', 9, 1536, scale=2.0)
    assert_close(ops.softmax_rows_(dev(s)), s.softmax(-1), 1e-7, 1e-5, 'softmax 1536')
    v = rnd('si', 1000, scale=3.0)
    assert_close(ops.silu(dev(v)), F.silu(v), 1e-6, 1e-6, 'silu')
    p = rnd('ps', 5, 7, 30, scale=2.0)
    ref = p.view(5, 7, 15, 2).softmax(-1).view(5, 7, 30)
    assert_close(ops.pair_softmax(dev(p), 15), ref, 1e-6, 1e-6, 'pair softmax')


@pytest.mark.parametrize('cfg', [(2, 16, 20, 64, 2, 1), (2, 33, 41, 32, 2, 4), (1, 24, 32, 32, 2, 8), (3, 2, 2, 64, 4, 1)])
def test_dwconv(cfg):
    B, H, W, C, mult, st = cfg
    x = rnd(('dx', cfg), B, C, H, W)
    w = rnd(('dw', cfg), C * mult, 1, 3, 3, scale=0.3)
    b = rnd(('db', cfg), C * mult, scale=0.1)
    ref = F.conv2d(x, w, b, stride=st, padding=1, groups=C)
    got = ops.dwconv3x3(nhwc(x), dev(w), dev(b), mult, st)
    assert_close(nchw(got), ref, 2e-6, 2e-6, f'dwconv {cfg}')
    if st == 1:
        film = rnd(('df', cfg), B, 2 * C * mult, H, W)
        ref2 = ref * film[:, :C * mult] + film[:, C * mult:]
        got2 = ops.dwconv3x3(nhwc(x), dev(w), dev(b), mult, st, film=nhwc(film).view(-1, 2 * C * mult))
        assert_close(nchw(got2), ref2, 4e-6, 4e-6, f'dwconv film {cfg}')


@pytest.mark.parametrize('shape', [(2, 9, 11), (1, 37, 300), (2, 375, 1024), (1, 8, 257)])
def test_stem7x7_folded(shape):
    """csrc/stem.hip: init_conv folded into conv1 + FrozenBN + ReLU vs torch on the 3-channel map (borders: the init_conv
    bias is only present inside the image)."""
    from birdsoundclassif_amd.nets import _prep
    B, H, W = shape
    x = rnd(('st', shape), B, 1, H, W)
    wi, bi = rnd('stw', 3, 1, 1, 1), rnd('stb', 3)
    w1 = rnd('stw1', 64, 3, 7, 7, scale=0.1)
    sc, sh = rnd('stsc', 64).abs() + 0.5, rnd('stsh', 64, scale=0.2)
    ref = F.relu(F.conv2d(F.conv2d(x.double(), wi.double(), bi.double()), w1.double(), None, 2, 3) * sc.double().view(1, -1, 1, 1)
                 + sh.double().view(1, -1, 1, 1)).float()
    got = ops.stem7x7(nhwc(x), *_prep.stem_fold(dev(w1), dev(wi), dev(bi)), dev(sc), dev(sh))
    assert_close(nchw(got), ref, 2e-5, 2e-5, f'stem7x7 {shape}')


def test_init_conv():
    x = rnd('ic', 2, 1, 9, 11)
    w = rnd('icw', 3, 1, 1, 1)
    b = rnd('icb', 3)
    got = ops.init_conv(nhwc(x), dev(w), dev(b))
    assert_close(nchw(got), F.conv2d(x, w, b), 1e-6, 1e-6, 'init_conv')


# --------------------------------------------------------------------------- proposal path (bit-exact)
def _rpn_inputs(B, seed):
    cfg = O.make_cfg()
    cls_raw = rnd(('pc', seed), B, 30, 24, 64, scale=1.5)
    cls = cls_raw.view(B, 15, 2, 24, 64).softmax(2).view(B, 30, 24, 64)
    reg = rnd(('pr', seed), B, 60, 24, 64, scale=0.25)
    return cfg, cls, reg


@pytest.mark.parametrize('training', [False, True])
def test_proposal_layer_bit_exact(training):
    from birdsoundclassif_amd.nets.layers import ProposalLayer
    from birdsoundclassif_amd.train import default_args
    cfg, cls, reg = _rpn_inputs(2, 3)
    ref_rois, ref_scores = O.proposal_layer(cfg, cls, reg, training=training)
    pl = ProposalLayer(default_args(), 5)
    pl.train(training)
    rois, scores = pl(cls.cuda(), reg.cuda())
    assert rois.shape == ref_rois.shape, (rois.shape, ref_rois.shape)
    mism = (rois.cpu() != ref_rois).any(-1)
    # expf on the device vs SLEEF on the CPU may differ by 1 ulp before a coordinate is rounded: allow a
    # vanishing fraction of 1-pixel flips, report them
    assert mism.float().mean() <= 1e-3, f'{int(mism.sum())} RoIs differ'
    assert (rois.cpu() - ref_rois).abs().max() <= 1.0
    assert_close(scores, ref_scores, 0, 0, 'roi scores')


def test_rpn_failed_path():
    from birdsoundclassif_amd.nets.layers import ProposalLayer
    from birdsoundclassif_amd.train import default_args
    cfg, cls, reg = _rpn_inputs(2, 4)
    reg = torch.full_like(reg, -20.0)              # every box collapses below min_threshold
    ref_rois, _ = O.proposal_layer(cfg, cls, reg, training=False)
    assert ref_rois.numel() == 0
    pl = ProposalLayer(default_args(), 5).eval()
    rois, scores = pl(cls.cuda(), reg.cuda())
    assert rois.numel() == 0 and scores.numel() == 0


def test_proposal_layer_independent_images_equal_one_image_per_call():
    """`independent=True` (bulk inference): every image of the launch gets the RoIs the reference gives it when it is run ALONE
    (one file per model call, reference nbm_detect.py:24-28) -- the batch-coupled minima of layers.py:287 / nets_utils.py:236 do
    not leak between launch-mates.  Image 1 keeps few anchors, image 2 fails ("RPN failed"), images 0 and 3 fill the top-N."""
    from birdsoundclassif_amd.nets.layers import ProposalLayer
    from birdsoundclassif_amd.train import default_args
    cfg, cls, reg = _rpn_inputs(4, 7)
    few = reg[1, :, 5:6, 10:13].clone()             # image 1: all boxes collapse below min_threshold but those of 3 positions
    reg[1] = -20.0
    reg[1, :, 5:6, 10:13] = few
    reg[2] = -20.0                                  # image 2: none survives
    pl = ProposalLayer(default_args(), 5).eval()
    rois, scores, n = pl.forward_device(cls.permute(0, 2, 3, 1).contiguous().cuda(), reg.permute(0, 2, 3, 1).contiguous().cuda(),
                                        independent=True)
    assert n.shape == (4,)
    n = n.cpu().tolist()
    coupled = pl.forward_device(cls.permute(0, 2, 3, 1).contiguous().cuda(), reg.permute(0, 2, 3, 1).contiguous().cuda())[2]
    assert int(coupled.item()) == 0                 # one call on the whole batch: image 2 fails everybody (reference semantics)
    seen = set()
    for b in range(4):
        ref_rois, ref_scores = O.proposal_layer(cfg, cls[b:b + 1], reg[b:b + 1], training=False)
        nb = ref_rois.shape[1] if ref_rois.numel() else 0
        assert n[b] == nb, (b, n[b], nb)
        seen.add(nb)
        if nb:
            assert torch.equal(rois[b, :nb].cpu(), ref_rois[0]) and torch.equal(scores[b, :nb].cpu(), ref_scores[0])
        assert not rois[b, nb:].any()
    assert 0 in seen and 50 in seen and len(seen) >= 3, seen


def test_nms_ties_and_order():
    # many identical boxes / scores: greedy NMS must walk in the given order
    boxes = torch.tensor([[10., 10, 50, 50], [10, 10, 50, 50], [12, 12, 52, 52], [200, 100, 260, 160],
                          [201, 101, 261, 161], [500, 300, 600, 370]])[None].repeat(2, 1, 1)
    boxes[1, :, 0] += 3
    scores = torch.linspace(0.9, 0.4, 6)[None].repeat(2, 1)
    ref_b, ref_s, _ = O.batched_nms(boxes, scores, 0.3, 50)
    bx = torch.zeros(2, 64, 4)
    sc = torch.zeros(2, 64)
    bx[:, :6], sc[:, :6] = boxes, scores
    n_in = torch.tensor([6], dtype=torch.int32).cuda()
    rois, rs, n_out = ops.nms_batched(bx.cuda(), sc.cuda(), n_in, 0.3, 50)
    n = int(n_out.item())
    assert n == ref_b.shape[1]
    assert torch.equal(rois[:, :n].cpu(), ref_b) and torch.equal(rs[:, :n].cpu(), ref_s)


def test_roi_pool_and_rcnn_post():
    cfg = O.make_cfg()
    B = 2
    fmaps = [rnd(('fm', i), B, 256, h, w) for i, (h, w) in enumerate([(188, 512), (94, 256), (47, 128), (24, 64), (12, 32)])]
    # RoIs of every size class, incl. degenerate / border boxes
    u = synth.uniform('rois', B * 40 * 4).reshape(B, 40, 4)
    x1 = np.floor(u[..., 0] * 1000)
    y1 = np.floor(u[..., 1] * 360)
    w = np.floor(2 + u[..., 2] ** 3 * 1000)
    h = np.floor(2 + u[..., 3] ** 3 * 370)
    rois = torch.tensor(np.stack([x1, y1, np.minimum(x1 + w, 1023), np.minimum(y1 + h, 374)], -1), dtype=torch.float32)
    rois[0, 0] = torch.tensor([0., 0, 1023, 374])
    rois[0, 1] = torch.tensor([5., 5, 5, 5])
    rois[0, 2] = torch.tensor([1000., 370, 1023, 374])
    ref_pool, ref_pe, ref_lvl = O.roi_pooling(cfg, rois, fmaps)
    from birdsoundclassif_amd.nets.layers import ROIPooling
    from birdsoundclassif_amd.train import default_args
    rp = ROIPooling(default_args())
    pool, pe, lvl = rp(rois.cuda(), [f.cuda() for f in fmaps])
    assert np.array_equal(lvl, ref_lvl.numpy())
    assert_close(pool, ref_pool, 2e-6, 2e-6, 'roi pool')
    assert_close(pe, ref_pe, 2e-6, 2e-6, 'roi pe')

    # post-processing on synthetic head outputs
    N = B * 40
    reg = rnd('post_reg', N, 604, scale=0.3)
    logits = rnd('post_cls', N, 151, scale=3.0)
    logits[:, 0] += 2.0
    cls = logits.softmax(-1)
    for ms in (0.05, 0.3):
        ref = O.fast_rcnn_post(cfg, rois, reg, cls, 0.3, ms)
        n_roi = torch.tensor([40], dtype=torch.int32).cuda()
        det, n_det = ops.rcnn_post(rois.cuda(), n_roi, reg.cuda(), cls.cuda(), 1024, 375, 0.3, ms, 50)
        from birdsoundclassif_amd.nets.layers import FastRCNN
        got = FastRCNN.dets_to_dicts(det, n_det, 150)
        from helpers import dets_to_rows
        r_ref, r_got = dets_to_rows(ref), dets_to_rows(got)
        assert r_ref.shape == r_got.shape and len(r_ref) > 0, (r_ref.shape, r_got.shape)
        assert np.array_equal(r_ref[:, :6], r_got[:, :6]), 'boxes / classes differ'
        assert np.abs(r_ref[:, 6] - r_got[:, 6]).max() == 0


def test_conv_with_fused_topdown_merge():
    """nbm_gemm_conv `up`: lateral 1x1 + bias + bilinear(align_corners) of the coarser map == the two separate kernels."""
    B, H, W, Ci, N = 2, 23, 37, 64, 384
    x = torch.from_numpy(synth.normal('upx', B * H * W * Ci).astype(np.float32).reshape(B, H, W, Ci)).cuda()
    w = torch.from_numpy((synth.normal('upw', N * Ci) * 0.1).astype(np.float32).reshape(N, Ci)).cuda()
    b = torch.from_numpy(synth.normal('upb', N).astype(np.float32)).cuda()
    coarse = torch.from_numpy(synth.normal('upc', B * 12 * 19 * N).astype(np.float32).reshape(B, 12, 19, N)).cuda()
    ref = ops.upsample_bilinear_add(coarse, H, W, add=ops.conv2d(x, w, shift=b, alpha=2.0))
    got = ops.conv2d(x, w, shift=b, alpha=2.0, up=coarse)
    assert (got - ref).abs().max().item() <= 1e-6
    t = F.interpolate(coarse.permute(0, 3, 1, 2).cpu(), size=(H, W), mode='bilinear', align_corners=True)
    t = t + F.conv2d(2.0 * x.permute(0, 3, 1, 2).cpu(), w.cpu()[:, :, None, None], b.cpu())
    assert (got.cpu().permute(0, 3, 1, 2) - t).abs().max().item() < 2e-5


def test_c_abi_from_plain_c(tmp_path):
    """examples/cabi_gemm_conv.c: the library driven from C with hipMalloc'd buffers -- no Python object crosses the ABI."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / 'cabi_gemm_conv')
    libdir = os.path.join(root, 'birdsoundclassif_amd')
    subprocess.run(['gcc', '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include', '-I' + os.path.join(root, 'include'),
                    os.path.join(root, 'examples', 'cabi_gemm_conv.c'), '-L' + libdir, '-lnbm_hip', '-L/opt/rocm/lib', '-lamdhip64',
                    '-lm', '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib', '-o', exe], check=True, capture_output=True, timeout=300)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'max |err|' in out.stdout


@pytest.mark.parametrize('cfg', [
    # B, H, W, Cin, Cout, k, stride, pad, groups      (K / 32 >= 9: the launches the deep-K kernel gets)
    (2, 24, 32, 512, 256, 1, 1, 0, 1),
    (3, 17, 23, 1024, 200, 1, 1, 0, 1),        # ragged M (1173 rows) and N
    (2, 21, 19, 128, 128, 3, 2, 1, 1),         # taps at the image border, stride 2
    (2, 12, 14, 512, 1024, 1, 2, 0, 1),        # 1x1 stride 2
    (1, 25, 33, 384, 256, 3, 1, 1, 1),         # 3x3 stride 1
    (1, 9, 11, 64, 192, 5, 4, 2, 1),           # 5x5 / stride 4 / pad 2 (the composed RPN reader's launches)
    (1, 1, 300, 320, 96, 1, 1, 0, 3),          # grouped plain GEMM (three groups of [300 x 320] x [96 x 320]^T)
])
def test_half_step_kernel_gives_the_bits_of_the_two_stage_kernel(cfg, monkeypatch):
    """csrc/igemm_h16.hip (half-step LDS stages, three workgroups per CU; library default for deep K) against igemm_kernel<128,128,...>
    (NBM_H16=0): the same products in the same order -- bit for bit, with every epilogue operand in play -- and a convolution (float64)."""
    B, H, W, Ci, Co, k, st, pad, G = cfg

    def nrm(key, *shape, scale=1.0):
        return torch.from_numpy((synth.normal((key, cfg), int(np.prod(shape))) * scale).astype(np.float32).reshape(shape))

    if G == 1:
        x = F.relu(nrm('hx', B, Ci, H, W))
        w = nrm('hw', Co, Ci, k, k, scale=(2.0 / (Ci * k * k)) ** 0.5)
        sc, sh = 1 + 0.1 * nrm('hs', Co), 0.1 * nrm('hb', Co)
        ref = F.conv2d(x.double(), w.double(), stride=st, padding=pad)
        res = nrm('hr', *ref.shape)
        mask = nrm('hm', *ref.shape)
        ref = F.relu(ref * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1) + res.double()) * (mask > 0)
        xd = x.permute(0, 2, 3, 1).contiguous().cuda()
        wd = w.permute(0, 2, 3, 1).reshape(Co, -1).contiguous().cuda()
        rd, md = res.permute(0, 2, 3, 1).contiguous().cuda(), mask.permute(0, 2, 3, 1).contiguous().cuda()
        Ho, Wo = ref.shape[2:]

        def run():
            y = torch.empty((B, Ho, Wo, Co), device='cuda')
            ops.gemm_conv(xd, wd, y, B=B, H=H, W=W, Cin=Ci, N=Co, kh=k, kw=k, stride=st, pad=pad, Ho=Ho, Wo=Wo, w_ld=wd.shape[1],
                          scale=sc.cuda(), shift=sh.cuda(), residual=rd, act=ops.ACT_RELU, mask=md, mask_ld=Co)
            return y
        ref = ref.permute(0, 2, 3, 1)
    else:
        a = nrm('ga', G, W, Ci).cuda()
        bm = nrm('gb', G, Co, Ci, scale=(1.0 / Ci) ** 0.5).cuda()
        ref = torch.einsum('gmk,gnk->gmn', a.double().cpu(), bm.double().cpu())

        def run():
            return ops.bgemm_nt(a, bm)
    outs = {}
    for mode in ('0', '3'):
        monkeypatch.setenv('NBM_H16', mode)
        outs[mode] = run()
    monkeypatch.delenv('NBM_H16')
    default = run()
    assert torch.equal(outs['0'], outs['3']) and torch.equal(default, outs['3']), cfg
    err = float((outs['3'].cpu().double() - ref).abs().max())
    assert err <= 2e-5 * (1 + float(ref.abs().max())), (cfg, err)


@pytest.mark.parametrize('cfg', [
    # B, H, W, Cin (channels of dX), N (channels of G), k, stride     -- launches the two-stage data-gradient kernel used to get
    (2, 24, 32, 256, 1024, 1, 1),          # deep 1x1, 128-wide tile
    (3, 17, 23, 200, 512, 1, 1),           # ragged rows (1173) and columns
    (1, 25, 33, 256, 384, 3, 1),           # 3x3 stride 1
    (2, 21, 19, 256, 256, 3, 2),           # stride 2 by parity class, odd map, 8..32 steps per class
    (2, 20, 24, 64, 96, 3, 1),             # 64-wide tile beyond the single-stage limit (27 steps)
    (2, 16, 20, 48, 704, 1, 1),            # 64-wide tile, 22 steps, ragged columns
])
def test_half_step_data_gradient_gives_the_bits_of_the_two_stage_kernel(cfg, monkeypatch):
    """`igemm_nn_kernel<BN, 2, true>` (half-step LDS stages, three or more workgroups per CU; library default for the deep-K data
    gradients) against the two-stage kernel (NBM_NN_H16=0) with every epilogue operand in play (BatchNorm scale on G, shortcut gradient,
    ReLU mask): the same products in the same order, bit for bit -- and the gradient of a convolution (float64)."""
    B, H, W, Ci, N, k, st = cfg
    pad = k // 2

    def nrm(key, *shape, scale=1.0):
        return torch.from_numpy((synth.normal((key, cfg), int(np.prod(shape))) * scale).astype(np.float32).reshape(shape))

    x = nrm('dx', B, Ci, H, W).double().requires_grad_(True)
    w = nrm('dw', N, Ci, k, k, scale=(2.0 / (Ci * k * k)) ** 0.5)
    sc = 1 + 0.1 * nrm('ds', N)
    y = F.conv2d(x, w.double(), stride=st, padding=pad) * sc.double().view(1, -1, 1, 1)
    g = nrm('dg', *y.shape, scale=0.1)
    y.backward(g.double())
    res, mask = nrm('dr', B, Ci, H, W), nrm('dm', B, Ci, H, W)
    ref = ((x.grad + res.double()) * (mask > 0)).permute(0, 2, 3, 1)
    gd = g.permute(0, 2, 3, 1).reshape(-1, N).contiguous().cuda()
    wk = w.permute(0, 2, 3, 1).reshape(N, -1).contiguous().cuda()
    rd, md = res.permute(0, 2, 3, 1).contiguous().cuda(), mask.permute(0, 2, 3, 1).contiguous().cuda()
    outs = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('NBM_NN_H16', mode)
        out = torch.empty((B, H, W, Ci), device='cuda')
        ops.conv_dgrad(gd, wk, out, B=B, H=H, W=W, Cin=Ci, N=N, kh=k, kw=k, stride=st, pad=pad, g_ld=N, w_ld=wk.shape[1], a_scale=sc.cuda(),
                       residual=rd, mask=md)
        outs[mode] = out
    monkeypatch.delenv('NBM_NN_H16')
    assert torch.equal(outs['0'], outs['1']), cfg
    err = float((outs['1'].cpu().double() - ref).abs().max())
    assert err <= 2e-5 * (1 + float(ref.abs().max())), (cfg, err)


@pytest.mark.parametrize('cfg', [
    # B, H, W, Cin, N, residual     -- one launch per kernel that finishes a bottleneck: streaming 1x1 (K = 64 / sliced 128 / 256), single-stage,
    (2, 70, 64, 64, 256, True),     # half-step deep K
    (2, 70, 64, 128, 512, True),
    (2, 70, 64, 256, 1024, True),
    (3, 17, 23, 256, 96, False),    # single-stage tiled kernel (M < 8192: not the streaming form), 96 = 3 words per row
    (1, 24, 32, 512, 2048, True),
])
def test_relu_bits_of_the_forward_epilogues_and_their_use_as_data_gradient_mask(cfg):
    """`nbm_gemm_desc.bits_out`: the epilogue that stores y also stores (y > 0), 32 channels per word -- checked against y itself for every
    kernel that finishes a ResNet bottleneck; `nbm_bwd_desc.mask_bits`: the data gradient masked by those bits is, bit for bit, the one
    masked by y."""
    B, H, W, Ci, N, with_res = cfg

    def nrm(key, *shape, scale=1.0):
        return torch.from_numpy((synth.normal((key, cfg), int(np.prod(shape))) * scale).astype(np.float32).reshape(shape)).cuda()

    x = torch.relu(nrm('bx', B, H, W, Ci))
    w = nrm('bw', N, Ci, scale=(2.0 / Ci) ** 0.5)
    sc, sh = 1 + 0.1 * nrm('bs', N), 0.1 * nrm('bb', N)
    res = nrm('br', B, H, W, N) if with_res else None
    bits = torch.full((B * H * W * N // 32,), -1, device='cuda', dtype=torch.int32)
    y = ops.conv2d(x, w, scale=sc, shift=sh, residual=res, act=ops.ACT_RELU, bits_out=bits)
    y0 = ops.conv2d(x, w, scale=sc, shift=sh, residual=res, act=ops.ACT_RELU)
    assert torch.equal(y, y0)
    want = (y.view(-1, 32) > 0).to(torch.int64)
    pos = torch.tensor([8 * (ch % 4) + ch // 4 for ch in range(32)], device='cuda')       # bit of channel ch within its word (nbm_hip.h)
    packed = (want << pos).sum(1)
    packed = torch.where(packed >= 2 ** 31, packed - 2 ** 32, packed).to(torch.int32)
    assert torch.equal(bits, packed), cfg
    assert 0.2 < float(want.float().mean()) < 0.8                  # (the mask is not trivial)
    # consumer: d/dy of a following 1x1 (y [.., N] -> P channels), masked by y / by the bits
    P = 64
    g = nrm('bg', B * H * W, P, scale=0.1)
    wk = nrm('bk', P, N, scale=(2.0 / N) ** 0.5)
    a_scale = 1 + 0.1 * nrm('ba', P)
    short = nrm('bh', B, H, W, N)
    outs = []
    for kw in (dict(mask=y), dict(mask=y, mask_bits=bits)):
        out = torch.empty((B, H, W, N), device='cuda')
        ops.conv_dgrad(g, wk, out, B=B, H=H, W=W, Cin=N, N=P, g_ld=P, w_ld=N, a_scale=a_scale, residual=short, **kw)
        outs.append(out)
    assert torch.equal(outs[0], outs[1]), cfg
    assert float((outs[1] == 0).float().mean()) > 0.15


def test_winograd_fused_block_shapes_give_the_same_bits(monkeypatch):
    """The fused F(2x2,3x3) kernel in its 96-row block shape (chosen for dense launches whose 128-row blocks leave the last round of
    workgroups half empty, csrc/wino_fused.hip) against the 128-row shape: every output sums its planes and K-steps in the same order, so
    the results are the same BITS -- ragged tile counts, odd map sizes, both epilogue flavours -- and both are a convolution (float64)."""
    from birdsoundclassif_amd.nets import _prep

    def nrm(key, *shape):
        return synth.normal(key, int(np.prod(shape))).reshape(shape)

    for B, H, W, Ci, Co, relu in ((3, 24, 64, 128, 256, True), (2, 11, 13, 96, 128, False), (5, 12, 32, 160, 384, True), (1, 6, 10, 64, 128, False)):
        x = torch.from_numpy(nrm(('wfx', B, H, Ci), B, H, W, Ci).astype(np.float32)).cuda()
        w = torch.from_numpy((nrm(('wfw', Co, Ci), Co, Ci, 3, 3) * (2.0 / (9 * Ci)) ** 0.5).astype(np.float32))
        b = torch.from_numpy((0.1 * nrm(('wfb', Co), Co)).astype(np.float32)).cuda()
        mask = torch.from_numpy(nrm(('wfm', B, H, Co), B, H, W, Co).astype(np.float32)).cuda() if relu else None
        U = _prep.wino23(w.cuda())
        outs = {}
        for bm in ('128', '96'):
            monkeypatch.setenv('NBM_WINO_BM', bm)
            outs[bm] = ops.conv3x3_winograd(x, U, b, relu=relu, mask=mask)
        monkeypatch.delenv('NBM_WINO_BM')
        auto = ops.conv3x3_winograd(x, U, b, relu=relu, mask=mask)
        assert torch.equal(outs['128'], outs['96']) and torch.equal(auto, outs['128']), (B, H, W, Ci, Co)
        ref = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double(), b.double().cpu(), padding=1)
        if relu:
            ref = F.relu(ref) * (mask.permute(0, 3, 1, 2).cpu() > 0)
        err = (outs['96'].permute(0, 3, 1, 2).cpu().double() - ref).abs().max().item()
        assert err <= 5e-6 * max(1.0, ref.abs().max().item()), (err, B, H, W, Ci, Co)


def test_winograd_weight_transform_kernels():
    """nbm_wino_weight / nbm_wino_weight_grad vs the float64 definition U = G g G^T, dW = G^T dU G (both tile sizes, the
    rotated / channel-swapped / scaled form of the data-gradient weights, the row scale of the gradient)."""
    from birdsoundclassif_amd import ops
    G = {2: torch.tensor([[1.0, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1.0]], dtype=torch.float64),
         4: torch.tensor([[1.0, 0, 0], [1, 1, 1], [1, -1, 1], [1, 0.5, 0.25], [1, -2, 4], [0, 0, 1.0]], dtype=torch.float64)}
    w = rnd(('wwt', 1), 96, 160, 3, 3)
    sc = rnd(('wsc', 1), 96).abs() + 0.5
    for m in (2, 4):
        a = m + 2
        u = ops.wino_weight(w.cuda(), m=m).cpu()
        ref = torch.einsum('ia,ncab,jb->ijnc', G[m], w.double(), G[m]).reshape(a * a, 96, 160)
        assert u.shape == ref.shape and torch.equal(u, ref.float())
        ut = ops.wino_weight(w.cuda(), transposed=True, m=m, scale=sc.cuda()).cpu()
        gt = (w.double() * sc.double().view(-1, 1, 1, 1)).flip(2, 3).transpose(0, 1)
        reft = torch.einsum('ia,ncab,jb->ijnc', G[m], gt, G[m]).reshape(a * a, 160, 96)
        assert (ut.double() - reft).abs().max() <= 1.2e-7 * reft.abs().max()
        dU = rnd(('wdu', m), a * a, 96, 160)
        dw = ops.wino_weight_grad(dU.cuda(), m, row_scale=sc.cuda()).cpu()
        refw = torch.einsum('ia,ijnc,jb->ncab', G[m], dU.double().view(a, a, 96, 160), G[m]) * sc.double().view(-1, 1, 1, 1)
        assert (dw.double() - refw).abs().max() <= 2e-6 * refw.abs().max()


def test_proposal_iou_bit_exact_and_criterion_handover():
    """nbm_proposal_iou == the reference-order NumPy IoU (nets_utils.py:103-126) bit for bit -- overlaps, exact ties (first best
    box), disjoint boxes, boxes equal to a ground-truth box -- and the training step's hand-over (device IoU + pinned copies ->
    host thresholds and draws) returns what the all-host ProposalTargetLayer returns, from the same RNG position."""
    from birdsoundclassif_amd.nets import targets
    from birdsoundclassif_amd.nets.criterion import SetCriterion
    from birdsoundclassif_amd.train import default_args
    from test_targets_host import _batch, _pre_from_numpy
    args = default_args(device='cuda')
    layer = targets.ProposalTargetLayer(args)
    for seed, (B, R, cap) in enumerate([(16, 1000, 1024), (8, 40, 64), (5, 17, 32)]):
        rois, gt, ids, lens = _batch(B, R, seed)
        rois[0, 1] = rois[0, 0]                                   # duplicate proposal
        rois[1, 2] = torch.tensor([5., 5., 4., 4.])               # degenerate (x2 < x1): area 0, the reference divides anyway
        gt_pad, _ = layer.pad_gt(gt.numpy(), lens)
        full = torch.full((B, cap, 4), 3.25)
        full[:, :R] = rois
        mx, asg = ops.proposal_iou(full.cuda(), torch.from_numpy(gt_pad).cuda(), torch.tensor(lens, dtype=torch.int32).cuda())
        _, mx_ref, asg_ref, _ = _pre_from_numpy(layer, rois, gt, lens, cap)
        rows = np.r_[0:R, cap:cap + gt_pad.shape[1]]
        # rows of GT boxes an image does not have compare -1 boxes with real ones: never read by the host, skip them
        for b in range(B):
            r = np.r_[0:R, cap:cap + lens[b]]
            assert np.array_equal(mx.cpu().numpy()[b, r].view(np.uint32), mx_ref[b, r].view(np.uint32)), (seed, b)
            assert np.array_equal(asg.cpu().numpy()[b, r], asg_ref[b, r]), (seed, b)
        # the criterion's flow, as train.step drives it
        crit = SetCriterion(args, {})
        dev_rois = full.cuda()
        crit.precompute_proposal_iou(dev_rois, gt, lens)
        np.random.seed(seed)
        got, s_got = crit.generate_all_rois(dev_rois[:, :R].contiguous(), gt, ids, lens, use_precomputed=True), np.random.get_state()
        np.random.seed(seed)
        ref, s_ref = layer(rois, gt, ids, lens), np.random.get_state()
        for k, a in zip(('rois', 'bbox_targets', 'labels'), ref):
            assert torch.equal(got[k].cpu(), a), (seed, k)
        assert np.array_equal(s_ref[1], s_got[1]) and s_ref[2] == s_got[2]


def test_anchor_targets_bit_exact_and_criterion_handover():
    """nbm_anchor_targets == the reference-order NumPy arithmetic of the AnchorTargetLayer (layers.py:150-179) on every inside anchor:
    labels before the subsampling (thresholds, best-anchor ties), first best box; a degenerate box raises the image's flag; and the
    training step's hand-over (side stream, pinned copies -> host draws) returns what the all-host layer returns, from the same RNG
    position."""
    from birdsoundclassif_amd.nets import targets
    from birdsoundclassif_amd.nets.criterion import SetCriterion
    from birdsoundclassif_amd.train import default_args
    from test_targets_host import _anchor_pre_from_numpy
    args = default_args(device='cuda')
    layer = targets.AnchorTargetLayer(args)
    anc = layer.anchors.cuda().contiguous()
    rng = np.random.default_rng(0)
    for seed, B in enumerate([8, 5, 16]):
        bbs, lens = [], []
        for i in range(B):
            bb, _, l = synth.label_batch((i + seed) % 8, 1)
            bbs.append(bb); lens += l
        gt = torch.cat(bbs).numpy().astype(np.float32)
        # boxes equal to anchors (IoU 1, exact ties between boxes), a box twice, a tiny box no anchor reaches 0.3 with
        gt[0] = layer.anchors_np[rng.integers(len(layer.anchors_np))]
        if lens[1] >= 2:
            gt[lens[0] + 1] = gt[lens[0]]
        gt[-1] = np.array([500., 200., 503., 202.], np.float32)
        G = max(lens)
        gt_pad = np.full((B, G, 4), -1, np.float32)
        gt_pad[np.arange(G)[None, :] < np.asarray(lens)[:, None]] = gt
        lab, amx, flag = ops.anchor_targets(anc, torch.from_numpy(gt_pad).cuda(), torch.tensor(lens, dtype=torch.int32).cuda(),
                                            args.rpn_neg_label, args.rpn_pos_label)
        ref = _anchor_pre_from_numpy(layer, gt, lens, args)
        assert int(flag.sum()) == 0
        assert np.array_equal(lab.cpu().numpy(), ref[0]), seed
        assert np.array_equal(amx.cpu().numpy(), ref[1]), seed
        assert int((ref[0] == 1).sum()) >= B and int((ref[0] == 0).sum()) > 0
        # the criterion's flow, as train.step drives it
        crit = SetCriterion(args, {})
        gt_t = torch.from_numpy(gt)
        crit.start_anchor_targets(gt_t, lens, 'cuda')
        assert crit._pre_anchor is not None
        np.random.seed(seed)
        pre = crit._take_anchor_pre(lens)
        got, s_got = layer(gt_t, lens, device='cpu', pre=pre), np.random.get_state()
        np.random.seed(seed)
        want, s_ref = layer(gt_t, lens, device='cpu'), np.random.get_state()
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
        assert np.array_equal(s_ref[1], s_got[1]) and s_ref[2] == s_got[2]
    # degenerate box of area -289 = minus the area of the 17 x 17 anchors: 0 / 0 = NaN there -> the flag of that image (only) is raised
    gt_pad[2, 0] = np.array([100., 10., 82., 26.], np.float32)
    assert np.isnan(targets.box_iou_incl(layer.anchors_np, gt_pad[2, :1])).any()
    _, _, flag = ops.anchor_targets(anc, torch.from_numpy(gt_pad).cuda(), torch.tensor(lens, dtype=torch.int32).cuda(),
                                    args.rpn_neg_label, args.rpn_pos_label)
    f = flag.cpu().numpy()
    assert f[2] == 1 and f.sum() == 1


def test_space_to_batch_makes_a_dilated_conv_an_ordinary_one_and_avgpool2x2():
    """`--dilation` building blocks: nbm_space_to_batch2 and its inverse are permutations; the ordinary 3x3 / pad-1 convolution on the
    space-to-batch form equals torch's 3x3 / dilation-2 / pad-2 convolution (forward and both gradients through the autograd
    Functions); nbm_avgpool2x2 == F.adaptive_avg_pool2d to half the size, and its gradient."""
    import torch.nn.functional as F
    from birdsoundclassif_amd.nets import functional as Fn
    B, H, W, Ci, Co = 2, 12, 16, 128, 128
    x = torch.from_numpy(synth.normal('s2b.x', B * H * W * Ci).astype(np.float32)).view(B, H, W, Ci).cuda()
    p = ops.space_to_batch2(x)
    assert tuple(p.shape) == (4 * B, H // 2, W // 2, Ci)
    for a in range(2):
        for b in range(2):
            assert torch.equal(p[(2 * a + b) * B:(2 * a + b + 1) * B], x[:, a::2, b::2])
    assert torch.equal(ops.space_to_batch2(p, inverse=True), x)
    w = (torch.from_numpy(synth.normal('s2b.w', Co * Ci * 9).astype(np.float32)).view(Co, Ci, 3, 3) * 0.05)
    xr = x.detach().cpu().permute(0, 3, 1, 2).clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, padding=2, dilation=2)
    gy = torch.from_numpy(synth.normal('s2b.g', ref.numel()).astype(np.float32)).view_as(ref)
    ref.backward(gy)
    xd, wd = x.clone().requires_grad_(True), w.cuda().requires_grad_(True)
    y = Fn.SpaceToBatch2.apply(Fn.conv(Fn.SpaceToBatch2.apply(xd, False), wd, kh=3, kw=3, pad=1), True)
    got = y.permute(0, 3, 1, 2)
    assert float((got.detach().cpu() - ref.detach()).abs().max()) < 2e-5 * float(ref.abs().max())
    y.backward(gy.permute(0, 2, 3, 1).contiguous().cuda())
    assert float((xd.grad.cpu().permute(0, 3, 1, 2) - xr.grad).abs().max()) < 1e-4 * float(xr.grad.abs().max())
    assert float((wd.grad.cpu() - wr.grad).abs().max()) < 1e-4 * float(wr.grad.abs().max())
    # 2x2 average pooling
    t = x.detach().cpu().permute(0, 3, 1, 2).clone().requires_grad_(True)
    r = F.adaptive_avg_pool2d(t, (H // 2, W // 2))
    g2 = torch.from_numpy(synth.normal('ap.g', r.numel()).astype(np.float32)).view_as(r)
    r.backward(g2)
    xa = x.clone().requires_grad_(True)
    ya = Fn.AvgPool2x2.apply(xa)
    assert float((ya.detach().cpu().permute(0, 3, 1, 2) - r.detach()).abs().max()) < 1e-6
    ya.backward(g2.permute(0, 2, 3, 1).contiguous().cuda())
    assert torch.equal(xa.grad.cpu().permute(0, 3, 1, 2), t.grad)
