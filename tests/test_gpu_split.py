"""GPU: the deep-K implicit GEMM on the bf16 matrix pipe through split fp32 operands (csrc/igemm_split.hip, NBM_SPLIT_BF16=1) against
float64 and against the fp32-matrix-instruction kernel of the same launch.  The split form must not be the less accurate of the two:
its error against float64 is asserted to stay within 1.05 x the fp32 kernel's (measured 0.4-0.6 x)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from birdsoundclassif_amd import ops, synth          # noqa: E402


def rnd(key, *shape, scale=1.0):
    return torch.from_numpy((synth.normal(key, int(np.prod(shape))) * scale).astype(np.float32).reshape(shape))


def krsc(w):
    co, ci, kh, kw = w.shape
    return w.permute(0, 2, 3, 1).reshape(co, -1).contiguous().cuda()


def both(monkeypatch, fn):
    monkeypatch.setenv('NBM_SPLIT_BF16', '0')
    a = fn()
    monkeypatch.setenv('NBM_SPLIT_BF16', '1')
    b = fn()
    torch.cuda.synchronize()
    return a, b


@pytest.mark.parametrize('cfg', [
    # B, H, W, Cin, Cout, k, stride, pad                 K32 steps -> stages mod 6
    (2, 24, 32, 512, 256, 1, 1, 0),        # 16 -> 2
    (2, 24, 32, 384, 384, 1, 1, 0),        # 12 -> 0, N = 3 tiles
    (3, 17, 23, 1024, 200, 1, 1, 0),       # 32 -> 4, ragged M (1173 rows) and N
    (2, 21, 19, 128, 128, 3, 2, 1),        # 36 -> 0, taps at the image border, stride 2
    (2, 12, 14, 512, 1024, 1, 2, 0),       # 1x1 stride 2
    (1, 25, 33, 384, 256, 3, 1, 1),        # 108 -> 0, 3x3 stride 1
    (1, 40, 64, 2048, 512, 1, 1, 0),       # 64 -> 2
])
def test_split_conv_against_float64_and_the_fp32_kernel(cfg, monkeypatch):
    B, H, W, Ci, Co, k, st, pad = cfg
    x = F.relu(rnd(('x', cfg), B, Ci, H, W))
    w = rnd(('w', cfg), Co, Ci, k, k, scale=(2.0 / (Ci * k * k)) ** 0.5)
    scale = 1 + 0.1 * rnd(('s', cfg), Co)
    shift = 0.1 * rnd(('b', cfg), Co)
    ref = F.conv2d(x.double(), w.double(), stride=st, padding=pad)
    res = rnd(('r', cfg), *ref.shape)
    ref = F.relu(ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1) + res.double())
    xd, wd = x.permute(0, 2, 3, 1).contiguous().cuda(), krsc(w)
    rd = res.permute(0, 2, 3, 1).contiguous().cuda()
    f32, split = both(monkeypatch, lambda: ops.conv2d(xd, wd, k, k, st, pad, scale=scale.cuda(), shift=shift.cuda(), residual=rd,
                                                      act=ops.ACT_RELU))
    ref = ref.permute(0, 2, 3, 1)
    e32 = (f32.cpu().double() - ref).abs()
    esp = (split.cpu().double() - ref).abs()
    assert float(esp.max()) <= 2e-5 * (1 + float(ref.abs().max())), (cfg, float(esp.max()))
    rms32, rmssp = float((e32 ** 2).mean().sqrt()), float((esp ** 2).mean().sqrt())
    assert rmssp <= 1.05 * rms32 + 1e-9, f'{cfg}: split rms error {rmssp:.3e} vs fp32 kernel {rms32:.3e}'
    assert float((split - f32).abs().max()) <= 2e-5 * (1 + float(ref.abs().max()))


def test_split_grouped_gemm_with_row_shift(monkeypatch):
    """The attention form: groups of plain GEMMs (blockIdx.z), shift per row."""
    G, M, K, N = 6, 1536, 512, 384
    x = rnd('gx', G, M, K).cuda()
    w = rnd('gw', G, N, K, scale=K ** -0.5).cuda()
    sh = rnd('gs', G * M).cuda()

    def run():
        y = torch.empty(G, M, N, device='cuda')
        ops.gemm_conv(x, w, y, B=1, H=M, W=1, Cin=K, N=N, groups=G, x_gs=M * K, w_gs=N * K, y_gs=M * N, alpha=0.5)
        return y
    f32, split = both(monkeypatch, run)
    ref = 0.5 * torch.einsum('gmk,gnk->gmn', x.double().cpu(), w.double().cpu())
    e32, esp = (f32.cpu().double() - ref).abs(), (split.cpu().double() - ref).abs()
    assert float(esp.max()) <= 2e-5 * (1 + float(ref.abs().max()))
    assert float((esp ** 2).mean().sqrt()) <= 1.05 * float((e32 ** 2).mean().sqrt()) + 1e-9
    del sh


def test_split_with_fused_topdown_merge(monkeypatch):
    B, H, W, Ci, N = 2, 23, 37, 512, 384
    x = rnd('upx', B, H, W, Ci).cuda()
    w = rnd('upw', N, Ci, scale=0.05).cuda()
    b = rnd('upb', N).cuda()
    coarse = rnd('upc', B, 12, 19, N).cuda()
    f32, split = both(monkeypatch, lambda: ops.conv2d(x, w, shift=b, alpha=2.0, up=coarse))
    assert float((split - f32).abs().max()) <= 2e-5 * (1 + float(f32.abs().max()))
    assert float((split - f32).abs().max()) > 0          # (the two kernels do differ in the last bits: the switch took effect)


@pytest.mark.parametrize('cfg', [
    # rows (B, H, W), Cin (= channels of dX), N (= channels of the incoming gradient = K), a_scale, residual, mask
    ((2, 24, 32), 256, 1024, True, False, True),      # layer3 1x1 256->1024: K = 1024 (32 steps -> 4)
    ((3, 17, 23), 200, 512, True, True, True),        # ragged rows and channels, shortcut gradient + ReLU mask
    ((1, 40, 64), 1024, 1408, False, False, False),   # the attention [q|k|v] projection: plain GEMM, K = 1408 (44 -> 4)
    ((2, 12, 16), 384, 288, False, True, False),      # K = 288 (9 steps -> 0): the shortest launch the split form takes
])
def test_split_data_gradient_against_float64_and_the_fp32_kernel(cfg, monkeypatch):
    """NN form: a deep-K 1x1 data gradient dX = mask(alpha (G * a_scale) W + residual) run as the forward GEMM of G with
    the transposed, scaled weights on the split-bf16 kernel (`ops.conv_dgrad`, NBM_SPLIT_BF16=1), against float64 and against the fp32
    data-gradient kernel (`igemm_nn_kernel`) of the same call: error <= 1.05 x the fp32 kernel's."""
    (B, H, W), Ci, N, use_scale, use_res, use_mask = cfg
    M = B * H * W
    g = rnd(('g', cfg), M, N).cuda()
    w = (rnd(('w', cfg), N, Ci) * (2.0 / N) ** 0.5).cuda()                 # KRSC rows of a 1x1 convolution [N][Cin]
    a_scale = (1 + 0.1 * rnd(('s', cfg), N)).cuda() if use_scale else None
    res = rnd(('r', cfg), M, Ci).cuda() if use_res else None
    mask = rnd(('m', cfg), M, Ci).cuda() if use_mask else None
    ref = (g.double() * (a_scale.double()[None, :] if use_scale else 1.0)) @ w.double()
    if use_res:
        ref = ref + res.double()
    if use_mask:
        ref = ref * (mask > 0).double()

    def run():
        out = torch.empty((M, Ci), device='cuda')
        ops.conv_dgrad(g, w, out, B=B, H=H, W=W, Cin=Ci, N=N, a_scale=a_scale, residual=res, mask=mask)
        return out
    f32, split = both(monkeypatch, run)
    assert not torch.equal(f32, split)                                     # two kernels, two summation orders
    e32, esp = (f32.double() - ref).abs(), (split.double() - ref).abs()
    assert float(esp.max()) <= 2e-5 * (1 + float(ref.abs().max())), (cfg, float(esp.max()))
    rms32, rmssp = float((e32 ** 2).mean().sqrt()), float((esp ** 2).mean().sqrt())
    assert rmssp <= 1.05 * rms32 + 1e-9, f'{cfg}: split rms error {rmssp:.3e} vs fp32 kernel {rms32:.3e}'
    if use_mask:
        assert torch.equal(split == 0, f32 == 0) or float(((split == 0) != (f32 == 0)).float().mean()) < 1e-5
    monkeypatch.setenv('NBM_SPLIT_NN', '0')                                # the switch keeps the call on the fp32 data-gradient kernel
    assert torch.equal(run(), f32)


def test_producer_mask_in_the_forward_epilogue_of_both_kernels(monkeypatch):
    """`nbm_gemm_desc.mask` (include/nbm_hip.h): y = 0 where mask <= 0, after scale / shift / residual / activation -- on the fp32 kernel
    (16-byte and scalar epilogue) and on the split kernel."""
    for (M, K, N) in ((700, 512, 256), (333, 64, 130)):                   # deep K, N % 4 == 0  |  short K, scalar epilogue (N % 4 != 0)
        x, w = rnd(('mx', M, K), M, K).cuda(), (rnd(('mw', N, K), N, K) * (1.0 / K) ** 0.5).cuda()
        res, mask = rnd(('mr', M, N), M, N).cuda(), rnd(('mm', M, N), M, N).cuda()
        ref = ((x.double() @ w.double().t()) * 0.5 + res.double()) * (mask > 0).double()

        def run():
            y = torch.empty((M, N), device='cuda')
            ops.gemm_conv(x, w, y, B=1, H=M, W=1, Cin=K, N=N, residual=res, res_ld=N, alpha=0.5, mask=mask, mask_ld=N)
            return y
        f32, split = both(monkeypatch, run)
        for got in (f32, split):
            assert float((got.double() - ref).abs().max()) <= 2e-5 * (1 + float(ref.abs().max()))
            assert torch.equal(got == 0, ref == 0) or float(((got == 0) != (ref == 0)).float().mean()) < 1e-4


@pytest.mark.parametrize('cfg', [
    # M (pixels), N (rows of dW = channels of G), K (columns = channels of X), groups, row_scale, bias
    (24 * 64 * 2, 256, 1024, 1, True, False),      # layer3 1x1 1024->256
    (3001, 200, 130 * 2, 1, True, True),           # ragged everything: pixels not a multiple of 16, N and K not multiples of the tile
    (1536 * 2, 1408, 1024, 1, False, True),        # the attention [q|k|v] projection, bias through nbm_colsum
    (700, 256, 448, 5, False, False),              # grouped (cell-domain planes): [5][M][N]^T [5][M][K]
    (40000, 384, 256, 1, False, False),            # many splits
])
def test_split_weight_gradient_against_float64_and_the_fp32_kernel(cfg, monkeypatch):
    """TN form: plain weight-gradient GEMMs dW = alpha row_scale G^T X on the bf16 matrix pipe through split fp32
    operands (csrc/igemm_split_tn.hip: pixel-major LDS image, transposed reads), accumulated into a non-zero dW, against float64 and
    against the fp32 kernel (`igemm_tn_kernel`) of the same call: error <= 1.05 x the fp32 kernel's."""
    M, N, K, groups, use_scale, use_bias = cfg
    g = rnd(('tg', cfg), groups, M, N).cuda()
    x = torch.relu(rnd(('tx', cfg), groups, M, K)).cuda()
    base = rnd(('tb', cfg), groups, N, K).cuda()
    row_scale = (1 + 0.1 * rnd(('ts', cfg), N)).cuda() if use_scale else None
    ref = torch.einsum('gmn,gmk->gnk', g.double(), x.double()) * 0.5
    if use_scale:
        ref = ref * row_scale.double()[None, :, None]
    ref = ref + base.double()
    bref = g.double().sum(1)[0] + 1.0 if use_bias else None

    def run():
        out = base.clone()
        gb = torch.ones((N,), device='cuda') if use_bias else None
        ops.conv_wgrad(g, x, out, B=1, H=M, W=1, Cin=K, N=N, groups=groups, g_gs=M * N, x_gs=M * K, out_gs=N * K, row_scale=row_scale,
                       alpha=0.5, bias_grad=gb)
        return out, gb
    (f32, gb32), (split, gbsp) = both(monkeypatch, run)
    assert not torch.equal(f32, split)
    scale = float(ref.abs().max())
    e32, esp = (f32.double() - ref).abs(), (split.double() - ref).abs()
    assert float(esp.max()) <= 3e-5 * (1 + scale), (cfg, float(esp.max()), scale)
    rms32, rmssp = float((e32 ** 2).mean().sqrt()), float((esp ** 2).mean().sqrt())
    assert rmssp <= 1.05 * rms32 + 1e-9, f'{cfg}: split rms error {rmssp:.3e} vs fp32 kernel {rms32:.3e}'
    if use_bias:
        for gb in (gb32, gbsp):
            assert float((gb.double() - bref).abs().max()) <= 1e-4 * (1 + float(bref.abs().max()))
    monkeypatch.setenv('NBM_SPLIT_TN', '0')                                # the switch keeps the call on the fp32 kernel (atomics: not bit-stable)
    again, _ = run()
    assert float((again.double() - f32.double()).abs().max()) <= 1e-5 * (1 + scale)


def test_split_data_gradient_with_groups_is_chosen_by_the_layer_not_by_the_batch(monkeypatch):
    """The attention's P V product (`ops.conv_dgrad`, one group per image): under NBM_SPLIT_BF16=1 an image's result must not depend on
    how many images share the launch -- one group and four groups take the same (split) kernel."""
    L, dv = 1536, 384
    monkeypatch.setenv('NBM_SPLIT_BF16', '1')
    pm = torch.softmax(rnd('pv-p', 4, L, L).cuda(), -1).contiguous()
    v = rnd('pv-v', 4, L, dv + 64).cuda()                                   # V is a column slice of a wider [q|k|v] matrix
    def run(b0, nb):
        out = torch.empty((nb, L, dv), device='cuda')
        ops.conv_dgrad(pm[b0:b0 + nb], v[b0:b0 + nb, :, 64:], out, B=1, H=L, W=1, Cin=dv, N=L, g_ld=L, w_ld=dv + 64, out_ld=dv, groups=nb,
                       g_gs=L * L, w_gs=L * (dv + 64), out_gs=L * dv)
        return out
    four = run(0, 4)
    for b in range(4):
        assert torch.equal(run(b, 1)[0], four[b])
    ref = torch.einsum('blm,bmc->blc', pm.double(), v[:, :, 64:].double())
    assert float((four.double() - ref).abs().max()) <= 2e-6 * (1 + float(ref.abs().max()))
