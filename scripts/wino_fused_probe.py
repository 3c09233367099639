"""GPU probe of the fused Winograd kernel (csrc/wino_fused.hip): error vs a float64 convolution on odd sizes with the
whole epilogue, then time per variant on the FPN shapes (HIP events, interleaved rounds)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import ops
from birdsoundclassif_amd.nets import _prep

torch.manual_seed(0)
for (B, H, W, C, N) in ((2, 47, 127, 64, 128), (3, 24, 64, 384, 256), (1, 12, 33, 96, 64)):
    x = torch.relu(torch.randn(B, H, W, C, device='cuda'))
    w = torch.randn(N, C, 3, 3, device='cuda') * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(N, device='cuda'); sc = torch.rand(N, device='cuda') + 0.5
    mask = (torch.rand(B, H, W, N, device='cuda') > 0.3).float()
    U = _prep.wino23(w)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), None, padding=1)
    ref = torch.relu(ref * sc.double().cpu().view(1, -1, 1, 1) + b.double().cpu().view(1, -1, 1, 1)) * mask.permute(0, 3, 1, 2).double().cpu()
    for var in (128, 64):
        ops.WINO_FUSED_VARIANT = var
        y = ops.conv3x3_winograd(x, U, b, scale=sc, relu=True, mask=mask)
        err = (y.permute(0, 3, 1, 2).double().cpu() - ref).abs().max().item()
        print(f'B{B} {H}x{W} C{C} N{N} variant {var}: max err {err:.2e} (scale {ref.abs().max().item():.2f})', flush=True)
        assert err < 2e-5

B = int(sys.argv[1]) if len(sys.argv) > 1 else 26
for (H, W, C, N) in ((188, 512, 384, 256), (94, 256, 384, 256)):
    x = torch.relu(torch.randn(B, H, W, C, device='cuda'))
    w = torch.randn(N, C, 3, 3, device='cuda') * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(N, device='cuda')
    U = _prep.wino23(w)
    TH, TW = (H + 1) // 2, (W + 1) // 2
    tiles = TH * TW * B
    R, _ = ops._wino_scratch(x.device, 4 * B * TH * (2 * TW + 2) * C, 0)
    y = torch.empty(B, H, W, N, device='cuda')
    st = ops._stream()
    ops.check(ops.lib().nbm_wino23_rows(ops._ptr(x), B, H, W, C, ops._ptr(R), st), 'rows')
    flop = 2.0 * 16 * tiles * C * N
    VARS = (128, 64, 1128, 2128, 3128, 7128)
    res = {v: [] for v in VARS}
    for rnd in range(4):
        for var in VARS:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ops.check(ops.lib().nbm_wino23_conv_fused(ops._ptr(R), ops._ptr(U), None, ops._ptr(b), None, 0, B, H, W, C, N,
                                                      ops._ptr(y), var, st), 'fused')
            e.record(); torch.cuda.synchronize()
            res[var].append(s.elapsed_time(e))
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); ops.check(ops.lib().nbm_wino23_rows(ops._ptr(x), B, H, W, C, ops._ptr(R), st), 'rows'); e.record(); torch.cuda.synchronize()
    t_in = s.elapsed_time(e)
    for var in VARS:
        t = min(res[var][1:])
        print(f'{H}x{W} B={B} variant {var}: fused GEMM+output {t:.2f} ms = {flop / t / 1e9:.1f} TFLOP/s executed '
              f'(all rounds {["%.2f" % v for v in res[var]]}); row transform {t_in:.2f} ms', flush=True)
