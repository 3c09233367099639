import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import ops
def t(f, n=3):
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)
B = 128
for (H, W, Cin, N, k, pad) in ((94, 256, 256, 512, 1, 0), (94, 256, 128, 128, 3, 1)):
    Ho, Wo = (H + 2 * pad - k) // 2 + 1, (W + 2 * pad - k) // 2 + 1
    g = torch.randn(B * Ho * Wo, N, device='cuda') * 0.01
    w = torch.randn(N, k * k * Cin, device='cuda') * 0.05
    out = torch.empty(B, H, W, Cin, device='cuda')
    ms = t(lambda: ops.conv_dgrad(g, w, out, B=B, H=H, W=W, Cin=Cin, N=N, kh=k, kw=k, stride=2, pad=pad))
    print(f'phased dgrad {H}x{W} {Cin}->{N} k{k}: {ms:.2f} ms', flush=True)
    if k == 1:
        oc = torch.empty(B, Ho, Wo, Cin, device='cuda')
        ms = t(lambda: ops.conv_dgrad(g, w, oc, B=B, H=Ho, W=Wo, Cin=Cin, N=N))
        print(f'  compact GEMM only: {ms:.2f} ms;  fill of the full map: {t(lambda: out.zero_()):.2f} ms')
