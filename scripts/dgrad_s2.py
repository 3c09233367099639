"""Data gradients of the stride-2 convolutions of a B = 128 training step (the 3x3 of the first block of layers 2-4), igemm_nn_kernel
grouped by parity class.  usage: [NBM_NN_SHORTK_PHASED=n] python scripts/dgrad_s2.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import ops
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
B = 128
outs = []
for (H, W, Cin, N, k, pad) in ((94, 256, 128, 128, 3, 1), (47, 128, 256, 256, 3, 1), (24, 64, 512, 512, 3, 1)):
    Ho, Wo = (H + 2 * pad - k) // 2 + 1, (W + 2 * pad - k) // 2 + 1
    g = torch.randn(B * Ho * Wo, N, device='cuda') * 0.01
    w = torch.randn(N, k * k * Cin, device='cuda') * 0.05
    out = torch.empty(B, H, W, Cin, device='cuda')
    sc = torch.rand(N, device='cuda') + 0.5
    mask = torch.randn(B, H, W, Cin, device='cuda')
    ms = t(lambda: ops.conv_dgrad(g, w, out, B=B, H=H, W=W, Cin=Cin, N=N, kh=k, kw=k, stride=2, pad=pad, a_scale=sc, mask=mask))
    gf = 2.0 * B * Ho * Wo * N * k * k * Cin / 1e9
    print(f'phased dgrad {H}x{W} {Cin}->{N} k{k} s2 (+ a_scale + mask): {ms:.3f} ms  {gf / ms:.1f} TF/s  checksum {float(out.double().sum()):.6e}', flush=True)
