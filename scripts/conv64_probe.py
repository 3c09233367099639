"""Forward 3x3 64 -> 64 @94x256 (ResNet layer1) and the other 64-wide implicit-GEMM launches: two LDS stages / two workgroups per CU
against one stage / three (NBM_S1_N64 = largest K / 32 that takes the single-stage kernel).  usage: python scripts/conv64_probe.py [B]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
torch.manual_seed(0)
for (H, W, Cin, N, k, stride) in ((94, 256, 64, 64, 3, 1), (188, 512, 64, 64, 3, 2)):
    x = torch.relu(torch.randn(B, H, W, Cin, device='cuda'))
    w = torch.randn(N, k * k * Cin, device='cuda') * 0.05
    sc, sh = torch.rand(N, device='cuda') + 0.5, torch.randn(N, device='cuda')
    y = ops.conv2d(x, w, k, k, stride, k // 2, scale=sc, shift=sh, act=ops.ACT_RELU)
    ms = t(lambda: ops.conv2d(x, w, k, k, stride, k // 2, scale=sc, shift=sh, act=ops.ACT_RELU, out=y))
    gf = 2.0 * y.numel() * k * k * Cin / 1e9
    print(f'fwd {k}x{k} s{stride} {Cin}->{N} @{H}x{W} B={B}: {ms:.3f} ms  {gf / ms:.1f} TF/s  checksum {float(y.double().sum()):.10e}', flush=True)
