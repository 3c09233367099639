"""The bottleneck's first data gradient (G P channels -> dX 4P channels + shortcut gradient + ReLU mask) with the mask read from y (fp32)
against the mask as bits, B = 128.  usage: python scripts/relu_bits_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import ops
B = 128
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for label, H, W, C, P in (('layer1 256->64', 94, 256, 256, 64), ('layer2 512->128', 47, 128, 512, 128), ('layer3 1024->256', 24, 64, 1024, 256)):
    y = torch.relu(torch.randn(B, H, W, C, device='cuda'))
    bits = torch.zeros(B * H * W * C // 32, device='cuda', dtype=torch.int32)
    pos = torch.tensor([8 * (ch % 4) + ch // 4 for ch in range(32)], device='cuda')
    packed = ((y.view(-1, 32) > 0).to(torch.int64) << pos).sum(1)
    bits.copy_(torch.where(packed >= 2 ** 31, packed - 2 ** 32, packed).to(torch.int32))
    del packed
    g = torch.randn(B * H * W, P, device='cuda') * 0.1
    wk = torch.randn(P, C, device='cuda') * 0.05
    sc = torch.rand(P, device='cuda') + 0.5
    short = torch.randn(B, H, W, C, device='cuda')
    out = torch.empty(B, H, W, C, device='cuda')
    res = []
    for kw in (dict(mask=y), dict(mask=y, mask_bits=bits), dict()):
        ms = t(lambda: ops.conv_dgrad(g, wk, out, B=B, H=H, W=W, Cin=C, N=P, g_ld=P, w_ld=C, a_scale=sc, residual=short, **kw))
        res.append(ms)
    gb = (g.numel() + out.numel() * 3) * 4 / 1e9
    print(f'{label:<18} mask from y {res[0]:.3f} ms ({gb / res[0] * 1e3 / 1e3:.2f} TB/s of {gb:.2f} GB)   mask as bits {res[1]:.3f} ms   no mask {res[2]:.3f} ms', flush=True)

# the producers: the bottleneck's last convolution (P -> 4P + shortcut + ReLU) without / with the bits output
for label, H, W, P in (('layer1 64->256', 94, 256, 64), ('layer2 128->512', 47, 128, 128), ('layer3 256->1024', 24, 64, 256), ('layer4 512->2048', 12, 32, 512)):
    C = 4 * P
    x = torch.relu(torch.randn(B, H, W, P, device='cuda'))
    w = torch.randn(C, P, device='cuda') * 0.05
    sc, sh = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')
    res = torch.randn(B, H, W, C, device='cuda')
    y = torch.empty(B, H, W, C, device='cuda')
    bits = torch.empty(B * H * W * C // 32, device='cuda', dtype=torch.int32)
    t0 = t(lambda: ops.conv2d(x, w, scale=sc, shift=sh, residual=res, act=ops.ACT_RELU, out=y))
    t1 = t(lambda: ops.conv2d(x, w, scale=sc, shift=sh, residual=res, act=ops.ACT_RELU, out=y, bits_out=bits))
    print(f'{label:<18} forward {t0:.3f} ms   with bits_out {t1:.3f} ms', flush=True)

# the inner mask of layer1: 3x3 64 -> 64 (+ BN + ReLU) writes a2 (+ bits); conv3's data gradient (G 256 -> dX 64) masks by a2
H, W, P = 94, 256, 64
x = torch.relu(torch.randn(B, H, W, P, device='cuda'))
w = torch.randn(P, 9 * P, device='cuda') * 0.05
sc, sh = torch.rand(P, device='cuda') + 0.5, torch.randn(P, device='cuda')
y = torch.empty(B, H, W, P, device='cuda')
bits = torch.empty(B * H * W * P // 32, device='cuda', dtype=torch.int32)
t0 = t(lambda: ops.conv2d(x, w, 3, 3, 1, 1, scale=sc, shift=sh, act=ops.ACT_RELU, out=y))
t1 = t(lambda: ops.conv2d(x, w, 3, 3, 1, 1, scale=sc, shift=sh, act=ops.ACT_RELU, out=y, bits_out=bits))
print(f'layer1 3x3 64->64  forward {t0:.3f} ms   with bits_out {t1:.3f} ms', flush=True)
g = torch.randn(B * H * W, 4 * P, device='cuda') * 0.1
wk = torch.randn(4 * P, P, device='cuda') * 0.05
s3 = torch.rand(4 * P, device='cuda') + 0.5
out = torch.empty(B, H, W, P, device='cuda')
t0 = t(lambda: ops.conv_dgrad(g, wk, out, B=B, H=H, W=W, Cin=P, N=4 * P, g_ld=4 * P, w_ld=P, a_scale=s3, mask=y))
t1 = t(lambda: ops.conv_dgrad(g, wk, out, B=B, H=H, W=W, Cin=P, N=4 * P, g_ld=4 * P, w_ld=P, a_scale=s3, mask=y, mask_bits=bits))
print(f'layer1 64->256 data gradient (G 256 -> dX 64): mask from a2 {t0:.3f} ms   as bits {t1:.3f} ms', flush=True)
