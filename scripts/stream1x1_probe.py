"""ResNet layer1's 1x1 layers (and their data-gradient shapes) through nbm_gemm_conv: time + CRC of the output, to be run once
with NBM_STREAM1X1=0 (tiled kernel) and once with =1 (streaming kernel): the CRCs must agree (bit-identical sums)."""
import os, sys, zlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(0)
for (K, N, res, act, hw) in [(64, 256, True, True, (94, 256)), (64, 256, False, False, (94, 256)), (256, 64, False, True, (94, 256)), (64, 64, False, True, (94, 256)),
                            (256, 64, True, False, (94, 256)), (128, 512, True, True, (47, 128)), (256, 1024, True, True, (24, 64))]:
    x = torch.randn(B, *hw, K, device='cuda').relu_()
    w = torch.randn(N, K, device='cuda') * (2.0 / K) ** 0.5
    sc, sh = torch.rand(N, device='cuda') + 0.5, torch.randn(N, device='cuda') * 0.1
    r = torch.randn(B, *hw, N, device='cuda') if res else None
    kw = dict(scale=sc, shift=sh, residual=r, act=ops.ACT_RELU if act else ops.ACT_NONE)
    y = ops.conv2d(x, w, **kw); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
    for s, e in ev:
        s.record(); y = ops.conv2d(x, w, **kw); e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in ev)[3]
    gb = (x.numel() + y.numel() * (2 if res else 1)) * 4e-9
    print(f'{K:4d} -> {N:4d} @{hw[0]}x{hw[1]} res={int(res)} relu={int(act)} B={B}: {t:.3f} ms  {gb / t:.2f} TB/s  crc {zlib.crc32(y.cpu().numpy().tobytes()):08x}', flush=True)
