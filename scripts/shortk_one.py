"""One K = 64 point-wise GEMM (64 -> 384 @188x512, B = 64), a few launches, for rocprofv3 --pmc."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import ops
B, H, W, C, N = 64, 188, 512, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 384
x = torch.randn(B, H, W, C, device='cuda'); w = torch.randn(N, C, device='cuda') * 0.1; b = torch.randn(N, device='cuda')
y = torch.empty(B, H, W, N, device='cuda')
for _ in range(3):
    ops.conv2d(x, w, shift=b, out=y)
torch.cuda.synchronize()
