"""One variant of the fused Winograd kernel, a few launches (for rocprofv3 --pmc): python wino_fused_one.py VARIANT [B]."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import ops
from birdsoundclassif_amd.nets import _prep
var = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 26
H, W, C, N = 188, 512, 384, 256
torch.manual_seed(0)
x = torch.relu(torch.randn(B, H, W, C, device='cuda'))
w = torch.randn(N, C, 3, 3, device='cuda') * (2.0 / (9 * C)) ** 0.5
b = torch.randn(N, device='cuda')
U = _prep.wino23(w)
TH, TW = (H + 1) // 2, (W + 1) // 2
R, _ = ops._wino_scratch(x.device, 4 * B * TH * (2 * TW + 2) * C, 0)
y = torch.empty(B, H, W, N, device='cuda')
st = ops._stream()
ops.check(ops.lib().nbm_wino23_rows(ops._ptr(x), B, H, W, C, ops._ptr(R), st), 'rows')
for _ in range(3):
    ops.check(ops.lib().nbm_wino23_conv_fused(ops._ptr(R), ops._ptr(U), None, ops._ptr(b), None, 0, B, H, W, C, N,
                                              ops._ptr(y), var, st), 'fused')
torch.cuda.synchronize()
