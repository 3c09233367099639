"""Backward of the dense 3x3 / stride-1 Winograd layers at B = 128: data gradient through F(4x4,3x3) (input transform + 36 grouped
GEMMs + output transform) vs F(2x2,3x3) (row transform + the fused kernel); weight gradient through F(4x4) vs F(2x2) TN GEMMs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import ops
from birdsoundclassif_amd.nets import _prep

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
shapes = [('P2', 94, 256, 384, 256), ('P3', 47, 128, 384, 256), ('P4', 24, 64, 384, 256), ('P5', 12, 32, 384, 256),
          ('layer2', 47, 128, 128, 128), ('layer3', 24, 64, 256, 256), ('layer4', 12, 32, 512, 512)]


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in ev)[n // 2]


torch.manual_seed(0)
print(f'B = {B}; ms per call (median of 5)')
print('layer    HxW      C->N     dgrad F(4x4)  dgrad F(2x2) fused   wgrad F(4x4)  wgrad F(2x2)')
for name, H, W, C, N in shapes:
    x = torch.randn(B, H, W, C, device='cuda')
    g = torch.randn(B, H, W, N, device='cuda')
    w = torch.randn(N, C, 3, 3, device='cuda') * 0.05
    mask = (torch.rand(B, H, W, C, device='cuda') > 0.4).float() if name.startswith('layer') else None
    res = []
    for m in (4, 2):
        Ut = _prep.wino23(w, transposed=True, m=m)
        res.append(timeit(lambda: ops.conv3x3_winograd(g, Ut, None, m=m, mask=mask)))
    for m in (4, 2):
        res.append(timeit(lambda: ops.conv3x3_winograd_wgrad(x, g, want_bias=True, m=m)))
    print(f'{name:8s} {H}x{W:<4d} {C}->{N:<4d}   {res[0]:8.2f}      {res[1]:8.2f}            {res[2]:8.2f}      {res[3]:8.2f}', flush=True)
    del x, g, mask
