"""Fused F(2x2,3x3) kernel: 128-row against 96-row blocks (NBM_WINO_BM) on the dense launches of the detect (B = 64) and the training
(B = 128) forward pass.  usage: python scripts/wino_block_ab.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from birdsoundclassif_amd import ops
from birdsoundclassif_amd.nets import _prep

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SHAPES = [('layer2 3x3 128->128 @47x128', 47, 128, 128, 128), ('layer3 3x3 256->256 @24x64', 24, 64, 256, 256),
          ('layer4 3x3 512->512 @12x32', 12, 32, 512, 512), ('FPN 3x3 384->256 @12x32', 12, 32, 384, 256),
          ('FPN 3x3 384->256 @24x64', 24, 64, 384, 256), ('FPN 3x3 384->256 @47x128', 47, 128, 384, 256)]


def bench(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


print(f'{"launch (fused kernel only, ms; TF/s executed)":<44}' + ''.join(f'{"B=%d bm=%s" % (B, bm):>26}' for B in (64, 128) for bm in ('128', '96', 'auto')))
for label, H, W, C, N in SHAPES:
    cells = []
    for B in (64, 128):
        x = torch.relu(torch.randn(B, H, W, C, device='cuda'))
        w = torch.randn(N, C, 3, 3, device='cuda') * (2.0 / (9 * C)) ** 0.5
        b = torch.randn(N, device='cuda')
        U = _prep.wino23(w)
        TH, TW = (H + 1) // 2, (W + 1) // 2
        R, _ = ops._wino_scratch(x.device, 4 * B * TH * (2 * TW + 2) * C, 0)
        y = torch.empty(B, H, W, N, device='cuda')
        st = ops._stream()
        ops.check(ops.lib().nbm_wino23_rows(ops._ptr(x), B, H, W, C, ops._ptr(R), st), 'rows')
        gflop = 2.0 * 16 * B * TH * TW * C * N / 1e9
        ys = {}
        for bm in ('128', '96', 'auto'):
            if bm == 'auto':
                os.environ.pop('NBM_WINO_BM', None)
            else:
                os.environ['NBM_WINO_BM'] = bm
            ms = bench(lambda: ops.check(ops.lib().nbm_wino23_conv_fused(ops._ptr(R), ops._ptr(U), None, ops._ptr(b), None, 1, B, H, W, C, N,
                                                                         ops._ptr(y), 0, st), 'fused'))
            ys[bm] = y.clone()
            cells.append(f'{ms:8.3f} ({gflop / ms:5.1f})')
        assert torch.equal(ys['128'], ys['96']) and torch.equal(ys['128'], ys['auto'])
        del x, y, ys
    print(f'{label:<44}' + ''.join(f'{c:>26}' for c in cells))
