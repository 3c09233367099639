"""Probe of the short-K 1x1 GEMM (fpn.pt_wise.0: 64 -> 384 @188x512, B = 64): time vs N and vs the kernel variant, then the
same layer inside the model (with the fused top-down merge epilogue) and on its captured input without it.
Round-2 reading: 4.1 ms without the merge, 6.6 ms with it (4 gathered loads per 16 output bytes); moving the merge into the
Winograd row transform of the consumer was measured and dropped: the lateral GEMM gains 2.8 ms, the row transform loses 2.8."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import ops
B, H, W, C = 64, 188, 512, 64
x = torch.randn(B, H, W, C, device='cuda')
for N in (384, 256, 128, 512):
    w = torch.randn(N, C, device='cuda') * 0.1
    b = torch.randn(N, device='cuda')
    y = torch.empty(B, H, W, N, device='cuda')
    ts = []
    for _ in range(4):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); ops.conv2d(x, w, shift=b, out=y); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    t = min(ts[1:])
    gb = 4.0 * B * H * W * (C + N) / 1e9
    print(f'N={N}: {t:.3f} ms  {gb / t * 1e3:.0f} GB/s  {2.0 * B * H * W * C * N / t / 1e9:.1f} TF/s  (NBM_SHORTK_MAX={os.environ.get("NBM_SHORTK_MAX", "default")})', flush=True)
# plain device copy of the same number of bytes for reference
src = torch.empty(int(4.0 * B * H * W * (C + 384) / 8), device='cuda'); dst = torch.empty_like(src)
for _ in range(3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); dst.copy_(src); e.record(); torch.cuda.synchronize()
print(f'copy of the same bytes: {s.elapsed_time(e):.3f} ms')

# the same layer inside the model and right after it, on the model's own tensors
import numpy as np
from birdsoundclassif_amd import synth
from birdsoundclassif_amd.nets import build_model, functional as Fn
from birdsoundclassif_amd.train import default_args
model, _ = build_model(default_args(device='cuda'))
model.load_state_dict(synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}))
model = model.cuda().eval()
xi = torch.from_numpy(np.tile(synth.image_batch(0, 8), (8, 1, 1))).cuda()[:, None]
with torch.no_grad():
    model.detect(xi); torch.cuda.synchronize()
    ops.PROFILE = []
    model.detect(xi); torch.cuda.synchronize()
    prof, ops.PROFILE = ops.PROFILE, None
    for tag, s, e in prof:
        if tag[:5] == (64, 384, 1, 188, 512):
            print('in model:', tag, f'{s.elapsed_time(e):.3f} ms')
    # capture the real input of fpn.pt_wise.0 by wrapping Fn.conv (no private layout assumptions)
    cap = {}
    real_conv = Fn.conv
    def spy(x, w, *a, **k):
        if tuple(x.shape[1:]) == (188, 512, 64) and w.shape[0] == 384:
            cap['x'], cap['k'] = x, {kk: (vv if not torch.is_tensor(vv) else 'tensor') for kk, vv in k.items()}
        return real_conv(x, w, *a, **k)
    Fn.conv = spy
    model.detect(xi); torch.cuda.synchronize()
    Fn.conv = real_conv
    tt = cap['x']
    print('captured input', tuple(tt.shape), tt.stride(), tt.is_contiguous(), cap['k'])
    c = model.fpn.pt_wise['0']
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); y = Fn.conv(tt, c.weight, bias=c.bias, alpha=2.0); e.record(); torch.cuda.synchronize()
        print(f'Fn.conv on the captured input: {s.elapsed_time(e):.3f} ms  min {float(tt.min()):.3g} max {float(tt.max()):.3g} zeros {float((tt == 0).float().mean()):.3f}')
    z = torch.randn_like(tt)
    for _ in range(2):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); y = Fn.conv(z, c.weight, bias=c.bias, alpha=2.0); e.record(); torch.cuda.synchronize()
        print(f'Fn.conv on randn of the same shape: {s.elapsed_time(e):.3f} ms')
    w2 = torch.randn_like(c.weight) * 0.1
    for _ in range(2):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); y = Fn.conv(tt, w2, bias=c.bias, alpha=2.0); e.record(); torch.cuda.synchronize()
        print(f'Fn.conv on the captured input with random weights: {s.elapsed_time(e):.3f} ms')
