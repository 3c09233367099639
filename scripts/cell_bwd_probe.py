"""Backward of the demand-driven finest FPN level (3x3 384 -> 256 @188x512) at B = 128, 16 sampled RoIs per image: the pattern
share of the data / weight gradient through the cell transforms (csrc/cellwino.hip) vs through the listed F(2x2,3x3) kernels."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birdsoundclassif_amd import ondemand, ops
from birdsoundclassif_amd.nets import _prep

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H, W, C, N = 188, 512, 384, 256
torch.manual_seed(0)
x = torch.randn(B, H, W, C, device='cuda')
w = torch.randn(N, C, 3, 3, device='cuda') * 0.02
b = torch.randn(N, device='cuda')
y, st = ondemand.conv3x3_winograd_lazy(x, _prep.wino23(w), b, 8)
st.keep = True
rng = np.random.default_rng(1)
x1 = rng.integers(0, 990, (B, 16)); y1 = rng.integers(0, 350, (B, 16)); bw = rng.integers(4, 30, (B, 16)); bh = rng.integers(4, 20, (B, 16))
rois = torch.from_numpy(np.stack([x1, y1, x1 + bw, y1 + bh], -1).astype(np.float32)).cuda()
fh, fw = [188, 94, 47, 24, 12], [512, 256, 128, 64, 32]
ondemand.lazy_complete(y, rois, torch.tensor([16], dtype=torch.int32, device='cuda'), list(zip(fh, fw)))
written = ~torch.isnan(y[..., 0]) if False else None
g = torch.zeros(B, H, W, N, device='cuda')
rows = torch.zeros(H, dtype=torch.bool); cols = torch.zeros(W, dtype=torch.bool)
for n_, v in ((H, rows), (W, cols)):
    for o in range((n_ - 1) // 8 + 1):
        for k in range(3):
            if 0 <= 8 * o - 1 + k < n_:
                v[8 * o - 1 + k] = True
m = (rows[:, None] & cols[None, :]).cuda()
g[:, m] = torch.randn(B, int(m.sum()), N, device='cuda')
Ut, Uc = _prep.wino23(w, transposed=True, m=2), _prep.cell_weight(w)


def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in ev)[n // 2]


res = {}
for cell in (False, True):
    def dg():
        st.vg = st.cell_gb = st.cell_gb_done = None
        return ondemand.conv3x3_winograd_dgrad_tiles(st, g, Ut, Uc if cell else None)
    res['dgrad', cell] = timeit(dg)
    st.vg = st.cell_gb = st.cell_gb_done = None
    res['wgrad', cell] = timeit(lambda: ondemand.conv3x3_winograd_wgrad_tiles(st, x, g, want_bias=True, cell=cell))
    st.vg = st.cell_gb = st.cell_gb_done = None
    def both():
        ondemand.conv3x3_winograd_dgrad_tiles(st, g, Ut, Uc if cell else None)
        ondemand.conv3x3_winograd_wgrad_tiles(st, x, g, want_bias=True, cell=cell)
    res['both', cell] = timeit(both)
a = ondemand.conv3x3_winograd_dgrad_tiles(st, g, Ut, None); st.vg = st.cell_gb = st.cell_gb_done = None
c = ondemand.conv3x3_winograd_dgrad_tiles(st, g, Ut, Uc); st.vg = st.cell_gb = st.cell_gb_done = None
print(f'B = {B}: data gradient   listed F(2x2,3x3) {res["dgrad", False]:7.2f} ms   cell transforms {res["dgrad", True]:7.2f} ms   '
      f'max |diff| {float((a - c).abs().max()):.2e} (scale {float(a.abs().max()):.2f})')
da = ondemand.conv3x3_winograd_wgrad_tiles(st, x, g, want_bias=True, cell=False)
dc = ondemand.conv3x3_winograd_wgrad_tiles(st, x, g, want_bias=True, cell=True)
ga = _prep.wino23_weight_grad(da[0], 2)
gc = _prep.wino23_weight_grad(dc[0], 2) + _prep.cell_weight_grad(dc[2])
print(f'B = {B}: weight gradient listed F(2x2,3x3) {res["wgrad", False]:7.2f} ms   cell transforms {res["wgrad", True]:7.2f} ms   '
      f'max |diff| {float((ga - gc).abs().max()):.2e} (scale {float(ga.abs().max()):.2f}); bias diff {float((da[1] - dc[1]).abs().max()):.2e}')
print(f'B = {B}: both (one backward pass)          {res["both", False]:7.2f} ms                   {res["both", True]:7.2f} ms')
