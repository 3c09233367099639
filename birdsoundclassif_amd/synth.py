"""Seeded synthetic inputs: a counter-based RNG defined here, so the GPU box and
the build container generate bit-identical waveforms, labels and filler weights without relying on
torch/NumPy generator internals.

No network => no real checkpoint (`model_weights/` holds LFS stubs) and no dataset; every parity
test and the benchmark run on these.
"""
import zlib

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform(key, n, seed=0):
    """n doubles in (0,1), a pure function of (key, seed, index)."""
    with np.errstate(over='ignore'):
        base = np.uint64(zlib.crc32(str(key).encode()) & 0xFFFFFFFF) * np.uint64(0x100000001B3) + np.uint64(seed)
        idx = np.arange(n, dtype=np.uint64)
        h = _splitmix64(_splitmix64(base) + idx)
    return ((h >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(key, n, seed=0):
    """n standard normals (Box-Muller on two independent counter streams)."""
    u1 = uniform(('n1', key), n, seed)
    u2 = uniform(('n2', key), n, seed)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


# --------------------------------------------------------------------------- waveforms
def clip_pcm16(index, n_samples=66150, sr=22050):
    """One synthetic 3 s mono clip as int16 PCM: low-level pink-ish noise plus 1-3 tones/chirps
    between 1 and 10 kHz, seed = clip index."""
    t = np.arange(n_samples) / sr
    from scipy.signal import lfilter
    white = normal(('w', index), n_samples)
    a = 0.85
    lp = lfilter([1 - a], [1, -a], white)          # one-pole low-pass => gently falling spectrum
    y = 0.03 * white + 0.06 * lp
    u = uniform(('c', index), 16)
    n_calls = 1 + int(u[0] * 3)
    for j in range(n_calls):
        f0 = 1000 + 8000 * u[1 + 4 * j]
        f1 = f0 * (0.8 + 0.5 * u[2 + 4 * j])
        t0 = 0.2 + 2.2 * u[3 + 4 * j]
        dur = 0.08 + 0.3 * u[4 + 4 * j]
        env = np.exp(-0.5 * ((t - t0) / (dur / 2.5)) ** 2)
        ph = 2 * np.pi * (f0 * (t - t0) + 0.5 * (f1 - f0) / dur * (t - t0) ** 2)
        y = y + 0.35 * env * np.sin(ph)
    y = np.clip(y, -0.999, 0.999)
    return np.round(y * 32767.0).astype(np.int16)


def clip_batch_pcm16(start, count, n_samples=66150, sr=22050):
    return np.stack([clip_pcm16(start + i, n_samples, sr) for i in range(count)])


def write_wav(path, pcm16, sr=22050):
    import wave
    with wave.open(path, 'wb') as f:
        f.setnchannels(1), f.setsampwidth(2), f.setframerate(sr)
        f.writeframes(np.asarray(pcm16, dtype='<i2').tobytes())


# --------------------------------------------------------------------------- images / labels
def image_batch(start, count, h=375, w=1024):
    """Synthetic [0,1] spectrogram-like images for detector-only tests (no front end)."""
    out = np.empty((count, h, w), dtype=np.float32)
    for i in range(count):
        base = uniform(('img', start + i), h * w).reshape(h, w)
        yy, xx = np.mgrid[0:h, 0:w]
        u = uniform(('blob', start + i), 12)
        img = 0.25 * base
        for j in range(3):
            cx, cy = 60 + 900 * u[4 * j], 30 + 300 * u[4 * j + 1]
            sx, sy = 15 + 60 * u[4 * j + 2], 8 + 30 * u[4 * j + 3]
            img = img + 0.7 * np.exp(-0.5 * (((xx - cx) / sx) ** 2 + ((yy - cy) / sy) ** 2))
        out[i] = np.clip(img, 0, 1).astype(np.float32)
    return out


def label_batch(start, count, num_classes=150, img_w=1024, img_h=375):
    """1-3 boxes per clip: x1 in [0,900], y1 in [0,300], w in [20,120], h in [20,70], class in 1..nc.
    Returns (bb_coord f32 [sum,4], bird_ids f32 [sum], lengths list) like `collate_fn`
    (reference nets_utils.py:159-166; class ids are float, Appendix C-10)."""
    boxes, ids, lengths = [], [], []
    for i in range(count):
        u = uniform(('lab', start + i), 16)
        n = 1 + int(u[0] * 3)
        for j in range(n):
            x1 = np.floor(900 * u[1 + 5 * j])
            y1 = np.floor(300 * u[2 + 5 * j])
            w = np.floor(20 + 100 * u[3 + 5 * j])
            h = np.floor(20 + 50 * u[4 + 5 * j])
            boxes.append([x1, y1, min(x1 + w, img_w - 1), min(y1 + h, img_h - 1)])
            ids.append(1 + int(u[5 + 5 * j] * num_classes) % num_classes)
        lengths.append(n)
    return (torch.tensor(boxes, dtype=torch.float32), torch.tensor(ids, dtype=torch.float32), lengths)


def write_image_dataset(root, encode_png, h=375, w=1024):
    """Synthetic `Img_dataset` directory (reference image_dataset.py:15-29 layout): two positive recordings (2 + 1
    windows, a class-0 label among them), ONE negative and ONE hard-negative image (so `np.random.choice` cannot depend
    on directory listing order).  `encode_png(uint8 [h,w]) -> bytes` is injected (the tests pass the oracle's encoder).
    Returns the list of positive file names."""
    import os
    def u8(i):
        return np.round(image_batch(i, 1, h, w)[0] * 255.0).astype(np.uint8)
    names = []
    plan = {'recA': [(0, [[40, 50, 160, 110], [500, 200, 620, 260], [800, 30, 860, 90]], [7, 0, 113]),
                     (1, [[300, 120, 420, 190]], [42])],
            'rec__B': [(0, [[100, 20, 230, 80], [640, 240, 700, 300]], [150, 3])]}
    for k, (rec, rows) in enumerate(plan.items()):
        d = os.path.join(root, 'positive_files', rec)
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, 'annotations.csv'), 'w') as f:
            f.write('index;coord;bird_id\n')
            for idx, boxes, ids in rows:
                f.write(f'{idx};{boxes};{ids}\n')
                name = f'{rec}__{idx}.png'
                with open(os.path.join(d, name), 'wb') as g:
                    g.write(encode_png(u8(500 + 10 * k + idx)))
                names.append(name)
    for sub, rec, i in (('negative_files', 'negA', 600), ('hard_neg', 'hardA', 700)):
        d = os.path.join(root, sub, rec)
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, f'{rec}__0.png'), 'wb') as g:
            g.write(encode_png(u8(i)))
    return sorted(names)


def metrics_cases(n_cases=12):
    """Deterministic (detections, ground truth) file lists for the AP metrics: jittered copies of the GT boxes (hits and
    near misses), spurious boxes, species missing on either side, unique scores.  -> list of `outputs` arguments of
    `compute_AP_scores` (reference nets_utils.py:454)."""
    cases = []
    for c in range(n_cases):
        n_files = 1 + c % 4
        outputs = []
        for f in range(n_files):
            u = uniform(('met', c, f), 400)
            k = 0
            det, gt = {}, {}
            for s in range(6):
                name = f'sp{s}'
                n_gt = int(u[k] * 4); k += 1
                boxes = []
                for _ in range(n_gt):
                    x1, y1 = np.floor(900 * u[k]), np.floor(300 * u[k + 1])
                    boxes.append([x1, y1, x1 + np.floor(20 + 100 * u[k + 2]), y1 + np.floor(15 + 50 * u[k + 3])])
                    k += 4
                if n_gt and u[k] < 0.85:
                    gt[name] = boxes
                k += 1
                preds, sc = [], []
                if u[k] < 0.75:
                    for b in boxes:
                        if u[k + 1] < 0.8:
                            j = (u[k + 2:k + 6] - 0.5) * (60 if u[k + 6] < 0.4 else 12)
                            preds.append([float(np.float32(b[i] + np.floor(j[i]))) for i in range(4)])
                            sc.append(float(np.float32(0.2 + 0.8 * u[k + 7])))
                        k += 8
                    for _ in range(int(u[k] * 3)):
                        x1, y1 = np.floor(900 * u[k + 1]), np.floor(300 * u[k + 2])
                        preds.append([float(x1), float(y1), float(x1 + 40), float(y1 + 30)])
                        sc.append(float(np.float32(0.2 + 0.8 * u[k + 3])))
                        k += 4
                k += 1
                if preds:
                    det[name] = {'bbox_coord': preds, 'scores': sc}
            outputs.append((det, gt))
        cases.append(outputs)
    cases.append([({}, {})])                                                  # nothing at all
    cases.append([({'sp0': {'bbox_coord': [[1., 2., 30., 40.]], 'scores': [0.9]}}, {})])       # only a false positive
    cases.append([({}, {'sp1': [[1., 2., 30., 40.], [5., 5., 9., 9.]]})])                      # only misses
    return cases


def annotation_text(seed=0, n=7):
    """An Audacity spectral-label file like the reference's test annotations (`t0\\tt1\\tspecies` / `\\\\\\tf0\\tf1`)."""
    u = uniform(('annot', seed), 5 * n)
    lines = []
    for i in range(n):
        t0 = 170.0 * u[5 * i]
        f0 = 300.0 + 9000.0 * u[5 * i + 2]
        lines.append(f'{t0:.6f}\t{t0 + 0.05 + 2.0 * u[5 * i + 1]:.6f}\tsp{int(u[5 * i + 4] * 3)}\n')
        lines.append(f'\\\t{f0:.6f}\t{f0 + 200.0 + 6000.0 * u[5 * i + 3]:.6f}\n')
    return ''.join(lines)


def label_rows(seed=0, n=40, filename='recA', duration=42.0):
    """Rows (t_start, t_end, f_start, f_end, species, filename, bird_id) of a synthetic annotation table: calls of
    0.05-4 s between 200 Hz and 15 kHz (some outside the image band, some degenerate), a few spanning several windows,
    noise labels (-1) and rows of another file."""
    u = uniform(('labels', seed), 8 * n)
    rows = []
    for i in range(n):
        t0 = duration * u[8 * i]
        dur = 0.05 + 4.0 * u[8 * i + 1] ** 3 + (9.0 if u[8 * i + 6] > 0.93 else 0.0)
        f0 = 200.0 + 13000.0 * u[8 * i + 2]
        bw = 0.0 if u[8 * i + 7] > 0.95 else 30.0 + 5000.0 * u[8 * i + 3]
        bird = -1 if u[8 * i + 4] > 0.85 else 1 + int(u[8 * i + 5] * 150) % 150
        rows.append((round(t0, 6), round(t0 + dur, 6), round(f0, 3), round(f0 + bw, 3), f'sp{bird}',
                     filename if u[8 * i + 6] > 0.1 else 'other_file', bird))
    return rows


# --------------------------------------------------------------------------- filler weights
def fill_state_dict(shapes, seed=0):
    """Deterministic filler for a {name: shape} mapping (SURVEY Appendix B layout): conv/linear
    weights ~ N(0, gain/fan_in), BatchNorm weight ~ 1, running_var ~ 1, small biases/means; the last
    norm of every residual branch is damped so activations stay O(1) through 16 bottlenecks.
    Returns {name: torch.float32 tensor} (int64 for num_batches_tracked)."""
    out = {}
    for name, shape in shapes.items():
        shape = tuple(shape)
        n = int(np.prod(shape)) if len(shape) else 1
        if name.endswith('num_batches_tracked'):
            out[name] = torch.zeros((), dtype=torch.int64)
            continue
        leaf = name.rsplit('.', 1)[-1]
        parent = name.rsplit('.', 1)[0]
        is_norm = ('.bn' in name or '.norm' in name or 'downsample.1' in name)
        if is_norm:
            if leaf == 'weight':
                scale = 0.35 if parent.endswith('bn3') else 1.0
                v = scale * (1.0 + 0.1 * normal(name, n, seed))
            elif leaf == 'bias':
                v = 0.05 * normal(name, n, seed)
            elif leaf == 'running_mean':
                v = 0.05 * normal(name, n, seed)
            else:  # running_var
                v = 0.9 + 0.2 * uniform(name, n, seed)
        elif leaf == 'weights':                          # BiFPN fusion weights (ReLU'd): mostly positive, some clipped
            v = 0.8 + 0.6 * normal(name, n, seed)
        elif leaf == 'bias':
            v = 0.02 * normal(name, n, seed)
            if 'bbox_classif_layer' in name:
                v = 0.5 * normal(name, n, seed)
                v[0] += 2.5                              # a fair share of RoIs should come out as background
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            gain = 2.0
            if 'init_conv' in name:
                v = 1.0 + 0.2 * normal(name, n, seed)
                out[name] = torch.from_numpy(v.astype(np.float32).reshape(shape))
                continue
            if 'attention_modules' in name:
                gain = 1.0
            if 'final_projection' in name:
                gain = 0.25
            if 'fpn.pt_wise' in name:
                gain = 0.5
            if 'fpn.out_convs' in name:
                gain = 0.1
            if 'cls_score' in name:
                gain = 1.5
            if 'bbox_reg' in name:
                gain = 0.04
            if 'bbox_classif_layer' in name:
                gain = 12.0
            v = np.sqrt(gain / fan_in) * normal(name, n, seed)
        out[name] = torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(shape))
    return out


def tame_bifpn(sd, factor=0.6):
    """The filler weights drive a 2-layer BiFPN to O(50) activations: the box regression saturates and the proposal order
    becomes one big tie.  Scaling every point-wise BiFPN weight by 0.6 keeps the maps O(1..10) (fixtures and tests of the
    `--fpn bifpn` variant use this)."""
    for k in sd:
        if k.startswith('fpn.') and k.endswith('pt_wise.weight'):
            sd[k] = sd[k] * factor
    return sd
