"""Training-input stage: the reference's `Img_dataset` (nbm_datasets/image_dataset.py:13-96) with the
pixel work on the device.

Same constructor, directory layout (`positive_files/<rec>/<rec>__<i>.png` + `annotations.csv`, `negative_files/`,
`hard_neg/`), `__len__` and `__getitem__ -> (img, neg_img, bboxes, bird_ids)` as the reference, and the same host RNG
call order (NumPy global generator: negative choice, gain, 4 coin flips, hard-negative choice, two mix weights, cut-off;
torch global generator: the noise field), so a seeded run draws what the reference draws.  What changes is where the
bytes are processed: a worker only inflates the PNG's zlib stream (`raw_item`); `DeviceCollate` ships the filtered
scanlines of the whole batch to the GPU where `nbm_png_unfilter_gray8`, `nbm_image_half_std_u8` and `nbm_augment_batch`
reconstruct the pixels and apply the augmentation, producing the `[img, neg_img, bb_coord, bird_ids, lengths]` batch of
`collate_fn` (nets_utils.py:159-166) with the images already resident in HBM.

`host_noise=True` draws the N(0,1) field with `torch.randn` on the host exactly like the reference (:66);
`host_noise=False` draws one seed from the torch generator and lets the device counter RNG produce the field (same
distribution, different stream, no 1.5 MB per image over PCIe).
"""
import ast
import ctypes as C
import glob
import os
import re
import struct
import zlib

import numpy as np
import pandas as pd
import torch
from torch.utils.data import Dataset

from .. import _lib, ops

FREQ_ACCURACY = 33.3
_SIG = b'\x89PNG\r\n\x1a\n'


def read_png_scanlines(path):
    """8-bit greyscale, non-interlaced PNG -> uint8 [H, W+1] filtered scanlines (filter byte + W bytes per line): the
    host half of `imageio.imread`; the reconstruction runs on the device."""
    with open(path, 'rb') as f:
        data = f.read()
    if data[:8] != _SIG:
        raise ValueError(f'{path}: not a PNG file')
    pos, parts, hdr = 8, [], None
    while pos + 8 <= len(data):
        n, kind = struct.unpack('>I4s', data[pos:pos + 8])
        if kind == b'IHDR':
            hdr = struct.unpack('>IIBBBBB', data[pos + 8:pos + 8 + n])
        elif kind == b'IDAT':
            parts.append(data[pos + 8:pos + 8 + n])
        elif kind == b'IEND':
            break
        pos += 12 + n
    if hdr is None or hdr[2:] != (8, 0, 0, 0, 0):
        raise NotImplementedError(f'{path}: only 8-bit greyscale non-interlaced PNG is supported (IHDR {hdr})')
    W, H = hdr[0], hdr[1]
    raw = np.frombuffer(zlib.decompress(b''.join(parts)), dtype=np.uint8)
    if raw.size != H * (W + 1):
        raise ValueError(f'{path}: inflated size {raw.size} != {H}x({W}+1)')
    return raw.reshape(H, W + 1)


def lowpass_curve(cutting_freq, n_rows):
    """0.5*log10(clip(|H|, 1e-9)) of the first-order analog Butterworth low-pass at the rows' frequencies
    (image_dataset.py:88-93), float64 on the host -> fp32."""
    w = 500.0 + np.arange(n_rows) * FREQ_ACCURACY
    mag = 1.0 / np.sqrt(1.0 + (w / float(cutting_freq)) ** 2)
    return (0.5 * np.log10(np.clip(mag, 1e-9, None))).astype(np.float32)


def _listing(root, sub):
    out = []
    for rec in os.listdir(os.path.join(root, sub)):
        out.extend(os.path.basename(p) for p in glob.glob(os.path.join(root, sub, rec) + '/*.png'))
    return out


def _split_name(png):
    parts = png.replace('.png', '').split('__')
    return '__'.join(parts[:-1]), parts[-1]


class Img_dataset(Dataset):
    """reference image_dataset.py:13-96."""

    def __init__(self, dataset_path, transform=False, device='cuda', host_noise=True):
        super().__init__()
        self.ds_p = dataset_path
        self.transform = transform
        self.host_noise = host_noise
        self.positive_files = _listing(dataset_path, 'positive_files')
        self.negative_files = _listing(dataset_path, 'negative_files')
        self.hard_negative_files = _listing(dataset_path, 'hard_neg')
        self._annot = {}
        self.collate = DeviceCollate(device)

    def __len__(self):
        return len(self.positive_files)

    def _annotations(self, rec):
        if rec not in self._annot:
            a = pd.read_csv(os.path.join(self.ds_p, 'positive_files', rec, 'annotations.csv'), sep=';')
            # files written by the reference under NumPy >= 2 spell scalars as `np.int64(12)` (it reads them with eval)
            lit = lambda t: ast.literal_eval(re.sub(r'np\.(?:int|float)\d+\((-?[\d.eE+-]+)\)', r'\1', t))
            self._annot[rec] = {int(i): (lit(c), lit(b))
                                for i, c, b in zip(a['index'], a['coord'], a['bird_id'])}
        return self._annot[rec]

    def raw_item(self, idx):
        """Host half of `__getitem__`: file bytes inflated, labels parsed, every random number drawn (reference call
        order), no pixel touched."""
        imgp = self.positive_files[idx]
        rec, fileidx = _split_name(imgp)
        item = {'pos': read_png_scanlines(os.path.join(self.ds_p, 'positive_files', rec, imgp)), 'transform': self.transform}
        bboxes, bird_ids = self._annotations(rec)[int(fileidx)]
        keep = np.array(bird_ids) != 0                                   # class 0 dropped, image_dataset.py:55-56
        item['bboxes'], item['bird_ids'] = torch.Tensor(bboxes)[keep], torch.Tensor(bird_ids)[keep]
        negp = np.random.choice(self.negative_files, 1)[0]                                            # :59
        item['neg'] = read_png_scanlines(os.path.join(self.ds_p, 'negative_files', _split_name(negp)[0], negp))
        if not self.transform:
            return item
        H, W = item['pos'].shape[0], item['pos'].shape[1] - 1
        if self.host_noise:
            item['noise'] = torch.randn((H, W))                                                       # :66
        else:
            item['noise_seed'] = int(torch.randint(0, 2 ** 62, (1,)).item())
        item['gain'] = np.random.uniform(-0.1, 0.35)                                                  # :68
        flips = np.random.randint(2, size=4)                                                          # :71
        item['flags'] = 0
        if flips[0] == 1:
            hardp = np.random.choice(self.hard_negative_files, 1)[0]                                  # :73
            item['hard'] = read_png_scanlines(os.path.join(self.ds_p, 'hard_neg', _split_name(hardp)[0], hardp))
            item['coef'] = np.random.uniform(0.1, 0.4)                                                # :79
            item['neg_coef'] = np.random.uniform(0.5, 0.99)                                           # :82
            item['flags'] |= 1
        if flips[1] == 1:
            item['cutting_freq'] = np.random.randint(500, 10000)                                      # :90
            item['flags'] |= 2
        return item

    def __getitem__(self, idx):
        img, neg, bb, ids, _ = self.collate([self.raw_item(idx)])
        return img[0], neg[0], bb, ids


class DeviceCollate:
    """list of `Img_dataset.raw_item` records -> [img_batch, neg_img_batch, bb_coord_batch, bird_ids, lengths]
    (collate_fn, nets_utils.py:159-166) with both image batches computed on and left on the device."""

    def __init__(self, device='cuda'):
        self.device = device

    def _upload(self, lines):
        host = torch.from_numpy(np.stack(lines))
        return host.pin_memory().to(self.device, non_blocking=True)

    def __call__(self, items):
        B = len(items)
        H, W = items[0]['pos'].shape[0], items[0]['pos'].shape[1] - 1
        hard_items = [i for i, it in enumerate(items) if 'hard' in it]
        raw = self._upload([it['pos'] for it in items] + [it['neg'] for it in items] + [items[i]['hard'] for i in hard_items])
        n_img = raw.shape[0]
        pix = torch.empty((n_img, H, W), dtype=torch.uint8, device=self.device)
        status = torch.zeros((1,), dtype=torch.int32, device=self.device)
        lib, st = ops.lib(), ops._stream()
        _lib.check(lib.nbm_png_unfilter_gray8(ops._ptr(raw), H * (W + 1), n_img, H, W, ops._ptr(pix), H * W,
                                              ops._ptr(status), st), 'nbm_png_unfilter_gray8')
        img = torch.empty((B, H, W), dtype=torch.float32, device=self.device)
        neg = torch.empty((B, H, W), dtype=torch.float32, device=self.device)
        if not items[0]['transform']:
            both = torch.empty((2 * B, H, W), dtype=torch.float32, device=self.device)
            _lib.check(lib.nbm_u8_to_unit(ops._ptr(pix), 2 * B * H * W, ops._ptr(both), st), 'nbm_u8_to_unit')
            img, neg = both[:B], both[B:]
        else:
            half_std = torch.empty((B,), dtype=torch.float32, device=self.device)
            _lib.check(lib.nbm_image_half_std_u8(ops._ptr(pix), H * W, B, H * W, ops._ptr(half_std), st),
                       'nbm_image_half_std_u8')
            prm = (_lib.AugmentParams * B)()
            curve = np.zeros((B, H), dtype=np.float32)
            for i, it in enumerate(items):
                p = prm[i]
                p.gain, p.flags = it['gain'], it['flags']
                if it['flags'] & 1:
                    p.coef, p.denom = it['coef'], 1 + it['coef']
                    p.neg_coef, p.neg_denom = it['neg_coef'], 1 + it['neg_coef']
                    p.hard_index = hard_items.index(i)
                if it['flags'] & 2:
                    curve[i] = lowpass_curve(it['cutting_freq'], H)
                p.noise_seed = it.get('noise_seed', 0)
            prm_dev = torch.frombuffer(bytearray(bytes(prm)), dtype=torch.uint8).to(self.device)
            curve_dev = torch.from_numpy(curve).to(self.device)
            noise = None
            if 'noise' in items[0]:
                noise = torch.stack([it['noise'] for it in items]).pin_memory().to(self.device, non_blocking=True)
            _lib.check(lib.nbm_augment_batch(ops._ptr(pix), ops._ptr(pix[B:]), ops._ptr(pix[2 * B:]) if hard_items else None,
                                             B, H, W, ops._ptr(prm_dev), ops._ptr(half_std), ops._ptr(noise),
                                             ops._ptr(curve_dev), ops._ptr(img), ops._ptr(neg), st), 'nbm_augment_batch')
        if int(status.item()):
            raise ValueError(f'corrupt PNG: bad filter byte on scanline {int(status.item()) - 1}')
        lengths = [len(it['bboxes']) for it in items]
        return [img, neg, torch.cat([it['bboxes'] for it in items], dim=0), torch.cat([it['bird_ids'] for it in items]),
                lengths]
