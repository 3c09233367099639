"""TEST INFRASTRUCTURE ONLY (oracle): CPU restatement of `Img_dataset.__getitem__`'s augmentation arithmetic
(reference nbm_datasets/image_dataset.py:36-96) in torch fp32, with every random draw passed in explicitly.

Pinned by `oracle/make_golden.py:img_dataset_golden`, which runs the REAL reference class (imported with an
`imageio.v2` stand-in that decodes through oracle/png_ref.py) on a synthetic dataset directory under seeded NumPy /
torch generators and stores its outputs in tests/golden/img_dataset.npz.
"""
import numpy as np
import torch

FREQ_ACCURACY = 33.3          # image_dataset.py:89


def lowpass_curve(cutting_freq, n_rows=375):
    """image_dataset.py:90-93: first-order analog Butterworth low-pass |H(j w)| sampled at the image rows' frequencies,
    as the additive term 0.5*log10(clip(|H|, 1e-9)).  (scipy.signal.butter(1, wc, analog=True) is H(s) = wc/(s+wc).)"""
    wc = 2.0 * np.pi * float(cutting_freq)
    w = 2.0 * np.pi * (500.0 + np.arange(n_rows) * FREQ_ACCURACY)
    mag = np.abs(wc / (1j * w + wc))
    return torch.Tensor(0.5 * np.log10(np.clip(mag, 1e-9, None)))


def to_float(img_u8):
    """image_dataset.py:44-45: `torch.Tensor(img / 255)` -- float64 quotient rounded to fp32."""
    return torch.Tensor(np.asarray(img_u8) / 255)


def augment(img_u8, neg_u8, hard_u8, noise_unit, gain, flags, coef, neg_coef, cutting_freq):
    """-> (img, neg_img) fp32.  `noise_unit` = the N(0,1) field of image_dataset.py:66 before scaling; `flags` =
    the 4 coin flips of :71 (only [0] hard-negative mix and [1] low-pass are live)."""
    img, neg = to_float(img_u8), to_float(neg_u8)
    noise = torch.clamp(noise_unit.clone().mul_(img.std().item() / 2), min=-0.5, max=0.5)     # :66
    img += gain                                                                                # :68
    img += noise                                                                               # :69
    if flags[0] == 1:                                                                          # :72-83
        hard = to_float(hard_u8)
        img = (img + coef * hard) / (1 + coef)
        neg = (neg + neg_coef * hard) / (1 + neg_coef)
    if flags[1] == 1:                                                                          # :86-94
        img = img + lowpass_curve(cutting_freq, img.shape[0])[:, None]
    return img, neg


def getitem(root, name, negative_files, hard_files, transform):
    """`Img_dataset.__getitem__` (image_dataset.py:36-96) for the positive file `name`, drawing from the global NumPy /
    torch generators in the reference's order.  -> (img, neg_img, bboxes, bird_ids)."""
    import ast
    import csv
    import os
    from . import png_ref

    def split(png):
        parts = png.replace('.png', '').split('__')
        return '__'.join(parts[:-1]), parts[-1]

    def read(sub, png):
        with open(os.path.join(root, sub, split(png)[0], png), 'rb') as f:
            return png_ref.decode_png_gray8(f.read())

    rec, fileidx = split(name)
    img_u8 = read('positive_files', name)
    with open(os.path.join(root, 'positive_files', rec, 'annotations.csv')) as f:
        rows = {int(r['index']): r for r in csv.DictReader(f, delimiter=';')}
    bboxes, ids = ast.literal_eval(rows[int(fileidx)]['coord']), ast.literal_eval(rows[int(fileidx)]['bird_id'])
    keep = np.array(ids) != 0                                                                  # :55-56
    bboxes, ids = torch.Tensor(bboxes)[keep], torch.Tensor(ids)[keep]
    neg_u8 = read('negative_files', np.random.choice(negative_files, 1)[0])                    # :59
    if not transform:
        return to_float(img_u8), to_float(neg_u8), bboxes, ids
    noise_unit = torch.randn(img_u8.shape)                                                     # :66
    gain = np.random.uniform(-0.1, 0.35)                                                       # :68
    flags = np.random.randint(2, size=4)                                                       # :71
    hard_u8 = coef = neg_coef = cut = None
    if flags[0] == 1:
        hard_u8 = read('hard_neg', np.random.choice(hard_files, 1)[0])                         # :73
        coef = np.random.uniform(0.1, 0.4)                                                     # :79
        neg_coef = np.random.uniform(0.5, 0.99)                                                # :82
    if flags[1] == 1:
        cut = np.random.randint(500, 10000)                                                    # :90
    img, neg = augment(img_u8, neg_u8, hard_u8, noise_unit, gain, flags, coef, neg_coef, cut)
    return img, neg, bboxes, ids
