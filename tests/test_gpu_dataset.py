"""GPU parity, training-input stage: device PNG reconstruction and the batch augmentation kernel
behind the reference's `Img_dataset` API, against the oracle and the REAL reference's outputs (img_dataset.npz)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from birdsoundclassif_amd import synth                                     # noqa: E402
from helpers import check_packed, load_golden                              # noqa: E402
from oracle import dataset_ref, png_ref                                    # noqa: E402


def _unfilter(raw_list):
    from birdsoundclassif_amd import _lib, ops
    raw = torch.from_numpy(np.stack(raw_list)).cuda()
    n, H, W1 = raw.shape
    out = torch.empty((n, H, W1 - 1), dtype=torch.uint8, device='cuda')
    status = torch.zeros((1,), dtype=torch.int32, device='cuda')
    _lib.check(ops.lib().nbm_png_unfilter_gray8(ops._ptr(raw), H * W1, n, H, W1 - 1, ops._ptr(out), H * (W1 - 1),
                                                ops._ptr(status), ops._stream()), 'unfilter')
    return out.cpu().numpy(), int(status.item())


def _scanlines(png_bytes):
    import zlib
    W, H, payload = png_ref.parse_png(png_bytes)
    return np.frombuffer(zlib.decompress(payload), dtype=np.uint8).reshape(H, W + 1).copy()


def test_png_unfilter_bit_exact():
    imgs = [np.round(synth.image_batch(9, 1)[0] * 255).astype(np.uint8),
            (np.arange(375 * 1024).reshape(375, 1024) * 37 % 256).astype(np.uint8),
            (synth.uniform('pngnoise', 375 * 1024) * 256).astype(np.uint8).reshape(375, 1024)]
    filt = [None, np.full(375, 4), (np.arange(375) * 3) % 5]
    raws = [_scanlines(png_ref.encode_png_gray8(im, f)) for im, f in zip(imgs, filt)]
    got, status = _unfilter(raws)
    assert status == 0
    for g, im in zip(got, imgs):
        assert np.array_equal(g, im)
    # other geometry (H not a multiple of the wave size, W odd)
    small = (synth.uniform('pngsmall', 70 * 33) * 256).astype(np.uint8).reshape(70, 33)
    got, status = _unfilter([_scanlines(png_ref.encode_png_gray8(small))])
    assert status == 0 and np.array_equal(got[0], small)
    bad = raws[0].copy()
    bad[17, 0] = 9
    assert _unfilter([bad])[1] == 18


def test_half_std_and_counter_randn():
    from birdsoundclassif_amd import _lib, ops
    u8 = torch.from_numpy((synth.uniform('std', 3 * 375 * 1024) ** 2 * 256).astype(np.uint8).reshape(3, 375, 1024)).cuda()
    hs = torch.empty((3,), device='cuda')
    _lib.check(ops.lib().nbm_image_half_std_u8(ops._ptr(u8), 375 * 1024, 3, 375 * 1024, ops._ptr(hs), ops._stream()), 'std')
    ref = torch.stack([dataset_ref.to_float(u8[i].cpu().numpy()).std() / 2 for i in range(3)])
    assert (hs.cpu() - ref).abs().max() < 1e-7
    n = 1 << 22
    z = torch.empty((n,), device='cuda')
    _lib.check(ops.lib().nbm_randn_fill(1234, n, ops._ptr(z), ops._stream()), 'randn')
    z2 = torch.empty((n,), device='cuda')
    _lib.check(ops.lib().nbm_randn_fill(1235, n, ops._ptr(z2), ops._stream()), 'randn')
    zd = z.double()
    assert abs(zd.mean().item()) < 3e-3 and abs(zd.var().item() - 1) < 5e-3
    assert abs((zd ** 3).mean().item()) < 1e-2 and abs((zd ** 4).mean().item() - 3) < 3e-2
    assert abs((zd[:-1] * zd[1:]).mean().item()) < 3e-3 and abs((zd * z2.double()).mean().item()) < 3e-3
    assert torch.isfinite(z).all() and z.abs().max() < 6.5


def test_img_dataset_vs_reference_golden(tmp_path):
    """Same seeds, same visiting order as oracle/make_golden.py: the product must reproduce the reference's items."""
    from birdsoundclassif_amd.nbm_datasets.image_dataset import Img_dataset
    g = load_golden('img_dataset.npz')
    names = synth.write_image_dataset(str(tmp_path), png_ref.encode_png_gray8)
    for transform in (False, True):
        ds = Img_dataset(str(tmp_path), transform=transform)
        assert len(ds) == 3
        for seed in range(4 if transform else 1):
            np.random.seed(100 + seed)
            torch.manual_seed(100 + seed)
            for name in names:
                img, neg, bb, ids = ds[ds.positive_files.index(name)]
                assert img.is_cuda and img.shape == (375, 1024)
                key = f't{int(transform)}.s{seed}.{name}'
                check_packed(g, key + '.img', img, atol=5e-7)
                check_packed(g, key + '.neg', neg, atol=5e-7)
                assert np.array_equal(bb.numpy(), g[key + '.bboxes']) and np.array_equal(ids.numpy(), g[key + '.bird_ids'])


def test_device_collate_batch_equals_items_and_device_noise(tmp_path):
    from birdsoundclassif_amd.nbm_datasets.image_dataset import Img_dataset
    synth.write_image_dataset(str(tmp_path), png_ref.encode_png_gray8)
    ds = Img_dataset(str(tmp_path), transform=True)
    np.random.seed(5), torch.manual_seed(5)
    items = [ds.raw_item(i) for i in (0, 1, 2, 1, 0)]
    img, neg, bb, ids, lengths = ds.collate(items)
    assert img.shape == (5, 375, 1024) and sum(lengths) == len(bb) == len(ids)
    assert any(it['flags'] & 1 for it in items) and any(it['flags'] & 2 for it in items)
    for i, it in enumerate(items):
        one = ds.collate([it])
        assert torch.equal(one[0][0], img[i]) and torch.equal(one[1][0], neg[i])
    # device counter noise: same law as the host field (mean, spread), clamp respected, train.step-compatible batch
    ds2 = Img_dataset(str(tmp_path), transform=True, host_noise=False)
    np.random.seed(5), torch.manual_seed(5)
    items2 = [ds2.raw_item(i) for i in (0, 1, 2, 1, 0)]
    img2 = ds2.collate(items2)[0]
    for i in range(5):
        assert items2[i]['flags'] == items[i]['flags'] and items2[i]['gain'] == items[i]['gain']
        d = (img2[i] - img[i]).double()
        assert abs(d.mean().item()) < 2e-3 and abs(img2[i].double().std().item() - img[i].double().std().item()) < 2e-3


def test_prepare_dataset_roundtrip(tmp_path):
    """wav + Audacity labels -> prepare_dataset (device front end, PNG + annotations.csv) -> Img_dataset (device PNG
    reconstruction): the images are the 8-bit rounding of the front end's windows and the boxes are the merged labels."""
    import pandas as pd
    from birdsoundclassif_amd.nbm_datasets.image_dataset import Img_dataset
    from birdsoundclassif_amd.nbm_datasets.prepare_dataset import (File_Processor, create_label_dataset, prepare_dataset)
    src = tmp_path / 'site1'
    src.mkdir()
    pcm = np.concatenate([synth.clip_pcm16(20 + i) for i in range(3)])             # 9 s @ 22.05 kHz -> 4 windows
    synth.write_wav(str(src / 'night#1.wav'), pcm)
    synth.write_wav(str(src / 'quiet.wav'), synth.clip_pcm16(30))
    (src / 'night#1.txt').write_text('0.400000\t1.100000\tsp3\n\\\t2000.0\t6000.0\n'
                                     '2.900000\t3.300000\tsp7\n\\\t900.0\t3000.0\n'
                                     '8.000000\t8.500000\tBackground\n\\\t600.0\t12000.0\n')
    (src / 'quiet.txt').write_text('0.5\t0.9\tBackground\n\\\t600.0\t900.0\n')
    birds = {'sp3': 3, 'sp7': 7, 'Other': 150}
    labels = create_label_dataset(str(src), birds, noise_labels=('Background',), suppress_noise=False)
    assert sorted(labels['bird_id'].tolist()) == [-1, -1, 3, 7]
    out = tmp_path / 'dataset'
    n_pos, n_neg = prepare_dataset(str(src), str(out), labels=labels)
    fp = File_Processor(str(src / 'night#1.wav'), '', labels)
    img_db, annots = fp.process_file()
    assert (n_pos, n_neg) == (len(annots), len(img_db) - len(annots) + 1)           # + the single window of quiet.wav
    assert sorted(os.listdir(out / 'positive_files')) == ['site1__night__1']
    assert sorted(os.listdir(out / 'negative_files')) == ['site1__night__1', 'site1__quiet']
    (out / 'hard_neg').mkdir()
    ds = Img_dataset(str(out), transform=False)
    assert len(ds) == n_pos
    np.random.seed(0)
    for idx, name in enumerate(ds.positive_files):
        win = int(name.replace('.png', '').split('__')[-1])
        img, neg, bb, ids = ds[idx]
        ref = np.round(img_db[win] * 255).astype(np.uint8)
        assert np.array_equal(np.round(img.cpu().numpy() * 255).astype(np.uint8), ref)
        row = annots.loc[annots['index'] == win].iloc[0]
        assert bb.numpy().tolist() == [list(map(float, c)) for c in row['coord']] and ids.numpy().tolist() == list(map(float, row['bird_id']))
    # second call: existing recordings are skipped (reference prepare_dataset.py:49-50)
    assert prepare_dataset(str(src), str(out), labels=labels) == (0, 0)
