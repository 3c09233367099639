"""CPU: the front-end oracle (oracle/frontend_ref.py) against the REAL `File_Processor.process_file` of the reference
(prepare_dataset.py:108-157, 228-294).  `tests/golden/frontend.npz` was written by `oracle/make_golden.py
--frontend-only`, which runs the reference's own class with `librosa.stft` / `librosa.core.load` stubbed by the oracle's
float64 STFT and wav reader: everything after the STFT (amp_to_db, crop, per-file min/max over chunks, chunk and window
bookkeeping, reflect padding with and without labels) and the constants are the reference's code.  What stays unpinned is
the STFT core and the resampler (third-party, absent)."""
import numpy as np
import pytest
import torch

from helpers import check_packed, load_golden
from oracle import frontend_ref as FR
from oracle import make_golden as MG, ref_import as R

CONSTS = ('W_PIX', 'HOP_SPECTRO', 'WIN_LENGTH', 'HOP_LENGTH', 'FREQ_ACCURACY', 'DT', 'LOW_IDX', 'HIGH_IDX', 'LOW_FREQ',
          'HIGH_FREQ', 'spectrogram_length')


def oracle_case(g, name, seed, n22, max_l, labels):
    pcm44, _ = MG.frontend_case_inputs(name, seed, n22, labels)
    y = pcm44.astype(np.float32) / np.float32(32768.0)
    kw = {} if max_l is None else {'max_l': max_l}
    if labels:
        kw['label_t_end_max'] = float(g[f'{name}.t_end_max'])
    return FR.process_waveform(y, **kw)


@pytest.mark.parametrize('case', MG.FRONTEND_CASES, ids=[c[0] for c in MG.FRONTEND_CASES])
def test_oracle_matches_reference_file_processor(case):
    g = load_golden('frontend.npz')
    name = case[0]
    imgs, c = oracle_case(g, *case)
    for k in CONSTS:
        assert float(c[k]) == float(g[f'{name}.{k}']), k
    assert len(imgs) == int(g[f'{name}.n_img'])
    for i, im in enumerate(imgs):
        # same float64 operations in the same order as the reference, cast to float32 at the end
        assert check_packed(g, f'{name}.img{i}', torch.from_numpy(im), atol=0.0) == 0.0
    assert np.array_equal(imgs[-1][[0, 187, 374]], g[f'{name}.last_rows'])


def test_window_bookkeeping_quirks_are_the_references():
    """The three behaviours the fixtures pin beyond the plain case."""
    c = FR.constants()
    # a window running past the end of the file is cut at the end of the chunk its first column is in
    cols = FR.window_columns([1516, 1516, 304], c)
    assert len(cols) == 4 and cols[3][0] == 2457 and cols[3][:575].tolist() == list(range(2457, 3032))
    assert cols[3].max() == 3031                       # chunk 2 (columns 3032..3335) never appears
    # windows straddling a chunk end are the plain slice of the concatenation
    cols = FR.window_columns([1516, 1516, 1137], c)
    assert all(np.array_equal(cols[k], np.arange(819 * k, 819 * k + 1024)) for k in range(4))
    # labels: the padding goes in steps of `empty_width`, which doubles
    cols = FR.window_columns([1516], c, label_t_end_max=4.4)
    tail = cols[1][697:]                               # int(4.4 / DT) = 1470 -> empty_width = 46, then 92, 184, ...
    assert tail[:46].tolist() == list(range(1514, 1468, -1))                     # reflect of 46 columns
    assert tail[46:46 + 92].tolist() == list(range(1470, 1516)) + list(range(1514, 1468, -1))


@pytest.mark.skipif(not R.available(), reason='/root/reference not present')
def test_every_pixel_live_against_the_reference(tmp_path):
    """Container only: all pixels of all windows, bit for bit, on two of the cases."""
    PD = MG.install_frontend_stubs()
    g = load_golden('frontend.npz')
    for case in MG.FRONTEND_CASES:
        if case[0] not in ('chunkq', 'labels', 'short'):
            continue
        _, ref_imgs, _ = MG.run_reference_frontend(PD, *case, str(tmp_path))
        imgs, _ = oracle_case(g, *case)
        assert len(imgs) == len(ref_imgs)
        for a, b in zip(imgs, ref_imgs):
            assert np.array_equal(a, np.asarray(b, dtype=np.float32))


def long_case_oracle(g):
    c = MG.LONG_CASE
    pcm44 = FR.upsample2x_pcm16(__import__('birdsoundclassif_amd.synth', fromlist=['x']).clip_pcm16(c['seed'], c['n22']))
    y = pcm44.astype(np.float32) / np.float32(32768.0)
    max_l = c['max_file'] - c['max_file'] % FR.FREQ
    return FR.process_long_waveform(y, max_l, labels=[r[:2] for r in MG.long_case_labels()])


def test_oracle_long_file_matches_reference_process_long_file():
    """prepare_dataset.py:187-225 with the 15e7 limit scaled down: three splits, each normalised and windowed on its own,
    annotations shifted into the split they START in, their end clipped to the split (which moves the padding of a split's
    last window, :283-287)."""
    g = load_golden('frontend.npz')
    out, kept = long_case_oracle(g)
    assert len(out) == int(g['longfile.n_split']) == 3
    for k, imgs in enumerate(out):
        assert len(imgs) == int(g[f'longfile.s{k}.n_img'])
        for i, im in enumerate(imgs):
            assert check_packed(g, f'longfile.s{k}.img{i}', torch.from_numpy(im), atol=0.0) == 0.0
    assert int(g['longfile.n_annot']) == sum(r is not None for r in kept)
