"""CPU (host logic): label merging and annotation parsing of the dataset-preparation path against the REAL reference
(tests/golden/labels.json from oracle/make_golden.py: prepare_dataset.py:297-375, utils.py:59-92)."""
import json
import os

import pandas as pd

from birdsoundclassif_amd import synth
from helpers import GOLD

COLS = ['t_start', 't_end', 'f_start', 'f_end', 'species', 'filename', 'bird_id']


def _processor(ext, labels):
    from birdsoundclassif_amd.nbm_datasets.prepare_dataset import File_Processor
    fp = File_Processor(f'/x/recA.{ext}', '', labels)
    fp.W_PIX, fp.HOP_SPECTRO, fp.DT, fp.FREQ_ACCURACY = 1024, 819, 132 / 44100, 44100 / 1324
    fp.LOW_FREQ, fp.HIGH_FREQ = 15 * fp.FREQ_ACCURACY, 390 * fp.FREQ_ACCURACY
    return fp


def test_merge_and_filter_labels_matches_reference():
    gold = json.load(open(os.path.join(GOLD, 'labels.json')))['merge']
    for (seed, ext, n_img), ref in zip(((0, 'wav', 16), (1, 'mp3', 16), (2, 'wav', 3), (3, 'wav', 1)), gold):
        r = _processor(ext, pd.DataFrame(synth.label_rows(seed), columns=COLS)).merge_and_filter_labels([None] * n_img)
        assert [int(i) for i in r['index']] == ref['index']
        assert [[[int(v) for v in box] for box in c] for c in r['coord']] == ref['coord']
        assert [[int(v) for v in b] for b in r['bird_id']] == ref['bird_id']
    assert any(len(c) > 1 for c in gold[0]['coord'])          # several boxes in one window
    # a file without labels: the reference raises and skips the file, the product returns None
    other = pd.DataFrame(synth.label_rows(0, filename='elsewhere'), columns=COLS)
    other = other.loc[other['filename'] != 'recA']
    assert _processor('wav', other).merge_and_filter_labels([None] * 4) is None


def test_read_txt_file_matches_reference(tmp_path):
    from birdsoundclassif_amd.nbm_datasets.prepare_dataset import create_label_dataset, read_txt_file
    gold = json.load(open(os.path.join(GOLD, 'labels.json')))['txt']
    for seed, ref in enumerate(gold):
        p = tmp_path / f'rec{seed}.txt'
        p.write_text(synth.annotation_text(seed))
        df = read_txt_file(str(p))
        got = [[float(r.t_start), float(r.t_end), float(r.f_start), float(r.f_end), str(r.species), str(r.filename)]
               for r in df.itertuples()]
        assert got == ref
    # duplicated frequency line and a record without one
    q = tmp_path / 'odd.txt'
    q.write_text('1.0\t2.0\tsp1\n\\\t1000.0\t2000.0\n\\\t5.0\t6.0\n3.0\t4.0\tsp2\n5.0\t6.0\tsp0\n\\\t-5.0\t-1.0\n')
    df = read_txt_file(str(q))
    assert df[['t_start', 'f_start', 'f_end', 'species']].values.tolist() == [[1.0, 1000.0, 2000.0, 'sp1'], [5.0, -5.0, -1.0, 'sp0']]
    lab = create_label_dataset(str(tmp_path), {'sp0': 3, 'sp1': 1, 'sp2': 2, 'Other': 150}, suppress_others=False)
    row = lab.loc[(lab['filename'] == 'odd') & (lab['species'] == 'sp0')].iloc[0]
    assert row['f_start'] == 0 and row['f_end'] == 20000 and row['bird_id'] == 3
