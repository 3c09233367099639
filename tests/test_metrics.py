"""CPU (host logic): annotation parsing and AP / mAP / recall metrics against the REAL reference's results
(tests/golden/metrics.json, written by oracle/make_golden.py from nets_utils.py:419-534)."""
import json
import math
import os

from birdsoundclassif_amd import synth
from birdsoundclassif_amd.nets.util.nets_utils import compute_AP_scores, format_txt_annots
from helpers import GOLD


def _close(a, b):
    return (math.isnan(a) and math.isnan(b)) or abs(a - b) < 1e-12


def test_ap_scores_match_reference():
    gold = json.load(open(os.path.join(GOLD, 'metrics.json')))
    cases = synth.metrics_cases()
    assert len(cases) == len(gold['cases'])
    n_nontrivial = 0
    for outputs, ref, ref_f in zip(cases, gold['cases'], gold['filtered']):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            got, got_f = compute_AP_scores(outputs), compute_AP_scores(outputs, filter_sp=['sp1', 'sp4'])
        for k in ('AP', 'mAP', 'Rec', 'mRec'):
            assert _close(float(got[k]), ref[k]), (k, got, ref)
            assert _close(float(got_f[k]), ref_f[k]), (k, got_f, ref_f)
        n_nontrivial += 0 < ref['AP'] < 1
    assert n_nontrivial >= 8


def test_format_txt_annots_matches_reference(tmp_path):
    gold = json.load(open(os.path.join(GOLD, 'metrics.json')))
    for seed, ref in enumerate(gold['annots']):
        p = tmp_path / f'a{seed}.txt'
        p.write_text(synth.annotation_text(seed))
        got = format_txt_annots(str(p))
        assert {k: [[float(z) for z in b] for b in v] for k, v in got.items()} == ref
