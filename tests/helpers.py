"""Shared helpers for the parity tests: golden-fixture access and filler state_dicts."""
import os

import numpy as np
import torch

from birdsoundclassif_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
N_SAMPLE = 4096


def load_golden(name):
    return dict(np.load(os.path.join(GOLD, name)))


def sample_idx(name, numel):
    return (synth.uniform(('gold', name), N_SAMPLE) * numel).astype(np.int64)


def check_packed(g, name, t, atol, rtol=0.0):
    """Compare tensor `t` with the packed golden entry `name` (full / samples+stats)."""
    t = t.detach().float().cpu().contiguous()
    assert list(t.shape) == g[name + '.shape'].tolist(), (name, t.shape, g[name + '.shape'])
    flat = t.flatten().numpy()
    if (name + '.full') in g:
        ref = g[name + '.full']
        got = flat
    else:
        ref = g[name + '.samples']
        got = flat[sample_idx(name, flat.size)]
        st = g[name + '.stats']
        n = flat.size
        assert abs(flat.astype(np.float64).sum() - st[0]) <= (atol + rtol * abs(st[1]) / n) * n, name
        assert abs(flat.min() - st[2]) <= atol + rtol * abs(st[2]) and abs(flat.max() - st[3]) <= atol + rtol * abs(st[3]), name
    err = np.abs(got - ref)
    tol = atol + rtol * np.abs(ref)
    assert (err <= tol).all(), f'{name}: max err {err.max():.3e} (ref scale {np.abs(ref).max():.3e})'
    return float(err.max())


def dets_to_rows(dets):
    rows = []
    for b, d in enumerate(dets):
        for k, v in d.items():
            bb = v['bbox_coord']
            if len(bb) == 0:
                continue
            sc = v['scores'].reshape(-1)
            for i in range(len(bb)):
                rows.append([b, int(k), *[float(z) for z in bb[i]], float(sc[i])])
    return np.array(rows, dtype=np.float64).reshape(-1, 7)


_SHAPES = {}


def state_dict_shapes(**overrides):
    """{name: shape} of the model (default config = SURVEY Appendix B), from the product nets package
    (construction only -- no compute)."""
    key = tuple(sorted(overrides.items()))
    if key not in _SHAPES:
        from birdsoundclassif_amd.nets import build_model
        from birdsoundclassif_amd.train import default_args
        model, _ = build_model(default_args(device='cpu', **overrides))
        _SHAPES[key] = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    return _SHAPES[key]


def filler_state_dict(seed=0, **overrides):
    sd = synth.fill_state_dict(state_dict_shapes(**overrides), seed)
    return synth.tame_bifpn(sd) if overrides.get('fpn') == 'bifpn' else sd


def assert_rois_equal_up_to_near_ties(got, ref, ref_scores, eps=2e-7, what='RoIs', max_pixel_flips=0):
    """RoIs [B,R,4] come out ordered by objectness (layers.py:292, an argsort of fp32 softmax outputs).  Two proposals whose
    scores differ by less than a couple of fp32 ulps of 1.0 have no defined order across implementations (the reference's
    own order depends on its BLAS build), so a run of such near-ties may come out permuted; everything else -- the rows
    themselves, their number, every other position -- must be bit-identical.  A near-tie that straddles the top-N cut would
    change the RoI SET, not just the order: that is never accepted (a permutation inside the list keeps the set).
    `max_pixel_flips` (per image, default 0): rows that differ from the reference's in ONE coordinate by exactly ONE pixel -- a
    box corner is `round()`-ed (nets_utils.py:186) and a pre-round value within fp32 noise of x.5 lands on either side,
    in the reference itself from one BLAS build to the next.  Returns the list of such flips (image, rank, got row, ref row)."""
    got, ref, sc = got.cpu(), ref.cpu(), ref_scores.cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    flips = []
    for b in range(ref.shape[0]):
        i, R = 0, ref.shape[1]
        n_flip = 0
        while i < R:
            if torch.equal(got[b, i], ref[b, i]):
                i += 1
                continue
            d = (got[b, i] - ref[b, i]).abs()
            if int((d != 0).sum()) == 1 and float(d.max()) == 1.0 and n_flip < max_pixel_flips:
                n_flip += 1
                flips.append((b, i, got[b, i].tolist(), ref[b, i].tolist()))
                i += 1
                continue
            j = i
            while j + 1 < R and abs(float(sc[b, j + 1]) - float(sc[b, j])) <= eps:
                j += 1
            assert j > i, f'{what}: image {b} rank {i}: {got[b, i].tolist()} != {ref[b, i].tolist()} (no score tie)'
            g = sorted(map(tuple, got[b, i:j + 1].tolist()))
            r = sorted(map(tuple, ref[b, i:j + 1].tolist()))
            assert g == r, f'{what}: image {b} ranks {i}..{j} differ beyond a permutation of near-ties'
            i = j + 1
    if flips:
        print(f'{what}: {len(flips)} one-pixel rounding flip(s): {flips}')
    return flips
