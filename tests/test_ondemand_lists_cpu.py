"""CPU: host logic of the demand-driven FPN level (ondemand.wino23_pattern): which tiles / planes / pixels the static lists hold, and
the contract fields of the committed bench line."""
import json
import os

import numpy as np
import pytest

from birdsoundclassif_amd import ondemand

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pattern_rows(n, stride):
    m = np.zeros(n, dtype=bool)
    for o in range((n + 2 - 3) // stride + 1):
        for k in range(3):
            if 0 <= o * stride - 1 + k < n:
                m[o * stride - 1 + k] = True
    return m


@pytest.mark.parametrize('geom', [(2, 188, 512, 8), (3, 47, 66, 8), (1, 94, 256, 4), (2, 25, 33, 8)])
def test_pattern_lists(geom):
    B, H, W, S = geom
    pat = ondemand.wino23_pattern(B, H, W, S, 'cpu')
    TH, TW = (H + 1) // 2, (W + 1) // 2
    tiles = pat.tiles.numpy().reshape(-1, 128)
    info = pat.blk_info.numpy().view(np.uint32)
    assert len(info) == len(tiles)
    ry, rx = pattern_rows(H, S), pattern_rows(W, S)
    stored = np.zeros((B, H, W), dtype=bool)
    seen = set()
    planes = 0
    for blk, word in zip(tiles, info):
        v = blk[blk >= 0]
        assert len(v) > 0 and np.all(np.diff(v) > 0), 'ascending inside a block, no empty block'
        assert np.all(blk[len(v):] == -1), '-1 only at the end of a block'
        pm, sm = int(word) & 0xffff, (int(word) >> 16) & 0xf
        planes += bin(pm).count('1') * len(v)
        b, rem = v // (TH * TW), v % (TH * TW)
        ty, tx = rem // TW, rem % TW
        assert b.max() - b.min() <= 7
        for p in range(2):
            for q in range(2):
                if (sm >> (2 * p + q)) & 1:
                    ok = (2 * ty + p < H) & (2 * tx + q < W)
                    stored[b[ok], 2 * ty[ok] + p, 2 * tx[ok] + q] = True
        # the planes of a block are exactly those its stored pixels need: output row p uses i in {0,1,2} (p = 0) / {1,2,3} (p = 1)
        need_i = set().union(*[{0, 1, 2} if p == 0 else {1, 2, 3} for p in range(2) if sm & (3 << (2 * p))])
        need_j = set().union(*[{0, 1, 2} if q == 0 else {1, 2, 3} for q in range(2) if sm & (5 << q)])
        assert pm == sum(1 << (4 * i + j) for i in need_i for j in need_j)
        for t in v.tolist():
            assert t not in seen, 'a tile is listed once'
            seen.add(t)
    assert pat.n == len(seen) and abs(pat.n_eff - planes / 16) < 1e-6
    assert np.array_equal(stored, np.broadcast_to(ry[:, None] & rx[None, :], (B, H, W))), 'exactly the pattern pixels are stored'
    # the lateral's pixel list = the pixels within one pixel of the pattern (the 3x3 convolution's reach), per image, ascending
    px = pat.px_rows.numpy()
    px = px[px >= 0]
    assert np.all(np.diff(px) > 0)
    dil = lambda m: m | np.r_[m[1:], False] | np.r_[False, m[:-1]]
    want = np.broadcast_to(dil(ry)[:, None] & dil(rx)[None, :], (B, H, W)).ravel()
    got = np.zeros(B * H * W, dtype=bool)
    got[px] = True
    assert np.array_equal(got, want)
    # most planes first inside each eighth of the list (one eighth per XCD)
    npl = np.array([bin(int(w) & 0xffff).count('1') for w in info])
    for x8 in range(8):
        seg = npl[len(npl) * x8 // 8:len(npl) * (x8 + 1) // 8]
        assert np.all(np.diff(seg) <= 0)


def test_committed_bench_line_carries_the_contract_fields():
    line = json.loads(open(os.path.join(ROOT, 'profiles', 'r02_bench_default.json')).read().strip().splitlines()[-1])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in line, k
    assert line['n_gpus'] == 1 and line['higher_is_better'] is True and line['scaling'] == 'weak' and line['vs_baseline'] is None
    assert 'workload' in line['config'] and 'configs[1]' in line['config']['workload']
    r = line['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert r['bound'] == 'mfma' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and 0 < r['frac'] <= 1
    c = line['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] in ('port', 'reference')
    assert abs(line['value'] - 64 * 1000.0 / line['ms_per_step']) < 1e-6 * line['value']


def test_pinned_counters_return_to_the_pool_when_the_map_dies():
    """Every (RoI pooling, chunk) entry of a LazyMap holds one pinned counter, two when the dilated list of a deferred lateral
    exists: all of them go back to ondemand._PINNED_FREE when the map's state is dropped (a pool that never refills allocates pinned
    memory -- a stream stall -- every few training steps)."""
    import gc
    import torch
    before = len(ondemand._PINNED_FREE)
    st = ondemand.LazyMap(None, None, None, 8)
    hosts = [torch.zeros(1, dtype=torch.int32) for _ in range(5)]
    st.roi.append([(None, hosts[0], None, None, None), (None, hosts[1], None, 'tiles_d', hosts[2])])
    st.roi.append([(None, hosts[3], None), (None, hosts[4], None, None, None)])          # the 3-tuples of the older layout too
    del st
    gc.collect()
    got = ondemand._PINNED_FREE[before:]
    assert len(got) == 5 and {id(h) for h in got} == {id(h) for h in hosts}
    del ondemand._PINNED_FREE[before:]
