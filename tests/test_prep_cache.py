"""CPU: the prepared-weight cache (nets/_prep.py) never serves one tensor's copy to another tensor that happens to
reuse its Python id / address, and forgets entries of dead tensors."""
import gc

import torch

from birdsoundclassif_amd.nets import _prep


def test_entries_die_with_their_tensor_and_ids_are_not_trusted():
    _prep.clear()
    w = torch.arange(2 * 3 * 3 * 3, dtype=torch.float32).view(2, 3, 3, 3)
    a = _prep.krsc(w)
    assert _prep.krsc(w) is a                                   # hit
    n_before = len(_prep._cache)
    key = next(iter(_prep._cache))
    # simulate id reuse: another live tensor filed under the dead tensor's key with matching version fields
    other = torch.ones(2, 3, 3, 3)
    ref, ver, val = _prep._cache[key]
    _prep._cache[(id(other), key[1])] = (ref, (other.data_ptr(), other._version, other.device, _prep._epoch), val)
    got = _prep.krsc(other)
    assert got.shape == (2, 32) and torch.equal(got[:, :27], torch.ones(2, 27))    # recomputed for `other`, not w's copy
    del w, a, val, ref
    gc.collect()
    assert all(k[0] != key[0] for k in _prep._cache), 'entry of the dead tensor was not evicted'
    assert len(_prep._cache) <= n_before


def test_two_models_in_one_process_get_their_own_prepared_weights():
    _prep.clear()
    outs = []
    for seed in (1, 2):
        torch.manual_seed(seed)
        conv = torch.nn.Conv2d(32, 4, 3)
        outs.append((_prep.krsc(conv.weight).clone(), conv.weight.detach().permute(0, 2, 3, 1).reshape(4, -1).clone()))
        del conv
        gc.collect()
    for got, want in outs:
        assert torch.equal(got, want)
