"""CPU, build container only: the oracle restatement beside the REAL reference (imported from
/root/reference with the torchvision stand-in) on inputs the golden files do not contain.  Skipped where
the reference is absent (always the case on the GPU box)."""
import warnings

import numpy as np
import pytest
import torch

from birdsoundclassif_amd import synth
from oracle import nets_ref as O, ref_import as R

pytestmark = pytest.mark.skipif(not R.available(), reason='/root/reference not present')


def test_eval_forward_other_seed():
    warnings.filterwarnings('ignore')
    a = R.default_args()
    m, _ = R.build_reference_model(a)
    sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=5)
    m.load_state_dict(sd), m.eval()
    cfg = O.make_cfg(a)
    x = torch.from_numpy(synth.image_batch(7, 1))[:, None]
    with torch.no_grad():
        ro, oo = m.forward_first_stage(x), O.forward_first_stage(sd, cfg, x)
        for p, q in zip(ro['fpn_out'], oo['fpn_out']):
            assert (p - q).abs().max() < 5e-5
        assert torch.equal(ro['rois'], oo['rois'])
        rd = m(x, min_score=0.05)
        od = O.forward(sd, cfg, x, min_score=0.05)
    for k in rd[0]:
        assert rd[0][k]['bbox_coord'].shape == od[0][k]['bbox_coord'].shape
        if rd[0][k]['bbox_coord'].numel():
            assert torch.equal(rd[0][k]['bbox_coord'], od[0][k]['bbox_coord'])


def test_anchors_match_reference():
    R.import_nets()
    from nbm_model.nets.util.nets_utils import generate_anchors_frcnn, get_anchor_shifts_frcnn
    from birdsoundclassif_amd.nets.util import nets_utils as mine
    cfg = O.make_cfg()
    ref = generate_anchors_frcnn(16, [0.5, 1, 2], 2 ** np.arange(5))
    assert np.array_equal(ref, O.base_anchors(cfg)) and np.array_equal(ref, mine.generate_anchors_frcnn(16, [0.5, 1, 2], 2 ** np.arange(5)))
    sh = get_anchor_shifts_frcnn(64, 24, 16)
    assert np.array_equal(sh, mine.get_anchor_shifts_frcnn(64, 24, 16))
    assert np.array_equal((ref + sh).reshape(-1, 4), O.all_anchors(cfg))


def test_learned_position_embedding_cannot_run_in_the_reference():
    """`--position_embedding learned` (position_encoding.py:59-83): the reference's Joiner evaluates the embedding for every
    pyramid level (backbone.py:139-148) and the module indexes nn.Embedding(50, .) with arange(w) -- the narrowest level of a
    375 x 1024 input is 32 columns wide, the next 64: its forward pass fails in the reference itself, so the build raises too."""
    warnings.filterwarnings('ignore')
    m, _ = R.build_reference_model(R.default_args(position_embedding='learned'))
    with torch.no_grad(), pytest.raises(IndexError):
        m.forward_first_stage(torch.zeros(1, 1, 375, 1024))
    from birdsoundclassif_amd.nets import build_model
    from birdsoundclassif_amd.train import default_args
    with pytest.raises(IndexError):
        build_model(default_args(device='cpu', position_embedding='learned'))
