"""GPU: the BASELINE.json configurations at their full sizes (configs[1]: B = 64 inference, configs[2]: B = 128 training
step).  The oracle cannot run these batches in test time, so they are tied to it through the batch: 8 distinct clips are
tiled, the B = 8 run is checked against the oracle (classes / boxes bit for bit), and every copy inside the big batch must
reproduce the B = 8 result.  That covers what the small parity tests cannot: the Winograd batch chunking (26 + 26 + 12
images forward, 46 + 46 + 36 backward), 64 / 128-image proposal and NMS launches, and the memory footprint."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from birdsoundclassif_amd import synth                                     # noqa: E402
from helpers import assert_rois_equal_up_to_near_ties, dets_to_rows, filler_state_dict                        # noqa: E402
from oracle import frontend_ref as FR, nets_ref as O                       # noqa: E402


def build(train=False):
    from birdsoundclassif_amd.nets import build_model
    from birdsoundclassif_amd.train import default_args
    m, crit = build_model(default_args(device='cuda'))
    m.load_state_dict(filler_state_dict())
    m = m.cuda()
    m.train(train), crit.train(train)
    return m, crit


def test_config1_b64_inference_equals_b8_equals_oracle():
    from birdsoundclassif_amd.nbm_datasets.prepare_dataset import SpectrogramFrontEnd
    model, _ = build()
    fe = SpectrogramFrontEnd('cuda')
    pcm8 = synth.clip_batch_pcm16(0, 8)
    pcm64 = torch.from_numpy(np.tile(pcm8, (8, 1))).cuda()
    with torch.no_grad():
        img8, _ = fe(torch.from_numpy(pcm8).cuda(), 22050)
        img64, _ = fe(pcm64, 22050)
        assert torch.equal(img64.view(8, 8, 375, 1024), img8[:, 0].expand(8, 8, 375, 1024))
        det8, n8 = model.detect(img8, min_score=0.05)
        torch.cuda.reset_peak_memory_stats()
        det64, n64 = model.detect(img64, min_score=0.05)
        peak = torch.cuda.max_memory_allocated() / 2 ** 30
        o8 = model.forward_first_stage(img8)
        o64 = model.forward_first_stage(img64)
        l8 = model.forward_first_stage(img8, lazy=True)['rois']           # the TIMED path: FPN levels 0 / 1 on demand
        l64 = model.forward_first_stage(img64, lazy=True)['rois']
    # every copy inside the 64-batch == the 8-batch, bit for bit (classes, boxes AND scores: same kernels, same data)
    assert torch.equal(n64.view(8, 8), n8.expand(8, 8))
    assert int(n8.sum()) > 0
    assert torch.equal(det64.view(8, 8, *det8.shape[1:]), det8.expand(8, *det8.shape))
    assert torch.equal(o64['rois'].view(8, 8, *o8['rois'].shape[1:]), o8['rois'].expand(8, *o8['rois'].shape))
    for a, b in zip(o64['fpn_out'], o8['fpn_out']):                       # the Winograd chunks 26 + 26 + 12 see the same data
        assert torch.equal(a.reshape(8, 8, *b.shape[1:]), b.expand(8, *b.shape))
    # the 8-batch == the oracle on the product's own images (front end parity is tested in test_gpu_e2e.py)
    sd = filler_state_dict()
    cfg = O.make_cfg()
    x = img8.cpu()
    with torch.no_grad():
        ref1 = O.forward_first_stage(sd, cfg, x)
        ref = O.forward(sd, cfg, x, min_score=0.05)
    assert_rois_equal_up_to_near_ties(o8['rois'], ref1['rois'], ref1['roi_scores'])
    # ... and so are the RoIs of the on-demand path (the one `detect` / the bench / training run), at both batch sizes
    assert_rois_equal_up_to_near_ties(l8, ref1['rois'], ref1['roi_scores'], what='RoIs of the on-demand path')
    assert torch.equal(l64.view(8, 8, *l8.shape[1:]), l8.expand(8, *l8.shape))
    got = dets_to_rows(model.head.fast_rcnn.dets_to_dicts(det8.cpu(), n8.cpu(), model.args.num_classes))
    want = dets_to_rows(ref)
    assert got.shape == want.shape and len(got) > 0
    assert np.array_equal(got[:, :6], want[:, :6]), 'class / box assignments differ from the oracle'
    assert np.abs(got[:, 6] - want[:, 6]).max() < 1e-4
    print(f'B=64 detect: {int(n64.sum())} detections, peak memory {peak:.1f} GiB')
    assert peak < 200


def oracle_detect_chunked(sd, cfg, x, min_score, chunk=8):
    """The oracle on a batch it cannot hold at once: everything per image (backbone, attention, FPN, RPN heads; eval mode, so no batch
    statistics) in chunks, then the ProposalLayer ONCE over the whole batch -- its two batch-coupled counts (reference layers.py:287
    `min` of the surviving anchors over the batch, nets_utils.py:236 `min` of the NMS keep counts) see all images, like in one reference
    call -- then the second stage chunk by chunk on the kept FPN maps (per RoI, no batch coupling).  -> (first-stage dict, detections)."""
    cls, reg, fpn = [], [], []
    with torch.no_grad():
        for b0 in range(0, x.shape[0], chunk):
            feats = O.backbone_forward(sd, x[b0:b0 + chunk])
            fo = O.fpn_forward(sd, O.sa_pyramid(sd, feats, cfg.pyramid_top_n_attn))
            c, r = O.rpn_forward(sd, cfg, fo, False, None)
            cls.append(c), reg.append(r), fpn.append(fo)
        rois, roi_scores = O.proposal_layer(cfg, torch.cat(cls), torch.cat(reg), False)
        dets = []
        for i, b0 in enumerate(range(0, x.shape[0], chunk)):
            dets += O.forward_second_stage(sd, cfg, fpn[i], rois[b0:b0 + chunk], 0.3, min_score, False)
    return {'rois': rois, 'roi_scores': roi_scores}, dets


def test_config1_b64_distinct_clips_equal_the_oracle():
    """configs[1] on 64 DIFFERENT clips in ONE call: the batch-coupled minima, the device RoI tile lists and the
    64-image NMS launches see 64 different images; RoIs and class / box assignments against the oracle over the same 64 images."""
    from birdsoundclassif_amd.nbm_datasets.prepare_dataset import SpectrogramFrontEnd
    model, _ = build()
    fe = SpectrogramFrontEnd('cuda')
    pcm = torch.from_numpy(synth.clip_batch_pcm16(0, 64)).cuda()
    assert len({bytes(r) for r in pcm.cpu().numpy()}) == 64
    with torch.no_grad():
        img, _ = fe(pcm, 22050)
        det, n = model.detect(img, min_score=0.05)
        rois = model.forward_first_stage(img, lazy=True)['rois']          # the timed path
    sd, cfg = filler_state_dict(), O.make_cfg()
    ref1, ref = oracle_detect_chunked(sd, cfg, img.cpu(), 0.05)
    assert ref1['rois'].shape == rois.shape and rois.shape[0] == 64
    exact = sum(bool(torch.equal(rois[b].cpu(), ref1['rois'][b])) for b in range(64))
    assert_rois_equal_up_to_near_ties(rois, ref1['rois'], ref1['roi_scores'], what='RoIs of 64 distinct clips')
    got = dets_to_rows(model.head.fast_rcnn.dets_to_dicts(det.cpu(), n.cpu(), model.args.num_classes))
    want = dets_to_rows(ref)
    assert got.shape == want.shape and len(got) > 64
    assert np.array_equal(got[:, :6], want[:, :6]), 'class / box assignments differ from the oracle'
    assert np.abs(got[:, 6] - want[:, 6]).max() < 1e-4
    print(f'B=64 distinct clips: {len(got)} detections == oracle; RoI lists identical row for row in {exact} of 64 images '
          f'(the others: permuted runs of score ties only), {rois.shape[1]} RoIs per image')


def test_config2_b128_train_steps_positive_and_negative():
    from birdsoundclassif_amd import train as T
    args = T.default_args(device='cuda')
    img8 = torch.from_numpy(synth.image_batch(0, 8))
    neg8 = torch.from_numpy(synth.image_batch(100, 8))
    bb8, ids8, len8 = synth.label_batch(0, 8)
    tile = lambda t, k: torch.cat([t] * k, 0) if torch.is_tensor(t) else list(t) * k
    batches = {8: [img8, neg8, bb8, ids8, len8],
               128: [tile(img8, 16), tile(neg8, 16), tile(bb8, 16), tile(ids8, 16), tile(len8, 16)]}
    out = {}
    for B in (8, 128):
        model, crit = build(train=True)
        opt, _ = T.build_optimizer(model, args)
        batch = batches[B]
        # RPN outputs of the step's forward (train mode: BatchNorm on batch statistics, identical for a tiled batch)
        with torch.no_grad():
            o1 = model.forward_first_stage(batch[0][:, None].cuda())
            rpn = (o1['rpn_cls_scores'].float().cpu(), o1['rpn_bbox_reg'].float().cpu())
            # the negative-image first-stage loss involves no sampling (top anchor per image, Appendix C-15a): evaluated
            # on the SAME fresh weights at both batch sizes it must agree
            o1 = model.forward_first_stage(batch[1][:, None].cuda())
            fresh_neg = float(crit.first_stage_loss(o1['rpn_cls_scores'], o1['rpn_bbox_reg'], batch[2], batch[4], True)
                              ['first_neg_class_loss'])
        del o1
        model, crit = build(train=True)                                  # fresh BatchNorm buffers for the measured steps
        opt, _ = T.build_optimizer(model, args)
        torch.cuda.reset_peak_memory_stats()
        np.random.seed(77)
        pos = T.train_one_step(model, crit, opt, batch, args.clip_max_norm, 'cuda', negative_sample=False)
        gn_pos = opt.grad_norm()
        neg = T.train_one_step(model, crit, opt, batch, args.clip_max_norm, 'cuda', negative_sample=True)
        gn_neg = opt.grad_norm()
        torch.cuda.synchronize()
        peak = torch.cuda.max_memory_allocated() / 2 ** 30
        pos = {k: float(v.detach() if torch.is_tensor(v) else v) for k, v in pos.items()}
        neg = {k: float(v.detach() if torch.is_tensor(v) else v) for k, v in neg.items()}
        assert all(np.isfinite(v) for v in pos.values()) and all(np.isfinite(v) for v in neg.values()), (pos, neg)
        assert {'first_class_loss', 'first_regression_loss', 'sec_class_loss', 'sec_regression_loss'} <= set(pos)
        assert {'first_neg_class_loss', 'sec_neg_class_loss'} <= set(neg)
        assert gn_pos > 0 and gn_neg > 0 and np.isfinite(gn_pos) and np.isfinite(gn_neg)
        # first-stage losses of the positive step == the oracle's target layer + loss on the same RPN outputs and seed
        np.random.seed(77)
        ref = O.first_stage_loss(O.make_cfg(), rpn[0], rpn[1], batch[2], batch[4])
        for k in ('first_class_loss', 'first_regression_loss'):
            assert abs(pos[k] - float(ref[k])) < 2e-4 * max(1.0, abs(float(ref[k]))), (B, k, pos[k], float(ref[k]))
        out[B] = dict(rpn=rpn, pos=pos, neg=neg, peak=peak, gn=(gn_pos, gn_neg), fresh_neg=fresh_neg)
        del model, crit, opt
        torch.cuda.empty_cache()
    # the tiled 128-batch reproduces the 8-batch: RPN outputs copy by copy, and the loss that involves no sampling
    for a, b in zip(out[128]['rpn'], out[8]['rpn']):
        assert float((a.view(16, *b.shape) - b[None]).abs().max()) < 1e-4
    assert abs(out[128]['fresh_neg'] - out[8]['fresh_neg']) < 2e-4 and out[8]['fresh_neg'] > 0
    print(f"B=128 train: positive {out[128]['pos']}, negative {out[128]['neg']}, grad norms {out[128]['gn']}, "
          f"peak memory {out[128]['peak']:.1f} GiB")
    assert out[128]['peak'] < 230          # 169-200 GiB measured (positive + negative step); 288 GB on the card
