"""GPU parity tests, path level: front end and detector forward through the reference-shaped API
(File_Processor / NbmModel) against the oracle and the golden fixtures generated from the real reference."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from birdsoundclassif_amd import synth                                     # noqa: E402
from helpers import (assert_flips_are_half_pixel_ties, assert_rois_equal_up_to_near_ties, check_packed, dets_to_rows,   # noqa: E402
                     filler_state_dict, load_golden)
from oracle import frontend_ref as FR, nets_ref as O                       # noqa: E402


@pytest.fixture(scope='module')
def model():
    from birdsoundclassif_amd.nets import build_model
    from birdsoundclassif_amd.train import default_args
    m, _ = build_model(default_args(device='cuda'))
    m.load_state_dict(filler_state_dict())
    return m.cuda().eval()


# --------------------------------------------------------------------------- front end
IMG_TOL = 1e-4      # north_star tolerance, max over every pixel of the [0,1] image (fp64 MFMA STFT: measured ~1e-6)


def test_upsample_bit_exact_and_image_parity(tmp_path):
    from birdsoundclassif_amd.nbm_datasets.prepare_dataset import SpectrogramFrontEnd, File_Processor
    from birdsoundclassif_amd import ops
    fe = SpectrogramFrontEnd('cuda')
    pcm = synth.clip_batch_pcm16(0, 3)
    # integer stage: bit exact
    lead = 662
    wave = ops.pcm16_to_wave(torch.from_numpy(pcm).cuda(), 2 * pcm.shape[1] + 2 * lead + 20, lead, True, fe.hq).cpu().numpy()
    for b in range(3):
        ref = FR.upsample2x_pcm16(pcm[b]).astype(np.float32) / np.float32(32768)
        assert np.array_equal(wave[b, lead:lead + len(ref)], ref)
        assert not wave[b, :lead].any() and not wave[b, lead + len(ref):].any()
    # whole front end: fp64-MFMA STFT vs the float64 FFT oracle, every bin of every frame in dB, every pixel of the image
    db, mm, Ls = fe.spectrogram_db(torch.from_numpy(pcm).cuda(), 22050)
    imgs, L2 = fe(torch.from_numpy(pcm).cuda(), 22050)
    assert Ls == [1003] and L2 == 1003 and tuple(imgs.shape) == (3, 1, 375, 1024)
    for b in range(3):
        y = FR.upsample2x_pcm16(pcm[b]).astype(np.float32) / np.float32(32768)
        ref_db = FR.amp_to_db(FR.stft_mag(y, 1324, 132)[16:391])
        err_db = np.abs(db[b].cpu().numpy().astype(np.float64) - ref_db)
        assert err_db.max() < 2e-4, err_db.max()            # fp32 rounding of a value of magnitude <= 100 + log10f
        ref_imgs, c = FR.process_waveform(y)
        assert len(ref_imgs) == 1 and c['spectrogram_length'] == 1003
        err = np.abs(imgs[b, 0].cpu().numpy() - ref_imgs[0])
        assert err.max() < IMG_TOL, err.max()
        assert imgs[b, 0].min() == 0.0 and imgs[b, 0].max() == 1.0
    # File_Processor interface on a wav file, 44.1 kHz multi-window file
    p = str(tmp_path / 'long.wav')
    long_pcm = np.concatenate([FR.upsample2x_pcm16(synth.clip_pcm16(10 + i)) for i in range(3)])
    synth.write_wav(p, long_pcm, 44100)
    fp = File_Processor(p)
    got, _ = fp.process_file()
    ref_imgs, c = FR.process_file(p)
    assert len(got) == len(ref_imgs) == 4 and fp.spectrogram_length == c['spectrogram_length']
    assert fp.W_PIX == 1024 and fp.HOP_SPECTRO == 819
    for a, b in zip(got, ref_imgs):
        assert np.abs(a - b).max() < IMG_TOL


def test_front_end_against_the_reference_file_processor(tmp_path, monkeypatch):
    """HIP front end vs tests/golden/frontend.npz = the REAL File_Processor.process_file (librosa.stft stubbed by the
    oracle STFT): constants, window count, every sampled pixel and the padding pattern of the last window, incl. the
    chunked STFT (chunk length scaled down on both sides), the chunk-end cut and the label-dependent padding."""
    import pandas as pd
    from birdsoundclassif_amd.nbm_datasets.prepare_dataset import SpectrogramFrontEnd, File_Processor
    from oracle import make_golden as MG
    g = load_golden('frontend.npz')
    for name, seed, n22, max_l, labels in MG.FRONTEND_CASES:
        pcm44, rows = MG.frontend_case_inputs(name, seed, n22, labels)
        path = str(tmp_path / (name + '.wav'))
        synth.write_wav(path, pcm44, 44100)
        monkeypatch.setattr(SpectrogramFrontEnd, 'MAX_CHUNK', int(5e7) if max_l is None else max_l)
        lab = None if rows is None else pd.DataFrame(rows, columns=['t_start', 't_end', 'f_start', 'f_end', 'species',
                                                                       'filename', 'bird_id'])
        fp = File_Processor(path, '', lab)
        imgs, annots = fp.process_file()
        for k in ('W_PIX', 'HOP_SPECTRO', 'WIN_LENGTH', 'HOP_LENGTH', 'FREQ_ACCURACY', 'DT', 'LOW_IDX', 'HIGH_IDX',
                  'spectrogram_length'):
            assert float(getattr(fp, k)) == float(g[f'{name}.{k}']), (name, k)
        assert len(imgs) == int(g[f'{name}.n_img']), name
        for i, im in enumerate(imgs):
            check_packed(g, f'{name}.img{i}', torch.from_numpy(im), atol=IMG_TOL)
        assert np.abs(imgs[-1][[0, 187, 374]] - g[f'{name}.last_rows']).max() < IMG_TOL, name
        if labels:
            assert [int(i) for i in annots['index']] == g[f'{name}.annot_index'].tolist()
            assert abs(fp.LOW_FREQ - float(g[f'{name}.LOW_FREQ'])) < 1e-9 and abs(fp.HIGH_FREQ - float(g[f'{name}.HIGH_FREQ'])) < 1e-9


def test_reflect_pad_mode_matches_the_oracle_switch():
    """librosa <= 0.9 pads every STFT chunk by reflection: only the first / last 6 frames differ."""
    from birdsoundclassif_amd.nbm_datasets.prepare_dataset import SpectrogramFrontEnd
    fe = SpectrogramFrontEnd('cuda', pad_mode='reflect')
    pcm = synth.clip_batch_pcm16(4, 2)
    imgs, L = fe(torch.from_numpy(pcm).cuda(), 22050)
    zero, _ = SpectrogramFrontEnd('cuda')(torch.from_numpy(pcm).cuda(), 22050)
    for b in range(2):
        y = FR.upsample2x_pcm16(pcm[b]).astype(np.float32) / np.float32(32768)
        ref, _ = FR.process_waveform(y, pad_mode='reflect')
        assert np.abs(imgs[b, 0].cpu().numpy() - ref[0]).max() < IMG_TOL
    assert (imgs[:, 0, :, :6] != zero[:, 0, :, :6]).any()


def test_other_rates_and_sample_formats(tmp_path):
    """F1 beyond 16-bit / 22.05 kHz: any rate through the rational polyphase resampler (device result == the oracle's int16
    signal), 24-bit / float / stereo files as float32 like librosa.load -- whole front end vs the oracle."""
    import struct
    from birdsoundclassif_amd import ops
    from birdsoundclassif_amd.nbm_datasets.prepare_dataset import SpectrogramFrontEnd, File_Processor
    from test_frontend_oracle import _write_wav
    fe = SpectrogramFrontEnd('cuda')
    for sr in (48000, 16000, 32000):
        pcm = synth.clip_pcm16(30, sr * 2, sr)                      # 2 s at the file's rate
        x = torch.from_numpy(pcm.astype(np.float32) / np.float32(32768))[None].cuda()
        kind, n44, (L, M, taps) = fe._source(torch.float32, len(pcm), sr)
        w = ops.resample_to_wave(x, n44 + 8 - n44 % 4, 0, L, M, taps)[0, :n44].cpu().numpy()
        ref = FR.resample_to_pcm16(pcm.astype(np.float64) / 32768.0, sr)
        d = np.abs(np.rint(w * 32768).astype(np.int64) - ref)
        assert len(ref) == n44 == 88200 and d.max() <= 1 and (d != 0).mean() < 1e-5, (sr, d.max(), (d != 0).mean())
        p = str(tmp_path / f'r{sr}.wav')
        synth.write_wav(p, pcm, sr)
        got, _ = File_Processor(p).process_file()
        ref_imgs, _ = FR.process_file(p)
        assert len(got) == len(ref_imgs) == 1 and np.abs(got[0] - ref_imgs[0]).max() < IMG_TOL, sr
    # 24-bit mono at 44.1 kHz, float32 stereo at 22.05 kHz (averaged, then resampled 1:2 by the generic filter)
    base = FR.upsample2x_pcm16(synth.clip_pcm16(31))
    p24 = str(tmp_path / 'a24.wav')
    _write_wav(p24, 1, 24, 44100, [[int(v) * 256 + 77 for v in base]])
    pf = str(tmp_path / 'af.wav')
    a, b = synth.clip_pcm16(32), synth.clip_pcm16(33)
    _write_wav(pf, 3, 32, 22050, [[float(v) / 32768 for v in a], [float(v) / 32768 * 0.5 for v in b]])
    for p in (p24, pf):
        got, _ = File_Processor(p).process_file()
        ref_imgs, _ = FR.process_file(p)
        assert len(got) == len(ref_imgs) == 1 and np.abs(got[0] - ref_imgs[0]).max() < IMG_TOL, p


def test_long_file_splits_against_the_reference(tmp_path, monkeypatch):
    """Recordings longer than 1.5e8 samples (here scaled to 250 000): nested per-split outputs of the REAL
    process_long_file (tests/golden/frontend.npz, 'longfile'), incl. the annotations moved into their split."""
    import pandas as pd
    from birdsoundclassif_amd.nbm_datasets.prepare_dataset import SpectrogramFrontEnd, File_Processor
    from oracle import make_golden as MG
    g = load_golden('frontend.npz')
    c = MG.LONG_CASE
    monkeypatch.setattr(SpectrogramFrontEnd, 'MAX_FILE', c['max_file'])
    path = str(tmp_path / (c['name'] + '.wav'))
    synth.write_wav(path, FR.upsample2x_pcm16(synth.clip_pcm16(c['seed'], c['n22'])), 44100)
    lab = pd.DataFrame(MG.long_case_labels(), columns=['t_start', 't_end', 'f_start', 'f_end', 'species', 'filename', 'bird_id'])
    img_db, annots = File_Processor(path, '', lab).process_file()
    assert len(img_db) == int(g['longfile.n_split']) and len(annots) == int(g['longfile.n_annot'])
    for k, imgs in enumerate(img_db):
        assert len(imgs) == int(g[f'longfile.s{k}.n_img'])
        for i, im in enumerate(imgs):
            check_packed(g, f'longfile.s{k}.img{i}', torch.from_numpy(im), atol=IMG_TOL)
    for k, a in enumerate(annots):
        assert [int(i) for i in a['index']] == g[f'longfile.a{k}.index'].tolist()
        assert [len(cc) for cc in a['coord']] == g[f'longfile.a{k}.n_boxes'].tolist()
        assert [list(cc[0]) for cc in a['coord']] == g[f'longfile.a{k}.first_box'].tolist()
    # the detection driver refuses the nested result instead of mis-indexing it
    from birdsoundclassif_amd import run_detection as RD
    with pytest.raises(NotImplementedError):
        RD.run_detection(None, None, path, 'unused')


def test_silent_file_is_nan_like_reference():
    from birdsoundclassif_amd.nbm_datasets.prepare_dataset import SpectrogramFrontEnd
    fe = SpectrogramFrontEnd('cuda')
    imgs, _ = fe(torch.zeros((1, 66150), dtype=torch.int16).cuda(), 22050)
    assert torch.isnan(imgs).all()          # (x - min)/(max - min) with max == min, Appendix C-8


# --------------------------------------------------------------------------- detector forward vs golden (real reference)
@pytest.fixture(params=[False, True], ids=['fp32pipe', 'splitbf16'])
def split(request, monkeypatch):
    """Both forms of the deep-K implicit GEMM against the reference's fixtures: the fp32 matrix instruction (default) and the bf16
    matrix pipe on split fp32 operands (csrc/igemm_split.hip, NBM_SPLIT_BF16=1; the library reads the switch per call)."""
    monkeypatch.setenv('NBM_SPLIT_BF16', '1' if request.param else '0')
    return request.param


# one-pixel RoI flips that are accepted -- ONLY with the pre-round coordinate asserted within 1e-4 of x.5: per kernel form
HALF_PIXEL_TIES = {False: {'posenc'}, True: {'posenc', 'bifpn'}}


def test_forward_matches_reference_golden(model, split):
    g = load_golden('eval_b2.npz')
    x = torch.from_numpy(synth.image_batch(0, 2))[:, None].cuda()
    with torch.no_grad():
        o = model.forward_first_stage(x)
    # logits-level tensors: 1e-4 absolute (BASELINE.json north_star tolerance) on O(1) activations
    for i, f in enumerate(o['fpn_out']):
        check_packed(g, f'fpn{i}', f, atol=1e-4, rtol=1e-4)
    check_packed(g, 'rpn_cls_scores', o['rpn_cls_scores'], atol=1e-4)
    check_packed(g, 'rpn_bbox_reg', o['rpn_bbox_reg'], atol=1e-4)
    # discrete outputs: bit exact box assignments
    ref_rois = g['rois.full'].reshape(g['rois.shape'])
    assert tuple(o['rois'].shape) == tuple(ref_rois.shape)
    n_bad = int((o['rois'].cpu().numpy() != ref_rois).any(-1).sum())
    assert n_bad == 0, f'{n_bad} RoIs differ from the reference'
    # the TIMED path (FPN levels 0 / 1 on demand: the RPN reads their pattern pixels through the cell transforms) must give
    # the reference's RoIs bit for bit too -- an RoI that flips without producing a detection would be invisible further down
    with torch.no_grad():
        ol = model.forward_first_stage(x, lazy=True)
    n_bad = int((ol['rois'].cpu().numpy() != ref_rois).any(-1).sum())
    assert n_bad == 0, f'{n_bad} RoIs of the on-demand (lazy) path differ from the reference'
    check_packed(g, 'rpn_cls_scores', ol['rpn_cls_scores'], atol=1e-4)
    check_packed(g, 'rpn_bbox_reg', ol['rpn_bbox_reg'], atol=1e-4)
    with torch.no_grad():
        s = model.forward_second_stage(o['fpn_out'], o['rois'], training=True)
    check_packed(g, 'bbox_reg', s['bbox_reg'], atol=1e-4)
    check_packed(g, 'bbox_classes', s['bbox_classes'], atol=1e-4)
    for ms in (0.05, 0.2, 0.5):
        with torch.no_grad():
            dets = model(x, min_score=ms)
        rows, ref = dets_to_rows(dets), g[f'dets_min{ms}']
        assert rows.shape == ref.shape, (ms, rows.shape, ref.shape)
        assert np.array_equal(rows[:, :6], ref[:, :6]), f'class / box assignment differs at min_score={ms}'
        assert np.abs(rows[:, 6] - ref[:, 6]).max() < 1e-4


@pytest.mark.parametrize('tag,pe_qk', [('std', False), ('peqk', True)])
def test_transformer_rcnn_head_vs_reference_golden(tag, pe_qk):
    """`--tf_rcnn` (reference layers.py:589-651), both encoder flavours, against the real reference's outputs."""
    from birdsoundclassif_amd.nets import build_model
    from birdsoundclassif_amd.train import default_args
    g = load_golden('tf_rcnn_b3.npz')
    m, _ = build_model(default_args(device='cuda', tf_rcnn=True, tf_pe_qk=pe_qk))
    m.load_state_dict(filler_state_dict(tf_rcnn=True, tf_pe_qk=pe_qk))
    m = m.cuda().eval()
    x = torch.from_numpy(synth.image_batch(0, 3))[:, None].cuda()
    with torch.no_grad():
        o = m.forward_first_stage(x)
        ref_rois = g[f'{tag}.rois.full'].reshape(g[f'{tag}.rois.shape'])
        assert np.array_equal(o['rois'].cpu().numpy(), ref_rois)
        s = m.forward_second_stage(o['fpn_out'], o['rois'], training=True)
        check_packed(g, f'{tag}.bbox_reg', s['bbox_reg'], atol=1e-4)
        check_packed(g, f'{tag}.bbox_classes', s['bbox_classes'], atol=1e-4)
        for ms in (0.05, 0.2):
            rows, ref = dets_to_rows(m(x, min_score=ms)), g[f'{tag}.dets_min{ms}']
            assert rows.shape == ref.shape, (ms, rows.shape, ref.shape)
            assert np.array_equal(rows[:, :6], ref[:, :6])
            assert np.abs(rows[:, 6] - ref[:, 6]).max() < 1e-4
        # sync-free path (device counters, padded RoI slots masked inside the attention)
        det, n_det = m.detect(x, 0.3, 0.2)
        from birdsoundclassif_amd.nets.layers import FastRCNN
        rows2 = dets_to_rows(FastRCNN.dets_to_dicts(det, n_det, 150))
        assert np.array_equal(rows2[:, :6], g[f'{tag}.dets_min0.2'][:, :6])


def test_mha_small_masks_padded_keys():
    """nbm_mha_small against torch on both token layouts, with a device-side valid-length counter."""
    from birdsoundclassif_amd import ops
    S, N, nh, E = 50, 5, 8, 512
    for seq_major in (True, False):
        qkv = torch.from_numpy(synth.normal(('mha', seq_major), S * N * 3 * E).astype(np.float32).reshape(S * N, 3 * E)).cuda()
        nv = 37
        cnt = torch.tensor([nv], dtype=torch.int32, device='cuda')
        ss, bs = (N, 1) if seq_major else (1, S)
        out = ops.mha_small(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], S, N, nh, ss, bs, cnt)
        t = qkv.view(S, N, 3 * E) if seq_major else qkv.view(N, S, 3 * E).transpose(0, 1)
        q, k, v = (t[..., i * E:(i + 1) * E].reshape(S, N * nh, E // nh).transpose(0, 1).double() for i in range(3))
        att = torch.softmax((q @ k[:, :nv].transpose(1, 2)) / (E // nh) ** 0.5, -1) @ v[:, :nv]
        ref = att.transpose(0, 1).reshape(S, N, E)
        got = out.view(S, N, E) if seq_major else out.view(N, S, E).transpose(0, 1)
        assert (got[:nv].double() - ref[:nv]).abs().max() < 2e-6


def test_intermediate_taps_vs_golden(model):
    g = load_golden('eval_b2.npz')
    x = torch.from_numpy(synth.image_batch(0, 2)).cuda()[..., None].contiguous()      # NHWC
    with torch.no_grad():
        taps, _ = model.backbone(x)
        for i, t in enumerate(taps):
            check_packed(g, f'tap{i}', t.permute(0, 3, 1, 2), atol=5e-5, rtol=5e-5)
        att = model.attn(taps)
        for i in (3, 4):
            check_packed(g, f'attn{i}', att[i].permute(0, 3, 1, 2), atol=1e-4, rtol=1e-4)


def test_forward_matches_oracle_other_seed(model):
    """Same comparison against the oracle restatement on inputs the golden file does not cover (B=3)."""
    sd = filler_state_dict()
    cfg = O.make_cfg()
    x = torch.from_numpy(synth.image_batch(40, 3))[:, None]
    with torch.no_grad():
        ref = O.forward_first_stage(sd, cfg, x)
        got = model.forward_first_stage(x.cuda())
    for a, b in zip(got['fpn_out'], ref['fpn_out']):
        assert (a.cpu() - b).abs().max() < 1e-4
    assert (got['rpn_cls_scores'].cpu() - ref['rpn_cls_scores']).abs().max() < 1e-4
    assert tuple(got['rois'].shape) == tuple(ref['rois'].shape)
    assert torch.equal(got['rois'].cpu(), ref['rois'])
    with torch.no_grad():
        ref_d = O.forward_second_stage(sd, cfg, ref['fpn_out'], ref['rois'], 0.3, 0.1)
        got_d = model(x.cuda(), min_score=0.1)
    r, q = dets_to_rows(ref_d), dets_to_rows(got_d)
    assert r.shape == q.shape and np.array_equal(r[:, :6], q[:, :6])


def test_checkpoint_layout_roundtrip(model, tmp_path):
    """{'checkpoints': state_dict} written like reference train.py:171-187 loads through initialize_model."""
    from birdsoundclassif_amd.nets.nbm_model import initialize_model
    from birdsoundclassif_amd.nets import build_model
    from birdsoundclassif_amd.train import default_args
    p = str(tmp_path / 'model_chkpt.pt')
    torch.save({'checkpoints': {k: v.cpu() for k, v in model.state_dict().items()}, 'steps': 1, 'epoch': 0,
                'best_val_cls_loss': 99}, p)
    m2, _ = build_model(default_args(device='cuda'))
    m2 = initialize_model(m2, p, train=False)
    x = torch.from_numpy(synth.image_batch(0, 1))[:, None].cuda()
    with torch.no_grad():
        a, b = model.detect(x, min_score=0.05), m2.cuda().detect(x, min_score=0.05)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_chunked_stft_of_a_long_row(monkeypatch):
    """Rows longer than the STFT chunk (5e7 samples in the reference) are transformed chunk by chunk, every chunk centre
    padded on its own, min/max over the whole row -- checked with the chunk size scaled down on both sides, 22.05 kHz
    input (resampled as a whole before it is cut) and a batch of two rows."""
    from birdsoundclassif_amd.nbm_datasets.prepare_dataset import SpectrogramFrontEnd
    fe = SpectrogramFrontEnd('cuda')
    monkeypatch.setattr(SpectrogramFrontEnd, 'MAX_CHUNK', 100000)
    pcm = np.stack([synth.clip_pcm16(20 + i, 125000) for i in range(2)])
    imgs, L = fe(torch.from_numpy(pcm).cuda(), 22050)
    for b in range(2):
        y = FR.upsample2x_pcm16(pcm[b]).astype(np.float32) / np.float32(32768)    # resample first, then cut
        ref, c = FR.process_waveform(y, max_l=100000)
        assert L == c['spectrogram_length'] and imgs.shape[1] == len(ref)
        for k, r in enumerate(ref):
            assert np.abs(imgs[b, k].cpu().numpy() - r).max() < IMG_TOL


@pytest.mark.parametrize('tag,kw', [('fpn_first', dict(fpn_first=True)), ('sandwich', dict(sandwich_attn=True)),
                                    ('posenc', dict(add_posenc=True)), ('bifpn', dict(fpn='bifpn', n_bifpn_layers=2)),
                                    ('attn5', dict(pyramid_top_n_attn=5)), ('dilation', dict(dilation=True))])
def test_composition_flags_vs_reference_golden(tag, kw, split):
    """--fpn_first / --sandwich_attn / --add_posenc (reference nbm_model.py:45-52) and --dilation (backbone.py:129-131: layer4 at
    layer3's resolution with 3x3 / dilation 2 -- here: the ordinary kernels on the space-to-batch form -- the RPN's adaptive pooling a
    real 2x2 average on that level, the RoI pooling's stride 32 kept) against the real reference's outputs, forward and one
    optimisation step's worth of backward (finite gradients everywhere)."""
    from birdsoundclassif_amd.nets import build_model
    from birdsoundclassif_amd.train import default_args, build_optimizer, train_one_step
    g = load_golden('variants_dilation_b2.npz' if tag == 'dilation' else 'variants_b2.npz')
    args = default_args(device='cuda', **kw)
    m, crit = build_model(args)
    m.load_state_dict(filler_state_dict(**kw))
    m = m.cuda().eval()
    x = torch.from_numpy(synth.image_batch(0, 2))[:, None].cuda()
    with torch.no_grad():
        o = m.forward_first_stage(x)
        for i, f in enumerate(o['fpn_out']):
            check_packed(g, f'{tag}.fpn{i}', f, atol=1e-4, rtol=1e-4)
        check_packed(g, f'{tag}.rpn_cls_scores', o['rpn_cls_scores'], atol=1e-4)
        check_packed(g, f'{tag}.rpn_bbox_reg', o['rpn_bbox_reg'], atol=1e-4)
        # RoIs: bit-identical to the reference's, up to permutations inside runs of proposals whose objectness differs by less
        # than 2e-7 (the reference's argsort is unstable, layers.py:292) -- the RoI SET of every image is the reference's
        ref_rois = torch.from_numpy(g[f'{tag}.rois.full'].reshape(g[f'{tag}.rois.shape']))
        ref_scores = torch.from_numpy(g[f'{tag}.roi_scores.full'].reshape(g[f'{tag}.roi_scores.shape']))
        # (and at most one corner per image that round() put on the other side of x.5 -- seen once in the suite: posenc, image 1,
        # x2 = 226 vs 227; reported, and the second stage is then checked stage-wise on the reference's RoIs)
        # -- tolerated for `posenc` ONLY, and only when the product's own pre-round coordinate is within 1e-4 of x.5 (diagnosis in
        # DESIGN 2 / scripts/posenc_flip.py: which reassociation crosses x.5); every other variant must match pixel for pixel
        flips = assert_rois_equal_up_to_near_ties(o['rois'], ref_rois, ref_scores, what=f'{tag} RoIs',
                                                  max_pixel_flips=1 if tag in HALF_PIXEL_TIES[split] else 0)
        assert_flips_are_half_pixel_ties(flips, o['rpn_bbox_reg'], args)
        # detections: the same (image, class, box) rows as the reference, scores within 1e-4
        dets = m.forward_second_stage(o['fpn_out'], ref_rois.cuda(), min_score=0.2, training=False) if flips else m(x, min_score=0.2)
        rows, ref = dets_to_rows(dets), g[f'{tag}.dets_min0.2']
        assert rows.shape == ref.shape, (rows.shape, ref.shape)
        rows, ref = rows[np.lexsort(rows[:, :6].T[::-1])], ref[np.lexsort(ref[:, :6].T[::-1])]
        assert np.array_equal(rows[:, :6], ref[:, :6]), f'{tag}: class / box assignments differ from the reference'
        assert np.abs(rows[:, 6] - ref[:, 6]).max() < 1e-4
    m.train(), crit.train()
    opt, _ = build_optimizer(m, args)
    bb, ids, lengths = synth.label_batch(0, 2)
    np.random.seed(3)
    loss = train_one_step(m, crit, opt, [x[:, 0], x[:, 0], bb, ids, lengths], args.clip_max_norm, 'cuda', negative_sample=False)
    assert all(np.isfinite(float(v.detach() if torch.is_tensor(v) else v)) for v in loss.values()) and np.isfinite(opt.grad_norm())
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in m.named_parameters()
               if n.startswith('attn') or n.startswith('fpn'))


@pytest.mark.parametrize('B', [1, 5])
def test_winograd_and_direct_convolution_paths_agree(model, B):
    """The FPN output convolutions through Winograd F(2x2,3x3) (default) and through the direct implicit-GEMM kernel:
    same proposals bit for bit, tensors within 5e-5 of O(4) values -- for batch sizes that the golden fixtures do not cover."""
    from birdsoundclassif_amd.nets import functional as Fn
    x = torch.from_numpy(synth.image_batch(70, B))[:, None].cuda()
    with torch.no_grad():
        a = model.forward_first_stage(x)
        da = model(x, min_score=0.1)
        Fn.WINOGRAD = False
        try:
            b = model.forward_first_stage(x)
            db = model(x, min_score=0.1)
        finally:
            Fn.WINOGRAD = True
    for fa, fb in zip(a['fpn_out'], b['fpn_out']):
        assert (fa - fb).abs().max().item() < 5e-5 * max(1.0, fb.abs().max().item() / 4)
    assert torch.equal(a['rois'], b['rois'])
    ra, rb = dets_to_rows(da), dets_to_rows(db)
    assert ra.shape == rb.shape and np.array_equal(ra[:, :6], rb[:, :6]) and np.abs(ra[:, 6] - rb[:, 6]).max() < 1e-5


def test_attention_projection_folded_into_the_fpn_laterals(model):
    """Evaluation mode (self_attention.Projected, _prep.lateral_of_projection): an attention level is handed to the FPN as (fm, ctx, W_o, b_o)
    and the lateral 1x1 computes W_l fm + (W_l W_o) ctx + (W_l b_o + b_l) instead of W_l (fm + W_o ctx + b_o) + b_l.  The handed-over level,
    materialised, IS the plain module's output bit for bit; the pyramid built from it equals the plain pyramid within fp32 rounding."""
    from birdsoundclassif_amd.nets import nbm_model
    from birdsoundclassif_amd.nets.self_attention import Projected, Scaled, materialize
    x = torch.from_numpy(synth.image_batch(0, 2))[:, None].cuda()
    with torch.no_grad():
        feats, _ = model.backbone(x.permute(0, 2, 3, 1).contiguous())
        plain = model.attn(feats)
        handed = model.attn(feats, defer_projection=True)
        assert [type(l).__name__ for l in handed] == ['Scaled', 'Scaled', 'Scaled', 'Projected', 'Projected']
        for a, b in zip(materialize(handed), materialize(plain)):
            assert torch.equal(a, b)
        lat = model.attn(feats, defer_projection=model.fpn.pt_wise)         # what the model passes: the laterals themselves
        assert all(l.lateral and l.ctx.shape[-1] == model.fpn.pt_wise['3'].weight.shape[0] for l in lat[3:])
        with pytest.raises(ValueError):
            materialize(lat)
        keep = nbm_model.DEFER_PROJECTION
        try:
            nbm_model.DEFER_PROJECTION = True
            on = model._fpn_nhwc(x)
            nbm_model.DEFER_PROJECTION = False
            off = model._fpn_nhwc(x)
        finally:
            nbm_model.DEFER_PROJECTION = keep
    for i, (a, b) in enumerate(zip(on, off)):
        err, ref = float((a - b).abs().max()), float(b.abs().max())
        assert err <= 2e-5 * max(1.0, ref), (i, err, ref)
        assert err > 0 or i < 0                                    # (the two routes do round differently: the switch took effect)
    # with a gradient to come the module keeps its own projection
    handed_grad = model.attn(feats, defer_projection=True)
    assert not any(isinstance(l, Projected) for l in handed_grad)
