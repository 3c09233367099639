"""GPU: the training driver (reference train.py:273-404) end to end on a synthetic dataset directory: device-decoded and
augmented batches -> train_one_step (a positive and a negative step) -> validation losses -> test-set AP -> checkpoint
-> resume."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from birdsoundclassif_amd import synth                                       # noqa: E402
from helpers import filler_state_dict                                         # noqa: E402
from oracle import png_ref                                                    # noqa: E402  (dataset writer)


def test_main_loop_trains_validates_tests_and_resumes(tmp_path, monkeypatch):
    from birdsoundclassif_amd import train as T
    data = tmp_path / 'dataset'
    synth.write_image_dataset(str(data), png_ref.encode_png_gray8)
    xc = data / 'test_files' / 'XC_annots'
    xc.mkdir(parents=True)
    synth.write_wav(str(xc / 'rec0.wav'), synth.clip_pcm16(3))
    (xc / 'rec0.txt').write_text('0.500000\t1.200000\tsp1\n\\\t2000.000000\t5000.000000\n')
    monkeypatch.chdir(tmp_path)
    (tmp_path / 'bird_dict.json').write_text(json.dumps({f'sp{i}': i for i in range(1, 151)}))
    args = T.default_args(device='cuda', data_path=str(data), save_dir=str(tmp_path / 'models'), model_name='m',
                          batch_size=1, num_workers=0, validation_prop=0.67, max_steps=4, neg_step_freq=2,
                          first_neg_step=0, seed=3)
    args.val_freq, args.log_freq, args.host_noise = 2, 1, False

    # start from the filler weights (a fresh init gives an RPN that may fail to propose anything)
    real_build = T.build_optimizer

    def build_with_filler(model, a):
        model.load_state_dict(filler_state_dict())
        from birdsoundclassif_amd.nets import _prep
        _prep.bump()
        return real_build(model, a)
    monkeypatch.setattr(T, 'build_optimizer', build_with_filler)
    steps = T.main(args)
    assert steps == 4
    save_dir = tmp_path / 'models' / 'm'
    rows = [json.loads(l) for l in open(save_dir / 'scalars.jsonl')]
    tags = {r['tag'] for r in rows}
    assert {'Training_Loss/first_class_loss', 'Training_Loss/sec_class_loss', 'Val_Loss/first_class_loss',
            'Test_metrics/AP', 'Test_metrics/mRec'} <= tags
    assert all(np.isfinite(r['value']) for r in rows if not r['tag'].startswith('Test_metrics/m'))
    assert any(r['tag'] == 'Training_Loss/first_neg_class_loss' and r['value'] > 0 for r in rows)   # step 2 was negative
    assert json.load(open(save_dir / 'args'))['batch_size'] == 1

    # checkpoint -> resume continues from the stored step counter with the stored split
    from birdsoundclassif_amd.nets import build_model
    model, crit = build_model(args)
    model.to('cuda')
    opt, sch = real_build(model, args)
    T.save(str(save_dir), model, 7, 123, 0.5, 'last', opt, sch, np.array([2]), np.array([0, 1]))
    args.max_steps = 125
    assert T.main(args) == 125


def test_resume_continues_the_uninterrupted_run(tmp_path):
    """N steps -> save -> resume into a FRESH model + optimizer -> one more step == N + 1 uninterrupted steps (the Adam
    moments live in the optimizer's flat buffers; `load_state_dict` must put the loaded ones there)."""
    from birdsoundclassif_amd import train as T
    from birdsoundclassif_amd.nets import build_model, _prep
    args = T.default_args(device='cuda')
    img = torch.from_numpy(synth.image_batch(0, 2))
    neg_img = torch.from_numpy(synth.image_batch(100, 2))
    bb, ids, lengths = synth.label_batch(0, 2)
    batch = [img, neg_img, bb, ids, lengths]

    def fresh():
        model, crit = build_model(args)
        model.load_state_dict(filler_state_dict())
        model = model.cuda().train()
        crit.train()
        opt, sch = T.build_optimizer(model, args)
        return model, crit, opt, sch

    def run(model, crit, opt, steps):
        for i in steps:
            np.random.seed(100 + i)
            T.train_one_step(model, crit, opt, batch, args.clip_max_norm, 'cuda', negative_sample=(i == 1))

    a, ca, oa, sa = fresh()
    run(a, ca, oa, range(2))                                   # one positive, one negative step
    T.save(str(tmp_path), a, 0, 2, 99, 'last', oa, sa, np.array([0]), np.array([1]))
    run(a, ca, oa, range(2, 3))

    b, cb, ob, sb = fresh()
    b, ob, sb, *_ = T.resume_training(str(tmp_path), b, ob, sb, args.lr_drop)
    f = ob._flat[0]
    st = ob.state[f['spans'][0][0]]
    assert st['exp_avg'].data_ptr() == f['m'].data_ptr() and float(f['m'].abs().sum()) > 0 and st['step'] == 2
    # buffers (BatchNorm running statistics) travel with the checkpoint too
    run(b, cb, ob, range(2, 3))
    # the weight-gradient kernels sum with fp32 atomics (arrival order), so two runs agree to rounding, not bit for bit;
    # a resume that lost the moments is off by ~3 lr = 3e-4 (no bias correction at step count 3 with m = v = 0)
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        assert float((p - q).abs().max()) < 2e-6, n
    for (n, p), q in zip(a.named_buffers(), b.named_buffers()):
        assert float((p.float() - q[1].float()).abs().max()) < 1e-5, n
