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
