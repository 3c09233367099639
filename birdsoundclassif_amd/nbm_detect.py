"""CLI of the reference's nbm_detect.py (nbm_model/nbm_detect.py:8-29): same flags, same `<wav>.txt = str(dict)`
outputs; `bird_dict.json` is looked up in the CWD like the reference, or given with --bird_dict.
Multi-GPU: launch one process per GPU (torchrun); files are sharded `files[rank::world]`, no collective."""
import argparse
import glob
import os


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--ckpt', type=str, help='directory with the json `args` and model_chkpt.pt')
    parser.add_argument('--audio_dir', type=str)
    parser.add_argument('--min_score', type=float, default=0.2)
    parser.add_argument('--batch', type=int, default=4)
    parser.add_argument('--bird_dict', type=str, default='bird_dict.json')
    args = parser.parse_args(argv)
    import torch
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
    from .run_detection import load_model, run_detection
    model, config = load_model(args.ckpt)
    files = sorted(glob.glob(os.path.join(args.audio_dir, '*.wav')))[rank::world]
    for i, wav_path in enumerate(files):
        output = run_detection(model, config, wav_path, args.bird_dict, min_score=args.min_score, bs=args.batch)
        print(f'{i + 1} / {len(files)} processed~')
        with open(wav_path.replace('.wav', '.txt'), 'w') as f:
            f.write(f'{str(output)}')


if __name__ == '__main__':
    main()
