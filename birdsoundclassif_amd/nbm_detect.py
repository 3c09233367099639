"""CLI of the reference's nbm_detect.py (nbm_model/nbm_detect.py:8-29): same flags, same `<wav>.txt = str(dict)`
outputs; `bird_dict.json` is looked up in the CWD like the reference, or given with --bird_dict.

Route: files that are equal-length single-window clips (mono 16-bit PCM, 22.05 / 44.1 kHz, <= 3.06 s: `bulk.bulk_groups`) go
through the pipelined hipGraph loop of `bulk.detect_files` in batches of --bulk_batch, every clip an independent batch of one --
exactly what the reference's per-file loop computes for them; everything else (long recordings, other formats) goes through the
per-file `run_detection` driver with --batch windows per model call, like the reference.  --no_bulk forces the per-file driver.
Multi-GPU: launch one process per GPU (torchrun); files are sharded `files[rank::world]`, no collective."""
import argparse
import glob
import json
import os

BULK_MIN_FILES = 8          # below this a graph capture (3 batch-sized steps) costs more than it saves


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--ckpt', type=str, help='directory with the json `args` and model_chkpt.pt')
    parser.add_argument('--audio_dir', type=str)
    parser.add_argument('--min_score', type=float, default=0.2)
    parser.add_argument('--batch', type=int, default=4)
    parser.add_argument('--bird_dict', type=str, default='bird_dict.json')
    parser.add_argument('--bulk_batch', type=int, default=64, help='clips per graph replay on the bulk route')
    parser.add_argument('--no_bulk', action='store_true', help='per-file driver for every file')
    args = parser.parse_args(argv)
    import torch
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
    from . import bulk
    from .run_detection import load_model, run_detection
    model, config = load_model(args.ckpt)
    files = sorted(glob.glob(os.path.join(args.audio_dir, '*.wav')))[rank::world]
    groups, rest = ({}, files) if args.no_bulk else bulk.bulk_groups(files)
    done = 0
    if groups:
        with open(args.bird_dict, 'r') as f:
            bird_dict = json.load(f)
        if getattr(config, 'tf_rcnn', False):       # per-image RoI counts are not built for the transformer head
            rest, groups = sorted(rest + [f for g in groups.values() for f in g]), {}
    for key, group in sorted(groups.items()):
        if len(group) < BULK_MIN_FILES:
            rest.extend(group)
            continue
        batch = min(args.bulk_batch, -(-len(group) // 8) * 8)
        try:
            try:
                bulk.detect_files(model, group, batch=batch, min_score=args.min_score, bird_dict=bird_dict, write_txt=True,
                                  keep_results=False)
            except torch.cuda.OutOfMemoryError:
                # two lanes = a second set of persistent scratch and graph-pool activations: degrade to one batch in flight
                torch.cuda.empty_cache()
                print(f'bulk route: out of device memory with two batches in flight; retrying the group of {len(group)} clips with one lane')
                bulk.detect_files(model, group, batch=batch, min_score=args.min_score, bird_dict=bird_dict, write_txt=True,
                                  keep_results=False, lanes=1)
        except (ValueError, NotImplementedError, OSError) as exc:
            # a file whose header disagrees with its data, a truncated or changing file, a decode the bulk reader does not do: the
            # per-file driver (= the reference's behaviour) takes the whole group (it rewrites the txt files the bulk route finished)
            print(f'bulk route gave up on a group of {len(group)} clips ({type(exc).__name__}: {exc}); they go through the per-file driver')
            rest.extend(group)
            continue
        done += len(group)
        print(f'{done} / {len(files)} processed~ (bulk route: {len(group)} clips of {key[1]} samples @ {key[0]} Hz)')
    for wav_path in sorted(rest):
        output = run_detection(model, config, wav_path, args.bird_dict, min_score=args.min_score, bs=args.batch)
        done += 1
        print(f'{done} / {len(files)} processed~')
        with open(wav_path.replace('.wav', '.txt'), 'w') as f:
            f.write(f'{str(output)}')


if __name__ == '__main__':
    main()
