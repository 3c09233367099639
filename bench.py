#!/usr/bin/env python3
"""Benchmark of the NBM detect hot path on MI355X (contract: see the task statement / DESIGN.md §Measurement).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 64]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic input per rank:
    PCM16 clips (3 s @ 22.05 kHz, already resident in HBM) -> HIP front end (2x up-sample, STFT-dB, normalise,
    window) -> HIP detector forward (ResNet-50 + attention + FPN + RPN + proposals + RoI pool + RCNN head +
    post-processing) -> per-clip detections on the host (reference output structure).
Workload = BASELINE.json configs[1] (batch 64 per GPU, fp32).  Multi-GPU: clips are independent, every rank
runs its own batch (weak scaling, no data-path collective); value = all clips of all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0, carrying `roofline` (dominant kernel: the deep-K instantiation of the fp32-MFMA implicit-GEMM
kernel, all its launches of a step timed live with HIP events on the launch stream), `bulk_inference` (configs[4] on this
rank's shard: wav files -> txt files end to end), `train_step` (configs[2] / [3]) and `cpu_baseline` (the oracle port on the
host cores, bounded sample, N=1 only).  `python bench.py --gpus N` starts its N ranks itself (see launch_ranks).

Order of the detect leg: `--spin-up` untimed steps (idle -> steady clocks), W warm-up steps, K eager steps with HIP events around
the dominant kernel's launches (-> `roofline`, `eager_with_events`), then the same step captured in a hipGraph: W warm-up replays and
K TIMED replays = `value` / `ms_per_step` (falls back to the eager figure, with a note in `config.launch`, if the capture fails).
Timed regions run with Python's cyclic garbage collector disabled, as `timeit` does (`gc_disabled_in_timed_regions`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402

PMC_DOMINANT = 'r05_pmc_dominant.json'    # committed rocprofv3 --pmc passes of `python bench.py` (scripts/pmc_dominant.py)
PMC_WGRAD = 'r05_pmc_wgrad.json'          # ... of scripts/trainbench.py: the weight-gradient kernels (scripts/r04_profiles.sh)
PMC_DGRAD = 'r05_pmc_dgrad.json'          # ... the data-gradient kernels
FP32_MFMA_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
BF16_MFMA_PEAK_TFLOPS = 2500.0         # dense bf16 MFMA (v_mfma_f32_32x32x16_bf16: 1024 FLOP/clk/SIMD), same guide
FP64_MFMA_PEAK_TFLOPS = 78.6           # v_mfma_f64_16x16x4_f64: half the fp32 rate on this part (64 cycles / 2048 FLOP / SIMD)
FPN0_GFLOP_PER_CLIP = 170.322          # fpn.out_convs.4, 3x3 384->256 @188x512
FWD_GFLOP_PER_CLIP = 325.56            # SURVEY.md §6 (conv + addmm + bmm, forward)
TRAIN_GFLOP_PER_CLIP = 993.45          # SURVEY.md §6 (fwd + bwd, positive step)


def conv_algorithmic_bytes(B, H, W, Cin, N, k, stride, groups, kind):
    """ALGORITHMIC HBM bytes of one implicit-GEMM launch (every operand once): forward / data gradient = activation in +
    weights + activation out; weight gradient = both activations + the weight-sized result.  Epilogue operands that depend on the call
    site (residual, ReLU mask, merged coarse map) are NOT included: the wasted-traffic ratio traffic / algorithmic is an upper bound."""
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    a_in, a_out, w = B * H * W * Cin, B * Ho * Wo * N, N * k * k * Cin
    return 4.0 * groups * (a_in + a_out + w)


def pmc_traffic(name, same_workload):
    """`traffic` (HBM bytes per launch, 2 x FETCH_SIZE + WRITE_SIZE) of a kernel family from THIS round's committed rocprofv3 --pmc
    passes (counters cannot be read live), or null with the reason."""
    path = os.path.join(ROOT, 'profiles', name)
    try:
        pj = json.load(open(path))
    except Exception as exc:
        return {'traffic': None, 'traffic_note': f'profiles/{name} not readable ({type(exc).__name__}): no PMC pass of this round for this kernel'}
    if not same_workload:
        return {'traffic': None, 'traffic_note': f'profiles/{name} was taken at another batch size'}
    return {'traffic': pj.get('traffic_bytes_per_launch'), 'traffic_unit': 'bytes/launch (2*FETCH_SIZE + WRITE_SIZE, mean over the launches)',
            'mfma_busy_frac_pmc': pj.get('mfma_busy_frac'),
            'traffic_note': f'profiles/{name}: separate rocprofv3 --pmc passes of scripts/trainbench.py (same step, same batch), not measured in this run'}


def train_bench(rank, world, dist, batch, steps, warmup, mix_steps=10):
    """BASELINE.json configs[2]/[3]: data-parallel training step, `batch` synthetic clips + random boxes/labels per GPU,
    HIP fwd/bwd + fused clip/AdamW, one RCCL all-reduce of the flat fp32 gradients per step when world > 1.
    Timed twice: `steps` positive steps (the headline of this leg), then `mix_steps` steps in the reference's schedule
    (train.py:343: every `neg_step_freq` = 10th step trains on the negative images, 1000 RoIs per image through the head)."""
    from birdsoundclassif_amd import ops, synth
    from birdsoundclassif_amd.nets import build_model
    from birdsoundclassif_amd.train import default_args, build_optimizer, train_one_step
    args = default_args(device='cuda')
    model, crit = build_model(args)
    model.load_state_dict(synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}))
    model = model.cuda().train()
    crit.train()
    opt, _ = build_optimizer(model, args)
    # `batch` DISTINCT images / label sets per rank (8 tiled clips never show the batch-coupled paths a mixed batch)
    img = torch.from_numpy(synth.image_batch(rank * batch, batch)).cuda()
    neg = torch.from_numpy(synth.image_batch(100000 + rank * batch, batch)).cuda()
    bb, ids, lens = synth.label_batch(rank * batch, batch)
    data = [img, neg, bb, ids, list(lens)]                        # labels stay on the host, as a DataLoader delivers them
    np.random.seed(1000 + rank)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    dt_local = [0.0]

    def timed(schedule):
        import gc
        gc.collect()
        gc.disable()                       # like timeit: a generation-2 collection inside a 3-5 step window is +30-90 ms on one step
        try:
            sync_all()
            t0 = time.perf_counter()
            for negative in schedule:
                loss = train_one_step(model, crit, opt, data, args.clip_max_norm, 'cuda', negative_sample=negative)
            torch.cuda.synchronize()
            dt_local[0] = time.perf_counter() - t0     # this rank's own time (before the closing barrier)
            sync_all()
            dt = time.perf_counter() - t0
        finally:
            gc.enable()
        if dist is not None:
            t = torch.tensor([dt], device='cuda', dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, loss

    for _ in range(warmup + 6):            # + 6: the persistent gradient maps, kept operands and pinned rings appear during the first
        train_one_step(model, crit, opt, data, args.clip_max_norm, 'cuda', negative_sample=False)      # steps (scripts/steptimes.py)
    # first with the instruments on (HIP events around every data- / weight-gradient launch, FLOP counters: ~600 event records and a
    # few dozen small copies per step): roofline of the weight-gradient kernels, executed FLOPs
    ops.FLOPS = [0.0]                      # executed MFMA FLOPs of every GEMM launch (Winograd-domain counts where that path runs)
    ops.PROFILE_BWD = []
    dt_instr, _ = timed([False] * steps)
    exec_gflop_per_clip = ops.flops_total() / (steps * batch) / 1e9
    prof, ops.PROFILE_BWD, ops.FLOPS = ops.PROFILE_BWD, None, None
    from birdsoundclassif_amd import train as T
    T.exchange_stats_reset(dist is not None)
    dt, loss = timed([False] * steps)      # the figure of this leg: nothing attached
    # data parallel: what the exchange cost THIS rank, and how far the ranks' own step times are apart (all-gathered below)
    exch = T.exchange_stats_summary()
    T.exchange_stats_reset(False)
    per_rank = None
    if dist is not None:
        mine = torch.tensor([dt_local[0] / steps * 1e3, exch['exchange_ms'] if exch else 0.0, exch['control_ms'] if exch else 0.0,
                             float(exch['overlapped_steps']) if exch else 0.0], device='cuda', dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rows = [t.tolist() for t in allr]
        per_rank = {'ms_per_step': [r[0] for r in rows], 'ms_per_step_min': min(r[0] for r in rows), 'ms_per_step_max': max(r[0] for r in rows),
                    'exchange_ms': [r[1] for r in rows], 'control_ms': [r[2] for r in rows],
                    'overlapped_steps': [int(r[3]) for r in rows], 'timed_steps': steps,
                    'note': 'per rank, means over the timed steps: ms_per_step = the rank\'s own wall time between the two barriers; '
                            'exchange_ms = HIP events from the start of the first flat-buffer all-reduce (inside the backward pass when '
                            'overlapped) to the end of the last; control_ms = host wall time of the touched-bitmap all-reduce over gloo '
                            '(includes waiting for the slowest rank\'s host)'}
    # dominant backward kernel: igemm_tn_kernel<128,0,0> (weight gradients); its largest launches are the 36
    # Winograd F(4x4,3x3)-domain TN GEMMs of fpn.out_convs.4 (groups = 36, one launch per batch chunk)
    wg = {}
    for tag, e0, e1 in prof:
        if tag[0] == 'wgrad':
            wg.setdefault(tag, []).append(e0.elapsed_time(e1))
    bwd_roof = None
    if wg:
        def gflop(t):
            _, b, H, W, Cin, N, k, stride, groups = t
            return 2.0 * b * ((H - 1) // stride + 1) * ((W - 1) // stride + 1) * N * Cin * k * k * groups / 1e9
        all_ms = sum(sum(v) for v in wg.values())
        all_gf = sum(gflop(t) * len(v) for t, v in wg.items())
        n_launch = sum(len(v) for v in wg.values())
        top = sorted(wg, key=lambda t: -sum(wg[t]))[:3]
        bwd_roof = {'bound': 'mfma', 'kernel': 'igemm_tn_kernel<BN,MODE,0>: every weight-gradient launch of a step (TN GEMMs, split-K over '
                                                'pixels / Winograd tiles with fp32 atomics)',
                    'achieved': all_gf / all_ms, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': all_gf / all_ms / FP32_MFMA_PEAK_TFLOPS,
                    'avg_launch_ms': all_ms / n_launch, 'launches': n_launch, 'executed_GFLOP_per_launch': all_gf / n_launch,
                    'all_wgrad_ms_per_step': all_ms / steps,
                    'largest_launches': [{'B,H,W,Cin,N,k,stride,groups': list(t[1:]), 'ms': sum(wg[t]) / len(wg[t]),
                                          'launches': len(wg[t]), 'executed_TFLOPs': gflop(t) * len(wg[t]) / sum(wg[t])} for t in top],
                    'algorithmic_bytes_per_launch': sum(conv_algorithmic_bytes(*t[1:6], t[6], t[7], t[8], 'wgrad') * len(v) for t, v in wg.items()) / n_launch,
                    **pmc_traffic(PMC_WGRAD, batch == 128)}
    dg = {}
    for tag, e0, e1 in prof:
        if tag[0] == 'dgrad':
            dg.setdefault(tag, []).append(e0.elapsed_time(e1))
    dgrad_roof = None
    if dg:
        def dgflop(t):
            _, b, H, W, Cin, N, k, stride, groups = t
            return 2.0 * b * ((H - 1) // stride + 1) * ((W - 1) // stride + 1) * N * Cin * k * k * groups / 1e9
        d_ms = sum(sum(v) for v in dg.values())
        d_gf = sum(dgflop(t) * len(v) for t, v in dg.items())
        d_n = sum(len(v) for v in dg.values())
        dtop = sorted(dg, key=lambda t: -sum(dg[t]))[:4]
        dgrad_roof = {'bound': 'mfma', 'kernel': 'igemm_nn_kernel<BN,STAGES>: every data-gradient launch of a step (NN GEMMs on the KRSC weights)',
                      'achieved': d_gf / d_ms, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': d_gf / d_ms / FP32_MFMA_PEAK_TFLOPS,
                      'avg_launch_ms': d_ms / d_n, 'launches': d_n, 'all_dgrad_ms_per_step': d_ms / steps,
                      'largest_launches': [{'B,H,W,Cin,N,k,stride,groups': list(t[1:]), 'ms': sum(dg[t]) / len(dg[t]), 'launches': len(dg[t]),
                                            'executed_TFLOPs': dgflop(t) * len(dg[t]) / sum(dg[t])} for t in dtop],
                      'algorithmic_bytes_per_launch': sum(conv_algorithmic_bytes(*t[1:6], t[6], t[7], t[8], 'dgrad') * len(v) for t, v in dg.items()) / d_n,
                      'algorithmic_bytes_note': 'G + weights + dX; the shortcut-gradient and ReLU-mask operands of the bottleneck epilogues (as '
                                                'large as dX each) are not included: profiles/r05_dgrad_attribution.txt has them per launch',
                      **pmc_traffic(PMC_DGRAD, batch == 128)}
    pos = {'ms_per_step': dt / steps * 1e3, 'clips_per_s': world * batch * steps / dt, 'ms_per_step_instrumented': dt_instr / steps * 1e3}
    # SURVEY 8d (ii), second figure: the same step with the front end fused in -- PCM16 clips resident in HBM -> 2x up-sample -> STFT-dB
    # -> normalise / window (the detect leg's front end) -> images -> training step.  The reference trains from PNG spectrograms
    # (image_dataset.py:43-44), so the headline of this leg stays the image-fed step.
    with_fe = None
    try:
        from birdsoundclassif_amd.nbm_datasets.prepare_dataset import SpectrogramFrontEnd
        fe = SpectrogramFrontEnd('cuda')
        pcm = torch.from_numpy(np.tile(synth.clip_batch_pcm16(rank * 8, 8), (-(-batch // 8), 1))[:batch].copy()).cuda()

        def fe_step():
            with torch.no_grad():
                im, _ = fe(pcm, 22050)
            d2 = [im[:, 0], neg, data[2], data[3], data[4]]
            return train_one_step(model, crit, opt, d2, args.clip_max_norm, 'cuda', negative_sample=False)
        fe_step()
        import gc
        gc.collect()
        gc.disable()
        try:
            sync_all()
            t0 = time.perf_counter()
            for _ in range(steps):
                fe_step()
            sync_all()
            dtf = time.perf_counter() - t0
        finally:
            gc.enable()
        if dist is not None:
            t = torch.tensor([dtf], device='cuda', dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtf = float(t.item())
        with_fe = {'ms_per_step': dtf / steps * 1e3, 'clips_per_s': world * batch * steps / dtf, 'steps': steps,
                   'what': 'PCM16 (3 s @ 22.05 kHz, resident in HBM) -> HIP front end -> images -> the same positive training step'}
    except Exception as exc:                      # informational figure: never lose the leg over it
        with_fe = {'error': f'{type(exc).__name__}: {exc}'[:300]}
    # the reference's schedule: one negative step in ten
    mix = None
    if mix_steps:
        # one untimed negative step first: its 1000 RoIs per image grow the allocator's pools by ~17 GiB (hipMalloc under a
        # running stream: 300 ms once per process, 35 ms / step if it lands inside a 10-step window)
        train_one_step(model, crit, opt, data, args.clip_max_norm, 'cuda', negative_sample=True)
        dtm, _ = timed([(i % 10) == 9 for i in range(mix_steps)])
        mix = {'steps': mix_steps, 'negative_every': 10, 'ms_per_step': dtm / mix_steps * 1e3,
               'clips_per_s': world * batch * mix_steps / dtm}
    # opt-in mode beside the headline of this leg (never IN it): the same positive step with every deep-K GEMM -- forward, 1x1 data gradients,
    # plain weight gradients -- on the bf16 matrix pipe through split fp32 operands (csrc/igemm_split.hip, igemm_split_tn.hip)
    split_leg = None
    if world == 1:
        was = os.environ.get('NBM_SPLIT_BF16')
        os.environ['NBM_SPLIT_BF16'] = '1'
        try:
            timed([False, False])
            dts, _ = timed([False] * steps)
            split_leg = {'default': False, 'switch': 'NBM_SPLIT_BF16=1', 'ms_per_step': dts / steps * 1e3, 'clips_per_s': batch * steps / dts,
                         'steps': steps, 'what': 'forward deep-K GEMMs (igemm_split_kernel), deep-K 1x1 data gradients (the same kernel on '
                                                 'transposed scaled weights, mask / shortcut in its epilogue), plain weight-gradient GEMMs with >= 192 '
                                                 'rows (igemm_split_tn_kernel); everything else on the fp32 matrix instruction'}
        except Exception as exc:
            split_leg = {'error': f'{type(exc).__name__}: {exc}'[:300]}
        finally:
            if was is None:
                os.environ.pop('NBM_SPLIT_BF16', None)
            else:
                os.environ['NBM_SPLIT_BF16'] = was
    if os.environ.get('NBM_BENCH_MEMLOG') == '1':
        st = torch.cuda.memory_stats()
        print(f'bench: train leg: alloc retries {st.get("num_alloc_retries")}, ooms {st.get("num_ooms")}, peak allocated '
              f'{torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB, peak reserved {torch.cuda.max_memory_reserved() / 2 ** 30:.1f} GiB',
              file=sys.stderr, flush=True)
    del model, opt
    torch.cuda.empty_cache()
    v = pos['clips_per_s']
    exec_tflops = v / world * exec_gflop_per_clip / 1e3
    return {'value': v, 'unit': 'clips/s', 'batch_per_gpu': batch, 'distinct_images': batch, 'global_batch': world * batch, 'steps': steps,
            'ms_per_step': pos['ms_per_step'], 'ms_per_step_with_the_instruments_on': pos['ms_per_step_instrumented'],
            'parallelism': f'dp{world}', 'per_rank': per_rank, 'split_bf16': split_leg,
            'exchange': None if world == 1 else ('one all-reduce (AVG) per flat gradient buffer; the non-backbone buffer starts inside the backward '
                                                 'pass (hook on the backbone\'s last tap), the backbone buffer after it' if T.DP_OVERLAP else
                                                 'one all-reduce (AVG) per flat gradient buffer, after the backward pass'),
            'executed_GFLOP_per_clip': exec_gflop_per_clip, 'executed_TFLOPs_per_gpu': exec_tflops,
            'executed_frac_of_mfma_peak': exec_tflops / FP32_MFMA_PEAK_TFLOPS,
            'direct_conv_equivalent_GFLOP_per_clip': TRAIN_GFLOP_PER_CLIP,
            'direct_conv_equivalent_TFLOPs_per_gpu': v / world * TRAIN_GFLOP_PER_CLIP / 1e3,
            'note': 'executed = MFMA FLOPs the launches really perform (Winograd-domain counts for the 3x3 / stride-1 layers: '
                    'F(2x2,3x3) forward, F(4x4,3x3) data and weight gradients; the finest FPN output map is computed on demand: tiles '
                    'with a reader, forward and weight gradient); direct_conv_equivalent = SURVEY 993.45 GFLOP/clip '
                    'and is NOT a roofline figure',
            'reference_schedule_9_positive_1_negative': mix,
            'roofline_backward': bwd_roof, 'roofline_backward_data_gradients': dgrad_roof, 'with_front_end': with_fe,
            'final_loss': {k: float(x.detach()) if torch.is_tensor(x) else float(x) for k, x in loss.items()},
            'workload': 'BASELINE.json configs[2]/[3]: positive training step (fwd + bwd + clip + AdamW), fp32'}


def bulk_bench(model, rank, world, dist, n_files, batch, min_score, headline_clips_per_s_per_gpu, lanes=2):
    """BASELINE.json configs[4] on this rank's shard, END TO END: `n_files` synthetic 3 s wav files on tmpfs -> `<wav>.txt` files
    through `bulk.detect_files` (reader thread -> pinned batch -> H2D -> hipGraph replay of front end + detector + device
    post-processing -> D2H of the compact rows -> writer thread: reference output dict -> txt).  Timed region = first file opened
    to last txt file closed (model load and the one-time graph capture are outside, as they are for a 100k-file shard); one
    untimed pass over two batches warms the page cache of nothing (tmpfs) but the threads / pinned slots."""
    import shutil
    import tempfile
    from birdsoundclassif_amd import bulk, synth
    root = tempfile.mkdtemp(prefix=f'nbm_bulk_r{rank}_', dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
    try:
        base = [synth.clip_pcm16(rank * 64 + i) for i in range(64)]      # 64 distinct clips, cycled: a batch of 64 holds 64 different files
        files = []
        for i in range(n_files):
            path = os.path.join(root, f'clip{i:06d}.wav')
            synth.write_wav(path, base[i % 64], 22050)
            files.append(path)
        names = {f'Species {i}': i for i in range(1, 151)}
        det = bulk.GraphedDetector(model, batch, 66150, 22050, min_score=min_score, independent=True, lanes=max(1, lanes))
        kw = dict(batch=batch, min_score=min_score, bird_dict=names, write_txt=True, keep_results=False, detector=det)
        bulk.detect_files(model, files[:2 * batch], **kw)
        for f in files[:2 * batch]:
            os.remove(bulk.txt_path(f))
        stats = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bulk.detect_files(model, files, stats=stats, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0          # this rank's shard; the caller takes the max over the ranks (no collective in here:
        n_txt = sum(1 for f in files if os.path.isfile(bulk.txt_path(f)))
        import ast
        n_det = sum(len(v['scores']) for f in files[:batch] for v in ast.literal_eval(open(bulk.txt_path(f)).read()).values())
        det.close()
        del det
        torch.cuda.empty_cache()
        # a rank that fails in this optional leg must not leave the others in a barrier)
        return {'value': None, 'unit': 'clips/s', 'files_per_gpu': n_files, 'txt_files_written_rank0': n_txt, 'batch': batch,
                'wall_s': dt, 'ratio_to_resident_hbm_headline': None,
                'detections_in_first_batch': n_det,
                'stages_rank0': {k: (round(x, 4) if isinstance(x, float) else x) for k, x in stats.items()},
                'workload': 'BASELINE.json configs[4] on one shard per GPU: synthetic 3 s 22.05 kHz PCM16 wav files on tmpfs -> '
                            'txt files (reference CLI output), hipGraph-captured detect loop, every clip an independent batch of '
                            'one (= the reference\'s per-file loop), file reads / H2D / replay / D2H / txt writes overlapped'}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def cpu_baseline(n_clips=16, batch=4, threads=32, threads8_clips=8):
    """Oracle port (numpy front end + pure-torch detector) on the host cores; bounded sample (~10-20 s).
    32 torch threads: measured fastest on the GPU box (8: 1.36, 16: 1.43, 32: 1.81, 64: 1.13, 128: 0.54 clips/s for
    the detector alone); one untimed warm-up batch."""
    from birdsoundclassif_amd import synth
    from oracle import frontend_ref as FR, nets_ref as O
    from birdsoundclassif_amd.nets import build_model
    from birdsoundclassif_amd.train import default_args
    model, _ = build_model(default_args(device='cpu'))
    sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    cfg = O.make_cfg()
    threads = min(threads, os.cpu_count() or threads)
    torch.set_num_threads(threads)

    def run(start, count):
        pcm = synth.clip_batch_pcm16(start, count)
        imgs = [FR.process_waveform(FR.upsample2x_pcm16(p).astype(np.float32) / np.float32(32768))[0][0] for p in pcm]
        O.forward(sd, cfg, torch.from_numpy(np.stack(imgs))[:, None], min_score=0.2)

    with torch.no_grad():
        run(900, 2)                                             # warm-up (thread pool, allocator)
        t0 = time.perf_counter()
        done = 0
        for s in range(0, n_clips, batch):
            run(1000 + s, batch)
            done += batch
        dt = time.perf_counter() - t0
    res = {'value': done / dt, 'unit': 'clips/s', 'cores': threads, 'kind': 'port',
           'sample': f'{done} synthetic 3 s clips in batches of {batch}: oracle front end (numpy float64 FFT, 1 thread) + '
                     f'oracle detector forward (torch CPU fp32, {threads} threads), {dt:.1f} s'}
    # the 8-thread figure (comparable with the survey container's 8 cores, BASELINE.md 2: ~1.1 clips/s), same code
    if threads != 8 and threads8_clips > 0:
        torch.set_num_threads(8)
        with torch.no_grad():
            run(900, 2)
            t0 = time.perf_counter()
            d8 = 0
            for s0 in range(0, threads8_clips, batch):
                run(1000 + s0, batch)
                d8 += batch
            dt8 = time.perf_counter() - t0
        res['threads8'] = {'value': d8 / dt8, 'unit': 'clips/s', 'cores': 8, 'sample': f'{d8} of the same clips, torch.set_num_threads(8), {dt8:.1f} s'}
        torch.set_num_threads(threads)
    return res


def graph_replay_agreed(dist, ok_here):
    """Multi-rank detect leg: did EVERY rank finish its timed replays?  (MIN over the ranks of a success flag.)"""
    flag = torch.tensor([1.0 if ok_here else 0.0], device='cuda')
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return float(flag.item()) > 0.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--spin-up', dest='spin_up', type=int, default=20,
                    help='untimed steps BEFORE the --warmup steps: the card takes ~20 steps (1.5 s) from idle to its steady clocks -- '
                         'scripts/detectloop.py: first block of 20 steps after 5 warm-up steps 1.5-3 %% slower than the next ones; '
                         'reported as spin_up_steps in the line')
    ap.add_argument('--min-score', type=float, default=0.2)
    ap.add_argument('--lanes', type=int, default=2,
                    help='detect steps in flight together on one GPU: parallel branches (one stream each) of ONE captured graph '
                         '(bulk.GraphedDetector(lanes=k)): the kernel tails of one step are filled by the other -- 65.2 against 68.9 ms '
                         'per batch (profiles/r04_two_lanes.txt); 1 = one step at a time')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--train-batch', type=int, default=128)
    ap.add_argument('--train-steps', type=int, default=8,
                    help='timed steps of the train leg (every timed region starts with an empty GPU queue: the host side of its first step '
                         '-- 160 ms of anchor targets at B = 128 -- is not hidden behind a previous step; 2-3 steps measured 470-500 ms / step '
                         'where 5-8 measure 427)')
    ap.add_argument('--no-train', action='store_true')
    ap.add_argument('--tile-clips', dest='tile_clips', type=int, default=0,
                    help='A/B switch: repeat the first N synthetic clips over the batch instead of B distinct clips (0 = distinct, the default)')
    ap.add_argument('--no-dense-reference', dest='no_dense_reference', action='store_true')
    ap.add_argument('--no-split-leg', dest='no_split_leg', action='store_true',
                    help='skip the extra detect leg with the deep-K GEMMs on the bf16 matrix pipe (split fp32 operands, opt-in mode)')
    ap.add_argument('--bulk-files', type=int, default=16384,
                    help='bulk_inference leg (configs[4] on this rank\'s shard): wav files on tmpfs -> txt files; 0 = skip')
    return ap.parse_args(argv)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def host_threads_per_rank(world):
    """Host threads one rank may use: usable cores (affinity mask where the platform has one) / ranks on this host, at least 1."""
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    return max(1, cores // max(1, world))


def launch_ranks(n, argv, script=None, timeout=None):
    """`python bench.py --gpus N` without torchrun: start N rank processes of this script (one per GPU; RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment) BEFORE this process has made any GPU call -- it never makes one: it only
    counts devices, waits, and relays rank 0's JSON line.  Nothing is re-exec'ed.  Returns the exit code: 0 only when every
    rank exited 0 and rank 0 printed its line; the first failing rank takes the others down."""
    import subprocess
    script = script or os.path.abspath(__file__)
    env = dict(os.environ)
    env.update(WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()), NBM_BENCH_LAUNCHER='self')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    # N ranks share one host: cap every rank's host thread pools (torch intra-op / OpenMP / MKL: the NumPy target layers and the
    # CPU side of torch each default to ALL cores) at cores / N, so that 8 ranks do not run 8 x 128 threads on 128 cores
    per_rank = host_threads_per_rank(n)
    for k in ('OMP_NUM_THREADS', 'MKL_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'NBM_HOST_THREADS'):
        env.setdefault(k, str(per_rank))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=e,
                                      stdout=subprocess.PIPE if r == 0 else None, text=True if r == 0 else None))
    out0, rc = '', 0
    t_end = time.time() + timeout if timeout else None
    try:
        import threading
        buf = []
        rd = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
        rd.start()
        alive = set(range(n))
        while alive:
            for r in sorted(alive):
                c = procs[r].poll()
                if c is None:
                    continue
                alive.discard(r)
                if c != 0 and rc == 0:
                    rc = c if c > 0 else 1
                    print(f'bench launcher: rank {r} exited with code {c}; stopping the other ranks', file=sys.stderr, flush=True)
                    for q in alive:
                        procs[q].terminate()
            if t_end and time.time() > t_end and alive:
                rc = rc or 124
                print('bench launcher: timeout; stopping the ranks', file=sys.stderr, flush=True)
                for q in alive:
                    procs[q].terminate()
                t_end = None
            time.sleep(0.05)
        rd.join(5)
        out0 = buf[0] if buf else ''
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    lines = [ln for ln in out0.splitlines() if ln.startswith('{')]
    if rc == 0 and not lines:
        print('bench launcher: rank 0 printed no JSON line', file=sys.stderr, flush=True)
        rc = 1
    sys.stdout.write(out0)
    sys.stdout.flush()
    return rc


def dist_setup(a):
    """Rank environment -> (rank, world, local, dist | None, info).  `info` goes into the JSON line: backend, ranks_seen (an
    all-reduce of ones: what the collective library really connected), the device index of every rank."""
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if a.gpus != world:
        raise SystemExit(f'--gpus {a.gpus} but WORLD_SIZE={world}')
    dry = os.environ.get('NBM_BENCH_DRY') == '1'       # tests/test_bench_launcher.py: rendezvous + collectives only, no GPU work
    # NBM_BENCH_SHARE_GPU=1 (functional rehearsal of the multi-rank path on a one-GPU box): every rank uses cuda:0 and
    # the collectives go through gloo, because RCCL refuses two ranks on one device.  Never set for measurements.
    share = os.environ.get('NBM_BENCH_SHARE_GPU') == '1'
    dev = 0 if share else local
    if not dry:
        torch.cuda.set_device(dev)
    dist, backend, seen, devices = None, None, 1, [dev]
    host_threads = None
    if world > 1:
        # under torchrun the launcher above did not run: cap this rank's host thread pools here (same rule)
        host_threads = int(os.environ.get('NBM_HOST_THREADS') or host_threads_per_rank(int(os.environ.get('LOCAL_WORLD_SIZE', world))))
        torch.set_num_threads(host_threads)
        import datetime
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if share or dry:
            backend = 'gloo'
            dist.init_process_group('gloo', timeout=datetime.timedelta(minutes=5))
        else:
            backend = 'nccl'                    # = RCCL on ROCm
            # a collective that cannot complete raises after 5 minutes instead of hanging the bench
            dist.init_process_group('nccl', device_id=torch.device('cuda', local), timeout=datetime.timedelta(minutes=5))
        where = 'cpu' if dry else 'cuda'
        one = torch.ones((1,), device=where, dtype=torch.float32)
        dist.all_reduce(one)
        seen = int(one.item())
        idx = torch.zeros((world,), device=where, dtype=torch.int32)
        idx[rank] = dev + 1
        dist.all_reduce(idx)
        devices = [int(v) - 1 for v in idx.tolist()]
        if seen != world:
            raise SystemExit(f'the {backend} all-reduce saw {seen} ranks, expected {world}')
        # the gloo group for host-side control data of the training exchange (touched bitmap), created collectively NOW -- not
        # inside the first training step (train.init_control_group)
        from birdsoundclassif_amd.train import init_control_group
        control = init_control_group(dist)
    info = {'backend': backend, 'ranks_seen': seen, 'rank_devices': devices,
            'host_threads_per_rank': host_threads,
            'control_group': None if world == 1 else ('gloo beside the RCCL group' if control is not None else 'the default (gloo) group'),
            'launcher': os.environ.get('NBM_BENCH_LAUNCHER', 'torchrun' if 'TORCHELASTIC_RUN_ID' in os.environ else 'external' if world > 1 else 'none')}
    return rank, world, local, dist, info


def main(argv=None):
    a = parse_args(argv)
    if 'WORLD_SIZE' not in os.environ and a.gpus > 1:
        # launcher mode: no GPU call in this process (device_count() does not initialise the runtime)
        dry = os.environ.get('NBM_BENCH_DRY') == '1'
        share = os.environ.get('NBM_BENCH_SHARE_GPU') == '1'
        have = torch.cuda.device_count()
        if not (dry or share) and have < a.gpus:
            print(f'bench: --gpus {a.gpus} but {have} GPU(s) visible; refusing to run a smaller job under that label',
                  file=sys.stderr, flush=True)
            raise SystemExit(2)
        raise SystemExit(launch_ranks(a.gpus, sys.argv[1:] if argv is None else argv))

    rank, world, local, dist, dist_info = dist_setup(a)
    if os.environ.get('NBM_BENCH_DRY') == '1':
        fail = os.environ.get('NBM_BENCH_DRY_FAIL_RANK')
        if fail is not None and int(fail) == rank:
            raise SystemExit(3)
        if dist is not None:
            dist.barrier()
        if rank == 0:
            print(json.dumps({'metric': 'dry run (launcher test)', 'n_gpus': world, **dist_info,
                              'torch_threads': torch.get_num_threads(), 'OMP_NUM_THREADS': os.environ.get('OMP_NUM_THREADS')}), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    from birdsoundclassif_amd import ondemand, ops, synth
    from birdsoundclassif_amd.nbm_datasets.prepare_dataset import SpectrogramFrontEnd
    from birdsoundclassif_amd.nets import build_model
    from birdsoundclassif_amd.train import default_args

    # the headline legs run the library's DEFAULT deep-K kernel (fp32 matrix instruction) whatever the caller's environment says; the opt-in
    # form (bf16 matrix pipe on split fp32 operands, csrc/igemm_split.hip) gets a leg of its own below (`split_bf16`)
    split_env_was = os.environ.get('NBM_SPLIT_BF16')
    os.environ['NBM_SPLIT_BF16'] = '0'
    B = a.batch
    model, _ = build_model(default_args(device='cuda'))
    model.load_state_dict(synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}))
    model = model.cuda().eval()
    fe = SpectrogramFrontEnd('cuda')
    # B DISTINCT clips per rank: the batch-coupled minima of the proposal stage, the device RoI tile lists and the NMS launches see a
    # mixed batch (tests/test_gpu_fullsize.py checks the same 64 clips against the oracle)
    pcm_host = synth.clip_batch_pcm16(rank * B, B)
    if a.tile_clips:                                          # A/B switch: the first `tile_clips` clips repeated (rounds 1-4 timed 8 tiled clips)
        pcm_host = np.tile(pcm_host[:a.tile_clips], (-(-B // a.tile_clips), 1))[:B].copy()
    distinct_clips = len({bytes(r) for r in pcm_host})
    pcm = torch.from_numpy(pcm_host).cuda()

    fast_rcnn = model.head.fast_rcnn

    copy_stream = torch.cuda.Stream()

    def launch():
        """GPU side of one step (asynchronous): front end + detector + device post-processing, then the D2H copy of
        the compact detection rows on a second stream so that it does not queue behind the next step's kernels."""
        imgs, _ = fe(pcm, 22050)                               # [B,1,375,1024]
        det, n_det = model.detect(imgs, min_score=a.min_score)  # [B,50,6], [B] on the device
        ready = torch.cuda.Event()
        ready.record()
        det_h = torch.empty(det.shape, dtype=det.dtype, pin_memory=True)
        n_h = torch.empty(n_det.shape, dtype=n_det.dtype, pin_memory=True)
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(ready)
            det_h.copy_(det, non_blocking=True)
            n_h.copy_(n_det, non_blocking=True)
            done = torch.cuda.Event()
            done.record(copy_stream)
        return det, n_det, det_h, n_h, done

    def finish(pending):
        """Host side of one step: wait for the copy, build the reference's per-clip output dictionaries."""
        pending[4].synchronize()
        return fast_rcnn.dets_to_dicts(pending[2], pending[3], model.args.num_classes)

    def step():
        return finish(launch())

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.spin_up):                                 # idle -> steady clocks (see --spin-up); not part of W, not timed
        step()
    for _ in range(a.warmup):
        step()
    import gc
    gc.collect()
    gc.disable()                                               # timed regions run without the cyclic collector (as timeit does)
    sync_all()
    ops.PROFILE = []                                           # live HIP-event timing of the dominant kernel's launches only:
    ops.PROFILE_ONLY = 'deepk'                                 # events around all ~250 launches cost the step 2.5 %
    t0 = time.perf_counter()
    n_det = 0
    pending = None
    for _ in range(a.steps):                                   # software pipeline: the GPU work of step i+1 is queued
        cur = launch()                                         # before the host finishes step i; all K steps complete
        if pending is not None:                                # inside the timed region
            out = finish(pending)
            n_det += sum(len(v['bbox_coord']) for d in out for v in d.values())
        pending = cur
    out = finish(pending)
    n_det += sum(len(v['bbox_coord']) for d in out for v in d.values())
    sync_all()
    dt = time.perf_counter() - t0
    gc.enable()
    prof, ops.PROFILE = ops.PROFILE, None
    ops.PROFILE_ONLY = None
    # ---- the headline: the SAME step (front end + detector + device post-processing, same kernels, same batch semantics) captured
    # once in a hipGraph and replayed -- no Python between the ~300 launches of a step.  The eager loop above (HIP events around
    # the dominant kernel's launches) stays in the line as `eager_with_events`: it carries the roofline measurement, and it is
    # bound by the HOST on boxes with slower cores (75.3 ms eager against 69.4 ms per replayed batch in one and the same run)
    eager = {'ms_per_step': dt / a.steps * 1e3, 'clips_per_s_per_gpu': B * a.steps / dt, 'detections_per_step': n_det / a.steps}
    graph_note = None
    n_lanes = max(1, a.lanes)

    def capture(lanes):
        """-> GraphedDetector with `lanes` parallel branches (or None + note).  Every rank ends up with a detector or none does."""
        nonlocal graph_note
        gd = None
        try:
            from birdsoundclassif_amd.bulk import GraphedDetector
            gd = GraphedDetector(model, B, pcm.shape[1], 22050, min_score=a.min_score, independent=False, lanes=lanes)
            for p_ in gd.pcms:
                p_.copy_(pcm)
        except Exception as exc:
            graph_note = f'hipGraph capture ({lanes} lane(s)) failed ({type(exc).__name__}: {exc})'[:300]
            gd = None
        if dist is not None:                                   # every rank replays, or none does (the loop below holds barriers)
            flag = torch.tensor([0.0 if gd is None else 1.0], device='cuda')
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if float(flag.item()) == 0.0 and gd is not None:
                gd, graph_note = None, 'hipGraph capture failed on another rank'
        return gd

    def replay_leg(gd, expect=None):
        """Timed replays of the captured step: `gd.lanes` batches per replay, in flight on the GPU together; EXACTLY K steps = K // lanes
        replays + (K mod lanes) eager steps behind them (in a lane of their own: the graph's branches own lanes 0 .. lanes-1).
        -> seconds, or None.  Every rank passes BOTH barriers whatever happens to it in between (a rank-local exception used to leave
        that rank two barriers short and its peers waiting for the collective timeout)."""
        nonlocal graph_note
        L = gd.lanes
        n_rep, n_tail = a.steps // L, a.steps % L
        last_copy = [None]

        def launch_g():
            with torch.cuda.stream(gd.stream):
                if last_copy[0] is not None:
                    gd.stream.wait_event(last_copy[0])          # the static outputs are overwritten: the previous D2H must be done
                gd.replay()
                ready = torch.cuda.Event()
                ready.record(gd.stream)
            hs = [(torch.empty(gd.dets[j].shape, dtype=gd.dets[j].dtype, pin_memory=True),
                   torch.empty(gd.n_dets[j].shape, dtype=gd.n_dets[j].dtype, pin_memory=True)) for j in range(L)]
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(ready)
                for j in range(L):
                    hs[j][0].copy_(gd.dets[j], non_blocking=True)
                    hs[j][1].copy_(gd.n_dets[j], non_blocking=True)
                done = torch.cuda.Event()
                done.record(copy_stream)
            last_copy[0] = done
            return hs, done

        def finish_g(pending):
            pending[1].synchronize()
            return [fast_rcnn.dets_to_dicts(h[0], h[1], model.args.num_classes) for h in pending[0]]

        ok, dt_leg = True, None
        try:
            for _ in range(max(1, -(-a.warmup // L))):
                finish_g(launch_g())
        except Exception as exc:
            ok, graph_note = False, f'hipGraph replay failed in the warm-up ({type(exc).__name__}: {exc}); value is the eager loop'[:300]
        gc.collect()
        gc.disable()
        try:
            sync_all()
            t0 = time.perf_counter()
            if ok:
                try:
                    n_det_g, pend = 0, None
                    for _ in range(n_rep):
                        cur = launch_g()
                        if pend is not None:
                            n_det_g += sum(len(v['bbox_coord']) for out in finish_g(pend) for d in out for v in d.values())
                        pend = cur
                    if pend is not None:
                        n_det_g += sum(len(v['bbox_coord']) for out in finish_g(pend) for d in out for v in d.values())
                    for _ in range(n_tail):                    # K is not a multiple of the lanes: the remaining steps, eagerly
                        torch.cuda.current_stream().wait_stream(gd.stream)
                        with ops.lane(max(gd.lane_ids) + 1):
                            out = step()
                        n_det_g += sum(len(v['bbox_coord']) for d in out for v in d.values())
                    want = n_det if expect is None else expect
                    if n_det_g != want:
                        raise RuntimeError(f'the replayed steps returned {n_det_g} detections, the eager ones {want}')
                except Exception as exc:
                    ok, graph_note = False, f'hipGraph replay failed ({type(exc).__name__}: {exc}); value is the eager loop'[:300]
            sync_all()
            if ok:
                dt_leg = time.perf_counter() - t0
        finally:
            gc.enable()
        # graph and eager times are never mixed in the MAX-reduce: every rank reports its replayed time, or every rank its eager one
        if dist is not None and not graph_replay_agreed(dist, dt_leg is not None):
            if dt_leg is not None:
                graph_note = 'hipGraph replay failed on another rank; value is the eager loop'
            dt_leg = None
        return dt_leg

    # the one-lane graph is measured and released before the multi-lane one exists (memory: every live detector owns its lanes' scratch)
    dt_g = dt_g1 = dt_g2 = None
    lanes_used = 0
    gd = capture(1)
    if gd is not None:
        dt_g1 = replay_leg(gd)
        dt_g, lanes_used = dt_g1, (1 if dt_g1 is not None else 0)
    gd = None
    gc.collect()
    torch.cuda.empty_cache()
    if n_lanes > 1 and dt_g1 is not None and a.steps >= n_lanes:
        note1 = graph_note
        gd = capture(n_lanes)
        if gd is not None:
            dt_g2 = replay_leg(gd)                              # `n_lanes` batches in flight together: the headline when it is the faster loop
            if dt_g2 is not None and dt_g2 < dt_g1:
                dt_g, lanes_used = dt_g2, n_lanes
        if dt_g2 is None and dt_g1 is not None:                 # the one-lane figure stands; keep the reason the multi-lane leg gave up
            graph_note = None if graph_note is None else 'multi-lane leg: ' + graph_note
        gd = None
        gc.collect()
        torch.cuda.empty_cache()
    single_lane = None if dt_g1 is None else {'ms_per_step': dt_g1 / a.steps * 1e3, 'clips_per_s_per_gpu': B * a.steps / dt_g1}
    multi_lane = None if dt_g2 is None else {'lanes': n_lanes, 'ms_per_step': dt_g2 / a.steps * 1e3, 'clips_per_s_per_gpu': B * a.steps / dt_g2}
    if dt_g is not None:
        dt = dt_g
    def gemm_gflop(tag):
        Cin, N, k, H, W, Bn, groups, stride = tag[:8]
        if len(tag) == 9 and isinstance(tag[8], tuple) and tag[8][0] == 'rpn-composite':      # k taps x 1, one output row: W cells
            return 2.0 * W * N * Cin * k * groups / 1e9
        return 2.0 * Bn * ((H - 1) // stride + 1) * ((W - 1) // stride + 1) * N * Cin * k * k * groups / 1e9

    def gemm_bytes(tag):
        """Algorithmic HBM bytes of one launch of the forward implicit GEMM (input + weights + output, each once)."""
        Cin, N, k, H, W, Bn, groups, stride = tag[:8]
        if len(tag) == 9 and isinstance(tag[8], tuple) and tag[8][0] == 'rpn-composite':      # k taps x 1 over W cells: [k][W][Cin] in, [W][N] out
            return 4.0 * groups * (k * W * Cin + W * N + N * k * Cin)
        return conv_algorithmic_bytes(Bn, H, W, Cin, N, k, stride, groups, 'fwd')

    def what(tag):
        Cin, N, k, H, W, Bn, groups, stride, label = tag
        g = f' x{groups} groups' if groups > 1 else ''
        lab = f' [{label[0]} {label[1]}x{label[2]}]' if isinstance(label, tuple) else ''
        if isinstance(label, tuple) and label[0] == 'rpn-composite':
            return f'{k} taps x {Cin}->{N} over {W} cells{lab}'
        return f'{k}x{k} s{stride} {Cin}->{N} @{H}x{W} B={Bn}{g}{lab}'
    # ---- opt-in mode, measured beside the headline (never IN it): the same step with every deep-K implicit GEMM on the bf16 matrix pipe
    # through split fp32 operands (x = hi + mid + lo, six products, two fp32 accumulators: csrc/igemm_split.hip; error against float64
    # <= the fp32 instruction's, asserted in tests/test_gpu_split.py).  Single-rank runs only, like the dense leg.
    split_leg = None
    if world == 1 and not a.no_split_leg and dt_g is not None:
        os.environ['NBM_SPLIT_BF16'] = '1'
        try:
            step()
            out_s = step()                                       # the eager step in THIS mode: what the replays of the leg must reproduce
            n_det_split = sum(len(v['bbox_coord']) for d in out_s for v in d.values()) * a.steps
            torch.cuda.synchronize()
            ops.PROFILE, ops.PROFILE_ONLY = [], 'deepk'
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            prof_s, ops.PROFILE, ops.PROFILE_ONLY = ops.PROFILE, None, None
            note0 = graph_note
            gd = capture(max(1, lanes_used))
            dt_s = replay_leg(gd, expect=n_det_split) if gd is not None else None
            gd = None
            gc.collect()
            torch.cuda.empty_cache()
            per_s = [(tag, s_.elapsed_time(e_)) for (tag, s_, e_) in prof_s if len(tag) == 9 and ops.is_deepk(tag[0], tag[1], tag[2], tag[2])]
            gf = sum(gemm_gflop(t) for t, _ in per_s)
            ms_s = sum(m for _, m in per_s)
            split_leg = {'default': False, 'switch': 'NBM_SPLIT_BF16=1 (read per call by nbm_gemm_conv)',
                         'value': None if dt_s is None else B * a.steps / dt_s, 'unit': 'clips/s',
                         'ms_per_step': None if dt_s is None else dt_s / a.steps * 1e3, 'lanes': max(1, lanes_used),
                         'launch_note': graph_note if dt_s is None else None,
                         'kernel': 'igemm_split_kernel (csrc/igemm_split.hip): the launches the default mode gives to the dominant kernel, here with '
                                   'fp32 operands split into three bf16 terms on their way into LDS and six v_mfma_f32_32x32x16_bf16 products per '
                                   'K-slice (fp32 accumulate, hi*hi in an accumulator of its own)',
                         'roofline': None if not per_s else {
                             'bound': 'mfma', 'achieved': 6.0 * gf / ms_s, 'peak': BF16_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s (bf16 MFMA, executed: six products per fp32 product)',
                             'frac': 6.0 * gf / ms_s / BF16_MFMA_PEAK_TFLOPS, 'fp32_equivalent_TFLOPs': gf / ms_s,
                             'fp32_equivalent_over_fp32_mfma_peak': gf / ms_s / FP32_MFMA_PEAK_TFLOPS,
                             'ms_per_step': ms_s / 3, 'launches_per_step': len(per_s) / 3, 'avg_launch_ms': ms_s / len(per_s),
                             'note': 'fp32_equivalent_over_fp32_mfma_peak may exceed 1: these FLOPs do not run on the fp32 pipe -- it is NOT a roofline '
                                     'fraction, `frac` (against the bf16 peak, six executed products per fp32 product) is'},
                         'parity': 'tests/test_gpu_split.py (error against float64 <= the fp32 kernel\'s), tests/test_gpu_e2e.py [splitbf16]: golden RoIs / '
                                   'detections of the reference bit for bit (one half-pixel tie of the bifpn variant, margin asserted)',
                         'why_not_default': 'the contract metric is quoted on the reference\'s fp32 arithmetic; this form is fp32-accurate but sums in '
                                            'another order on another pipe -- kept opt-in until a judge round has looked at it (DESIGN 4e)'}
            graph_note = note0
        except Exception as exc:                  # informational leg: never lose the headline line over it
            split_leg = {'error': f'{type(exc).__name__}: {exc}'[:300]}
        finally:
            os.environ['NBM_SPLIT_BF16'] = '0'
            ops.PROFILE, ops.PROFILE_ONLY = None, None
    ops.PROFILE = []                                           # one extra, untimed step with events around every GEMM-type launch
    ops.FLOPS = [0.0]                                          # ... and the executed-MFMA-FLOP counter of every GEMM launch
    step()
    torch.cuda.synchronize()
    exec_gflop_per_clip = ops.flops_total() / B / 1e9 + 2.0 * 2 * 384 * 664 * 1024 / 1e9      # + the fp64 DFT GEMMs of the front end
    prof_all, ops.PROFILE, ops.FLOPS = ops.PROFILE, None, None
    # for the record: the same step with the finest FPN level computed densely, as the reference does -- untimed
    # for the headline, 2 warm-up + 10 steps; single-rank runs only (a failure of this optional leg on one rank must not leave
    # the other ranks in a barrier)
    dense_ref = None
    if ondemand.LAZY_FINEST and not a.no_dense_reference and world == 1:
        ondemand.LAZY_FINEST = False
        try:
            step()
            step()
            torch.cuda.synchronize()
            td = time.perf_counter()
            pend = None
            for _ in range(10):
                cur = launch()
                if pend is not None:
                    finish(pend)
                pend = cur
            finish(pend)
            torch.cuda.synchronize()
            td = (time.perf_counter() - td) / 10
            dense_ref = {'ms_per_step': td * 1e3, 'clips_per_s_per_gpu': B / td, 'steps': 10,
                         'note': 'NBM_LAZY_FINEST=0: every pixel of the finest FPN map and of its lateral is computed; identical classes and boxes, scores within 1e-5 (tests/test_gpu_lazy.py)'}
        except Exception as exc:                  # informational leg: never lose the headline line over it
            dense_ref = {'error': f'{type(exc).__name__}: {exc}'[:300]}
        finally:
            ondemand.LAZY_FINEST = True
    if dist is not None:
        t = torch.tensor([dt], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # roofline of the dominant kernel = igemm_h16_kernel<3> (csrc/igemm_h16.hip; round 5: the half-step form of
    # igemm_kernel<128,128,64,64,A_FAST,EPI_STD,2>, csrc/igemm.hip), the deep-K kernel of the implicit GEMM: every launch of a step that nbm_gemm_conv dispatches to it -- the ResNet 1x1 / strided 3x3 layers
    # with K > 256, the attention GEMMs, the FPN laterals of levels 1-4 and the launches of the composed RPN reader (FPN levels 0 and
    # 1 on demand).  `achieved` = MFMA FLOPs these launches EXECUTE (2 * M * N * K * groups) over their HIP-event
    # time on the launch stream inside the timed loop; avg_launch_ms is what rocprofv3 --stats reports as the kernel's average.
    deep = [(tag, s.elapsed_time(e)) for (tag, s, e) in prof if len(tag) == 9 and ops.is_deepk(tag[0], tag[1], tag[2], tag[2])]
    all_ms = sum(s.elapsed_time(e) for (tag, s, e) in prof_all if len(tag) == 9)
    fused_ms = sum(s.elapsed_time(e) for (tag, s, e) in prof_all
                   if len(tag) == 9 and isinstance(tag[8], tuple) and tag[8][0] in ('wino23', 'wino23-rois'))
    roof = None
    traffic = None                      # HBM bytes per launch of the dominant kernel: PMC counters cannot be read live;
    traffic_why = None
    try:                                # the value comes from the committed rocprofv3 --pmc passes of this same command
        pj = json.load(open(os.path.join(ROOT, 'profiles', PMC_DOMINANT)))
        if B == 64 and pj.get('lazy_finest') == bool(ondemand.LAZY_FINEST) and pj.get('kernel_family') == 'igemm deep-K':
            traffic = pj['traffic_bytes_per_launch']
        else:
            traffic_why = 'batch / on-demand mode differ from that profile'
    except Exception as exc:
        traffic, traffic_why = None, f'profiles/{PMC_DOMINANT} not readable ({type(exc).__name__})'
    if deep:
        per = [(tag, ms, gemm_gflop(tag)) for tag, ms in deep]
        gflop, ms = sum(g for _, _, g in per), sum(m for _, m, _ in per)
        ach = gflop / ms                                       # GFLOP/ms == TFLOP/s
        big = {}
        for tag, m, g in per:                                  # per distinct launch (layer): time and rate
            slot = big.setdefault(what(tag), [0.0, 0.0, 0])
            slot[0] += m
            slot[1] += g
            slot[2] += 1
        top = sorted(big.items(), key=lambda kv: -kv[1][0])[:4]
        roof = {'bound': 'mfma', 'kernel': 'igemm_h16_kernel<3> (csrc/igemm_h16.hip): fp32-MFMA implicit GEMM, the deep-K kernel (128x128 tiles, '
                                           'double-buffered half-step LDS stages, three workgroups per CU, fused epilogue; the same products in the same '
                                           'order as igemm_kernel<128,128,64,64,A_FAST,EPI_STD,2>, which NBM_H16=0 selects); all its launches of a '
                                           'step: ResNet 1x1 / strided 3x3 layers with K > 256, attention, FPN laterals, the 5x5 / stride-S '
                                           'launches of the RPN reader composed with the output convolution of the on-demand FPN levels (DESIGN 4f)',
                'achieved': ach, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / FP32_MFMA_PEAK_TFLOPS,
                'traffic': traffic, 'traffic_unit': 'bytes/launch (2*FETCH_SIZE + WRITE_SIZE, mean over the launches)',
                'algorithmic_bytes_per_launch': sum(gemm_bytes(tag) for tag, _, _ in per) / len(per),
                'wasted_traffic_ratio': None if traffic is None else traffic / (sum(gemm_bytes(tag) for tag, _, _ in per) / len(per)),
                'traffic_source': f'profiles/{PMC_DOMINANT}: separate rocprofv3 --pmc passes of this command, NOT measured in this run '
                                  '(counters cannot be read live); null when batch / on-demand mode differ from that profile'
                                  + (f' -- null here: {traffic_why}' if traffic_why else ''),
                'avg_launch_ms': ms / len(per), 'launches': len(per), 'launches_per_step': len(per) / a.steps,
                'executed_GFLOP_per_launch': gflop / len(per), 'ms_per_step': ms / a.steps,
                'largest_launches': [{'what': k, 'ms': v[0] / v[2], 'launches_per_step': v[2] / a.steps, 'executed_TFLOPs': v[1] / v[0],
                                      'frac': v[1] / v[0] / FP32_MFMA_PEAK_TFLOPS} for k, v in top],
                'all_gemm_type_launches_ms_per_step': all_ms,
                'fused_winograd_kernel_ms_per_step': fused_ms,
                'whole_step_executed_GFLOP_per_clip': exec_gflop_per_clip,
                'whole_step_executed_TFLOPs': exec_gflop_per_clip * B * a.steps / (dt * 1e3),
                'whole_step_executed_frac_of_mfma_peak': exec_gflop_per_clip * B * a.steps / (dt * 1e3) / FP32_MFMA_PEAK_TFLOPS,
                'whole_step_direct_conv_equivalent_TFLOPs': FWD_GFLOP_PER_CLIP * B * a.steps / (dt * 1e3),
                'whole_step_note': 'executed = MFMA FLOPs the launches of a step really perform (Winograd- / cell-domain counts, listed '
                                   'tiles only, fp64 DFT GEMMs counted at face value) over the wall time of the whole step incl. every '
                                   'non-GEMM kernel; direct_conv_equivalent uses SURVEY 325.56 GFLOP/clip and is NOT a roofline figure',
                'fpn_levels_0_1': 'on demand (pattern pixels of the RPN readers through the cell transforms + tiles under the RoIs; the '
                                  'other pixels have no reader)' if ondemand.LAZY_FINEST else 'dense'}
    # front end alone (HBM-bound stage of the path): live HIP events around K replays
    fe_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for s0, e0 in fe_ev:
        s0.record()
        fe(pcm, 22050)
        e0.record()
    torch.cuda.synchronize()
    fe_ms = sorted(s0.elapsed_time(e0) for s0, e0 in fe_ev)[len(fe_ev) // 2]
    fe_flop = 2.0 * 2 * 384 * 664 * 1024            # per clip: Re and Im GEMMs, 384 x 664 x 1024 (bins x k x frames, padded)
    frontend = {'ms_per_batch': fe_ms, 'clips_per_s': B / fe_ms * 1e3,
                'hbm_algorithmic_GBps': B * 1.668e6 / (fe_ms * 1e-3) / 1e9, 'hbm_peak_GBps': 8000.0,
                'dft_gemm_executed_TFLOPs_f64': B * fe_flop / (fe_ms * 1e-3) / 1e12, 'mfma_f64_peak_TFLOPs': FP64_MFMA_PEAK_TFLOPS,
                'note': 'PCM16 -> 2x up-sample -> STFT-dB (folded real DFT on the fp64 MFMA, csrc/stft.hip) -> normalise/window; '
                        'algorithmic bytes 1.668 MB/clip (SURVEY 8d); the stage is bound by the DFT-GEMMs, not by HBM'}
    bulk_leg = None
    if a.bulk_files > 0:
        try:
            bulk_leg = bulk_bench(model, rank, world, dist, a.bulk_files, B, a.min_score, B * a.steps / dt, lanes=a.lanes)
        except Exception as exc:                  # never lose the headline line over an extra leg
            bulk_leg = {'error': f'{type(exc).__name__}: {exc}'[:500]}
        ok, wall = (0.0, 0.0) if 'error' in bulk_leg else (1.0, bulk_leg['wall_s'])
        if dist is not None:                      # every rank gets here, failed or not: max time, min ok over the ranks
            tt = torch.tensor([wall, -ok], device='cuda', dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            wall, ok = float(tt[0].item()), -float(tt[1].item())
        if ok > 0 and 'error' not in bulk_leg:
            v = world * a.bulk_files / wall
            bulk_leg.update(value=v, wall_s=wall, ratio_to_resident_hbm_headline=v / world / (B * a.steps / dt))
        elif 'error' not in bulk_leg:
            bulk_leg = {'error': 'the bulk_inference leg failed on another rank'}
    train = None
    if not a.no_train:
        del model
        import gc
        gc.collect()                              # the captured graphs of the detect / bulk legs own private pools: let them go
        ops.release_lane_scratch(keep=(0,))       # ... and the second lane's persistent scratch (no graph is alive any more)
        torch.cuda.empty_cache()
        if os.environ.get('NBM_BENCH_MEMLOG') == '1':
            print(f'bench: before the train leg: allocated {torch.cuda.memory_allocated() / 2 ** 30:.1f} GiB, reserved '
                  f'{torch.cuda.memory_reserved() / 2 ** 30:.1f} GiB', file=sys.stderr, flush=True)
        try:
            train = train_bench(rank, world, dist, a.train_batch, a.train_steps, 2)     # 2 warm-up steps: allocator, lists, pinned ring
        except Exception as exc:                  # the detect line above is the contract metric: never lose it
            train = {'error': f'{type(exc).__name__}: {exc}'[:500]}
    if rank == 0:
        line = {'metric': 'clips/sec (3 s @ 22.05 kHz) detect fwd', 'value': world * B * a.steps / dt, 'unit': 'clips/s',
                'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'spin_up_steps': a.spin_up, 'gc_disabled_in_timed_regions': True, 'ms_per_step': dt / a.steps * 1e3,
                'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                'config': {'workload': 'BASELINE.json configs[1]: 1xMI355X inference, batch=64 synthetic 3 s clips '
                                       '(PCM16 @22.05 kHz resident in HBM) through the HIP STFT front end + detector '
                                       'forward + device post-processing, detections returned to the host',
                           'batch_per_gpu': B, 'distinct_clips': distinct_clips, 'min_score': a.min_score, 'detections_per_step': n_det / a.steps,
                           'launch': (('hipGraph replay of ONE graph whose capture forks into %d parallel detect steps (one stream each): '
                                       '%d batches in flight together per replay' % (lanes_used, lanes_used) if lanes_used > 1 else
                                       'hipGraph replay of the captured step') if dt_g is not None else (graph_note or 'eager loop')),
                           'lanes': lanes_used, 'launch_note': graph_note},
                'eager_with_events': eager, 'single_lane_graph_replay': single_lane, 'multi_lane_graph_replay': multi_lane,
                'roofline': roof, 'split_bf16': split_leg, 'frontend': frontend, 'dense_finest_map': dense_ref, 'bulk_inference': bulk_leg,
                'train_step': train, 'deep_k_gemm': 'fp32 matrix instruction (library default; NBM_SPLIT_BF16 pinned to 0 for every leg but `split_bf16`'
                               + (f'; the caller had set it to {split_env_was!r}' if split_env_was not in (None, '0') else '') + ')', **dist_info}
        if world == 1 and not a.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
