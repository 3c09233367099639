"""Bulk inference over many equal-length clips (BASELINE.json configs[4]): wav shard per GPU, the whole detect step
(front end + detector + device post-processing) captured once in a hipGraph and replayed per batch.

The graph is legal because `NbmModel.detect` never syncs with the host: every data-dependent size (kept anchors, NMS
survivors, RoI count, detections per clip) lives in fixed-capacity device buffers with device-side counters.

`detect_files` is a three-stage software pipeline around the graph, so that the GPU never waits for a file or a dict:

    reader thread   wav files -> rows of a pinned int16 batch buffer (a ring of `depth` slots)
    main thread     H2D copy of the slot -> graph replay -> D2H copy of the compact [B,50,6] detection rows + counts into the
                    slot's pinned result buffers, all on the graph's stream; batch i+1 is queued before the host waits for
                    batch i
    writer thread   rows -> the reference's per-file output dictionary (single-window `merge_images`) -> `<wav>.txt`

The clips of a batch are INDEPENDENT (`NbmModel.detect(..., independent=True)`): the reference CLI runs one file per model call
(nbm_detect.py:24-28 -> run_detection.py:49-55, a 3 s clip is a batch of one window), so the batch-coupled proposal counts of
the reference's ProposalLayer / nms (min over the batch, layers.py:287, nets_utils.py:236) must not couple files that merely
share a launch here: every clip keeps its own counts, exactly as if it had been run alone.

Multi-GPU: one process per GPU, files sharded `files[rank::world]`, no data-path collective (`nbm_detect.py` does the sharding).
"""
import os
import queue
import struct
import threading
import time

import numpy as np
import torch

from .nbm_datasets.prepare_dataset import SpectrogramFrontEnd, read_wav_pcm16


class GraphedDetector:
    """Captures `front end -> model.detect` for a fixed (batch, n_samples, sample rate) and replays it.

    `lanes` > 1: ONE graph whose capture forks into that many parallel branches (one stream each, joined before the capture ends),
    every branch a complete detect step on its own static input / outputs and its own persistent scratch (`ops.lane`): a replay
    processes `lanes` batches that are in flight on the GPU TOGETHER, so the tail of one step's kernels (a launch's last, partly
    filled round of workgroups; the latency-bound proposal / NMS kernels) is filled by the other's: 65.2 instead of 68.9 ms per
    B = 64 batch (profiles/r04_two_lanes.txt).

    Fence (profiles/r05_graph_pair.txt): the HIP runtime that torch 2.10+rocm7.0 bundles (7.0.51831) loses MEMSET nodes of a
    replayed graph exec -- round 4's "second graph exec computes garbage / faults" was the proposal stage's counters, zeroed by
    hipMemsetAsync = memset nodes, keeping the previous replay's values.  The library zeroes with kernels now, and the capture is
    refused unless its graph consists of kernel (and empty fork / join) nodes only (`ops.graph_census`), whoever issued the others.
    The persistent scratch / tile-list buffers of the lanes are held as captured (`self._held`), so no later growth or release can
    recycle memory this graph writes to.  Several detectors may be alive at a time (`tests/test_gpu_detect_cli.py`)."""

    def __init__(self, model, batch, n_samples, sr, min_score=0.2, nms_thresh=0.3, device='cuda', independent=False, lanes=1):
        from . import ops
        self.model, self.batch, self.sr, self.lanes = model.eval(), batch, sr, max(1, int(lanes))
        self.fes = [SpectrogramFrontEnd(device) for _ in range(self.lanes)]
        self.fe = self.fes[0]
        self.min_score, self.nms_thresh, self.independent = min_score, nms_thresh, independent
        self.n_img = self.fe.n_images(self.fe.n_frames(n_samples * (2 if sr * 2 == self.fe.FREQ else 1)))
        if self.n_img != 1:
            raise NotImplementedError('GraphedDetector handles clips that fit one 1024-column window (<= 3.06 s)')
        self.pcms = [torch.zeros((batch, n_samples), dtype=torch.int16, device=device) for _ in range(self.lanes)]   # static graph inputs
        self.pcm = self.pcms[0]
        self.stream = torch.cuda.Stream()
        self.side = [torch.cuda.Stream() for _ in range(self.lanes - 1)]
        self.lane_ids = self._claim_lanes(self, self.lanes)
        try:
            self._capture(ops)
        except BaseException:
            self.close()                                          # a refused / failed capture leaves no lane claimed and no graph behind
            raise

    def _capture(self, ops):
        with torch.no_grad(), torch.cuda.stream(self.stream):
            for _ in range(2):                                   # warm-up: fills every weight / anchor / table cache, sizes the lanes' scratch
                self._run_all()
            self.stream.synchronize()
            for s_ in self.side:
                s_.synchronize()
            self.graph = torch.cuda.CUDAGraph(keep_graph=True)
            with torch.cuda.graph(self.graph, stream=self.stream):
                outs = self._run_all()                            # static graph outputs
        self.census = ops.graph_census(self.graph.raw_cuda_graph())
        bad = {k: v for k, v in self.census.items() if v and k not in ('kernel', 'empty')}
        if bad or not self.census['kernel']:
            raise RuntimeError(f'the captured detect step holds graph nodes other than kernels: {bad} (census {self.census}).  Memset '
                               'nodes are not replayed reliably by the HIP runtime torch bundles (DESIGN 4d): refusing to replay this graph')
        self.graph.instantiate()
        self._held = ops.lane_buffers(set(self.lane_ids))
        self.dets, self.n_dets = [o[0] for o in outs], [o[1] for o in outs]
        self.det, self.n_det = self.dets[0], self.n_dets[0]

    # lanes in use by live detectors: a second detector alive beside the first gets scratch of its own instead of sharing buffers
    # whose addresses both graphs would write through (two replays on two streams would race on them)
    _LANES_IN_USE = {}

    @classmethod
    def _claim_lanes(cls, owner, n):
        import weakref
        for k in [k for k, ref in cls._LANES_IN_USE.items() if ref() is None]:
            del cls._LANES_IN_USE[k]
        ids, k = [], 0
        while len(ids) < n:
            if k not in cls._LANES_IN_USE:
                ids.append(k)
                cls._LANES_IN_USE[k] = weakref.ref(owner)
            k += 1
        return ids

    def close(self):
        """Drops the graph, its static buffers and its hold on the lanes' scratch; lanes other than 0 are released to the allocator."""
        from . import ops
        self.graph = None
        self._held = []
        self.pcms = self.dets = self.n_dets = []
        self.pcm = self.det = self.n_det = None
        for k in getattr(self, 'lane_ids', []):
            ref = self._LANES_IN_USE.get(k)
            if ref is not None and ref() in (self, None):
                del self._LANES_IN_USE[k]
        self.lane_ids = []
        others = set(self._LANES_IN_USE)
        ops.release_lane_scratch(keep=tuple(others | {0}))

    def _run(self, k=0):
        imgs, _ = self.fes[k](self.pcms[k], self.sr)
        return self.model.detect(imgs[:, 0][:, None].contiguous(), self.nms_thresh, self.min_score, independent=self.independent)

    def _run_all(self):
        """Lane 0 on the current (main) stream, every other lane on its side stream between a fork and a join."""
        from . import ops
        main = torch.cuda.current_stream()
        outs = [None] * self.lanes
        for k, s_ in enumerate(self.side, start=1):
            s_.wait_stream(main)                                  # fork
            with torch.cuda.stream(s_), ops.lane(self.lane_ids[k]):
                outs[k] = self._run(k)
        with ops.lane(self.lane_ids[0]):
            outs[0] = self._run(0)
        for s_ in self.side:
            main.wait_stream(s_)                                  # join
        return outs

    def replay(self):
        """Runs the captured step(s) on the current content of `self.pcms`; results land in `self.dets`, `self.n_dets`."""
        self.graph.replay()

    def __call__(self, pcm):
        """pcm int16 [batch, n] (host or device) -> list[batch] of the reference's per-clip dictionaries (lane 0)."""
        from .nets.layers import FastRCNN
        self.pcm.copy_(pcm, non_blocking=True)
        self.replay()
        return FastRCNN.dets_to_dicts(self.det, self.n_det, self.model.args.num_classes)


def wav_header(path):
    """(format tag, channels, sample rate, bits, n_samples per channel, byte offset of the samples) of a RIFF/WAVE file, reading
    the chunk headers only."""
    with open(path, 'rb') as f:
        head = f.read(12)
        if len(head) < 12 or head[:4] != b'RIFF' or head[8:12] != b'WAVE':
            raise ValueError(f'{path}: not a RIFF/WAVE file')
        fmt = None
        while True:
            ch = f.read(8)
            if len(ch) < 8:
                raise ValueError(f'{path}: wav file without fmt / data chunk')
            cid, size = ch[:4], struct.unpack('<I', ch[4:])[0]
            if cid == b'fmt ':
                fmt = f.read(size + (size & 1))[:size]
            elif cid == b'data':
                if fmt is None:
                    raise ValueError(f'{path}: data chunk in front of the fmt chunk')
                tag, nch, sr, _, _, bits = struct.unpack('<HHIIHH', fmt[:16])
                if tag == 0xFFFE and len(fmt) >= 26:
                    tag = struct.unpack('<H', fmt[24:26])[0]
                off = f.tell()
                size = min(size, os.fstat(f.fileno()).st_size - off)
                return tag, nch, sr, bits, size // max(1, nch * (bits // 8)), off
            else:
                f.seek(size + (size & 1), 1)


def bulk_groups(files):
    """Splits a file list into {(sample rate, n_samples): [files]} of the clips the graphed path takes -- mono 16-bit PCM at
    22.05 or 44.1 kHz that fill exactly one spectrogram window -- and the rest (any other format, length or an unreadable
    header), which goes through the per-file driver."""
    groups, rest = {}, []
    for f in files:
        try:
            tag, nch, sr, bits, n, _ = wav_header(f)
        except (OSError, ValueError, struct.error):
            rest.append(f)
            continue
        ok = tag == 1 and nch == 1 and bits == 16 and sr in (22050, 44100) and n > 0
        # one window <=> 1 + n44 // HOP_LENGTH <= W_PIX frames (prepare_dataset.py:126,267 with the default dt / w_pix)
        ok = ok and 1 + (n * (2 if sr == 22050 else 1)) // int(44100 * 0.003) <= 1024
        if ok:
            groups.setdefault((sr, n), []).append(f)
        else:
            rest.append(f)
    return groups, rest


def rows_to_result(rows, n, w_pix, hop, spectrogram_length, names=None):
    """Detection rows of ONE single-window file -> the per-file output dictionary of run_detection (reference
    run_detection.py:69-84 after `merge_images`, :163-249).  rows: float32 [>= n, 6] = {class, x1, y1, x2, y2, score} sorted by
    (class, score desc).  For a file that is one window `merge_images` applies the first-window border rule (:195-196) and the
    end-of-file rule (:213); its file-level NMS (:233) cannot suppress anything, because the detector's own class-agnostic NMS ran
    at the same threshold on the same boxes (all surviving pairs have IoU < 0.3).
    -> {species | class id: {'bbox_coord': [[x1,y1,x2,y2]...], 'scores': [...]}}, classes ascending."""
    res = {}
    if n <= 0:
        return res
    r = rows[:n]
    keep = ~((r[:, 3] >= w_pix - 5) & ((r[:, 3] - r[:, 1]) < np.float32(0.9 * (w_pix - hop)))) & ~(r[:, 3] >= spectrogram_length)
    r = r[keep]
    if len(r) == 0:
        return res
    cls = r[:, 0].astype(np.int64)
    starts = np.flatnonzero(np.r_[True, cls[1:] != cls[:-1]])
    ends = np.r_[starts[1:], len(r)]
    for s0, e0 in zip(starts.tolist(), ends.tolist()):
        k = int(cls[s0])
        res[names[k] if names else str(k)] = {'bbox_coord': r[s0:e0, 1:5].tolist(), 'scores': r[s0:e0, 5].tolist()}
    return res


def single_window_merge(d, w_pix, hop, spectrogram_length, names=None):
    """`rows_to_result` for one entry of `FastRCNN.dets_to_dicts` (the reference's per-image dictionary)."""
    rows = [np.concatenate([np.full((len(v['bbox_coord']), 1), float(k), dtype=np.float32), v['bbox_coord'].numpy(),
                            v['scores'].reshape(-1, 1).numpy()], 1) for k, v in d.items() if len(v['bbox_coord'])]
    if not rows:
        return {}
    rows = np.concatenate(rows).astype(np.float32)
    return rows_to_result(rows, len(rows), w_pix, hop, spectrogram_length, names)


def txt_path(wav_path):
    return wav_path.replace('.wav', '.txt')          # like the reference CLI (nbm_detect.py:27)


def detect_files(model, files, batch=64, min_score=0.2, bird_dict=None, write_txt=True, depth=5, keep_results=True,
                 independent=True, stats=None, detector=None, lanes=None):
    """Detects over equal-length mono 16-bit PCM wav files (single-window clips, see `bulk_groups`): -> list of per-file output
    dicts in `files` order (None entries with keep_results=False); `<wav>.txt = str(dict)` written when `write_txt`.
    The last, partial batch is padded with silence and its padding results are dropped.  `stats` (dict) receives the stage
    times.  `detector`: a GraphedDetector to reuse (same batch / clip length / rate / lanes).
    `lanes` (default: the detector's, else NBM_BULK_LANES, else 2 when the shard has at least 4 batches): batches go through the graph
    in groups of that many, in flight on the GPU together (see GraphedDetector)."""
    if not files:
        return []
    _, _, sr, _, n, _ = wav_header(files[0])
    n_batches = -(-len(files) // batch)
    if lanes is None:
        lanes = detector.lanes if detector is not None else int(os.environ.get('NBM_BULK_LANES', '2' if n_batches >= 4 else '1'))
    lanes = max(1, int(lanes))
    det = detector or GraphedDetector(model, batch, n, sr, min_score=min_score, independent=independent, lanes=lanes)
    own_det = detector is None
    if (det.batch, det.pcm.shape[1], det.sr) != (batch, n, sr) or det.lanes != lanes:
        raise ValueError('the GraphedDetector handed in was captured for another batch / clip length / sample rate / number of lanes')
    fe = det.fe
    L = fe.n_frames(n * (2 if sr * 2 == fe.FREQ else 1))
    names = None
    if bird_dict is not None:
        names = {v: k for k, v in bird_dict.items()}
        names[0] = 'Non bird sound'
    depth = max(3, depth, 3 * lanes)
    cap = det.det.shape[1]
    slots = [(torch.zeros((batch, n), dtype=torch.int16).pin_memory(), torch.zeros((batch, cap, 6), dtype=torch.float32).pin_memory(),
              torch.zeros((batch,), dtype=torch.int32).pin_memory()) for _ in range(depth)]
    free_q, ready_q, done_q = queue.Queue(), queue.Queue(), queue.Queue()
    for s in range(depth):
        free_q.put(s)
    out = [None] * len(files)
    err = []
    t_read, t_write = [0.0], [0.0]

    def reader():
        try:
            for i in range(n_batches):
                s = free_q.get()
                if s is None:                        # the writer failed (or the main loop is shutting down): pass it on
                    ready_q.put(None)
                    return
                t0 = time.perf_counter()
                chunk = files[i * batch:(i + 1) * batch]
                host = slots[s][0].numpy()
                for j, f in enumerate(chunk):
                    p, sr_i = read_wav_pcm16(f)
                    if sr_i != sr or len(p) != n:
                        raise ValueError(f'{f}: bulk detection needs clips of identical length and rate')
                    host[j] = p
                if len(chunk) < batch:
                    host[len(chunk):] = 0
                t_read[0] += time.perf_counter() - t0
                ready_q.put((i, s, len(chunk)))
        except BaseException as exc:                 # noqa: BLE001 -- handed to the main thread
            err.append(exc)
            ready_q.put(None)

    def writer():
        try:
            while True:
                item = done_q.get()
                if item is None:
                    return
                i, s, cnt = item
                t0 = time.perf_counter()
                rows, nd = slots[s][1].numpy(), slots[s][2].numpy()
                for j in range(cnt):
                    res = rows_to_result(rows[j], int(nd[j]), fe.W_PIX, fe.HOP_SPECTRO, L, names)
                    f = files[i * batch + j]
                    if keep_results:
                        out[i * batch + j] = res
                    if write_txt:
                        with open(txt_path(f), 'w') as fh:
                            fh.write(str(res))
                t_write[0] += time.perf_counter() - t0
                free_q.put(s)
        except BaseException as exc:                 # noqa: BLE001
            err.append(exc)
            free_q.put(None)

    th_r, th_w = threading.Thread(target=reader, daemon=True), threading.Thread(target=writer, daemon=True)
    t_start = time.perf_counter()
    th_r.start(), th_w.start()
    inflight = []
    t_wait_in = 0.0
    try:
        with torch.no_grad(), torch.cuda.stream(det.stream):
            k, stop = 0, False
            while k < n_batches and not stop:
                group = []
                for j in range(min(lanes, n_batches - k)):      # the lanes' inputs: H2D on the graph's stream, in front of the replay
                    t0 = time.perf_counter()
                    item = ready_q.get()
                    t_wait_in += time.perf_counter() - t0
                    if item is None or err:
                        stop = True
                        break
                    det.pcms[j].copy_(slots[item[1]][0], non_blocking=True)
                    group.append(item)
                if not group:
                    break
                det.replay()                                    # a lane without a batch (odd tail) recomputes its previous input: ignored
                for j, (i, s, cnt) in enumerate(group):
                    slots[s][1].copy_(det.dets[j], non_blocking=True)
                    slots[s][2].copy_(det.n_dets[j], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(det.stream)
                inflight.append((group, ev))
                k += len(group)
                if len(inflight) >= 2:             # the GPU has the next group queued: now wait for the previous one
                    g0, e0 = inflight.pop(0)
                    e0.synchronize()
                    for it in g0:
                        done_q.put(it)
            for g0, e0 in inflight:
                e0.synchronize()
                for it in g0:
                    done_q.put(it)
    finally:
        done_q.put(None)
        free_q.put(None)                            # unblocks a reader that waits for a slot after an error
        th_w.join()
        th_r.join(timeout=5)
        if own_det:                                 # graph, static buffers and the extra lanes' scratch (tens of GB at B = 64) go now,
            torch.cuda.synchronize()                # not whenever the caller's frame dies: the per-file driver may run right behind this
            det.close()
    if err:
        raise err[0]
    if stats is not None:
        stats.update(wall_s=time.perf_counter() - t_start, reader_busy_s=t_read[0], writer_busy_s=t_write[0],
                     gpu_loop_waited_for_input_s=t_wait_in, batches=n_batches, depth=depth, lanes=lanes)
    return out
