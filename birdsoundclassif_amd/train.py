"""Training driver (reference nbm_model/train.py): the flag system and, in a later milestone, the
optimisation step on HIP backward kernels."""
import argparse

from .nets.util.nets_utils import setattr_others


def get_args_parser():
    """Same flags / defaults as the reference (train.py:21-168)."""
    p = argparse.ArgumentParser('Set detector', add_help=False)
    a = p.add_argument
    a('--lr', default=1e-4, type=float); a('--lr_backbone', default=1e-5, type=float)
    a('--batch_size', default=2, type=int); a('--weight_decay', default=1e-4, type=float)
    a('--lr_drop', default=383, type=int); a('--clip_max_norm', default=0.1, type=float)
    a('--model_name', default='new_model', type=str); a('--data_path', default='dataset', type=str)
    a('--save_dir', default='models', type=str); a('--max_steps', default=5e5, type=float)
    a('--first_neg_step', default=0, type=float); a('--neg_step_freq', default=10, type=int)
    a('--save_step', default=None, type=float)
    a('--img_width', default=1024, type=int); a('--img_height', default=375, type=int)
    a('--inpt_channels', default=1, type=int)
    a('--backbone', default='resnet50', type=str); a('--dilation', action='store_true')
    a('--position_embedding', default='sine', type=str, choices=('sine', 'learned'))
    a('--add_posenc', action='store_true', default=False)
    a('--one_dim_posenc', default=True, action='store_false')
    a('--norm_layer_backbone', default='frozen_batchnorm', type=str)
    for k in ('fs_cls', 'fs_neg_cls', 'fs_reg', 'sec_cls', 'sec_neg_cls', 'sec_reg'):
        a(f'--{k}_loss_coef', default=1, type=float)
    a('--focal_loss', action='store_true', default=False)
    a('--device', default='cuda'); a('--seed', default=42, type=int); a('--num_workers', default=4, type=int)
    a('--n_ratios', default=3, type=int); a('--anchor_stride', default=16, type=int); a('--base_size', default=16, type=int)
    a('--rpn_neg_label', default=0.3, type=float); a('--rpn_pos_label', default=0.7, type=float)
    a('--rpn_batchsize', default=16, type=int); a('--rpn_fg_fraction', default=0.5, type=float)
    a('--rcnn_batch_size', default=16, type=int); a('--rcnn_fg_prop', default=0.4, type=float)
    a('--fg_threshold', default=0.5, type=float); a('--bg_threshold_lo', default=0.1, type=float)
    a('--bg_threshold_hi', default=0.5, type=float); a('--depth_rcnn', default=3, type=int)
    a('--pre_nms_topN', default=3000, type=int); a('--min_threshold', default=5, type=int)
    a('--nms_thresh', default=0.7, type=float); a('--post_nms_topN', default=1000, type=int)
    a('--post_nms_topN_eval', default=50, type=int); a('--pre_nms_topN_eval', default=500, type=int)
    a('--roi_pool_h', default=2, type=int); a('--roi_pool_w', default=2, type=int)
    a('--hidden_size_rcnn', default=512, type=int); a('--dropout', default=0, type=float)
    a('--proposal_number', default=50, type=int)
    a('--fpn', default='fpn', type=str); a('--n_bifpn_layers', default=5, type=int)
    a('--fpn_p_chan', default=384, type=int); a('--out_fpn_chan', default=256, type=int)
    a('--fpn_first', action='store_true', default=False); a('--sandwich_attn', action='store_true', default=False)
    a('--tf_rcnn', action='store_true', default=False); a('--tf_pe_qk', action='store_true', default=False)
    a('--tf_model_dim', default=512, type=int); a('--tf_nhead', default=8, type=int)
    a('--tf_num_encoder_layers', default=6, type=int); a('--tf_dim_feedforward', default=1024, type=int)
    a('--pyramid_top_n_attn', default=2, type=int); a('--num_classes', default=150, type=int)
    a('--validation_prop', default=0.03, type=float)
    return p


def default_args(device='cuda', **overrides):
    """Namespace with the reference defaults + `setattr_others` derived fields."""
    args = get_args_parser().parse_args([])
    args.device = device
    for k, v in overrides.items():
        setattr(args, k, v)
    setattr_others(args)
    return args


# =========================================================================== optimisation step
import os  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (reference train.py:302-303) on the HIP kernels `nbm_sqnorm_accum` +
    `nbm_adamw_step`, over FLAT buffers: at construction every parameter of a group is re-homed into one contiguous
    fp32 buffer (values preserved; `p.data`, `p.grad`, `exp_avg`, `exp_avg_sq` become views), so that

      * `zero_grad` is one memset per group, the squared gradient norm one reduction launch per group, the update one
        launch per contiguous run of parameters that received a gradient (torch skips parameters whose grad is None:
        e.g. `bbox_reg_layer` in a negative step -- tracked with post-accumulate hooks),
      * gradient clipping (torch.nn.utils.clip_grad_norm_, train.py:213-214) is folded into the update: the clip
        coefficient is evaluated on the device from the accumulated squared norm, no host sync,
      * the data-parallel exchange is ONE all-reduce per group on the flat gradient buffer (`allreduce_grads`).
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._sq = None
        self._flat = []          # per group: dict(p, g, m, v, spans=[(param, offset, numel)], step)
        self._touched = set()
        for group in self.param_groups:
            ps = [p for p in group['params']]
            if not ps:
                self._flat.append(None)
                continue
            dev = ps[0].device
            self._check_device(dev)
            al = lambda k: (k + 63) // 64 * 64          # every view starts 256-byte aligned (kernels need 16 B)
            n = sum(al(p.numel()) for p in ps)
            fp = torch.zeros(n, device=dev, dtype=torch.float32)
            fg = torch.zeros(n, device=dev, dtype=torch.float32)
            fm = torch.zeros(n, device=dev, dtype=torch.float32)
            fv = torch.zeros(n, device=dev, dtype=torch.float32)
            spans, off = [], 0
            for p in ps:
                k = p.numel()
                with torch.no_grad():
                    fp[off:off + k].copy_(p.data.reshape(-1))
                p.data = fp[off:off + k].view(p.shape)
                p.grad = fg[off:off + k].view(p.shape)
                self.state[p] = {'step': 0, 'exp_avg': fm[off:off + k].view(p.shape),
                                 'exp_avg_sq': fv[off:off + k].view(p.shape)}
                p.register_post_accumulate_grad_hook(self._mark)
                spans.append((p, off, al(k)))       # the zero padding rides along (stays exactly zero)
                off += al(k)
            self._flat.append(dict(p=fp, g=fg, m=fm, v=fv, spans=spans))
        from .nets import _prep
        _prep.bump()
        try:                                     # weight-gradient kernels may accumulate straight into these views (functional.grad_sink)
            import weakref
            from .nets import functional as Fn
            for f in self._flat:
                for (p, _, _) in (f['spans'] if f is not None else []):
                    if p.is_cuda:
                        Fn.GRAD_SINK[p.data_ptr()] = (weakref.ref(p), self._mark)
        except (ImportError, OSError, RuntimeError):     # the CPU twin of the optimiser in the gloo tests: no HIP library, no sink
            pass

    @staticmethod
    def _check_device(dev):
        if dev.type != 'cuda':
            raise RuntimeError('FusedAdamW needs the parameters on the GPU (no CPU fallback)')

    def _mark(self, p):
        self._touched.add(id(p))

    def all_params(self):
        """Every parameter in flat-buffer order (the order of the `touched` bitmap exchanged between ranks)."""
        return [p for f in self._flat if f is not None for (p, _, _) in f['spans']]

    def touched_bitmap(self):
        """int32 [n_params] on the host: 1 where the parameter received a gradient in the last backward."""
        return torch.tensor([1 if id(p) in self._touched else 0 for p in self.all_params()], dtype=torch.int32)

    def set_touched_bitmap(self, bits):
        self._touched = {id(p) for p, b in zip(self.all_params(), bits.tolist()) if b}

    def load_state_dict(self, state_dict):
        """torch's loader replaces `state[p]` by fresh tensors; the kernels read the FLAT moment buffers, so copy the
        loaded moments into their spans and point the state back at the views (step counts are kept).  Without this a
        resumed run would continue with m = v = 0 and a large step count, i.e. without bias correction."""
        super().load_state_dict(state_dict)
        for f in self._flat:
            if f is None:
                continue
            for (p, off, _) in f['spans']:
                st = self.state.get(p)
                if not st:                             # parameter never stepped before the checkpoint
                    self.state[p] = st = {'step': 0}
                k = p.numel()
                for name, buf in (('exp_avg', f['m']), ('exp_avg_sq', f['v'])):
                    view = buf[off:off + k].view(p.shape)
                    if name in st and st[name].data_ptr() != view.data_ptr():
                        view.copy_(st[name].to(view.device, torch.float32))
                    elif name not in st:
                        view.zero_()
                    st[name] = view
                st['step'] = int(st.get('step', 0))

    def mark_all(self):
        """Treat every parameter as having received a gradient (for gradients written into `p.grad` by hand)."""
        for f in self._flat:
            if f is not None:
                self._touched.update(id(p) for (p, _, _) in f['spans'])

    def zero_grad(self, set_to_none=True):
        """Gradients live in the flat buffers: zero them in place (never set to None)."""
        for f in self._flat:
            if f is not None:
                f['g'].zero_()
        self._touched.clear()

    def flat_grads(self):
        return [f['g'] for f in self._flat if f is not None]

    @torch.no_grad()
    def step(self, max_norm=0.0):
        from . import ops
        from .nets import _prep
        sq = None
        if max_norm and max_norm > 0:
            sq = torch.zeros((1,), device=self._flat[0]['g'].device, dtype=torch.float64)
            for f in self._flat:
                if f is not None:
                    ops.sqnorm_accum(f['g'], sq)      # untouched parameters hold zeros
            self._sq = sq
        for group, f in zip(self.param_groups, self._flat):
            if f is None:
                continue
            b1, b2 = group['betas']
            # contiguous runs of parameters that received a gradient in this backward and share the step count
            runs, cur = [], None
            for (p, off, k) in f['spans']:
                if id(p) not in self._touched:
                    cur = None
                    continue
                st = self.state[p]
                st['step'] = int(st['step']) + 1
                step = st['step']
                if cur is not None and cur[2] == step and cur[0] + cur[1] == off:
                    cur[1] += k
                else:
                    cur = [off, k, step]
                    runs.append(cur)
            for off, k, step in runs:
                ops.adamw_step(f['p'][off:off + k], f['g'][off:off + k], f['m'][off:off + k], f['v'][off:off + k],
                               group['lr'], b1, b2, group['eps'], group['weight_decay'], step, sqnorm=sq,
                               max_norm=max_norm or 0.0)
        _prep.bump()

    def grad_norm(self):
        """Total gradient norm seen by the last clipped step (one host sync; for logging / tests)."""
        return None if self._sq is None else float(self._sq.sqrt().item())


def build_optimizer(model, args):
    """Two parameter groups exactly like reference train.py:295-304."""
    param_dicts = [
        {'params': [p for n, p in model.named_parameters() if 'backbone' not in n and p.requires_grad]},
        {'params': [p for n, p in model.named_parameters() if 'backbone' in n and p.requires_grad], 'lr': args.lr_backbone},
    ]
    optimizer = FusedAdamW(param_dicts, lr=args.lr, weight_decay=args.weight_decay)
    lr_scheduler = torch.optim.lr_scheduler.StepLR(optimizer, args.lr_drop)
    return optimizer, lr_scheduler


# The first-stage loss is back-propagated through the RPN head while the host is still busy with the proposal targets (see
# functional.EARLY): the GPU used to idle ~12 ms per step behind that sync.  Needs `optimizer.zero_grad()` BEFORE `step` (train_one_step
# does that); the first-stage losses come back detached (their gradient is already in the parameters / parked at the FPN maps).
SPLIT_BACKWARD = os.environ.get('NBM_SPLIT_BACKWARD', '1') != '0'


# The IoU / arg-max / threshold half of the AnchorTargetLayer on the device (nbm_anchor_targets); the NumPy draws stay on the host.
ANCHOR_TARGETS_ON_DEVICE = os.environ.get('NBM_ANCHOR_DEVICE', '1') != '0'


def step(model, criterion, batch, device, negative_sample, early_backward=False):
    """One forward + loss evaluation (reference train.py:220-257).  `early_backward` (train_one_step): see SPLIT_BACKWARD."""
    img, neg_img, bb_coord, bird_ids, lengths = batch
    img, neg_img = img.to(device), neg_img.to(device)
    # Boxes and class ids are consumed by the host-side target layers (NumPy RNG, like the reference's `.cpu().numpy()`
    # round trips): keep them on the host.  A device copy here would make every later `.cpu()` a stream-ordered wait
    # for ALL queued GPU work (it cost 160 ms per step at B=128); loaders hand them over as CPU tensors anyway.
    bb_coord, bird_ids = bb_coord.cpu(), bird_ids.cpu()
    loss = {}
    inpt = (neg_img if negative_sample else img)[:, None]
    host_work = None
    if not negative_sample and hasattr(criterion, 'start_anchor_targets') and ANCHOR_TARGETS_ON_DEVICE:
        criterion.start_anchor_targets(bb_coord, lengths, inpt.device)       # labels only: queued on a side stream before the forward pass
    if not negative_sample and hasattr(criterion, 'precompute_first_stage_loss'):
        # AnchorTargetLayer (host, NumPy RNG) runs while the GPU executes the first-stage forward queued before it, and the
        # first-stage loss kernels are queued behind that forward pass before the host waits for the RoI count
        def host_work(cls, reg, rois=None):
            if rois is not None and hasattr(criterion, 'precompute_proposal_iou'):
                criterion.precompute_proposal_iou(rois, bb_coord, lengths)      # device IoU + D2H, queued right behind the proposals
            criterion.precompute_first_stage_loss(cls, reg, bb_coord, lengths)
            if early_backward and SPLIT_BACKWARD and torch.is_grad_enabled():
                _early_rpn_backward(model, criterion)
        host_work.wants_rois = True
    # lazy=True: the finest FPN map is computed where it is read; it goes straight into forward_second_stage below
    # (also in a negative step, whose 1000 RoIs per image list nearly every tile: 620-660 ms against 757 ms with dense maps at B = 128,
    # scripts/negstep_ab.py)
    out_first_stage = model.forward_first_stage(inpt, host_work, lazy=True)
    loss.update(criterion.first_stage_loss(out_first_stage['rpn_cls_scores'], out_first_stage['rpn_bbox_reg'],
                                           bb_coord, lengths, negative_sample))
    if len(out_first_stage['rois']) == 0:            # "RPN failed": first-stage loss only
        return loss
    if not negative_sample:
        proposal_tgt_out = criterion.generate_all_rois(out_first_stage['rois'], bb_coord, bird_ids, lengths,
                                                       **({'use_precomputed': True} if host_work is not None else {}))
        if proposal_tgt_out['rois'] is None:
            return loss
    else:
        proposal_tgt_out = {'rois': out_first_stage['rois'], 'bbox_targets': None, 'labels': None}
    out_second_stage = model.forward_second_stage(out_first_stage['fpn_out'], proposal_tgt_out['rois'], training=True)
    loss.update(criterion.second_stage_loss(out_second_stage['bbox_reg'], out_second_stage['bbox_classes'],
                                            proposal_tgt_out['bbox_targets'], proposal_tgt_out['labels'], negative_sample))
    if not negative_sample:
        loss.update(criterion.loss_cardinality(out_second_stage['bbox_classes'], proposal_tgt_out['labels']))
    return loss


def _early_rpn_backward(model, criterion):
    """Queue the backward pass of the (pre-queued) first-stage loss through the RPN head now, before the host waits for the RoI
    count: gradients of the RPN parameters accumulate, the shares of d/d(FPN maps) are parked (functional._PARKED) for the RoI pooling's
    backward pass -- or `parked_flush`, if the step ends after the first stage."""
    from .nets import functional as Fn
    pre = criterion._pre_loss
    w = criterion.weight_dict
    keys = [k for k in pre if k in w and pre[k].requires_grad]
    rpn = getattr(getattr(model, 'head', None), 'rpn', None)
    if not keys or rpn is None or not Fn.GRAD_SHARE or not Fn._FPN_OUT:
        return
    Fn.pass_check_owner(model, 'early RPN backward')
    params = [p for p in rpn.parameters() if p.requires_grad]
    loss = sum(pre[k] * w[k] for k in keys)
    Fn.EARLY = True
    try:
        torch.autograd.backward(loss, inputs=params)       # stops at the RPN's first operators: they park d/d(FPN map) themselves
    finally:
        Fn.EARLY = False
    if len(Fn._PARKED) != len(Fn._FPN_OUT):
        raise RuntimeError('early RPN backward: not every FPN map received its share (functional._PARKED)')
    criterion._pre_loss = {k: (v.detach() if k in keys else v) for k, v in pre.items()}


_CONTROL_GROUP = {}
# Per-rank timings of the last data-parallel exchange (bench.py reports them; reset by `exchange_stats_reset`):
#   control_ms   host wall time of the touched-bitmap all-reduce over the gloo control group (includes waiting for the slowest rank's HOST)
#   exchange_ms  device time between the first and the last flat-buffer all-reduce of the step finishing, HIP events on the launch stream
#                (CPU tensors: host wall time); with the overlapped form this window runs beside the backbone's backward kernels
#   overlapped   1 when the non-backbone buffer's all-reduce was started from inside the backward pass (see DP_OVERLAP)
EXCHANGE_STATS = None


def exchange_stats_reset(enabled=True):
    global EXCHANGE_STATS
    EXCHANGE_STATS = {'steps': 0, 'control_ms': [], 'exchange_events': [], 'exchange_ms_host': [], 'overlapped': 0} if enabled else None


def exchange_stats_summary():
    """-> dict of means over the recorded steps (synchronises the recorded events), or None."""
    st = EXCHANGE_STATS
    if not st or not st['steps']:
        return None
    ms = [a.elapsed_time(b) for a, b in st['exchange_events']] + list(st['exchange_ms_host'])
    n = st['steps']
    return {'steps': n, 'control_ms': sum(st['control_ms']) / max(1, len(st['control_ms'])),
            'exchange_ms': sum(ms) / max(1, len(ms)), 'overlapped_steps': st['overlapped']}


def _device_backend(dist):
    """Backend of the default process group ('nccl' = RCCL on ROCm).  A function of its own so that the CPU test can force the
    RCCL-side control flow (separate gloo control group) over a gloo default group."""
    return dist.get_backend()


def init_control_group(dist=None):
    """Create the process group for small HOST tensors NOW -- collectively, on every rank, right after `init_process_group` (train
    `__main__`, `bench.dist_setup`): the default group itself under gloo (returns None), a gloo group beside an RCCL default group.
    Creating it lazily inside the first training step would put a collective `new_group` behind data-dependent control flow."""
    if dist is None:
        import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or _device_backend(dist) != 'nccl':
        return None
    key = id(dist.group.WORLD)
    if key not in _CONTROL_GROUP:
        _CONTROL_GROUP[key] = dist.new_group(backend='gloo')
    return _CONTROL_GROUP[key]


def _control_group(dist):
    """The group of `init_control_group` (created there; a caller that skipped the eager call still gets it here, at the first
    data-parallel step, where every rank arrives)."""
    return init_control_group(dist)


def _flat_allreduce(dist, buf, world, async_op=False):
    """Average one flat fp32 gradient buffer over the ranks.  -> work handle (async_op) or None."""
    if dist.get_backend() == 'nccl':
        return dist.all_reduce(buf, op=dist.ReduceOp.AVG, async_op=async_op)          # RCCL averages in the collective itself
    w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=async_op)
    if async_op:
        return (w, buf, world)
    buf.div_(world)
    return None


def _flat_wait(handle):
    if handle is None:
        return
    if isinstance(handle, tuple):
        w, buf, world = handle
        w.wait()
        buf.div_(world)
    else:
        handle.wait()


# Overlap of the exchange with the backward pass ("bucketed in reverse-autograd order, overlapped with backward"), in the
# two buckets the optimiser already has: the NON-backbone flat buffer (FPN, attention, RPN, RCNN head: ~20 M gradients, 80 MB) is
# final when autograd hands the gradient of the backbone's LAST tap (c5) to the backbone -- every node of the heads, the FPN
# (top-down chain: the coarsest lateral runs last) and the attention pyramid was created after every backbone node and c5's gradient
# is the sum of their shares -- so `NbmModel._fpn_nhwc` puts a tensor hook on that tap and the hook starts the first all-reduce
# (async; RCCL's stream waits for the kernels queued so far) while the backbone's backward kernels (~40 % of the backward pass) are
# still to run.  The backbone buffer follows after `backward()`.  The ORDER of collectives is the same on every rank whatever its
# control flow (buffer 0, then buffer 1; a rank whose hook never fired -- no backward pass at all -- starts buffer 0 afterwards).
DP_OVERLAP = os.environ.get('NBM_DP_OVERLAP', '1') != '0'
_PENDING = {}            # id(optimizer) -> {'handles': [...], 'started': int, 't0': event | float}


def _dist_active():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def exchange_begin(optimizer):
    """Called by `train_one_step` before the step: arms the overlapped exchange for this optimiser (FusedAdamW + world > 1 only)."""
    _PENDING.pop(id(optimizer), None)
    if DP_OVERLAP and hasattr(optimizer, 'flat_grads') and _dist_active():
        _PENDING[id(optimizer)] = {'handles': [], 'started': 0, 't0': None, 'opt': optimizer}


def _mark_t0(st, like):
    if st['t0'] is None:
        if like.is_cuda:
            st['t0'] = torch.cuda.Event(enable_timing=True)
            st['t0'].record()
        else:
            import time
            st['t0'] = time.perf_counter()


def backbone_boundary_hook(grad):
    """Tensor hook on the backbone's last tap (registered by NbmModel._fpn_nhwc while an exchange is armed): every non-backbone
    gradient is final -> start the all-reduce of flat buffer 0 beside the backbone's backward pass."""
    import torch.distributed as dist
    from .nets import functional as Fn
    if Fn._PARKED:
        # an RPN share of d/d(FPN map) that no RoI pooling picked up is still parked: `parked_flush` will back-propagate it into FPN /
        # attention gradients of buffer 0 AFTER this hook -- the buffer is not final, `allreduce_grads` starts it behind the flush
        return None
    for st in _PENDING.values():
        if st['started'] == 0:
            bufs = st['opt'].flat_grads()
            _mark_t0(st, bufs[0])
            st['handles'].append(_flat_allreduce(dist, bufs[0], dist.get_world_size(), async_op=True))
            st['started'] = 1
            if EXCHANGE_STATS is not None:
                EXCHANGE_STATS['overlapped'] += 1
    return None


def exchange_armed():
    return bool(_PENDING)


class PeerStepError(RuntimeError):
    """Raised by `allreduce_grads` on EVERY rank of a data-parallel job when any rank's step raised (the error bit of the control
    all-reduce): all ranks leave the same step with an exception instead of one rank leaving and its peers blocking in the collective."""


def allreduce_grads(optimizer_or_model, failed=None):
    """Data-parallel exchange step (NEW capability): average the fp32 gradients of all ranks -- one
    collective per flat gradient buffer (RCCL all-reduce over xGMI when the backend is nccl; gloo in the CPU tests).
    No-op when torch.distributed is not initialised or world_size == 1.

    With `FusedAdamW` the set of parameters that received a gradient is made the UNION over the ranks first: the soft
    failure paths of `step` ("RPN failed", proposal batch cannot be filled -- data dependent, reference train.py:232-247)
    leave the second-stage parameters without a gradient on one rank only; that rank must still apply the averaged
    gradient and advance its Adam step count like its peers, or the replicas drift apart for good.  The bitmap has ~400
    int32 entries and is a HOST tensor reduced over gloo in either case (`_control_group`).

    When `exchange_begin` armed the overlapped form, buffer 0's all-reduce may already be in flight (started by
    `backbone_boundary_hook` inside the backward pass); this call starts whatever has not been started, in buffer order, and waits.

    `failed`: the exception this rank's step ended in, or None.  A rank whose step raised still takes part in every collective of the
    step (its gradient buffers hold whatever the step left there: nobody applies them) and sets the ERROR BIT that rides behind the
    touched bitmap; every rank that sees the bit raises `PeerStepError` after the collectives have completed -- so a Python-level failure
    on one rank ends the same step on all ranks within the time of one exchange, instead of leaving the peers in the RCCL all-reduce
    until its timeout."""
    import time
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    world = dist.get_world_size()
    stats = EXCHANGE_STATS
    if hasattr(optimizer_or_model, 'flat_grads'):
        bufs = optimizer_or_model.flat_grads()
        st = _PENDING.pop(id(optimizer_or_model), None) or {'handles': [], 'started': 0, 't0': None}
        _mark_t0(st, bufs[0])
        for b in bufs[st['started']:]:                    # same order on every rank: buffer 0 (unless the hook started it), buffer 1
            st['handles'].append(_flat_allreduce(dist, b, world, async_op=True))
        bits = torch.cat([optimizer_or_model.touched_bitmap(), torch.tensor([0 if failed is None else 1], dtype=torch.int32)])
        # the bitmap is host data and decides host control flow: it travels through a gloo group of its own (a host tensor over
        # loopback / TCP, ~0.1 ms), NOT through RCCL -- a device round trip here is a stream synchronisation per step, i.e. the
        # host loses its run-ahead and the anchor targets of the next step (160 ms of NumPy at B = 128) stop being hidden
        tc = time.perf_counter()
        dist.all_reduce(bits, op=dist.ReduceOp.MAX, group=_control_group(dist))
        optimizer_or_model.set_touched_bitmap(bits[:-1])
        peer_failed = bool(bits[-1])
        control_ms = (time.perf_counter() - tc) * 1e3
        for h in st['handles']:
            _flat_wait(h)                                  # RCCL: the compute stream waits for the collective (no host block)
        if stats is not None:
            stats['steps'] += 1
            stats['control_ms'].append(control_ms)
            if bufs[0].is_cuda:
                t1 = torch.cuda.Event(enable_timing=True)
                t1.record()
                stats['exchange_events'].append((st['t0'], t1))
            else:
                stats['exchange_ms_host'].append((time.perf_counter() - st['t0']) * 1e3)
        if peer_failed:
            raise PeerStepError('the training step raised on at least one rank of the data-parallel job (error bit of the control '
                                'all-reduce): every rank leaves this step; the failing rank re-raises its own exception')
    else:
        # plain module (torch optimiser; the gloo CPU tests): EVERY trainable parameter in module order, zeros where this rank
        # has no gradient, plus one flag per parameter -- so that all ranks reduce buffers of the same length whatever soft
        # failure one of them went through, and a parameter that got a gradient on ANY rank gets the average on every rank
        params = [p for p in optimizer_or_model.parameters() if p.requires_grad]
        if not params:
            return
        dev = params[0].device
        flags = torch.tensor([0.0 if p.grad is None else 1.0 for p in params] + [0.0 if failed is None else 1.0], device=dev)
        flat = torch.cat([(torch.zeros_like(p) if p.grad is None else p.grad).reshape(-1).to(torch.float32) for p in params] + [flags])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if float(flat[-1]) > 0:
            raise PeerStepError('the training step raised on at least one rank of the data-parallel job (error flag of the exchange)')
        flat = flat[:-1]
        touched = flat[-len(params):] > 0
        flat.div_(world)
        off = 0
        for p, t in zip(params, touched.tolist()):
            n = p.numel()
            if t:
                avg = flat[off:off + n].view_as(p).to(p.dtype)
                if p.grad is None:
                    p.grad = avg.clone()
                else:
                    p.grad.copy_(avg)
            off += n


def train_one_step(model, criterion, optimizer, batch, max_norm, device, negative_sample):
    """reference train.py:205-217."""
    from .nets import functional as Fn
    optimizer.zero_grad()                  # before the step: part of the backward pass runs inside it (SPLIT_BACKWARD)
    exchange_begin(optimizer)              # data parallel: arm the overlapped exchange (no-op for one rank / a torch optimiser)
    try:
        loss_dict = step(model, criterion, batch, device, negative_sample, early_backward=True)
        weight_dict = criterion.weight_dict
        losses = sum(loss_dict[k] * weight_dict[k] for k in loss_dict.keys() if k in weight_dict)
        if losses.requires_grad:
            losses.backward()
        Fn.parked_flush()                  # the step ended after the first stage: the RPN branch's gradients still have to reach the FPN
        Fn.stash_check_empty()             # a handed-over gradient that nobody picked up would be a silently dropped gradient
    except BaseException as exc:
        Fn.pass_abandon()                  # whatever this step parked / stashed dies with it: the next step starts clean
        if isinstance(exc, Exception) and _dist_active():
            # data parallel: the peers are in (or on their way into) this step's collectives -- take part with the error bit set so that
            # every rank raises in this same step (PeerStepError there, this rank's own exception here)
            try:
                allreduce_grads(optimizer if isinstance(optimizer, FusedAdamW) else model, failed=exc)
            except PeerStepError:
                pass
        _PENDING.pop(id(optimizer), None)
        raise
    allreduce_grads(optimizer if isinstance(optimizer, FusedAdamW) else model)
    if isinstance(optimizer, FusedAdamW):
        optimizer.step(max_norm=max_norm)
    else:
        if max_norm > 0:
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
        optimizer.step()
    return loss_dict


def seed_everything(seed):
    import random
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def save(out_dir, model, epoch, steps, best_val_cls_loss, label, optim=None, scheduler=None, train_indices=None,
         val_indices=None):
    """Checkpoint layout of reference train.py:171-187."""
    save_dict = dict(checkpoints={k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}, steps=steps,
                     epoch=epoch, best_val_cls_loss=best_val_cls_loss)
    if optim is not None:
        save_dict['optim'] = optim.state_dict()
    if scheduler is not None:
        save_dict['scheduler'] = scheduler.state_dict()
    if train_indices is not None:
        save_dict['train_indices'] = train_indices
    if val_indices is not None:
        save_dict['val_indices'] = val_indices
    torch.save(save_dict, os.path.join(out_dir, 'model_chkpt_' + label + '.pt'))


def resume_training(out_dir, model, optim, scheduler, lr_drop):
    """reference train.py:190-202."""
    from .nets import _prep
    save_dict = torch.load(os.path.join(out_dir, 'model_chkpt_last.pt'), map_location='cpu', weights_only=False)
    model.load_state_dict(save_dict['checkpoints'])
    _prep.bump()
    model.train()
    optim.load_state_dict(save_dict['optim'])
    sd = save_dict['scheduler']
    sd['step_size'] = lr_drop
    scheduler.load_state_dict(sd)
    return (model, optim, scheduler, save_dict.get('train_indices'), save_dict.get('val_indices'), save_dict['epoch'],
            save_dict['steps'], save_dict['best_val_cls_loss'])


# =========================================================================== training driver (caller of the hot path)
class ScalarLog:
    """Stand-in for `SummaryWriter.add_scalar` (tensorboard is not part of this build): one JSON object per line in
    `<save_dir>/scalars.jsonl`, written by rank 0 only."""

    def __init__(self, save_dir, enabled=True):
        self.f = open(os.path.join(save_dir, 'scalars.jsonl'), 'a') if enabled else None

    def add_scalar(self, tag, value, global_step):
        if self.f is not None:
            import json
            self.f.write(json.dumps({'tag': tag, 'value': float(value), 'step': int(global_step)}) + '\n')
            self.f.flush()


class _RawItems(torch.utils.data.Dataset):
    """DataLoader view of `Img_dataset` whose workers do the host half only (inflate, labels, RNG draws)."""

    def __init__(self, dataset):
        self.dataset = dataset

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, idx):
        return self.dataset.raw_item(idx)


def evaluate_test_files(model, args, wav_paths, bird_dicts_path='bird_dict.json', min_score=0.02):
    """reference train.py:389-394: run_detection on every annotated test recording -> AP / mAP / Rec / mRec."""
    from .nets.util.nets_utils import compute_AP_scores, format_txt_annots
    from .run_detection import run_detection
    outputs = [(run_detection(model, args, w, bird_dicts_path, min_score=min_score), format_txt_annots(w.replace('.wav', '.txt')))
               for w in wav_paths]
    return compute_AP_scores(outputs)


def main(args):
    """Training loop of reference train.py:273-404 around `train_one_step`: same checkpoint / resume protocol, loss
    bookkeeping, StepLR stepped every 1000 iterations, validation (eval-mode `step`) and test-set AP every `val_freq`
    (500) iterations.  Differences: images are decoded and augmented on the device (`Img_dataset` + `DeviceCollate`),
    scalars go to `scalars.jsonl`, and under torchrun every rank trains on its own slice of the training indices with
    the gradient all-reduce of `train_one_step` (rank 0 validates, tests and saves)."""
    import glob
    import json
    import torch.distributed as dist
    from torch.utils.data import DataLoader, SubsetRandomSampler
    from .nbm_datasets.image_dataset import Img_dataset
    from .nets import build_model
    from .nets.util.nets_utils import train_test_split

    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
    device = torch.device(args.device)
    seed_everything(args.seed)                      # the train / validation split must be the same on every rank
    save_dir = os.path.join(args.save_dir, args.model_name)
    os.makedirs(save_dir, exist_ok=True)
    resume = os.path.isfile(os.path.join(save_dir, 'model_chkpt_last.pt'))
    if rank == 0:
        with open(os.path.join(save_dir, 'args'), 'w') as f:
            json.dump({k: v for k, v in args.__dict__.items() if isinstance(v, (int, float, str, bool, type(None)))}, f)
    setattr_others(args)
    val_freq, log_freq = getattr(args, 'val_freq', 500), getattr(args, 'log_freq', 50)

    model, criterion = build_model(args)
    model.to(device)
    optimizer, lr_scheduler = build_optimizer(model, args)
    dataset = Img_dataset(args.data_path, transform=True, device=args.device, host_noise=getattr(args, 'host_noise', False))
    if resume:
        model, optimizer, lr_scheduler, train_indices, val_indices, epoch, steps, best_val_cls_loss = \
            resume_training(save_dir, model, optimizer, lr_scheduler, args.lr_drop)
        print('Resuming training~~~~')
    else:
        train_indices, val_indices = train_test_split(len(dataset), val_prop=args.validation_prop)
        epoch, steps, best_val_cls_loss = 0, 0, 99
    if world > 1:
        seed_everything(args.seed + rank)           # per-rank sampling / augmentation streams
    raw = _RawItems(dataset)
    loader = lambda idx, bs, drop: DataLoader(raw, batch_size=bs, sampler=SubsetRandomSampler(idx), collate_fn=list,
                                              num_workers=args.num_workers, drop_last=drop)
    train_loader = loader(list(train_indices)[rank::world], args.batch_size, False)
    validation_loader = loader(val_indices, 2 * args.batch_size, True) if len(val_indices) > 0 else None
    writer = ScalarLog(save_dir, rank == 0)
    loss_keys = ['first_class_loss', 'first_regression_loss', 'sec_class_loss', 'sec_regression_loss',
                 'first_neg_class_loss', 'sec_neg_class_loss', 'cardinality_error']
    train_losses = {k: 0 for k in loss_keys}
    save_steps = [180e3, 190e3, 200e3]
    as_float = lambda k, v: float(v) if k == 'cardinality_error' else v.item()
    model.train(), criterion.train()
    # everything built so far (modules, parameters, dataset index, caches) is long-lived: take it out of the cyclic collector's view,
    # so that a generation-2 pass inside the loop walks the step's garbage only (30-90 ms pauses on single steps otherwise: bench.py)
    import gc
    gc.collect()
    gc.freeze()
    print('Start training')
    while steps < args.max_steps:
        for raw_batch in train_loader:
            batch = dataset.collate(raw_batch)
            negative = (steps % args.neg_step_freq == 0) and (steps > args.first_neg_step)
            losses = train_one_step(model, criterion, optimizer, batch, args.clip_max_norm, device, negative_sample=negative)
            for k, v in losses.items():
                train_losses[k] += as_float(k, v)
            if steps % log_freq == 0:
                for k in loss_keys:
                    freq = log_freq / args.neg_step_freq if 'neg' in k else log_freq
                    writer.add_scalar(f'Training_Loss/{k}', train_losses[k] / freq, steps)
                    train_losses[k] = 0
            if steps in save_steps and rank == 0:
                save(save_dir, model, epoch, steps, best_val_cls_loss, str(steps), optimizer, lr_scheduler, train_indices, val_indices)
            steps += 1
            if steps % 1000 == 0:
                lr_scheduler.step()
                writer.add_scalar('Lr', lr_scheduler.get_last_lr()[0], steps)
            if steps % val_freq == 0 and rank == 0:
                model.eval(), criterion.eval()
                if validation_loader is not None and len(validation_loader) > 0:
                    val_losses = {k: 0 for k in loss_keys}
                    n_val = 0
                    for raw_val in validation_loader:
                        valid_batch = dataset.collate(raw_val)
                        with torch.no_grad():
                            loss_dict = step(model, criterion, valid_batch, device, negative_sample=False)
                        for k, v in loss_dict.items():
                            val_losses[k] += as_float(k, v)
                        n_val += 1
                    for k in loss_keys:
                        val_losses[k] /= max(1, n_val)
                    with torch.no_grad():
                        loss_dict = step(model, criterion, valid_batch, device, negative_sample=True)
                    for k, v in loss_dict.items():
                        val_losses[k] += as_float(k, v)
                    for k in loss_keys:
                        writer.add_scalar(f'Val_Loss/{k}', val_losses[k], steps)
                    if (steps / 1000 > args.lr_drop) and (val_losses['sec_class_loss'] < best_val_cls_loss):
                        best_val_cls_loss = val_losses['sec_class_loss']
                        save(save_dir, model, epoch, steps, best_val_cls_loss, 'best')
                wavs = sorted(glob.glob(os.path.join(args.data_path, 'test_files', 'XC_annots') + '/*.wav'))
                if wavs:
                    for k, v in evaluate_test_files(model, args, wavs).items():
                        writer.add_scalar(f'Test_metrics/{k}', v, steps)
                model.train(), criterion.train()
            if steps >= args.max_steps:
                break
        if (epoch > 0) and (epoch % 10 == 0) and rank == 0:
            save(save_dir, model, epoch, steps, best_val_cls_loss, 'last', optimizer, lr_scheduler, train_indices, val_indices)
        epoch += 1
    return steps


if __name__ == '__main__':
    _p = argparse.ArgumentParser('NbmModel training and evaluation script', parents=[get_args_parser()])
    _args = _p.parse_args()
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        import datetime
        import torch.distributed as _dist
        _lr = int(os.environ.get('LOCAL_RANK', '0'))
        torch.cuda.set_device(_lr)
        # ranks > 0 wait in the next all-reduce while rank 0 validates and runs the test-set detection: give the
        # collective watchdog room for that instead of its 10-minute default
        _dist.init_process_group('nccl', timeout=datetime.timedelta(hours=2), device_id=torch.device('cuda', _lr))
        init_control_group(_dist)          # the gloo group for host-side control data, created collectively before any step
    main(_args)
