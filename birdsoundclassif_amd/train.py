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
    `nbm_adamw_step`; gradient clipping (torch.nn.utils.clip_grad_norm_, train.py:213-214) is folded into the step:
    the clip coefficient is evaluated on the device from the accumulated squared norm, no host sync."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._sq = None
        self.last_grad_norm = None

    @torch.no_grad()
    def step(self, max_norm=0.0):
        from . import ops
        from .nets import _prep
        items = []
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is None:
                    continue
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st['step'] = int(st['step']) + 1
                items.append((p, st, group))
        if not items:
            return
        sq = None
        if max_norm and max_norm > 0:
            sq = torch.zeros((1,), device=items[0][0].device, dtype=torch.float64)
            for p, _, _ in items:
                ops.sqnorm_accum(p.grad, sq)
            self._sq = sq
        for p, st, group in items:
            b1, b2 = group['betas']
            ops.adamw_step(p.data, p.grad, st['exp_avg'], st['exp_avg_sq'], group['lr'], b1, b2, group['eps'],
                           group['weight_decay'], st['step'], sqnorm=sq, max_norm=max_norm or 0.0)
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


def step(model, criterion, batch, device, negative_sample):
    """One forward + loss evaluation (reference train.py:220-257)."""
    img, neg_img, bb_coord, bird_ids, lengths = batch
    img, neg_img, bb_coord, bird_ids = img.to(device), neg_img.to(device), bb_coord.to(device), bird_ids.to(device)
    loss = {}
    inpt = (neg_img if negative_sample else img)[:, None]
    out_first_stage = model.forward_first_stage(inpt)
    loss.update(criterion.first_stage_loss(out_first_stage['rpn_cls_scores'], out_first_stage['rpn_bbox_reg'],
                                           bb_coord, lengths, negative_sample))
    if len(out_first_stage['rois']) == 0:            # "RPN failed": first-stage loss only
        return loss
    if not negative_sample:
        proposal_tgt_out = criterion.generate_all_rois(out_first_stage['rois'], bb_coord, bird_ids, lengths)
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


def allreduce_grads(model, world_size=None):
    """Data-parallel exchange step (NEW capability): average the fp32 gradients of all ranks with ONE
    collective over a flat buffer (RCCL all-reduce over xGMI when the backend is nccl; gloo in the CPU tests)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(dist.get_world_size())
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def train_one_step(model, criterion, optimizer, batch, max_norm, device, negative_sample):
    """reference train.py:205-217."""
    loss_dict = step(model, criterion, batch, device, negative_sample)
    weight_dict = criterion.weight_dict
    losses = sum(loss_dict[k] * weight_dict[k] for k in loss_dict.keys() if k in weight_dict)
    optimizer.zero_grad()
    losses.backward()
    allreduce_grads(model)
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
