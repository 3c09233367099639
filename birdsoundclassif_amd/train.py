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
