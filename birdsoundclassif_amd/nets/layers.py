"""Detector head layers (reference nets/layers.py), HIP forward.

Same class names, constructor arguments and state_dict keys as the reference.  Public `forward`s keep the
reference's NCHW-shaped return values (as permuted views of the NHWC buffers the kernels work on); the
`*_device` methods are the sync-free internal path (fixed-capacity buffers + device-side counters) used by
`NbmModel.forward`.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import ops, ondemand
from . import _prep, functional as Fn
from .position_encoding import one_dimension_positional_encoding
from .targets import AnchorTargetLayer, ProposalTargetLayer   # noqa: F401  (reference exports them from layers.py)
from .util.nets_utils import (weight_init, generate_anchors_frcnn as generate_anchors,
                              get_anchor_shifts_frcnn as get_anchor_shifts)


def _nhwc(t):
    """NCHW-shaped tensor (any strides) -> contiguous NHWC tensor (no copy for our own permuted views)."""
    return t.permute(0, 2, 3, 1).contiguous()


def _pow2_cap(n):
    c = 64
    while c < n:
        c <<= 1
    return c


class DepthwiseSepConv2d(nn.Module):
    """Inverted depthwise-separable block (reference layers.py:13-46): [bilinear x(1/stride)] -> depthwise 3x3
    (channel multiplier = expansion_fact) -> [FiLM from the positional encoding] -> 1x1 -> BatchNorm -> SiLU."""

    def __init__(self, indim, outdim, kernel=3, stride=1, expansion_fact=4, bias_out=True, pe_channels=None):
        super().__init__()
        if kernel != 3:
            raise NotImplementedError('only 3x3 depthwise kernels are on the hot path')
        self.stride = stride
        self.expansion_fact = expansion_fact
        self.depth_wise = nn.Conv2d(indim, expansion_fact * indim, kernel, stride=int(max(1, stride)), padding=1,
                                    groups=indim)
        if pe_channels is not None:
            self.pe_proj = nn.Conv2d(pe_channels, 2 * expansion_fact * indim, 1)
        self.pt_wise = nn.Conv2d(expansion_fact * indim, outdim, 1, bias=bias_out)
        self.norm = nn.BatchNorm2d(outdim)
        self.act = nn.SiLU()

    def forward(self, x, pe_act=None):
        """x NHWC; pe_act = SiLU(pe) NHWC (the activation is shared by the blocks of the RCNN) or None.
        eval: BatchNorm (running statistics) + SiLU are folded into the 1x1 GEMM's epilogue;
        train: batch statistics (nn.BatchNorm2d semantics) through the differentiable ops."""
        st = int(max(1, self.stride))
        if self.training:
            if pe_act is None and Fn.LAZY_DGRAD and Fn.LAZY_WGRAD and ondemand.train_composite_ready(x, self) is not None:
                # a demand-driven FPN map with a backward pass to come: this block (up to its BatchNorm) composed with the map's own
                # convolution in the cell domain -- the map's pattern pixels are never formed
                out = Fn.RpnComposite.apply(x, self.depth_wise.weight, self.depth_wise.bias, self.pt_wise.weight, self.pt_wise.bias)
                self.norm.num_batches_tracked += 1
                out = Fn.BatchNormTrain.apply(out, self.norm.weight, self.norm.bias, self.norm.running_mean,
                                              self.norm.running_var, self.norm.eps, self.norm.momentum)
                return Fn.Silu.apply(out)
            ondemand.pattern_materialize(x)          # (the map's pattern pixels were left pending and this block cannot take them)
            if self.stride < 1:
                size = ((1 / self.stride) * np.array(x.shape[1:3])).astype(np.int64).tolist()
                x = Fn.UpsampleAdd.apply(x, None, size[0], size[1])
            out = Fn.DwConv.apply(x, self.depth_wise.weight, self.depth_wise.bias, self.expansion_fact, st)
            if pe_act is not None:
                film = Fn.conv(pe_act, self.pe_proj.weight, bias=self.pe_proj.bias)
                out = Fn.Film.apply(out, film)
            out = Fn.conv(out, self.pt_wise.weight, bias=self.pt_wise.bias)
            self.norm.num_batches_tracked += 1
            out = Fn.BatchNormTrain.apply(out, self.norm.weight, self.norm.bias, self.norm.running_mean,
                                          self.norm.running_var, self.norm.eps, self.norm.momentum)
            return Fn.Silu.apply(out)
        if pe_act is None and self.stride >= 1:
            # a demand-driven FPN map read for the first time (evaluation mode): this block composed with the map's own convolution is
            # one 5x5 / stride convolution of that convolution's input -- the map's pattern pixels are never formed
            f = ondemand.rpn_composite(x, self)
            if f is not None:
                return f
        ondemand.pattern_materialize(x)              # (no-op unless x is such a map and this block could not take it)
        if self.stride < 1:
            size = ((1 / self.stride) * np.array(x.shape[1:3])).astype(np.int64).tolist()
            x = ops.upsample_bilinear_add(x, size[0], size[1])
        film = None
        if pe_act is not None:
            film = ops.conv2d(pe_act, _prep.krsc(self.pe_proj.weight), shift=self.pe_proj.bias.detach())
            film = film.view(-1, film.shape[-1])
        out = ops.dwconv3x3(x, self.depth_wise.weight.detach(), self.depth_wise.bias.detach(), self.expansion_fact,
                            st, film=film)
        s, b = _prep.bn_affine(self.norm.weight, self.norm.bias, self.norm.running_mean, self.norm.running_var,
                               self.norm.eps, conv_bias=self.pt_wise.bias)
        return ops.conv2d(out, _prep.krsc(self.pt_wise.weight), scale=s, shift=b, act=ops.ACT_SILU)


MERGE_RPN_HEADS = os.environ.get('NBM_MERGE_RPN_HEADS', '1') != '0'     # training: class and box heads of an RPN level as one GEMM


class RegionProposalNetwork(nn.Module):
    """reference layers.py:49-99."""

    def __init__(self, args, n_layers, top_layer_size):
        super().__init__()
        in_cn = args.out_fpn_chan
        self.A = args.n_ratios
        self.n_layers = n_layers
        self.top_size = tuple(top_layer_size)
        self.convs = nn.ModuleDict({
            str(i): DepthwiseSepConv2d(in_cn, in_cn, stride=(args.anchor_stride / (2 ** (i + 1))), expansion_fact=2)
            for i in range(n_layers)})
        self.avgpool = nn.AdaptiveAvgPool2d(top_layer_size)          # identity at the reference geometry
        self.cls_score = nn.ModuleDict({str(i): nn.Conv2d(in_cn, self.A * 2, 1) for i in range(n_layers)})
        self.bbox_reg = nn.ModuleDict({str(i): nn.Conv2d(in_cn, self.A * 4, 1) for i in range(n_layers)})
        self.apply(weight_init)

    def forward_nhwc(self, x):
        """x: list of NHWC FPN maps -> (cls softmaxed [B,h,w,n_layers*A*2], reg [B,h,w,n_layers*A*4], raw cls)."""
        A, nl = self.A, self.n_layers
        feats = []
        for i, fm in enumerate(x):
            f = self.convs[str(i)](fm)
            if tuple(f.shape[1:3]) == (2 * self.top_size[0], 2 * self.top_size[1]):
                # `--dilation`: the coarsest level keeps layer3's resolution, the RPN up-samples it by 2 (stride 16 / 32 = 0.5) and
                # nn.AdaptiveAvgPool2d(top_size) (layers.py:84,94) is then a real 2x2 average
                f = Fn.AvgPool2x2.apply(f) if (torch.is_grad_enabled() and f.requires_grad) else ops.avgpool2x2(f)
            if tuple(f.shape[1:3]) != self.top_size:
                raise NotImplementedError(f'RPN map {tuple(f.shape[1:3])} != top_size {self.top_size}: adaptive '
                                          'average pooling to a size other than the map itself or its half is outside the hot-path scope')
            feats.append(f)
        B, h, w, cn = feats[0].shape
        if torch.is_grad_enabled() and any(f.requires_grad for f in feats):
            cls_l, reg_l = [], []
            for i, f in enumerate(feats):
                c, r = self.cls_score[str(i)], self.bbox_reg[str(i)]
                if MERGE_RPN_HEADS:
                    # both heads of a level read the same map: ONE GEMM with the 6 + 12 output channels side by side -- forward, data
                    # gradient and weight gradient each run once instead of twice (N = 6 / 12 GEMMs sit at 8-14 TF/s: 7 ms of a step)
                    y = Fn.conv(f, torch.cat([c.weight, r.weight], 0), bias=torch.cat([c.bias, r.bias], 0))
                    cls_l.append(y[..., :A * 2])
                    reg_l.append(y[..., A * 2:])
                    continue
                cls_l.append(Fn.conv(f, c.weight, bias=c.bias))
                reg_l.append(Fn.conv(f, r.weight, bias=r.bias))
            cls_raw = torch.cat(cls_l, -1)
            reg = torch.cat(reg_l, -1)
            return Fn.PairSoftmax.apply(cls_raw, nl * A), reg, cls_raw
        cls_raw = torch.empty((B, h, w, nl * A * 2), device=feats[0].device, dtype=torch.float32)
        reg = torch.empty((B, h, w, nl * A * 4), device=feats[0].device, dtype=torch.float32)
        for i, f in enumerate(feats):
            c, r = self.cls_score[str(i)], self.bbox_reg[str(i)]
            ops.gemm_conv(f, _prep.krsc(c.weight), cls_raw[..., i * A * 2:], B=1, H=B * h * w, W=1, Cin=cn,
                          N=A * 2, y_ld=nl * A * 2, shift=c.bias.detach())
            ops.gemm_conv(f, _prep.krsc(r.weight), reg[..., i * A * 4:], B=1, H=B * h * w, W=1, Cin=cn,
                          N=A * 4, y_ld=nl * A * 4, shift=r.bias.detach())
        return ops.pair_softmax(cls_raw, nl * A), reg, cls_raw

    def forward(self, x):
        """-> (cls_scores [B, n_layers*A*2, h, w], bbox_reg [B, n_layers*A*4, h, w]) like the reference."""
        cls, reg, _ = self.forward_nhwc(x)
        return cls.permute(0, 3, 1, 2), reg.permute(0, 3, 1, 2)


class ProposalLayer(nn.Module):
    """reference layers.py:219-303: decode, clip, size filter, top-N by objectness, greedy NMS 0.7,
    batch-coupled truncation -- all on device."""

    def __init__(self, config, n_layers):
        super().__init__()
        self.n_layers = n_layers
        self.config = config
        self._anchors = {}

    def anchors(self, height, width, device):
        key = (height, width, str(device))
        if key not in self._anchors:
            cfg = self.config
            a = generate_anchors(base_size=cfg.base_size, ratios=cfg.ratios, scales=2 ** np.arange(self.n_layers))
            s = get_anchor_shifts(width, height, cfg.anchor_stride)
            self._anchors[key] = torch.from_numpy((a + s).reshape(-1, 4).astype(np.float32)).to(device)
        return self._anchors[key]

    def forward_device(self, cls_nhwc, reg_nhwc, independent=False):
        """-> (rois [B,post_n,4], scores [B,post_n], n_roi int32[1] on device); n_roi = 0 <=> "RPN failed".
        `independent`: n_roi int32 [B], the batch-coupled minima of layers.py:287 / nets_utils.py:236 taken per image."""
        cfg = self.config
        B, h, w, c2 = cls_nhwc.shape
        n_anchor = c2 // 2
        pre, post = (cfg.pre_nms_topN, cfg.post_nms_topN) if self.training else \
            (cfg.pre_nms_topN_eval, cfg.post_nms_topN_eval)
        anchors = self.anchors(h, w, cls_nhwc.device)
        boxes, keys, cnt = ops.rpn_decode(cls_nhwc, reg_nhwc, anchors, n_anchor, cfg.img_width, cfg.img_height,
                                          cfg.min_threshold)
        cap = _pow2_cap(pre)
        sb, ss, n_sel = ops.rpn_select(boxes, keys, cnt, pre, cfg.rcnn_batch_size, cap, per_image=independent)
        return ops.nms_batched(sb, ss, n_sel, cfg.nms_thresh, post)

    def forward(self, labels_pred, bbox_reg):
        rois, scores, n = self.forward_device(_nhwc(labels_pred), _nhwc(bbox_reg))
        n = int(n.item())
        if n == 0:
            print('Not enough possible RoIs, RPN failed')
            return torch.tensor([]).to(rois.device), torch.tensor([]).to(rois.device)
        return rois[:, :n].contiguous(), scores[:, :n].contiguous()


class ROIPooling(nn.Module):
    """reference layers.py:399-497."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self._pe = {}

    def pe_tables(self, device):
        key = str(device)
        if key not in self._pe:
            cfg = self.config
            self._pe[key] = (one_dimension_positional_encoding(cfg.img_height, cfg.out_fpn_chan // 2).to(device).contiguous(),
                             one_dimension_positional_encoding(cfg.img_width, cfg.out_fpn_chan // 2).to(device).contiguous())
        return self._pe[key]

    def forward_device(self, rois, n_roi, fmaps_nhwc):
        """rois [B,cap,4], n_roi device int32[1] -> pool, pe NHWC [B*cap,2,2,C], level int32 [B,cap]."""
        cfg = self.config
        if (cfg.roi_pool_h, cfg.roi_pool_w) != (2, 2):
            raise NotImplementedError('roi_pool 2x2 (reference default) only')
        pe_f, pe_t = self.pe_tables(rois.device)
        for lvl, fm in enumerate(fmaps_nhwc):        # demand-driven maps: compute the tiles under these RoIs first
            if ondemand.lazy_pending(fm):
                with torch.no_grad():
                    ondemand.lazy_complete(fm, rois.detach(), n_roi, [tuple(f.shape[1:3]) for f in fmaps_nhwc], level=lvl)
        return Fn.RoiPool.apply(rois, n_roi, pe_f, pe_t, cfg.img_height, cfg.img_width, *fmaps_nhwc)

    def forward(self, rois, conv_out):
        B, R = rois.shape[:2]
        n = torch.full((1,), R, device=rois.device, dtype=torch.int32)
        pool, pe, lvl = self.forward_device(rois.contiguous(), n, [_nhwc(f) for f in conv_out])
        C_ = pool.shape[-1]
        return (pool.view(B, R, 2, 2, C_).permute(0, 1, 4, 2, 3), pe.view(B, R, 2, 2, C_).permute(0, 1, 4, 2, 3),
                lvl.cpu().numpy())


class RCNN(nn.Module):
    """reference layers.py:500-586 (the positional-encoding / FiLM variant, the only live branch)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        cn = config.out_fpn_chan
        hidden_size = cn * config.roi_pool_h * config.roi_pool_w
        self.pe_proj = nn.Conv2d(cn, cn, 1)
        self.rcnn = nn.ModuleList([DepthwiseSepConv2d(cn, cn, pe_channels=cn) for _ in range(config.depth_rcnn)])
        self.bbox_reg_layer = nn.Linear(hidden_size, 4 * (1 + config.num_classes))
        self.bbox_classif_layer = nn.Linear(hidden_size, 1 + config.num_classes)
        self.softmax_layer = nn.Softmax(dim=-1)
        self.apply(weight_init)

    @staticmethod
    def _hwc_weight(lin, cn):
        """Linear weight over flatten([C,h,w]) -> the same weight over our (h,w,C) feature order."""
        def make():
            o = lin.weight.shape[0]
            return lin.weight.detach().view(o, cn, -1).permute(0, 2, 1).reshape(o, -1).contiguous()
        return _prep._cached(lin.weight, 'hwc', make)

    def forward_nhwc(self, pool, pe):
        """pool, pe: NHWC [N,2,2,C] -> (bbox_reg [N, 4(1+nc)], bbox_classes [N, 1+nc] softmaxed)."""
        cn = pool.shape[-1]
        if torch.is_grad_enabled() and (self.training or pool.requires_grad):
            if not self.training:
                raise NotImplementedError('gradients through the eval-mode (running-statistics) head are not implemented')
            roi_pe = Fn.conv(pe, self.pe_proj.weight, bias=self.pe_proj.bias)
            pe_act = Fn.Silu.apply(roi_pe)
            out = pool
            for blk in self.rcnn:
                out = blk(out, pe_act)
            feat = out.reshape(out.shape[0], -1)
            wr = self.bbox_reg_layer.weight.view(-1, cn, 4).permute(0, 2, 1).reshape(-1, 4 * cn)
            wc = self.bbox_classif_layer.weight.view(-1, cn, 4).permute(0, 2, 1).reshape(-1, 4 * cn)
            reg = Fn.linear(feat, wr, self.bbox_reg_layer.bias)
            cls = Fn.linear(feat, wc, self.bbox_classif_layer.bias)
            return reg, Fn.SoftmaxRows.apply(cls)
        roi_pe = ops.conv2d(pe, _prep.krsc(self.pe_proj.weight), shift=self.pe_proj.bias.detach())
        pe_act = ops.silu(roi_pe)
        out = pool
        for blk in self.rcnn:
            out = blk(out, pe_act)
        feat = out.view(out.shape[0], -1)
        reg = ops.linear(feat, self._hwc_weight(self.bbox_reg_layer, cn), self.bbox_reg_layer.bias.detach())
        cls = ops.linear(feat, self._hwc_weight(self.bbox_classif_layer, cn), self.bbox_classif_layer.bias.detach())
        return reg, ops.softmax_rows_(cls)

    def forward(self, roi_pool_out, roi_pe_out):
        """[B,R,C,2,2] x2 -> (bbox_reg [B*R, 4(1+nc)], bbox_classes [B*R, 1+nc])."""
        p = roi_pool_out.flatten(end_dim=1).permute(0, 2, 3, 1).contiguous()
        e = roi_pe_out.flatten(end_dim=1).permute(0, 2, 3, 1).contiguous()
        return self.forward_nhwc(p, e)


class _EncoderLayer(nn.Module):
    """Parameter container with the state_dict keys of both encoder flavours the reference can build: torch's
    nn.TransformerEncoderLayer (layers.py:618-621) and its DETR-style layer (self_attention.py:110-139)."""

    def __init__(self, d_model, nhead, dim_feedforward):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)


class _Encoder(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([_EncoderLayer(d_model, nhead, dim_feedforward) for _ in range(num_layers)])


class Transformer_RCNN(nn.Module):
    """reference layers.py:589-651 (`--tf_rcnn`).  Default flavour: post-norm ReLU encoder fed
    rois_embed + pos_embed with batch_first=False, i.e. the reference attends ACROSS THE IMAGES OF THE BATCH for each RoI
    slot (sequence axis = bs, <= 128 here); `tf_pe_qk`: DETR flavour, attention across the RoIs of one image with
    q = k = src + pos and LeakyReLU.  Dropout is inactive in both (eval)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        in_dim = config.out_fpn_chan * config.roi_pool_h * config.roi_pool_w
        E = config.tf_model_dim
        self.tf_pe_qk = config.tf_pe_qk
        if getattr(config, 'dropout', 0):
            raise NotImplementedError('Transformer_RCNN: dropout > 0 is not implemented')
        if not self.tf_pe_qk and E != 512:
            raise ValueError('the reference hard-codes d_model=512 for the non-pe_qk encoder (layers.py:619)')
        self.nhead = config.tf_nhead
        if E % self.nhead or E // self.nhead > 64:
            raise NotImplementedError('Transformer_RCNN: head dim must divide tf_model_dim and be <= 64')
        self.pos_embedding = nn.Sequential(nn.Linear(in_dim, E), nn.LeakyReLU())
        self.rois_embedding = nn.Sequential(nn.Linear(in_dim, E), nn.LeakyReLU())
        self.encoder = _Encoder(E, self.nhead, config.tf_dim_feedforward, config.tf_num_encoder_layers)
        self.bbox_reg_layer = nn.Linear(E, 4 * (1 + config.num_classes))
        self.bbox_classif_layer = nn.Linear(E, 1 + config.num_classes)
        self.softmax_layer = nn.Softmax(dim=-1)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward_nhwc(self, pool, pe, B, R, n_valid=None):
        """pool, pe: NHWC [B*R,2,2,C]; n_valid: device int32[1] RoIs per image that are real (pe_qk masks the rest)
        -> (bbox_reg [B*R, 4(1+nc)], bbox_classes [B*R, 1+nc] softmaxed)."""
        if torch.is_grad_enabled() and (self.training or pool.requires_grad):
            return self._forward_train(pool, pe, B, R, n_valid)
        cn, M = pool.shape[-1], B * R
        lin = lambda m: (m.weight.detach(), m.bias.detach())
        pos = ops.linear(pe.view(M, -1), RCNN._hwc_weight(self.pos_embedding[0], cn), self.pos_embedding[0].bias.detach(),
                         act=ops.ACT_LEAKY)
        x = ops.linear(pool.view(M, -1), RCNN._hwc_weight(self.rois_embedding[0], cn),
                       self.rois_embedding[0].bias.detach(), act=ops.ACT_LEAKY)
        E = x.shape[1]
        if self.tf_pe_qk:
            geom = dict(S=R, N=B, seq_stride=1, batch_stride=R, n_valid=n_valid)
            ff_act = ops.ACT_LEAKY
        else:
            x = ops.axpby(x, pos)
            geom = dict(S=B, N=R, seq_stride=R, batch_stride=1)
            ff_act = ops.ACT_RELU
        for l in self.encoder.layers:
            w, b = l.self_attn.in_proj_weight.detach(), l.self_attn.in_proj_bias.detach()
            if self.tf_pe_qk:
                qk = ops.linear(ops.axpby(x, pos), w[:2 * E], b[:2 * E])
                q, k, v = qk[:, :E], qk[:, E:], ops.linear(x, w[2 * E:], b[2 * E:])
            else:
                qkv = ops.linear(x, w, b)
                q, k, v = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]
            a = ops.mha_small(q, k, v, nhead=self.nhead, **geom)
            x = ops.layernorm(ops.linear(a, *lin(l.self_attn.out_proj), residual=x), *lin(l.norm1), eps=l.norm1.eps)
            h = ops.linear(x, *lin(l.linear1), act=ff_act)
            x = ops.layernorm(ops.linear(h, *lin(l.linear2), residual=x), *lin(l.norm2), eps=l.norm2.eps)
        reg = ops.linear(x, *lin(self.bbox_reg_layer))
        return reg, ops.softmax_rows_(ops.linear(x, *lin(self.bbox_classif_layer)))

    def _forward_train(self, pool, pe, B, R, n_valid=None):
        """Same computation on the differentiable operator layer (`functional`): every forward and backward step is a
        HIP kernel; torch.autograd only sequences them."""
        cn, M, E = pool.shape[-1], B * R, self.config.tf_model_dim
        hwc = lambda lin: lin.weight.view(-1, cn, 4).permute(0, 2, 1).reshape(-1, 4 * cn)
        pos = Fn.linear(pe.reshape(M, -1), hwc(self.pos_embedding[0]), self.pos_embedding[0].bias, act=Fn.ACT_LEAKY)
        x = Fn.linear(pool.reshape(M, -1), hwc(self.rois_embedding[0]), self.rois_embedding[0].bias, act=Fn.ACT_LEAKY)
        if self.tf_pe_qk:
            geom, ff_act = (R, B, self.nhead, 1, R, n_valid), Fn.ACT_LEAKY
        else:
            x = Fn.Add.apply(x, pos)
            geom, ff_act = (B, R, self.nhead, R, 1, None), Fn.ACT_RELU
        for l in self.encoder.layers:
            w, b = l.self_attn.in_proj_weight, l.self_attn.in_proj_bias
            if self.tf_pe_qk:
                qk = Fn.linear(Fn.Add.apply(x, pos), w[:2 * E], b[:2 * E])
                q, k, v = qk[:, :E], qk[:, E:], Fn.linear(x, w[2 * E:], b[2 * E:])
            else:
                qkv = Fn.linear(x, w, b)
                q, k, v = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]
            a = Fn.MhaSmall.apply(q, k, v, *geom)
            x = Fn.LayerNorm.apply(Fn.linear(a, l.self_attn.out_proj.weight, l.self_attn.out_proj.bias, residual=x),
                                   l.norm1.weight, l.norm1.bias, l.norm1.eps)
            h = Fn.linear(x, l.linear1.weight, l.linear1.bias, act=ff_act)
            x = Fn.LayerNorm.apply(Fn.linear(h, l.linear2.weight, l.linear2.bias, residual=x),
                                   l.norm2.weight, l.norm2.bias, l.norm2.eps)
        reg = Fn.linear(x, self.bbox_reg_layer.weight, self.bbox_reg_layer.bias)
        cls = Fn.linear(x, self.bbox_classif_layer.weight, self.bbox_classif_layer.bias)
        return reg, Fn.SoftmaxRows.apply(cls)

    def forward(self, rois, pos):
        """[B,R,C,2,2] x2 -> (bbox_reg [B*R, 4(1+nc)], bbox_classes [B*R, 1+nc])."""
        B, R = rois.shape[:2]
        p = rois.flatten(end_dim=1).permute(0, 2, 3, 1).contiguous()
        e = pos.flatten(end_dim=1).permute(0, 2, 3, 1).contiguous()
        return self.forward_nhwc(p, e, B, R)


class FastRCNN(nn.Module):
    """reference layers.py:654-778."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.roi_pooling = ROIPooling(config)
        self.rcnn = Transformer_RCNN(config) if config.tf_rcnn else RCNN(config)

    def _head(self, pool, pe, rois, n_roi):
        if self.config.tf_rcnn:
            if n_roi.numel() != 1:
                raise NotImplementedError('Transformer_RCNN with per-image RoI counts (independent detection)')
            return self.rcnn.forward_nhwc(pool, pe, rois.shape[0], rois.shape[1], n_roi)
        return self.rcnn.forward_nhwc(pool, pe)

    def detect_device(self, fmaps_nhwc, rois, n_roi, nms_thresh=0.3, min_score=0.5):
        """Sync-free eval path -> (det [B,cap,6] rows {class,x1,y1,x2,y2,score}, n_det int32 [B])."""
        cfg = self.config
        pool, pe, _ = self.roi_pooling.forward_device(rois, n_roi, fmaps_nhwc)
        reg, cls = self._head(pool, pe, rois, n_roi)
        return ops.rcnn_post(rois, n_roi, reg, cls, cfg.img_width, cfg.img_height, nms_thresh, min_score,
                             cfg.proposal_number)

    _EMPTY = {}

    @staticmethod
    def dets_to_dicts(det, n_det, num_classes):
        """One D2H copy (if the rows are still on the device), then the reference's output structure: list[B] of
        {'1'..'nc': {'bbox_coord': f32[n,4], 'scores': f32[1,n]}} with `torch.Tensor()` for empty classes
        (layers.py:749-776).  Rows arrive sorted by (class, score desc), so every class is one contiguous slice."""
        det = det.cpu() if det.is_cuda else det
        n_det = (n_det.cpu() if n_det.is_cuda else n_det).tolist()
        tmpl = FastRCNN._EMPTY.get(num_classes)
        if tmpl is None:
            empty = dict(bbox_coord=torch.Tensor(), scores=torch.Tensor())
            tmpl = FastRCNN._EMPTY[num_classes] = {str(c): empty for c in range(1, num_classes + 1)}
        cls_all = det[..., 0].to(torch.int64).numpy()
        out = []
        for b, n in enumerate(n_det):
            res = dict(tmpl)
            if n:
                rows, cls = det[b, :n], cls_all[b, :n]
                starts = np.flatnonzero(np.r_[True, cls[1:] != cls[:-1]])
                ends = np.r_[starts[1:], n]
                for s0, e0 in zip(starts.tolist(), ends.tolist()):
                    res[str(int(cls[s0]))] = dict(bbox_coord=rows[s0:e0, 1:5].clone(), scores=rows[s0:e0, 5][None].clone())
            out.append(res)
        return out

    def forward(self, conv_out, rois, nms_thresh=0.3, min_score=0.5, training=None):
        if training is None:
            training = self.training
        B, R = rois.shape[:2]
        fm = [_nhwc(f) for f in conv_out]
        rois = rois.contiguous()
        n = torch.full((1,), R, device=rois.device, dtype=torch.int32)
        if training:
            pool, pe, _ = self.roi_pooling.forward_device(rois, n, fm)
            return self._head(pool, pe, rois, n)
        det, n_det = self.detect_device(fm, rois, n, nms_thresh, min_score)
        return self.dets_to_dicts(det, n_det, self.config.num_classes)
