"""Positional encodings (reference nets/position_encoding.py).

Only `one_dimension_positional_encoding` is on the hot path (RoI positional encoding,
reference layers.py:437-438).  The sine image encoding that the reference's `Joiner` evaluates for
every level (backbone.py:139-148) is discarded by the default config (nbm_model.py:45,
`add_posenc=False`), so `build_position_encoding` returns a marker object and
nothing is computed for it unless `--add_posenc` asks for it (nbm_model.py:45-46); `learned` is rejected loudly.
"""
import torch


def one_dimension_positional_encoding(length, cn, temp=10000):
    """[length, cn] sinusoid table, position_encoding.py:10-15.  Built once on the host (tiny) and
    uploaded; the RoI kernel averages its rows."""
    pos = torch.arange(1, length + 1, dtype=torch.float32)
    dt = temp ** (2 * torch.div(torch.arange(cn, dtype=torch.float32), 2, rounding_mode='trunc') / cn)
    posenc = pos[:, None] / dt[None, :]
    return torch.stack([posenc[:, 0::2].sin(), posenc[:, 1::2].cos()], dim=2).flatten(start_dim=1)


class PositionEmbeddingSine(torch.nn.Module):
    """reference position_encoding.py:18-56 with normalize=True, only_y_scale=True (the only variant whose channel count
    matches the feature map it is added to -- the 2-D variant fails inside the reference itself).  No parameters; the
    encoding depends on the map shape only, so it is built once per (h, w, C) on the host and kept on the device as an
    NHWC table [h, w, C] that `functional.AddConst` broadcasts over the batch."""

    def __init__(self):
        super().__init__()
        self._tables = {}

    def table(self, h, w, c, device):
        key = (h, w, c, str(device))
        if key not in self._tables:
            import math
            y = torch.arange(1, h + 1, dtype=torch.float32)
            y = y / (y[-1] + 1e-6) * (2 * math.pi)
            dim_t = 10000 ** (2 * torch.div(torch.arange(c, dtype=torch.float32), 2, rounding_mode='trunc') / c)
            e = y[:, None] / dim_t
            pos = torch.stack((e[:, 0::2].sin(), e[:, 1::2].cos()), dim=2).flatten(1)               # [h, c]
            self._tables[key] = pos[:, None, :].expand(h, w, c).contiguous().to(device)
        return self._tables[key]

    def forward(self, x):
        """x NHWC [B,h,w,C] -> the [h,w,C] table on x's device."""
        return self.table(x.shape[1], x.shape[2], x.shape[3], x.device)


def build_position_encoding(args):
    if args.position_embedding in ('v3', 'learned'):
        # reference position_encoding.py:59-83 + backbone.py:139-148: Joiner evaluates the embedding for EVERY pyramid level on every
        # forward pass and PositionEmbeddingLearned indexes nn.Embedding(50, .) with arange(w) -- w = 512 ... 32 columns at the
        # reference's own 375 x 1024 input, so its first forward pass dies with an IndexError.  Nothing to be faithful to.
        raise IndexError('--position_embedding learned: the reference indexes nn.Embedding(50, .) with arange(width of every '
                         'pyramid level) (position_encoding.py:63-80, backbone.py:146) and fails on its first forward pass at any '
                         'input wider than 100 px; use the sine encoding')
    if args.position_embedding not in ('v2', 'sine'):
        raise ValueError(f'not supported {args.position_embedding}')
    if getattr(args, 'add_posenc', False) and not getattr(args, 'one_dim_posenc', True):
        raise ValueError('--add_posenc needs the one-dimensional encoding: the 2-D variant has twice the channels of the map '
                         'it is added to and fails in the reference too (position_encoding.py:52)')
    return PositionEmbeddingSine()
