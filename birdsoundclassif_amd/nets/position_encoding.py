"""Positional encodings (reference nets/position_encoding.py).

Only `one_dimension_positional_encoding` is on the hot path (RoI positional encoding,
reference layers.py:437-438).  The sine image encoding that the reference's `Joiner` evaluates for
every level (backbone.py:139-148) is discarded by the default config (nbm_model.py:45,
`add_posenc=False`), so `build_position_encoding` returns a marker object and
nothing is computed for it; `--add_posenc` / `learned` are rejected loudly.
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
    """Placeholder with no parameters (keeps `Joiner`'s two-slot layout so that state_dict keys keep the
    `backbone.0.` prefix)."""

    def forward(self, x):
        raise NotImplementedError('add_posenc is not part of the accelerated path (default config discards it)')


def build_position_encoding(args):
    if getattr(args, 'add_posenc', False):
        raise NotImplementedError('--add_posenc is outside the hot-path scope (SURVEY.md §8)')
    if args.position_embedding not in ('v2', 'sine'):
        raise ValueError(f'not supported {args.position_embedding}')
    return PositionEmbeddingSine()
