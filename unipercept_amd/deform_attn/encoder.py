"""Pixel-decoder encoder built on ``MSDeformAttn``: the direct caller of the hot path.

The reference snapshot holds no pixel decoder (its model code that would call
``unipercept.nn.layers.deform_attn`` is absent); the stack follows the Deformable-DETR /
Mask2Former convention: per layer ``x = LN(x + MSDeformAttn(x + pos, ref, x))`` then
``x = LN(x + FFN(x))``, reference points at the pixel centres of every level, sine position
embeddings as in the reference's ``TrigonometricEmbed2d``
(sources/unipercept/nn/layers/position.py:63-83,94-122).  Everything here is PyTorch-ROCm
plumbing (hipBLASLt GEMMs, LayerNorm); the sampling itself runs in the HIP kernels.
"""

from __future__ import annotations

import math

import torch
from torch import nn

from .layer import MSDeformAttn, token_linear


def sine_position_embedding(
    height: int,
    width: int,
    num_pos_feats: int = 128,
    temperature: float = 10000.0,
    normalize: bool = True,
    scale: float | None = None,
    device=None,
    dtype: torch.dtype = torch.float32,
) -> torch.Tensor:
    """(2 * num_pos_feats, H, W) sine/cosine embedding of the pixel grid: channels
    ``[0, F)`` encode y, ``[F, 2F)`` encode x; even channels sine, odd channels cosine of
    ``coordinate / temperature ** (2 * (c // 2) / F)``; coordinates count from 1 and, when
    ``normalize``, are divided by the last coordinate (+1e-6) and multiplied by ``scale``
    (default 2*pi).  Same arithmetic as ``trigonometric_embed_2d`` of the reference
    (sources/unipercept/nn/layers/position.py:63-83)."""
    if num_pos_feats % 2:
        raise ValueError("num_pos_feats must be even")
    if scale is not None and not normalize:
        raise ValueError("scale needs normalize=True")
    scale = 2.0 * math.pi if scale is None else scale
    y = torch.arange(1, height + 1, dtype=torch.float32, device=device)
    x = torch.arange(1, width + 1, dtype=torch.float32, device=device)
    if normalize:
        y = y / (y[-1] + 1e-6) * scale
        x = x / (x[-1] + 1e-6) * scale
    c = torch.arange(num_pos_feats, dtype=torch.float32, device=device)
    dim_t = temperature ** (2.0 * torch.div(c, 2, rounding_mode="floor") / num_pos_feats)
    ang_y = y[:, None] / dim_t[None, :]  # (H, F)
    ang_x = x[:, None] / dim_t[None, :]  # (W, F)
    even = (torch.arange(num_pos_feats, device=device) % 2 == 0)[None, :]
    pos_y = torch.where(even, ang_y.sin(), ang_y.cos())  # (H, F)
    pos_x = torch.where(even, ang_x.sin(), ang_x.cos())  # (W, F)
    out = torch.cat(
        (
            pos_y.t()[:, :, None].expand(num_pos_feats, height, width),
            pos_x.t()[:, None, :].expand(num_pos_feats, height, width),
        ),
        dim=0,
    )
    return out.to(dtype)


def encoder_reference_points(
    spatial_shapes, valid_ratios: torch.Tensor | None = None, device=None, dtype=torch.float32, batch: int = 1
) -> torch.Tensor:
    """(N, S, L, 2): every position's own pixel centre, normalised by its level's valid extent,
    then expressed on every level (multiplied by that level's valid ratio).  Without
    ``valid_ratios`` (no padding) all ratios are 1 and the point is the same on all levels."""
    levels = [(int(h), int(w)) for h, w in spatial_shapes]
    if valid_ratios is None:
        valid_ratios = torch.ones(batch, len(levels), 2, device=device, dtype=dtype)
    refs = []
    for lvl, (h, w) in enumerate(levels):
        ys = torch.arange(h, device=device, dtype=dtype) + 0.5
        xs = torch.arange(w, device=device, dtype=dtype) + 0.5
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        ry = yy.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * h)
        rx = xx.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * w)
        refs.append(torch.stack((rx, ry), -1))
    ref = torch.cat(refs, 1)  # (N, S, 2)
    return ref[:, :, None, :] * valid_ratios[:, None, :, :]


class MSDeformAttnEncoderLayer(nn.Module):
    def __init__(
        self,
        d_model: int = 256,
        d_ffn: int = 1024,
        dropout: float = 0.0,
        n_levels: int = 3,
        n_heads: int = 8,
        n_points: int = 4,
    ):
        super().__init__()
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)

    def forward(
        self,
        src: torch.Tensor,
        pos: torch.Tensor | None,
        reference_points: torch.Tensor,
        spatial_shapes: torch.Tensor,
        level_start_index: torch.Tensor,
        padding_mask: torch.Tensor | None = None,
    ) -> torch.Tensor:
        query = src if pos is None else src + pos
        attn = self.self_attn(
            query, reference_points, src, spatial_shapes, level_start_index, padding_mask
        )
        src = self.norm1(src + self.dropout(attn))
        hidden = torch.relu(token_linear(src, self.linear1.weight, self.linear1.bias))
        ffn = token_linear(self.dropout(hidden), self.linear2.weight, self.linear2.bias)
        return self.norm2(src + self.dropout(ffn))


class MSDeformAttnEncoder(nn.Module):
    """``num_layers`` encoder layers over the flattened multi-level feature maps."""

    def __init__(self, num_layers: int = 6, **layer_kwargs):
        super().__init__()
        self.layers = nn.ModuleList(MSDeformAttnEncoderLayer(**layer_kwargs) for _ in range(num_layers))

    def forward(
        self,
        src: torch.Tensor,
        spatial_shapes: torch.Tensor,
        level_start_index: torch.Tensor,
        pos: torch.Tensor | None = None,
        valid_ratios: torch.Tensor | None = None,
        padding_mask: torch.Tensor | None = None,
        spatial_shapes_list=None,
    ) -> torch.Tensor:
        # the level shapes as Python ints: pass spatial_shapes_list to avoid a device read
        levels = spatial_shapes_list if spatial_shapes_list is not None else spatial_shapes.tolist()
        ref = encoder_reference_points(
            levels, valid_ratios, device=src.device, dtype=src.dtype, batch=src.shape[0]
        )
        if ref.shape[0] != src.shape[0]:
            ref = ref.expand(src.shape[0], -1, -1, -1)
        ref = ref.contiguous()
        for layer in self.layers:
            src = layer(src, pos, ref, spatial_shapes, level_start_index, padding_mask)
        return src


def flatten_levels(features, position_feats: int | None = None):
    """Flattens (N, C, H_l, W_l) maps into (N, S, C) plus the level tables, and, when
    ``position_feats`` is given, the matching sine position embeddings (N, S, 2*position_feats)."""
    shapes = [(f.shape[-2], f.shape[-1]) for f in features]
    src = torch.cat([f.flatten(2).transpose(1, 2) for f in features], dim=1)
    dev = src.device
    spatial_shapes = torch.as_tensor(shapes, dtype=torch.long, device=dev)
    level_start_index = torch.cat(
        (spatial_shapes.new_zeros((1,)), spatial_shapes.prod(1).cumsum(0)[:-1])
    )
    pos = None
    if position_feats is not None:
        pos = torch.cat(
            [
                sine_position_embedding(h, w, position_feats, device=dev, dtype=src.dtype)
                .flatten(1)
                .t()
                for h, w in shapes
            ],
            dim=0,
        )[None].expand(src.shape[0], -1, -1)
    return src, spatial_shapes, level_start_index, shapes, pos
