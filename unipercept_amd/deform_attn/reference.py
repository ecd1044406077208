"""Pure-PyTorch evaluation of multi-scale deformable attention (``reference.deform_attn``).

Mirrors the module the reference's test imports as ``reference``
(tests/unipercept/nn/layers/test_nn_layers_deform_attn.py:6-10,40-49): per level, bilinear
``grid_sample`` with zero padding and ``align_corners=False``, then the attention-weighted sum
over levels and points.  It runs on any device PyTorch supports and is NOT used by the HIP path;
it exists because callers of the reference package expect it.
"""

from __future__ import annotations

import torch
import torch.nn.functional as F


def deform_attn(
    value: torch.Tensor,
    spatial_shapes: torch.Tensor,
    sampling_locations: torch.Tensor,
    attention_weights: torch.Tensor,
) -> torch.Tensor:
    batch, _, heads, dim = value.shape
    _, queries, _, levels, points, _ = sampling_locations.shape
    sizes = [(int(h), int(w)) for h, w in spatial_shapes.tolist()]
    out = value.new_zeros(batch * heads, dim, queries)
    start = 0
    # (N, Q, M, L, P) -> (N*M, 1, Q, L, P)
    weights = attention_weights.transpose(1, 2).reshape(batch * heads, 1, queries, levels, points)
    for level, (h, w) in enumerate(sizes):
        feat = value[:, start : start + h * w]  # (N, H*W, M, D)
        start += h * w
        feat = feat.flatten(2).transpose(1, 2).reshape(batch * heads, dim, h, w)
        grid = sampling_locations[:, :, :, level].transpose(1, 2).flatten(0, 1) * 2 - 1
        picked = F.grid_sample(feat, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
        out = out + (picked * weights[:, :, :, level]).sum(-1)
    return out.view(batch, heads * dim, queries).transpose(1, 2).contiguous()
