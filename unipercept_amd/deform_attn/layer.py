"""``MSDeformAttn`` layer: linear projections on PyTorch-ROCm GEMMs around the HIP op.

The reference snapshot holds no such module (its ``nn/layers/deform_attn`` package is absent);
the interface follows the Deformable-DETR / Mask2Former pixel-decoder convention that the
reference's models call: ``forward(query, reference_points, input_flatten,
input_spatial_shapes, input_level_start_index, input_padding_mask=None)``.
"""

from __future__ import annotations

import math
import os

import torch
from torch import nn

from .. import _lib
from . import _ops
from ._function import (
    route_flags,
    HeadMajorValueFunction,
    MultiScaleRawPointsAttnFunction,
    MultiScaleDeformAttnFunction,
    MultiScaleFlashAttnFunction,
    SamplingPointsFunction,
)


def _split_k(tokens: int) -> int:
    """Slices for the weight-gradient GEMM of a projection over `tokens` rows (0: leave it to
    one GEMM).  The output is only d_model x d_model, so a single GEMM over hundreds of thousands
    of tokens runs on a few dozen workgroups; as a batched GEMM over slices of the token
    dimension plus a sum it fills the chip (fp32, 344 064 tokens, 256 x 256: 0.84 -> 0.33 ms on
    MI355X with 64 slices)."""
    if tokens < 32768:
        return 0
    for sk in (64, 32, 16, 8):
        if tokens % sk == 0 and tokens // sk >= 1024:
            return sk
    return 0


class _TokenLinearFunction(torch.autograd.Function):
    """``F.linear`` for (tokens, features) activations with many tokens: same forward GEMM and
    input-gradient GEMM, weight gradient by a batched GEMM over slices of the token dimension.
    Plain PyTorch-ROCm GEMMs throughout."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.in_dtypes = (x.dtype, weight.dtype, bias.dtype)
        if torch.is_autocast_enabled("cuda"):
            dt = torch.get_autocast_dtype("cuda")
            x, weight, bias = x.to(dt), weight.to(dt), bias.to(dt)
        ctx.save_for_backward(x, weight)
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, grad):
        x, weight = ctx.saved_tensors
        xd, wd, bd = ctx.in_dtypes
        g2 = grad.reshape(-1, grad.shape[-1]).to(x.dtype)
        x2 = x.reshape(-1, x.shape[-1])
        grad_x = grad_w = grad_b = None
        if ctx.needs_input_grad[0]:
            grad_x = (g2 @ weight).view_as(x).to(xd)
        if ctx.needs_input_grad[1]:
            tokens, sk = x2.shape[0], _split_k(x2.shape[0])
            if sk:
                parts = torch.bmm(
                    g2.view(sk, tokens // sk, -1).transpose(1, 2), x2.view(sk, tokens // sk, -1)
                )
                grad_w = parts.sum(0, dtype=torch.float32).to(wd)
            else:
                grad_w = (g2.t() @ x2).to(wd)
        if ctx.needs_input_grad[2]:
            grad_b = g2.sum(0, dtype=torch.float32).to(bd)
        return grad_x, grad_w, grad_b


def token_linear(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """``F.linear(x, weight, bias)``; on a ROCm device with enough tokens the weight gradient takes
    the sliced route of ``_TokenLinearFunction``."""
    tokens = x.numel() // max(x.shape[-1], 1)
    if x.is_cuda and bias is not None and torch.is_grad_enabled() and _split_k(tokens):
        return _TokenLinearFunction.apply(x, weight, bias)
    return torch.nn.functional.linear(x, weight, bias)


class MSDeformAttn(nn.Module):
    def __init__(
        self,
        d_model: int = 256,
        n_levels: int = 4,
        n_heads: int = 8,
        n_points: int = 4,
        im2col_step: int = 64,
        fused_points: bool = True,
    ):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError(f"d_model ({d_model}) must be divisible by n_heads ({n_heads})")
        self.d_model = d_model
        self.n_levels = n_levels
        self.n_heads = n_heads
        self.n_points = n_points
        self.im2col_step = im2col_step
        # One projection for offsets and logits + one HIP pass for softmax / reference-point
        # arithmetic + the packed-points kernels, where the shapes allow it (2-d reference
        # points on a ROCm device, even L*P, tuned head dim); the parameters stay the two
        # Linear modules of the canonical layer, so state dicts are unchanged.
        self.fused_points = fused_points

        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        # the sampling op, with the reference Function's positional signature
        self.op = MultiScaleDeformAttnFunction.apply
        self.reset_parameters()

    def reset_parameters(self) -> None:
        # offsets start as a ring of directions per head, growing with the point index
        nn.init.zeros_(self.sampling_offsets.weight)
        angle = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        ring = torch.stack((angle.cos(), angle.sin()), dim=-1)
        ring = ring / ring.abs().amax(dim=-1, keepdim=True)
        ring = ring.view(self.n_heads, 1, 1, 2).repeat(1, self.n_levels, self.n_points, 1)
        ring = ring * torch.arange(1, self.n_points + 1, dtype=torch.float32).view(1, 1, -1, 1)
        with torch.no_grad():
            self.sampling_offsets.bias.copy_(ring.reshape(-1))
        nn.init.zeros_(self.attention_weights.weight)
        nn.init.zeros_(self.attention_weights.bias)
        nn.init.xavier_uniform_(self.value_proj.weight)
        nn.init.zeros_(self.value_proj.bias)
        nn.init.xavier_uniform_(self.output_proj.weight)
        nn.init.zeros_(self.output_proj.bias)

    def sampling_points(self, query, reference_points, input_spatial_shapes):
        """Offsets / logits projections and reference-point arithmetic in plain PyTorch:
        returns (locations (N,Lq,M,L,P,2), weights (N,Lq,M,L,P))."""
        n, lq, _ = query.shape
        offsets = self.sampling_offsets(query).view(
            n, lq, self.n_heads, self.n_levels, self.n_points, 2
        )
        weights = self.attention_weights(query).view(n, lq, self.n_heads, self.n_levels * self.n_points)
        weights = weights.softmax(-1).view(n, lq, self.n_heads, self.n_levels, self.n_points)
        if reference_points.shape[-1] == 2:
            # (W, H) per level: offsets are in pixels of the level they sample
            wh = torch.stack((input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]), -1)
            locations = (
                reference_points[:, :, None, :, None, :]
                + offsets / wh.to(offsets.dtype)[None, None, None, :, None, :]
            )
        elif reference_points.shape[-1] == 4:
            locations = (
                reference_points[:, :, None, :, None, :2]
                + offsets / self.n_points * reference_points[:, :, None, :, None, 2:] * 0.5
            )
        else:
            raise ValueError(
                f"last dim of reference_points must be 2 or 4, got {reference_points.shape[-1]}"
            )
        return locations, weights

    def sampling_inputs(
        self,
        query: torch.Tensor,
        reference_points: torch.Tensor,
        input_flatten: torch.Tensor,
        input_spatial_shapes: torch.Tensor,
        input_padding_mask: torch.Tensor | None = None,
    ):
        """Projections and reference-point arithmetic: returns (value, locations, weights)."""
        value = self.project_value(input_flatten, input_padding_mask)
        locations, weights = self.sampling_points(query, reference_points, input_spatial_shapes)
        return value, locations, weights

    def project_value(self, input_flatten, input_padding_mask=None):
        n, s, _ = input_flatten.shape
        value = token_linear(input_flatten, self.value_proj.weight, self.value_proj.bias)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], 0.0)
        return value.view(n, s, self.n_heads, self.d_model // self.n_heads)

    def fused_projection(self):
        """Weight (M*3*L*P, C) and bias of the single projection whose output rows are, per head,
        the 2*L*P offset components followed by the L*P logits."""
        m, lp, c = self.n_heads, self.n_levels * self.n_points, self.d_model
        weight = torch.cat(
            (self.sampling_offsets.weight.view(m, 2 * lp, c), self.attention_weights.weight.view(m, lp, c)),
            dim=1,
        ).reshape(m * 3 * lp, c)
        bias = torch.cat(
            (self.sampling_offsets.bias.view(m, 2 * lp), self.attention_weights.bias.view(m, lp)), dim=1
        ).reshape(m * 3 * lp)
        return weight, bias

    def can_fuse_points(self, query, reference_points, value) -> bool:
        if not self.fused_points or self.op != MultiScaleDeformAttnFunction.apply:  # op replaced by a caller
            return False
        if not query.is_cuda or reference_points.shape[-1] != 2:
            return False
        # kernels forced through the environment (generic, atomic scatter) exist only behind the
        # unfused op, which honours them
        if route_flags() & (_lib.FLAG_GENERIC | _lib.FLAG_ATOMIC_SCATTER):
            return False
        n, lq, _ = query.shape
        width = 3 * self.n_levels * self.n_points
        # the packed locations+weights are fp32 (see forward), whatever the activations' type
        probe = torch.empty((n, lq, self.n_heads, width), dtype=torch.float32, device="meta")
        return _ops.packed_points_supported(value, probe, self.n_levels)

    def forward(
        self,
        query: torch.Tensor,
        reference_points: torch.Tensor,
        input_flatten: torch.Tensor,
        input_spatial_shapes: torch.Tensor,
        input_level_start_index: torch.Tensor,
        input_padding_mask: torch.Tensor | None = None,
    ) -> torch.Tensor:
        fuse_mask = input_padding_mask is not None and self.fused_points and input_flatten.is_cuda
        value = self.project_value(input_flatten, None if fuse_mask else input_padding_mask)
        if self.can_fuse_points(query, reference_points, value):
            n, lq, _ = query.shape
            _lib.check(
                _lib.load().msda_check_im2col_step(int(n), int(self.im2col_step)), "im2col_step check"
            )
            weight, bias = self.fused_projection()
            raw = token_linear(query, weight, bias).view(n, lq, self.n_heads, -1)
            # fp32 reference points make the packed locations and weights fp32 even when the
            # projection ran in 16 bits (autocast, .bfloat16()): a bf16 location near 1.0 has a
            # resolution of 1/256, a whole pixel on a 256-wide level
            probe = torch.empty(
                (n, lq, self.n_heads, self.n_levels, self.n_points, 2), dtype=torch.float32, device="meta"
            )
            if os.environ.get("MSDA_NO_HEAD_MAJOR") != "1" and _ops.head_major_supported(value, probe):
                # epilogue of the value projection: per-head layout + padding mask in one pass;
                # the op then gathers from that copy in both directions without making its own
                value_hm = HeadMajorValueFunction.apply(value, input_padding_mask if fuse_mask else None)
                if os.environ.get("MSDA_NO_RAW_POINTS") != "1" and _ops.raw_points_supported(
                    value_hm, raw, self.n_levels, head_major=True
                ):
                    # softmax and reference-point arithmetic inside the attention kernels: the
                    # locations and weights are never written out
                    out = MultiScaleRawPointsAttnFunction.apply(
                        value_hm, input_spatial_shapes, input_level_start_index, raw,
                        reference_points.float(),
                    )
                    return token_linear(out, self.output_proj.weight, self.output_proj.bias)
                packed = SamplingPointsFunction.apply(raw, reference_points.float(), input_spatial_shapes)
                out = MultiScaleFlashAttnFunction.apply(
                    value_hm, input_spatial_shapes, input_level_start_index, packed,
                    self.im2col_step, True,
                )
                return token_linear(out, self.output_proj.weight, self.output_proj.bias)
            if fuse_mask:
                value = value.masked_fill(input_padding_mask[..., None, None], 0.0)
            packed = SamplingPointsFunction.apply(raw, reference_points.float(), input_spatial_shapes)
            out = MultiScaleFlashAttnFunction.apply(
                value, input_spatial_shapes, input_level_start_index, packed, self.im2col_step
            )
            return token_linear(out, self.output_proj.weight, self.output_proj.bias)
        if fuse_mask:
            value = value.masked_fill(input_padding_mask[..., None, None], 0.0)
        locations, weights = self.sampling_points(query, reference_points, input_spatial_shapes)
        out = self.op(
            value, input_spatial_shapes, input_level_start_index, locations, weights,
            self.im2col_step,
        )
        return token_linear(out, self.output_proj.weight, self.output_proj.bias)
