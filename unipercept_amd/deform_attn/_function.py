"""Autograd entry points with the reference's signatures.

``MultiScaleDeformAttnFunction.apply(value, spatial_shapes, level_start_index,
sampling_locations, attention_weights, im2col_step)`` is the call the reference's test makes
(tests/unipercept/nn/layers/test_nn_layers_deform_attn.py:51-58,143-159).
"""

from __future__ import annotations

import os

import torch
from torch.autograd.function import once_differentiable

from .. import _lib
from . import _ops

# Reproducible grad_value (MSDA_FLAG_DETERMINISTIC).  Shapes the tuned kernels bin (every benchmark
# shape) keep the binned path and swap in a tile kernel that adds every position's contributions in
# an order fixed by the data: about twice that kernel's time, any size.  Other shapes (fp64, odd
# head dims) fall back to a per-destination gather that walks all Lq*P points of a level for every
# grad_value element: fine for gradcheck-sized problems, effectively a hang at benchmark sizes,
# hence the work limit.  fp64 exists for gradcheck; below the limit it takes that path unasked, so
# that two backward passes agree bitwise (gradcheck's default nondet_tol is 0).
_DET_WORK_LIMIT = 1 << 31


def route_flags() -> int:
    """Experiment switches of the Python layer (environment): kernels to force for A/B runs."""
    flags = 0
    if os.environ.get("MSDA_FORCE_GENERIC") == "1":
        flags |= _lib.FLAG_GENERIC
    if os.environ.get("MSDA_ATOMIC_SCATTER") == "1":
        flags |= _lib.FLAG_ATOMIC_SCATTER
    if os.environ.get("MSDA_TILE_1D") == "1":
        flags |= _lib.FLAG_TILE_1D
    return flags


def deterministic_requested() -> bool:
    """MSDA_DETERMINISTIC=1, or torch.use_deterministic_algorithms(True) unless MSDA_DETERMINISTIC=0."""
    det = os.environ.get("MSDA_DETERMINISTIC")
    return det == "1" or (det != "0" and torch.are_deterministic_algorithms_enabled())


def backward_flags(
    value: torch.Tensor, sampling_locations: torch.Tensor, needs_backward: bool = True
) -> int:
    """Flags of the backward launch.  Evaluated ONCE, in the forward: the layout choices made
    there (per-head value copy, packed points) must match the kernels the backward will run.
    `needs_backward`: whether a backward can follow at all (an input requires grad) — a forward
    under no_grad never alerts about the backward's reproducibility."""
    flags = route_flags()
    lq, p = sampling_locations.shape[1], sampling_locations.shape[4]
    small = value.numel() * lq * p <= _DET_WORK_LIMIT
    binned = (
        value.is_cuda
        and not flags & _lib.FLAG_GENERIC
        and _ops.head_major_supported(value, sampling_locations)
    )
    if deterministic_requested():
        if binned or small or os.environ.get("MSDA_DETERMINISTIC") == "1":
            flags |= _lib.FLAG_DETERMINISTIC
        elif needs_backward:
            # no reproducible kernel for this shape at this size: say so the way PyTorch's own
            # non-deterministic ops do (raises unless warn_only)
            _alert_not_deterministic(
                "ms_deform_attn backward for this shape at this size (grad_value by float "
                "atomics; set MSDA_DETERMINISTIC=1 to force the slow reproducible gather)"
            )
    elif os.environ.get("MSDA_DETERMINISTIC") != "0" and value.dtype == torch.float64 and small:
        flags |= _lib.FLAG_DETERMINISTIC
    return flags


def _alert_not_deterministic(what: str) -> None:
    """Raises (or warns, under warn_only) like a PyTorch op without a deterministic kernel."""
    msg = (
        f"{what} does not have a deterministic implementation, but you set "
        "'torch.use_deterministic_algorithms(True)'."
    )
    if torch.is_deterministic_algorithms_warn_only_enabled():
        import warnings

        warnings.warn(msg + " (warn_only=True: continuing with the non-reproducible path)")
    else:
        raise RuntimeError(msg)


def forward_flags() -> int:
    return _lib.FLAG_GENERIC if os.environ.get("MSDA_FORCE_GENERIC") == "1" else 0


def fused_route_flags() -> int:
    """Backward flags of the routes that hand the kernels a per-head value copy (made by the
    caller) and packed or raw points: those routes exist only for shapes the tuned kernels bin, so
    a request for reproducible gradients keeps them and selects the reproducible tile kernel."""
    flags = _lib.FLAG_VALUE_HEAD_MAJOR | (route_flags() & _lib.FLAG_TILE_1D)
    if deterministic_requested():
        flags |= _lib.FLAG_DETERMINISTIC
    return flags


def use_head_major(value: torch.Tensor, sampling_locations: torch.Tensor, bwd_flags: int) -> bool:
    """The fast path re-lays value out per head once and gathers from that copy in both passes."""
    if os.environ.get("MSDA_NO_HEAD_MAJOR") == "1":
        return False
    if bwd_flags & (_lib.FLAG_GENERIC | _lib.FLAG_ATOMIC_SCATTER):
        return False
    # (with MSDA_FLAG_DETERMINISTIC the shapes that pass this test run the binned path too)
    return value.is_cuda and _ops.head_major_supported(value, sampling_locations)


class MultiScaleDeformAttnFunction(torch.autograd.Function):
    @staticmethod
    def forward(
        ctx,
        value: torch.Tensor,
        spatial_shapes: torch.Tensor,
        level_start_index: torch.Tensor,
        sampling_locations: torch.Tensor,
        attention_weights: torch.Tensor,
        im2col_step: int = 64,
    ) -> torch.Tensor:
        _lib.check(
            _lib.load().msda_check_im2col_step(int(value.shape[0]), int(im2col_step)),
            "im2col_step check",
        )
        ctx.im2col_step = int(im2col_step)
        fflags = forward_flags()
        bflags = backward_flags(value, sampling_locations, any(ctx.needs_input_grad))
        ctx.head_major = use_head_major(value, sampling_locations, bflags)
        if ctx.head_major:
            # one re-layout serves both passes: the copy is what gets saved for the backward
            value = _ops.ms_deform_attn_value_head_major(value)
            fflags |= _lib.FLAG_VALUE_HEAD_MAJOR
            bflags |= _lib.FLAG_VALUE_HEAD_MAJOR
        ctx.bwd_flags = bflags
        out = _ops.ms_deform_attn_fwd(
            value, spatial_shapes, level_start_index, sampling_locations, attention_weights, fflags
        )
        ctx.save_for_backward(
            value, spatial_shapes, level_start_index, sampling_locations, attention_weights
        )
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output: torch.Tensor):
        value, spatial_shapes, level_start_index, sampling_locations, attention_weights = (
            ctx.saved_tensors
        )
        grad_value, grad_loc, grad_attn = _ops.ms_deform_attn_bwd(
            grad_output, value, spatial_shapes, level_start_index, sampling_locations,
            attention_weights, ctx.bwd_flags,
        )
        return grad_value, None, None, grad_loc, grad_attn, None


class MultiScaleFlashAttnFunction(torch.autograd.Function):
    """The packed-input entry point of the reference module
    (tests/unipercept/nn/layers/test_nn_layers_deform_attn.py:105-111): locations and attention
    weights arrive as one tensor ``cat([loc.flatten(-3), attn.flatten(-2)], -1)`` of shape
    ``(N, Lq, M, L*P*3)``.  As in that test, the weights are used as given (the test passes
    already normalised weights and compares with ``reference.deform_attn``), i.e. no softmax is
    applied here.  The tuned HIP kernels read the packed tensor directly and write the gradient
    packed the same way; shapes they do not cover (odd L*P, head dims without a tuned kernel,
    fp64) are unpacked and go through the same path as ``MultiScaleDeformAttnFunction``.
    """

    @staticmethod
    def forward(
        ctx,
        value: torch.Tensor,
        spatial_shapes: torch.Tensor,
        level_start_index: torch.Tensor,
        packed: torch.Tensor,
        im2col_step: int = 64,
        value_head_major: bool = False,
    ) -> torch.Tensor:
        """``value_head_major``: ``value`` is already the per-head copy (N, M, S, D) (made by
        ``HeadMajorValueFunction``); its gradient then has that shape too."""
        levels = int(spatial_shapes.shape[0])
        if packed.dim() != 4 or packed.shape[-1] % (3 * levels) != 0:
            raise ValueError(
                f"packed locations+weights must be (N, Lq, M, L*P*3) with L = {levels}; got "
                f"{tuple(packed.shape)}"
            )
        n, lq, m, width = packed.shape
        points = width // (3 * levels)
        lp = levels * points
        _lib.check(
            _lib.load().msda_check_im2col_step(int(value.shape[0]), int(im2col_step)),
            "im2col_step check",
        )
        loc = packed[..., : 2 * lp].reshape(n, lq, m, levels, points, 2)  # a view when possible
        fflags = forward_flags()
        ctx.value_head_major = bool(value_head_major)
        if value_head_major:
            # the caller made the per-head copy: only the packed-points kernels take it
            ctx.direct, ctx.head_major = True, True
            fflags |= _lib.FLAG_VALUE_HEAD_MAJOR
            ctx.bwd_flags = fused_route_flags()
            out = _ops.ms_flash_attn_fwd(value, spatial_shapes, level_start_index, packed, fflags)
            ctx.save_for_backward(value, spatial_shapes, level_start_index, packed)
            return out
        bflags = backward_flags(value, loc, any(ctx.needs_input_grad))
        ctx.head_major = use_head_major(value, loc, bflags)
        # the kernels read the packed tensor in place; the reproducible backward does so only on
        # its binned path (the shapes that also take the per-head copy)
        ctx.direct = (
            os.environ.get("MSDA_NO_PACKED_POINTS") != "1"
            and not (fflags | bflags) & _lib.FLAG_GENERIC
            and (not bflags & _lib.FLAG_DETERMINISTIC or _ops.head_major_supported(value, loc))
            and value.is_cuda
            and _ops.packed_points_supported(value, packed, levels)
        )
        if ctx.head_major:
            value = _ops.ms_deform_attn_value_head_major(value)
            fflags |= _lib.FLAG_VALUE_HEAD_MAJOR
            bflags |= _lib.FLAG_VALUE_HEAD_MAJOR
        ctx.bwd_flags = bflags
        if ctx.direct:
            out = _ops.ms_flash_attn_fwd(value, spatial_shapes, level_start_index, packed, fflags)
            ctx.save_for_backward(value, spatial_shapes, level_start_index, packed)
            return out
        attn = packed[..., 2 * lp :].reshape(n, lq, m, levels, points)
        out = _ops.ms_deform_attn_fwd(value, spatial_shapes, level_start_index, loc, attn, fflags)
        ctx.save_for_backward(value, spatial_shapes, level_start_index, loc, attn)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output: torch.Tensor):
        if ctx.direct:
            value, spatial_shapes, level_start_index, packed = ctx.saved_tensors
            grad_value, grad_packed = _ops.ms_flash_attn_bwd(
                grad_output, value, spatial_shapes, level_start_index, packed, ctx.bwd_flags
            )
            if ctx.value_head_major:
                # grad_value is produced position-major (N,S,M,D), the layout the projection's
                # weight- and input-gradient GEMMs want; seen through the input's (N,M,S,D) shape
                grad_value = grad_value.permute(0, 2, 1, 3)
            return grad_value, None, None, grad_packed, None, None
        value, spatial_shapes, level_start_index, loc, attn = ctx.saved_tensors
        grad_value, grad_loc, grad_attn = _ops.ms_deform_attn_bwd(
            grad_output, value, spatial_shapes, level_start_index, loc, attn, ctx.bwd_flags
        )
        grad_packed = torch.cat((grad_loc.flatten(-3), grad_attn.flatten(-2)), dim=-1)
        return grad_value, None, None, grad_packed, None, None


class MultiScaleRawPointsAttnFunction(torch.autograd.Function):
    """``apply(value_head_major (N,M,S,D), spatial_shapes, level_start_index, raw (N,Lq,M,3*L*P),
    reference_points (N,Lq,L,2)) -> (N, Lq, M*D)``: the op on the output of ONE projection that
    computes sampling offsets and attention logits together (per row L*P (dx, dy), then L*P
    logits).  Softmax and ``ref + offset / (W_l, H_l)`` happen inside the attention kernels in both
    directions; ``value`` is the per-head copy made by ``HeadMajorValueFunction``.  The layer of
    the canonical pixel decoder, not an entry point of the reference's test module."""

    @staticmethod
    def forward(ctx, value_hm, spatial_shapes, level_start_index, raw, reference_points):
        flags = _lib.FLAG_VALUE_HEAD_MAJOR
        out = _ops.ms_deform_raw_attn_fwd(
            value_hm, spatial_shapes, level_start_index, raw, reference_points, flags
        )
        ctx.save_for_backward(value_hm, spatial_shapes, level_start_index, raw, reference_points)
        ctx.bwd_flags = fused_route_flags()
        ctx.ref_needs_grad = reference_points.requires_grad
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value_hm, spatial_shapes, level_start_index, raw, reference_points = ctx.saved_tensors
        grad_value, grad_raw = _ops.ms_deform_raw_attn_bwd(
            grad_output, value_hm, spatial_shapes, level_start_index, raw, reference_points,
            ctx.bwd_flags,
        )
        grad_ref = None
        if ctx.ref_needs_grad:
            # x = ref + dx / W: d ref = sum over heads and points of d x = W * d(dx)
            n, lq, m, width = raw.shape
            levels = int(spatial_shapes.shape[0])
            lp = width // 3
            wh = torch.stack((spatial_shapes[:, 1], spatial_shapes[:, 0]), -1).to(grad_raw.device, torch.float32)
            g = grad_raw[..., : 2 * lp].reshape(n, lq, m, levels, lp // levels, 2).float()
            grad_ref = (g * wh[None, None, None, :, None, :]).sum(dim=(2, 4)).to(reference_points.dtype)
        return grad_value.permute(0, 2, 1, 3), None, None, grad_raw, grad_ref


class HeadMajorValueFunction(torch.autograd.Function):
    """Epilogue of the value projection: ``(N,S,M,D) -> (N,M,S,D)`` with the padding mask applied
    in the same pass (one HIP row copy instead of ``masked_fill`` + a re-layout inside the op).
    The result goes to ``MultiScaleFlashAttnFunction(..., value_head_major=True)``.  Backward: the
    incoming gradient is the op's position-major grad_value seen through the (N,M,S,D) shape; it is
    handed on in (N,S,M,D) with the masked positions zeroed."""

    @staticmethod
    def forward(ctx, value, padding_mask=None):
        ctx.save_for_backward(padding_mask) if padding_mask is not None else None
        ctx.has_mask = padding_mask is not None
        return _ops.ms_deform_attn_value_head_major(value, padding_mask)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_hm):
        grad = grad_hm.permute(0, 2, 1, 3)  # contiguous again when it came from the op
        if ctx.has_mask:
            (mask,) = ctx.saved_tensors
            grad = grad.masked_fill(mask[..., None, None], 0.0)
        return grad.contiguous(), None


class SamplingPointsFunction(torch.autograd.Function):
    """``packed = points(raw, reference_points, spatial_shapes)``: raw (N, Lq, M, 3*L*P) holds per
    (batch, query, head) row the L*P (dx, dy) offsets and then the L*P attention logits of ONE
    fused projection; the result is the packed tensor ``MultiScaleFlashAttnFunction`` takes:
    locations ``ref + offset / (W_l, H_l)`` and softmax weights.  One HIP pass each way."""

    @staticmethod
    def forward(ctx, raw, reference_points, spatial_shapes):
        packed = _ops.ms_deform_points_fwd(raw, reference_points, spatial_shapes)
        ctx.save_for_backward(packed, spatial_shapes)
        ctx.raw_dtype = raw.dtype
        ctx.ref_needs_grad = reference_points.requires_grad
        ctx.ref_dtype = reference_points.dtype
        return packed

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_packed):
        packed, spatial_shapes = ctx.saved_tensors
        grad_raw = _ops.ms_deform_points_bwd(grad_packed, packed, spatial_shapes, ctx.raw_dtype)
        grad_ref = None
        if ctx.ref_needs_grad:
            n, lq, m, width = packed.shape
            levels = int(spatial_shapes.shape[0])
            lp = width // 3
            g = grad_packed[..., : 2 * lp].reshape(n, lq, m, levels, lp // levels, 2)
            grad_ref = g.sum(dim=(2, 4)).to(ctx.ref_dtype)
        return grad_raw, grad_ref, None


def ms_deform_attn(
    value: torch.Tensor,
    spatial_shapes: torch.Tensor,
    level_start_index: torch.Tensor,
    sampling_locations: torch.Tensor,
    attention_weights: torch.Tensor,
    im2col_step: int = 64,
) -> torch.Tensor:
    """Functional form: (N,S,M,D), (L,2), (L,), (N,Lq,M,L,P,2), (N,Lq,M,L,P) -> (N,Lq,M*D)."""
    return MultiScaleDeformAttnFunction.apply(
        value, spatial_shapes, level_start_index, sampling_locations, attention_weights, im2col_step
    )
