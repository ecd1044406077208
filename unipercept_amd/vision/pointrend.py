"""Drop-in for the point-sampling part of ``unipercept.vision.pointrend`` on MI355X.

``sample`` replaces the reference's wrapper around ``grid_sample`` with border padding
(sources/unipercept/vision/pointrend.py:25-76 of the reference) by the HIP kernels behind
``msda_point_sample_forward/_backward`` (``include/msda.h``).  ``random_points``,
``logit_uncertainty`` and ``random_points_with_importance`` (same file, :79-300) are the callers
either side of it and stay plain PyTorch; they are here so that a model's point-based mask loss
runs unchanged with the sampling step on the HIP path.  The value-bin helpers of the same module
(``bins_by_values``, ``map_in_bins``, ``random_points_with_bins``, :303-542) are host-side
PyTorch logic around ``random_points``; they are restated so that the module is whole.

There is no CPU implementation of ``sample``: tensors that are not on a ROCm device raise.
"""

from __future__ import annotations

import enum
import math
from collections.abc import Callable

import torch
from torch.autograd.function import once_differentiable

from .. import _lib

_DTYPE_CODE = {
    torch.float32: _lib.F32,
    torch.float64: _lib.F64,
    torch.bfloat16: _lib.BF16,
    torch.float16: _lib.F16,
}
_MODES = {"bilinear": _lib.SAMPLE_BILINEAR, "nearest": _lib.SAMPLE_NEAREST}


def _check(input: torch.Tensor, coords: torch.Tensor) -> None:
    if input.dim() != 4:
        raise ValueError(f"input must be (N, C, H, W), got {tuple(input.shape)}")
    if coords.dim() != 3 or coords.shape[-1] != 2 or coords.shape[0] != input.shape[0]:
        raise ValueError(
            f"point_coords must be (N, P, 2) with N = {input.shape[0]}, got {tuple(coords.shape)}"
        )
    if input.dtype not in _DTYPE_CODE:
        raise TypeError(f"unsupported input dtype {input.dtype}")
    if coords.dtype != input.dtype:
        # the same rule as grid_sample's
        raise TypeError(
            f"input and point_coords must have the same dtype, got {input.dtype} and {coords.dtype}"
        )
    if input.shape[2] == 0 or input.shape[3] == 0:
        # grid_sample's own condition: only the spatial dimensions must be non-empty
        raise ValueError(f"input must have non-empty spatial dimensions, got {tuple(input.shape)}")
    if coords.device != input.device:
        raise RuntimeError(f"input and point_coords are on {input.device} and {coords.device}")
    if input.device.type != "cuda" and not _is_empty(input, coords):  # nothing to launch when empty
        raise RuntimeError(
            "pointrend.sample runs only on a ROCm GPU (HIP kernels); got tensors on "
            f"{input.device}"
        )


def _is_empty(input: torch.Tensor, coords: torch.Tensor) -> bool:
    """No samples, channels or points: grid_sample returns an empty (N, C, P) tensor then (a
    point-based mask loss on a batch without instances), and so does this drop-in, without a
    launch."""
    return input.shape[0] == 0 or input.shape[1] == 0 or coords.shape[1] == 0


@torch.library.custom_op("unipercept::point_sample_fwd", mutates_args=())
def point_sample_fwd(input: torch.Tensor, point_coords: torch.Tensor, mode: int,
                     align_corners: bool) -> torch.Tensor:
    _check(input, point_coords)
    n, c, h, w = input.shape
    p = point_coords.shape[1]
    if _is_empty(input, point_coords):
        return input.new_empty((n, c, p))
    lib = _lib.load()
    input = input.contiguous()
    coords = point_coords.contiguous()
    out = torch.empty((n, c, p), dtype=input.dtype, device=input.device)
    with torch.cuda.device_of(input):
        rc = lib.msda_point_sample_forward(
            input.data_ptr(), coords.data_ptr(), out.data_ptr(), n, c, h, w, p,
            _DTYPE_CODE[input.dtype], mode, int(align_corners),
            torch.cuda.current_stream(input.device).cuda_stream,
        )
    _lib.check(rc, "msda_point_sample_forward")
    return out


@point_sample_fwd.register_fake
def _(input, point_coords, mode, align_corners):
    return input.new_empty((input.shape[0], input.shape[1], point_coords.shape[1]))


@torch.library.custom_op("unipercept::point_sample_bwd", mutates_args=())
def point_sample_bwd(grad_out: torch.Tensor, input: torch.Tensor, point_coords: torch.Tensor,
                     mode: int, align_corners: bool) -> tuple[torch.Tensor, torch.Tensor]:
    _check(input, point_coords)
    if _is_empty(input, point_coords):
        return torch.zeros_like(input), torch.zeros_like(point_coords)
    lib = _lib.load()
    dtype = input.dtype
    # the kernels accumulate in the tensors' own type: 16-bit inputs go through fp32
    work = torch.float32 if dtype in (torch.bfloat16, torch.float16) else dtype
    input_w = input.to(work).contiguous()
    coords_w = point_coords.to(work).contiguous()
    grad_out_w = grad_out.to(work).contiguous()
    n, c, h, w = input_w.shape
    p = coords_w.shape[1]
    if tuple(grad_out_w.shape) != (n, c, p):
        raise ValueError(f"grad_out must be {(n, c, p)}, got {tuple(grad_out_w.shape)}")
    grad_input = torch.empty_like(input_w)
    grad_coords = torch.empty_like(coords_w)
    with torch.cuda.device_of(input):
        nbytes = lib.msda_point_sample_workspace_bytes(n, c, h, w, p, _DTYPE_CODE[work])
        workspace = torch.empty(nbytes, dtype=torch.uint8, device=input.device) if nbytes else None
        rc = lib.msda_point_sample_backward(
            grad_out_w.data_ptr(), input_w.data_ptr(), coords_w.data_ptr(), grad_input.data_ptr(),
            grad_coords.data_ptr(), workspace.data_ptr() if nbytes else None, nbytes,
            n, c, h, w, p, _DTYPE_CODE[work], mode, int(align_corners),
            torch.cuda.current_stream(input.device).cuda_stream,
        )
    _lib.check(rc, "msda_point_sample_backward")
    return grad_input.to(dtype), grad_coords.to(dtype)


@point_sample_bwd.register_fake
def _(grad_out, input, point_coords, mode, align_corners):
    return torch.empty_like(input), torch.empty_like(point_coords)


class PointSampleFunction(torch.autograd.Function):
    """``apply(input (N,C,H,W), point_coords (N,P,2), mode, align_corners) -> (N,C,P)``."""

    @staticmethod
    def forward(ctx, input, point_coords, mode: int, align_corners: bool):
        ctx.save_for_backward(input, point_coords)
        ctx.mode, ctx.align_corners = mode, align_corners
        return point_sample_fwd(input, point_coords, mode, align_corners)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        input, point_coords = ctx.saved_tensors
        grad_input, grad_coords = point_sample_bwd(
            grad_out, input, point_coords, ctx.mode, ctx.align_corners
        )
        return (
            grad_input if ctx.needs_input_grad[0] else None,
            grad_coords if ctx.needs_input_grad[1] else None,
            None,
            None,
        )


def sample(input: torch.Tensor, point_coords: torch.Tensor, align_corners: bool = False,
           **kwargs) -> torch.Tensor:
    """Samples ``input`` (N, C, H, W) at ``point_coords`` in the unit square.

    ``point_coords`` is (N, P, 2) -> result (N, C, P), or (N, Hgrid, Wgrid, 2) -> result
    (N, C, Hgrid, Wgrid); the last dimension is (x, y).  Values outside the map are those of its
    border.  Keyword arguments as for ``grid_sample``: ``mode`` = "bilinear" (default) or
    "nearest"; the padding mode is fixed (border), as in the reference
    (sources/unipercept/vision/pointrend.py:67-73).
    """
    mode = kwargs.pop("mode", "bilinear")
    if kwargs:
        raise TypeError(f"unsupported arguments for pointrend.sample: {sorted(kwargs)}")
    if mode not in _MODES:
        raise NotImplementedError(f"pointrend.sample: mode {mode!r} (supported: {sorted(_MODES)})")
    if input.dim() != 4:
        raise ValueError(f"input must be (N, C, H, W), got {tuple(input.shape)}")
    if point_coords.shape[-1:] != (2,) or point_coords.dim() not in (3, 4):
        raise ValueError(f"point_coords must be (N, P, 2) or (N, Hg, Wg, 2), got {tuple(point_coords.shape)}")
    grid_shape = None
    if point_coords.dim() == 4:
        grid_shape = point_coords.shape[1:3]
        point_coords = point_coords.flatten(1, 2)
    out = PointSampleFunction.apply(input, point_coords, _MODES[mode], bool(align_corners))
    if grid_shape is not None:
        out = out.unflatten(2, grid_shape)
    return out


@torch.no_grad()
def logit_uncertainty(logits: torch.Tensor) -> torch.Tensor:
    """PointRend's uncertainty of a logit map: the closer to 0, the less certain (reference:
    sources/unipercept/vision/pointrend.py:79-105)."""
    return logits.detach().abs().neg()


def random_points(source: torch.Tensor, n_points: int, d_coord: int = 2,
                  mask: torch.Tensor | None = None) -> torch.Tensor:
    """Uniform points (N, n_points, d_coord) in the unit cube; with a boolean ``mask``
    (N, 1, H, W) every point falls inside a pixel where the mask is set (reference:
    sources/unipercept/vision/pointrend.py:108-202: a masked pixel is drawn through the
    cumulative sum of the flattened mask, the point is jittered uniformly inside it)."""
    if source.dim() != 2 + d_coord:
        raise ValueError(f"source must have {2 + d_coord} dimensions, got {tuple(source.shape)}")
    n = source.shape[0]
    jitter = torch.rand(n, n_points, d_coord, dtype=torch.float32, device=source.device)
    if mask is None:
        return jitter
    if d_coord != 2:
        raise NotImplementedError("Mask constraints are only supported for 2D coordinates.")
    h, w = mask.shape[-2:]
    running = mask.reshape(n, -1).cumsum(dim=1)
    draws = torch.rand(n, n_points, device=source.device) * running[:, -1:]
    flat = torch.searchsorted(running, draws)
    col = (flat % w).float()
    row = torch.div(flat, w, rounding_mode="trunc").float()
    size = torch.tensor([w, h], dtype=torch.float32, device=source.device)
    pixel_centre = (torch.stack((col, row), dim=-1) + 0.5) / size
    return pixel_centre + (jitter - 0.5) / size


def random_points_with_importance(
    source: torch.Tensor,
    n_points: int,
    d_coord: int = 2,
    mask: torch.Tensor | None = None,
    oversample_ratio: float = 3,
    importance_fn: Callable[[torch.Tensor], torch.Tensor] = logit_uncertainty,
    importance_sample_ratio: float = 0.75,
    **kwargs,
) -> torch.Tensor:
    """PointRend's importance sampling (reference: sources/unipercept/vision/pointrend.py:205-300):
    draw ``oversample_ratio * n_points`` candidates, sample ``source`` there (on the HIP path),
    keep the ``importance_sample_ratio * n_points`` most important ones and fill up with fresh
    uniform points."""
    if oversample_ratio < 1 or not 0 <= importance_sample_ratio <= 1:
        raise ValueError("need oversample_ratio >= 1 and 0 <= importance_sample_ratio <= 1")
    n = source.shape[0]
    n_candidates = int(n_points * oversample_ratio)
    candidates = random_points(source, n_candidates, d_coord, mask)
    importance = importance_fn(sample(source, candidates, **kwargs))
    n_important = int(importance_sample_ratio * n_points)
    top = torch.topk(importance[:, 0, :], k=n_important, dim=1).indices
    chosen = torch.gather(candidates, 1, top.unsqueeze(-1).expand(n, n_important, d_coord))
    n_random = n_points - n_important
    if n_random > 0:
        chosen = torch.cat((chosen, random_points(source, n_random, d_coord, mask)), dim=1)
    return chosen


class BinsMode(str, enum.Enum):
    """How ``bins_by_values`` places its boundaries (reference:
    sources/unipercept/vision/pointrend.py:303-309)."""

    QUANTILE = "quantile"
    LINEAR = "linear"

    def __str__(self) -> str:  # behaves like the reference's StrEnum in format strings
        return self.value


@torch.no_grad()
def bins_by_values(source: torch.Tensor, n_bins: int, mask: torch.Tensor | None = None,
                   use_batch: bool = False, mode: BinsMode | str = BinsMode.QUANTILE,
                   eps: float = 1e-6) -> torch.Tensor:
    """Upper boundaries (B, n_bins) of ``n_bins`` value bins of ``source`` (B, C, ...): equally
    populated ("quantile") or equally wide ("linear"), over the whole batch or per sample
    (reference: sources/unipercept/vision/pointrend.py:312-391).  ``mask`` selects the values
    that count; without it NaNs are left out."""
    if source.dim() < 2:
        raise ValueError(f"source must be (B, C, ...), got {tuple(source.shape)}")
    n_batch = source.shape[0]
    if use_batch:
        flat = source.flatten(1)
        flat_mask = None if mask is None else mask.flatten(1)  # same cut as the values
        rows = [
            bins_by_values(flat[i][None], n_bins, None if flat_mask is None else flat_mask[i][None],
                           False, mode, eps)
            for i in range(n_batch)
        ]
        return torch.cat(rows, dim=0)
    values = source[mask] if mask is not None else source[~source.isnan()]
    mode = BinsMode(mode)
    if mode is BinsMode.QUANTILE:
        q = torch.linspace(0, 1, n_bins + 1, device=values.device)[1:]
        edges = torch.quantile(values, q) + eps
    else:
        edges = torch.linspace(float(values.min()), float(values.max()) + eps, n_bins + 1,
                               device=values.device)[1:]
    return edges[None].repeat(n_batch, 1)


def map_in_bins(source: torch.Tensor, bins: torch.Tensor,
                fn: Callable[[torch.Tensor, torch.Tensor], torch.Tensor],
                mask: torch.Tensor | None = None, values: torch.Tensor | None = None,
                **kwargs) -> torch.Tensor:
    """Calls ``fn(source, in_bin_mask)`` for every non-empty bin of ``values`` (default:
    ``source``) and concatenates the results along dim 1 (reference:
    sources/unipercept/vision/pointrend.py:394-462)."""
    if source.dim() < 2 or bins.dim() not in (1, 2):
        raise ValueError("expected source (B, C, ...) and bins (B, N) (or (N,) for one sample)")
    if values is None:
        values = source
    which = torch.searchsorted(bins, values, right=True)
    outputs = []
    for i in range(bins.shape[-1]):
        in_bin = which == i
        if mask is not None:
            in_bin = in_bin & mask
        if in_bin.any():
            outputs.append(fn(source, in_bin))
    return torch.cat(outputs, dim=1)


def random_points_with_bins(source: torch.Tensor, n_points: int, d_coord: int = 2,
                            mask: torch.Tensor | None = None, n_bins: int = 10,
                            use_batch: bool = False, mode: BinsMode | str = BinsMode.QUANTILE,
                            values: torch.Tensor | None = None) -> torch.Tensor:
    """Points (N, n_points, d_coord) drawn evenly over the value bins of every sample: the
    values are binned, every non-empty bin contributes the same number of points inside its own
    pixels, and the list is cut to ``n_points`` (reference:
    sources/unipercept/vision/pointrend.py:465-542)."""
    if values is None:
        values = source
    if mask is None:
        mask = torch.ones_like(values, dtype=torch.bool)
    bins = bins_by_values(source, n_bins, mask=mask, mode=mode, use_batch=use_batch)
    points = []
    for s, m, b, v in zip(source, mask, bins, values, strict=True):
        # s, m, v: one sample (C, ...); b: its (n_bins,) boundaries
        non_empty = int(map_in_bins(s, b, lambda _, in_bin: in_bin.any()[None, None], mask=m, values=v).sum())
        per_bin = math.ceil(n_points / max(non_empty, 1))
        chosen = map_in_bins(
            s, b, lambda src, in_bin: random_points(src[None], per_bin, d_coord, mask=in_bin[None]),
            mask=m, values=v,
        )
        points.append(chosen[0, :n_points])
    return torch.stack(points, dim=0)
