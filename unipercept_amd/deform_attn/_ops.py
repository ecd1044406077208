"""PyTorch-ROCm custom-op layer over the C ABI (``include/msda.h``).

Registers ``unipercept::ms_deform_attn_fwd`` / ``unipercept::ms_deform_attn_bwd`` with
``torch.library`` (the reference's custom-op idiom: sources/unipercept/vision/inpaint.py:94-140),
so autograd, DDP and ``torch.compile`` see two opaque ops.  Both run only on a ROCm device
through ``libmsda.so``; there is no CPU implementation.
"""

from __future__ import annotations

import ctypes
import os

import torch

from .. import _lib

_DTYPE_CODE = {
    torch.float32: _lib.F32,
    torch.float64: _lib.F64,
    torch.bfloat16: _lib.BF16,
    torch.float16: _lib.F16,
}

# device level tables already checked: key -> (shapes tensor, lsi tensor, host copy).  The
# tensors are kept alive so that their addresses cannot be recycled while the entry exists.
_level_cache: dict[tuple, tuple] = {}


def _dims(value, loc, head_major: bool = False):
    if head_major:
        n, m, s, d = value.shape
    else:
        n, s, m, d = value.shape
    _, lq, _, l, p, _ = loc.shape
    return n, s, m, d, lq, l, p


def _checked_host_levels(hw, st, s: int):
    run = 0
    for (h, w), start in zip(hw, st):
        if h < 1 or w < 1:
            raise ValueError(f"spatial_shapes must be positive, got {hw}")
        if start != run:
            raise ValueError(
                f"level_start_index {st} is not the exclusive prefix sum of H*W of spatial_shapes {hw}"
            )
        run += h * w
    if run != s:
        raise ValueError(f"value has {s} positions but spatial_shapes {hw} cover {run}")
    flat = [int(x) for pair in hw for x in pair]
    return (ctypes.c_int64 * len(flat))(*flat)


def _check_levels(shapes_in, lsi_in, shapes: torch.Tensor, lsi: torch.Tensor, s: int):
    """Host-side consistency check of the level table.  Returns a host copy of the (H, W) pairs
    (a ctypes int64 array; the backward sizes its 2-D scatter tiles from it) or None when the
    table cannot be read without a forbidden sync.  A table handed in on the host is read
    directly; a device table costs one sync the first time and is cached afterwards."""
    if os.environ.get("MSDA_SKIP_LEVEL_CHECK") == "1":
        return None
    if shapes_in.device.type == "cpu" and lsi_in.device.type == "cpu":
        return _checked_host_levels(shapes_in.tolist(), lsi_in.tolist(), s)
    key = (shapes.data_ptr(), shapes._version, lsi.data_ptr(), lsi._version, s, shapes.device)
    hit = _level_cache.get(key)
    if hit is not None:
        return hit[2]
    if shapes.is_cuda and torch.cuda.is_current_stream_capturing():
        return None  # cannot synchronise while a graph is being captured; the kernels guard themselves
    host = _checked_host_levels(shapes.tolist(), lsi.tolist(), s)
    if len(_level_cache) > 256:
        _level_cache.clear()
    _level_cache[key] = (shapes, lsi, host)
    return host


def _validate(value, shapes, lsi, loc, attn, head_major: bool = False):
    if value.dim() != 4 or loc.dim() != 6 or attn.dim() != 5 or loc.shape[-1] != 2:
        raise ValueError(
            "expected value (N,S,M,D), sampling_locations (N,Lq,M,L,P,2), attention_weights "
            f"(N,Lq,M,L,P); got {tuple(value.shape)}, {tuple(loc.shape)}, {tuple(attn.shape)}"
        )
    n, s, m, d, lq, l, p = _dims(value, loc, head_major)
    if loc.shape[0] != n or loc.shape[2] != m or tuple(attn.shape) != (n, lq, m, l, p):
        raise ValueError(
            f"inconsistent shapes: value {tuple(value.shape)}, sampling_locations "
            f"{tuple(loc.shape)}, attention_weights {tuple(attn.shape)}"
        )
    if tuple(shapes.shape) != (l, 2) or tuple(lsi.shape) != (l,):
        raise ValueError(
            f"spatial_shapes must be ({l}, 2) and level_start_index ({l},); got "
            f"{tuple(shapes.shape)} and {tuple(lsi.shape)}"
        )
    if value.dtype not in _DTYPE_CODE:
        raise TypeError(f"unsupported value dtype {value.dtype}")
    if value.device.type != "cuda":
        raise RuntimeError(
            "ms_deform_attn runs only on a ROCm GPU (HIP kernels); got a tensor on "
            f"{value.device}. Use `reference.deform_attn` for a pure-PyTorch CPU evaluation."
        )
    for t in (loc, attn):
        if t.device != value.device:
            raise RuntimeError("all tensors must live on the same device")


def _prepare(value, shapes, lsi, loc, attn, head_major: bool = False):
    """Returns contiguous, dtype-consistent tensors on value's device and the host copy of the
    level shapes (or None)."""
    _validate(value, shapes, lsi, loc, attn, head_major)
    dev = value.device
    shapes_in, lsi_in = shapes, lsi
    shapes = shapes.to(device=dev, dtype=torch.int64).contiguous()
    lsi = lsi.to(device=dev, dtype=torch.int64).contiguous()
    vdt = value.dtype
    # supported: loc/attn in value's dtype, or fp32 next to a 16-bit value
    if loc.dtype == torch.float32 and vdt in (torch.bfloat16, torch.float16):
        ldt = torch.float32
    else:
        ldt = vdt
    loc = loc.to(ldt).contiguous()
    attn = attn.to(ldt).contiguous()
    value = value.contiguous()
    levels_host = _check_levels(shapes_in, lsi_in, shapes, lsi, value.shape[2 if head_major else 1])
    return value, shapes, lsi, loc, attn, levels_host


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


@torch.library.custom_op("unipercept::ms_deform_attn_fwd", mutates_args=())
def ms_deform_attn_fwd(
    value: torch.Tensor,
    spatial_shapes: torch.Tensor,
    level_start_index: torch.Tensor,
    sampling_locations: torch.Tensor,
    attention_weights: torch.Tensor,
    flags: int,
) -> torch.Tensor:
    lib = _lib.load()
    hm = bool(flags & _lib.FLAG_VALUE_HEAD_MAJOR)
    value, shapes, lsi, loc, attn, _ = _prepare(
        value, spatial_shapes, level_start_index, sampling_locations, attention_weights, hm
    )
    n, s, m, d, lq, l, p = _dims(value, loc, hm)
    out = torch.empty((n, lq, m * d), dtype=value.dtype, device=value.device)
    with torch.cuda.device_of(value):
        rc = lib.msda_forward(
            value.data_ptr(), shapes.data_ptr(), lsi.data_ptr(), loc.data_ptr(), attn.data_ptr(),
            None, out.data_ptr(), n, s, m, d, lq, l, p, _DTYPE_CODE[value.dtype], _DTYPE_CODE[loc.dtype],
            flags, _stream_ptr(value.device),
        )
    _lib.check(rc, "msda_forward")
    return out


@ms_deform_attn_fwd.register_fake
def _(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, flags):
    n, d = value.shape[0], value.shape[3]
    m = sampling_locations.shape[2]
    return value.new_empty((n, sampling_locations.shape[1], m * d))


@torch.library.custom_op("unipercept::ms_deform_attn_bwd", mutates_args=())
def ms_deform_attn_bwd(
    grad_output: torch.Tensor,
    value: torch.Tensor,
    spatial_shapes: torch.Tensor,
    level_start_index: torch.Tensor,
    sampling_locations: torch.Tensor,
    attention_weights: torch.Tensor,
    flags: int,
) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    lib = _lib.load()
    loc_dtype, attn_dtype = sampling_locations.dtype, attention_weights.dtype
    hm = bool(flags & _lib.FLAG_VALUE_HEAD_MAJOR)
    value, shapes, lsi, loc, attn, levels_host = _prepare(
        value, spatial_shapes, level_start_index, sampling_locations, attention_weights, hm
    )
    n, s, m, d, lq, l, p = _dims(value, loc, hm)
    grad_output = grad_output.to(value.dtype).contiguous()
    if tuple(grad_output.shape) != (n, lq, m * d):
        raise ValueError(f"grad_output must be {(n, lq, m * d)}, got {tuple(grad_output.shape)}")
    grad_value = torch.empty((n, s, m, d), dtype=value.dtype, device=value.device)
    grad_loc = torch.empty_like(loc)
    grad_attn = torch.empty_like(attn)
    vcode, lcode = _DTYPE_CODE[value.dtype], _DTYPE_CODE[loc.dtype]
    with torch.cuda.device_of(value):
        nbytes = lib.msda_backward_workspace_bytes(
            n, s, m, d, lq, l, p, vcode, lcode, flags, levels_host
        )
        workspace = (
            torch.empty(nbytes, dtype=torch.uint8, device=value.device) if nbytes else None
        )
        rc = lib.msda_backward(
            grad_output.data_ptr(), value.data_ptr(), shapes.data_ptr(), lsi.data_ptr(),
            loc.data_ptr(), attn.data_ptr(), None, grad_value.data_ptr(), grad_loc.data_ptr(),
            grad_attn.data_ptr(), workspace.data_ptr() if nbytes else None, nbytes,
            n, s, m, d, lq, l, p, vcode, lcode, flags, levels_host, _stream_ptr(value.device),
        )
    _lib.check(rc, "msda_backward")
    return grad_value, grad_loc.to(loc_dtype), grad_attn.to(attn_dtype)


@ms_deform_attn_bwd.register_fake
def _(grad_output, value, spatial_shapes, level_start_index, sampling_locations, attention_weights, flags):
    gv = value.transpose(1, 2) if flags & _lib.FLAG_VALUE_HEAD_MAJOR else value
    return (
        torch.empty_like(gv, memory_format=torch.contiguous_format),
        torch.empty_like(sampling_locations),
        torch.empty_like(attention_weights),
    )


def _packed_dims(value, packed, levels: int, head_major: bool):
    if head_major:
        n, m, s, d = value.shape
    else:
        n, s, m, d = value.shape
    lq = packed.shape[1]
    p = packed.shape[3] // (3 * levels)
    return n, s, m, d, lq, levels, p


def _prepare_packed(value, shapes, lsi, packed, head_major: bool):
    """Packed-points counterpart of `_prepare`: packed is (N, Lq, M, 3*L*P), per row the L*P
    (x, y) pairs followed by the L*P weights."""
    levels = int(shapes.shape[0])
    if value.dim() != 4 or packed.dim() != 4 or packed.shape[-1] % (3 * levels) != 0:
        raise ValueError(
            f"expected value (N,S,M,D) and packed locations+weights (N,Lq,M,L*P*3) with L = "
            f"{levels}; got {tuple(value.shape)}, {tuple(packed.shape)}"
        )
    n, s, m, d, lq, l, p = _packed_dims(value, packed, levels, head_major)
    if packed.shape[0] != n or packed.shape[2] != m or p < 1:
        raise ValueError(f"inconsistent shapes: value {tuple(value.shape)}, packed {tuple(packed.shape)}")
    if tuple(shapes.shape) != (l, 2) or tuple(lsi.shape) != (l,):
        raise ValueError(f"spatial_shapes must be ({l}, 2) and level_start_index ({l},)")
    if value.dtype not in _DTYPE_CODE:
        raise TypeError(f"unsupported value dtype {value.dtype}")
    if value.device.type != "cuda" or packed.device != value.device:
        raise RuntimeError("ms_flash_attn runs only on a ROCm GPU, with all tensors on one device")
    dev = value.device
    shapes_in, lsi_in = shapes, lsi
    shapes = shapes.to(device=dev, dtype=torch.int64).contiguous()
    lsi = lsi.to(device=dev, dtype=torch.int64).contiguous()
    vdt = value.dtype
    ldt = torch.float32 if packed.dtype == torch.float32 and vdt in (torch.bfloat16, torch.float16) else vdt
    packed = packed.to(ldt).contiguous()
    value = value.contiguous()
    levels_host = _check_levels(shapes_in, lsi_in, shapes, lsi, s)
    return value, shapes, lsi, packed, levels_host


@torch.library.custom_op("unipercept::ms_flash_attn_fwd", mutates_args=())
def ms_flash_attn_fwd(
    value: torch.Tensor,
    spatial_shapes: torch.Tensor,
    level_start_index: torch.Tensor,
    packed: torch.Tensor,
    flags: int,
) -> torch.Tensor:
    """Forward on the packed locations+weights tensor itself (no unpacking pass)."""
    lib = _lib.load()
    hm = bool(flags & _lib.FLAG_VALUE_HEAD_MAJOR)
    value, shapes, lsi, packed, _ = _prepare_packed(
        value, spatial_shapes, level_start_index, packed, hm
    )
    n, s, m, d, lq, l, p = _packed_dims(value, packed, int(shapes.shape[0]), hm)
    out = torch.empty((n, lq, m * d), dtype=value.dtype, device=value.device)
    with torch.cuda.device_of(value):
        rc = lib.msda_forward(
            value.data_ptr(), shapes.data_ptr(), lsi.data_ptr(), packed.data_ptr(), None, None,
            out.data_ptr(), n, s, m, d, lq, l, p, _DTYPE_CODE[value.dtype],
            _DTYPE_CODE[packed.dtype], flags | _lib.FLAG_PACKED_POINTS, _stream_ptr(value.device),
        )
    _lib.check(rc, "msda_forward (packed points)")
    return out


@ms_flash_attn_fwd.register_fake
def _(value, spatial_shapes, level_start_index, packed, flags):
    d = value.shape[3]
    return value.new_empty((packed.shape[0], packed.shape[1], packed.shape[2] * d))


@torch.library.custom_op("unipercept::ms_flash_attn_bwd", mutates_args=())
def ms_flash_attn_bwd(
    grad_output: torch.Tensor,
    value: torch.Tensor,
    spatial_shapes: torch.Tensor,
    level_start_index: torch.Tensor,
    packed: torch.Tensor,
    flags: int,
) -> tuple[torch.Tensor, torch.Tensor]:
    """Backward on the packed tensor: returns grad_value (N,S,M,D) and the packed gradient."""
    lib = _lib.load()
    packed_dtype = packed.dtype
    hm = bool(flags & _lib.FLAG_VALUE_HEAD_MAJOR)
    value, shapes, lsi, packed, levels_host = _prepare_packed(
        value, spatial_shapes, level_start_index, packed, hm
    )
    n, s, m, d, lq, l, p = _packed_dims(value, packed, int(shapes.shape[0]), hm)
    grad_output = grad_output.to(value.dtype).contiguous()
    if tuple(grad_output.shape) != (n, lq, m * d):
        raise ValueError(f"grad_output must be {(n, lq, m * d)}, got {tuple(grad_output.shape)}")
    grad_value = torch.empty((n, s, m, d), dtype=value.dtype, device=value.device)
    grad_packed = torch.empty_like(packed)
    vcode, lcode = _DTYPE_CODE[value.dtype], _DTYPE_CODE[packed.dtype]
    flags |= _lib.FLAG_PACKED_POINTS
    with torch.cuda.device_of(value):
        nbytes = lib.msda_backward_workspace_bytes(
            n, s, m, d, lq, l, p, vcode, lcode, flags, levels_host
        )
        workspace = (
            torch.empty(nbytes, dtype=torch.uint8, device=value.device) if nbytes else None
        )
        rc = lib.msda_backward(
            grad_output.data_ptr(), value.data_ptr(), shapes.data_ptr(), lsi.data_ptr(),
            packed.data_ptr(), None, None, grad_value.data_ptr(), grad_packed.data_ptr(), None,
            workspace.data_ptr() if nbytes else None, nbytes,
            n, s, m, d, lq, l, p, vcode, lcode, flags, levels_host, _stream_ptr(value.device),
        )
    _lib.check(rc, "msda_backward (packed points)")
    return grad_value, grad_packed.to(packed_dtype)


@ms_flash_attn_bwd.register_fake
def _(grad_output, value, spatial_shapes, level_start_index, packed, flags):
    gv = value.transpose(1, 2) if flags & _lib.FLAG_VALUE_HEAD_MAJOR else value
    return torch.empty_like(gv, memory_format=torch.contiguous_format), torch.empty_like(packed)


def _prepare_raw(value, shapes, lsi, raw, reference_points, head_major: bool):
    """Raw-points counterpart of `_prepare_packed`: raw (N, Lq, M, 3*L*P) holds offsets and
    logits, reference_points (N, Lq, L, 2) the reference point per level (handed on in fp32)."""
    value, shapes, lsi, raw, levels_host = _prepare_packed(value, shapes, lsi, raw, head_major)
    n, lq, levels = raw.shape[0], raw.shape[1], int(shapes.shape[0])
    if tuple(reference_points.shape) != (n, lq, levels, 2):
        raise ValueError(
            f"reference_points must be {(n, lq, levels, 2)}, got {tuple(reference_points.shape)}"
        )
    ref = reference_points.to(device=value.device, dtype=torch.float32).contiguous()
    return value, shapes, lsi, raw, ref, levels_host


@torch.library.custom_op("unipercept::ms_deform_raw_attn_fwd", mutates_args=())
def ms_deform_raw_attn_fwd(
    value: torch.Tensor,
    spatial_shapes: torch.Tensor,
    level_start_index: torch.Tensor,
    raw: torch.Tensor,
    reference_points: torch.Tensor,
    flags: int,
) -> torch.Tensor:
    """Forward on raw points: offsets + logits as the projection left them and the reference
    points; the kernels do the softmax and the location arithmetic as they read them."""
    lib = _lib.load()
    hm = bool(flags & _lib.FLAG_VALUE_HEAD_MAJOR)
    value, shapes, lsi, raw, ref, _ = _prepare_raw(
        value, spatial_shapes, level_start_index, raw, reference_points, hm
    )
    n, s, m, d, lq, l, p = _packed_dims(value, raw, int(shapes.shape[0]), hm)
    out = torch.empty((n, lq, m * d), dtype=value.dtype, device=value.device)
    with torch.cuda.device_of(value):
        rc = lib.msda_forward(
            value.data_ptr(), shapes.data_ptr(), lsi.data_ptr(), raw.data_ptr(), None,
            ref.data_ptr(), out.data_ptr(), n, s, m, d, lq, l, p, _DTYPE_CODE[value.dtype],
            _DTYPE_CODE[raw.dtype], flags | _lib.FLAG_RAW_POINTS, _stream_ptr(value.device),
        )
    _lib.check(rc, "msda_forward (raw points)")
    return out


@ms_deform_raw_attn_fwd.register_fake
def _(value, spatial_shapes, level_start_index, raw, reference_points, flags):
    d = value.shape[3]
    return value.new_empty((raw.shape[0], raw.shape[1], raw.shape[2] * d))


@torch.library.custom_op("unipercept::ms_deform_raw_attn_bwd", mutates_args=())
def ms_deform_raw_attn_bwd(
    grad_output: torch.Tensor,
    value: torch.Tensor,
    spatial_shapes: torch.Tensor,
    level_start_index: torch.Tensor,
    raw: torch.Tensor,
    reference_points: torch.Tensor,
    flags: int,
) -> tuple[torch.Tensor, torch.Tensor]:
    """Backward on raw points: grad_value (N,S,M,D) and the gradient of the raw tensor."""
    lib = _lib.load()
    raw_dtype = raw.dtype
    hm = bool(flags & _lib.FLAG_VALUE_HEAD_MAJOR)
    value, shapes, lsi, raw, ref, levels_host = _prepare_raw(
        value, spatial_shapes, level_start_index, raw, reference_points, hm
    )
    n, s, m, d, lq, l, p = _packed_dims(value, raw, int(shapes.shape[0]), hm)
    grad_output = grad_output.to(value.dtype).contiguous()
    if tuple(grad_output.shape) != (n, lq, m * d):
        raise ValueError(f"grad_output must be {(n, lq, m * d)}, got {tuple(grad_output.shape)}")
    grad_value = torch.empty((n, s, m, d), dtype=value.dtype, device=value.device)
    grad_raw = torch.empty_like(raw)
    vcode, lcode = _DTYPE_CODE[value.dtype], _DTYPE_CODE[raw.dtype]
    flags |= _lib.FLAG_RAW_POINTS
    with torch.cuda.device_of(value):
        nbytes = lib.msda_backward_workspace_bytes(
            n, s, m, d, lq, l, p, vcode, lcode, flags, levels_host
        )
        workspace = (
            torch.empty(nbytes, dtype=torch.uint8, device=value.device) if nbytes else None
        )
        rc = lib.msda_backward(
            grad_output.data_ptr(), value.data_ptr(), shapes.data_ptr(), lsi.data_ptr(),
            raw.data_ptr(), None, ref.data_ptr(), grad_value.data_ptr(), grad_raw.data_ptr(), None,
            workspace.data_ptr() if nbytes else None, nbytes,
            n, s, m, d, lq, l, p, vcode, lcode, flags, levels_host, _stream_ptr(value.device),
        )
    _lib.check(rc, "msda_backward (raw points)")
    return grad_value, grad_raw.to(raw_dtype)


@ms_deform_raw_attn_bwd.register_fake
def _(grad_output, value, spatial_shapes, level_start_index, raw, reference_points, flags):
    gv = value.transpose(1, 2) if flags & _lib.FLAG_VALUE_HEAD_MAJOR else value
    return torch.empty_like(gv, memory_format=torch.contiguous_format), torch.empty_like(raw)


def raw_points_supported(value: torch.Tensor, raw: torch.Tensor, levels: int, head_major: bool = False) -> bool:
    """Whether the kernels take this raw offsets+logits tensor (softmax and location arithmetic
    inside the attention kernels)."""
    if value.dtype not in _DTYPE_CODE or value.dim() != 4 or raw.dim() != 4:
        return False
    if levels < 1 or raw.shape[-1] % (3 * levels) != 0:
        return False
    n, s, m, d, lq, l, p = _packed_dims(value, raw, levels, head_major)
    vdt = value.dtype
    ldt = torch.float32 if raw.dtype == torch.float32 and vdt in (torch.bfloat16, torch.float16) else vdt
    return bool(
        _lib.load().msda_raw_points_supported(
            n, s, m, d, lq, l, p, _DTYPE_CODE[vdt], _DTYPE_CODE[ldt]
        )
    )


def packed_points_supported(value: torch.Tensor, packed: torch.Tensor, levels: int) -> bool:
    """Whether the kernels read this packed locations+weights tensor directly."""
    if value.dtype not in _DTYPE_CODE or value.dim() != 4 or packed.dim() != 4:
        return False
    if levels < 1 or packed.shape[-1] % (3 * levels) != 0:
        return False
    n, s, m, d, lq, l, p = _packed_dims(value, packed, levels, False)
    vdt = value.dtype
    ldt = torch.float32 if packed.dtype == torch.float32 and vdt in (torch.bfloat16, torch.float16) else vdt
    return bool(
        _lib.load().msda_packed_points_supported(
            n, s, m, d, lq, l, p, _DTYPE_CODE[vdt], _DTYPE_CODE[ldt]
        )
    )


_POINT_DTYPES = (torch.float32, torch.bfloat16, torch.float16)


def _points_dims(raw: torch.Tensor, shapes: torch.Tensor):
    levels = int(shapes.shape[0])
    if raw.dim() != 4 or levels < 1 or raw.shape[-1] % (3 * levels) != 0:
        raise ValueError(
            f"expected raw offsets+logits (N, Lq, M, 3*L*P) with L = {levels}; got {tuple(raw.shape)}"
        )
    n, lq, m, width = raw.shape
    return n, lq, m, levels, width // (3 * levels)


def _packed_point_dtype(raw_dtype: torch.dtype, reference_points: torch.Tensor) -> torch.dtype:
    """Element type of the packed locations+weights: fp32 whenever the reference points are fp32
    (a 16-bit projection output must not quantise the sampling locations), else raw's."""
    return torch.float32 if reference_points.dtype == torch.float32 else raw_dtype


@torch.library.custom_op("unipercept::ms_deform_points_fwd", mutates_args=())
def ms_deform_points_fwd(
    raw: torch.Tensor, reference_points: torch.Tensor, spatial_shapes: torch.Tensor
) -> torch.Tensor:
    """Raw (dx, dy) offsets + attention logits -> packed locations + softmax weights, one pass.
    The result is fp32 when the reference points are (also next to a 16-bit `raw`)."""
    lib = _lib.load()
    n, lq, m, l, p = _points_dims(raw, spatial_shapes)
    if raw.dtype not in _POINT_DTYPES:
        raise TypeError(f"unsupported dtype {raw.dtype}")
    if raw.device.type != "cuda":
        raise RuntimeError("ms_deform_points runs only on a ROCm GPU")
    if tuple(reference_points.shape) != (n, lq, l, 2):
        raise ValueError(
            f"reference_points must be {(n, lq, l, 2)}, got {tuple(reference_points.shape)}"
        )
    pdt = _packed_point_dtype(raw.dtype, reference_points)
    raw = raw.contiguous()
    ref = reference_points.to(device=raw.device, dtype=pdt).contiguous()
    shapes = spatial_shapes.to(device=raw.device, dtype=torch.int64).contiguous()
    packed = torch.empty(raw.shape, dtype=pdt, device=raw.device)
    with torch.cuda.device_of(raw):
        rc = lib.msda_points_forward(
            raw.data_ptr(), ref.data_ptr(), shapes.data_ptr(), packed.data_ptr(), n, lq, m, l, p,
            _DTYPE_CODE[raw.dtype], _DTYPE_CODE[pdt], _stream_ptr(raw.device),
        )
    _lib.check(rc, "msda_points_forward")
    return packed


@ms_deform_points_fwd.register_fake
def _(raw, reference_points, spatial_shapes):
    return torch.empty(
        raw.shape, dtype=_packed_point_dtype(raw.dtype, reference_points), device=raw.device
    )


@torch.library.custom_op("unipercept::ms_deform_points_bwd", mutates_args=())
def ms_deform_points_bwd(
    grad_packed: torch.Tensor, packed: torch.Tensor, spatial_shapes: torch.Tensor, raw_dtype: torch.dtype
) -> torch.Tensor:
    """Gradient of ms_deform_points_fwd with respect to its raw input (in `raw_dtype`)."""
    lib = _lib.load()
    n, lq, m, l, p = _points_dims(packed, spatial_shapes)
    if packed.device.type != "cuda" or grad_packed.device != packed.device:
        raise RuntimeError("ms_deform_points runs only on a ROCm GPU")
    if raw_dtype not in _POINT_DTYPES or packed.dtype not in (raw_dtype, torch.float32):
        raise TypeError(f"unsupported dtypes: raw {raw_dtype}, packed {packed.dtype}")
    packed = packed.contiguous()
    grad_packed = grad_packed.to(packed.dtype).contiguous()
    if grad_packed.shape != packed.shape:
        raise ValueError(f"grad must be {tuple(packed.shape)}, got {tuple(grad_packed.shape)}")
    shapes = spatial_shapes.to(device=packed.device, dtype=torch.int64).contiguous()
    grad_raw = torch.empty(packed.shape, dtype=raw_dtype, device=packed.device)
    with torch.cuda.device_of(packed):
        rc = lib.msda_points_backward(
            grad_packed.data_ptr(), packed.data_ptr(), shapes.data_ptr(), grad_raw.data_ptr(),
            n, lq, m, l, p, _DTYPE_CODE[raw_dtype], _DTYPE_CODE[packed.dtype],
            _stream_ptr(packed.device),
        )
    _lib.check(rc, "msda_points_backward")
    return grad_raw


@ms_deform_points_bwd.register_fake
def _(grad_packed, packed, spatial_shapes, raw_dtype):
    return torch.empty(packed.shape, dtype=raw_dtype, device=packed.device)


@torch.library.custom_op("unipercept::ms_deform_attn_value_head_major", mutates_args=())
def ms_deform_attn_value_head_major(
    value: torch.Tensor, padding_mask: torch.Tensor | None = None
) -> torch.Tensor:
    """(N,S,M,D) -> a new contiguous (N,M,S,D) tensor (one slab per head), by a HIP row copy;
    positions where the optional boolean ``padding_mask`` (N,S) is set come out as zeros (the
    layer's ``masked_fill`` in the same pass)."""
    lib = _lib.load()
    if value.dim() != 4 or value.dtype not in _DTYPE_CODE:
        raise ValueError(f"expected a floating (N,S,M,D) tensor, got {tuple(value.shape)} {value.dtype}")
    if value.device.type != "cuda":
        raise RuntimeError("ms_deform_attn_value_head_major runs only on a ROCm GPU")
    value = value.contiguous()
    n, s, m, d = value.shape
    mask_ptr = None
    if padding_mask is not None:
        if tuple(padding_mask.shape) != (n, s):
            raise ValueError(f"padding_mask must be {(n, s)}, got {tuple(padding_mask.shape)}")
        padding_mask = padding_mask.to(device=value.device, dtype=torch.bool).contiguous()
        mask_ptr = padding_mask.data_ptr()
    out = torch.empty((n, m, s, d), dtype=value.dtype, device=value.device)
    with torch.cuda.device_of(value):
        rc = lib.msda_value_relayout(
            value.data_ptr(), out.data_ptr(), mask_ptr, n, s, m, d, _DTYPE_CODE[value.dtype], 1,
            _stream_ptr(value.device),
        )
    _lib.check(rc, "msda_value_relayout")
    return out


@ms_deform_attn_value_head_major.register_fake
def _(value, padding_mask=None):
    n, s, m, d = value.shape
    return value.new_empty((n, m, s, d))


def head_major_supported(value: torch.Tensor, sampling_locations: torch.Tensor) -> bool:
    """Whether forward and backward of this problem take a head-major value tensor."""
    if value.dtype not in _DTYPE_CODE or value.dim() != 4 or sampling_locations.dim() != 6:
        return False
    n, s, m, d, lq, l, p = _dims(value, sampling_locations)
    vdt = value.dtype
    ldt = sampling_locations.dtype
    if not (ldt == torch.float32 and vdt in (torch.bfloat16, torch.float16)):
        ldt = vdt
    return bool(
        _lib.load().msda_head_major_supported(n, s, m, d, lq, l, p, _DTYPE_CODE[vdt], _DTYPE_CODE[ldt])
    )
