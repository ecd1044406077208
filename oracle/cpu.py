"""TEST INFRASTRUCTURE: ctypes binding of the scalar C oracles (``msda_cpu.c``,
``point_sample_cpu.c``)."""

from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None


def build(force: bool = False) -> Path:
    so = _HERE / "libmsda_cpu.so"
    srcs = [_HERE / "msda_cpu.c", _HERE / "point_sample_cpu.c"]
    if force or not so.exists() or any(so.stat().st_mtime < src.stat().st_mtime for src in srcs):
        subprocess.run(["make", "-C", str(_HERE), "libmsda_cpu.so"], check=True, capture_output=True)
    return so


def lib(path: str | None = None) -> ctypes.CDLL:
    global _LIB
    if path is not None:
        return ctypes.CDLL(path)
    if _LIB is None:
        _LIB = ctypes.CDLL(str(build()))
    return _LIB


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _prep(value, shapes, lsi, loc, attn):
    dt = value.dtype
    if dt not in (np.float32, np.float64):
        raise TypeError(f"oracle supports float32/float64, got {dt}")
    value = np.ascontiguousarray(value)
    loc = np.ascontiguousarray(loc, dtype=dt)
    attn = np.ascontiguousarray(attn, dtype=dt)
    shapes = np.ascontiguousarray(shapes, dtype=np.int64)
    if lsi is None:
        hw = shapes[:, 0] * shapes[:, 1]
        lsi = np.concatenate([[0], np.cumsum(hw)[:-1]])
    lsi = np.ascontiguousarray(lsi, dtype=np.int64)
    n, s, m, d = value.shape
    _, q, _, l, p, _ = loc.shape
    return value, shapes, lsi, loc, attn, (n, s, m, d, q, l, p), ("f32" if dt == np.float32 else "f64")


def forward(value, shapes, loc, attn, lsi=None, library=None) -> np.ndarray:
    value, shapes, lsi, loc, attn, dims, sfx = _prep(value, shapes, lsi, loc, attn)
    n, s, m, d, q, l, p = dims
    out = np.empty((n, q, m * d), dtype=value.dtype)
    fn = getattr(library or lib(), f"msda_cpu_forward_{sfx}")
    fn.restype = None
    fn(_ptr(value), _ptr(shapes), _ptr(lsi), _ptr(loc), _ptr(attn), _ptr(out), *map(ctypes.c_int, dims))
    return out


def backward(grad_out, value, shapes, loc, attn, lsi=None, library=None):
    value, shapes, lsi, loc, attn, dims, sfx = _prep(value, shapes, lsi, loc, attn)
    grad_out = np.ascontiguousarray(grad_out, dtype=value.dtype)
    gv = np.empty_like(value)
    gl = np.empty_like(loc)
    ga = np.empty_like(attn)
    fn = getattr(library or lib(), f"msda_cpu_backward_{sfx}")
    fn.restype = None
    fn(
        _ptr(grad_out), _ptr(value), _ptr(shapes), _ptr(lsi), _ptr(loc), _ptr(attn),
        _ptr(gv), _ptr(gl), _ptr(ga), *map(ctypes.c_int, dims),
    )
    return gv, gl, ga


_SAMPLE_MODES = {"bilinear": 0, "nearest": 1}


def _prep_points(input, coords):
    dt = input.dtype
    if dt not in (np.float32, np.float64):
        raise TypeError(f"oracle supports float32/float64, got {dt}")
    input = np.ascontiguousarray(input)
    coords = np.ascontiguousarray(coords, dtype=dt)
    n, c, h, w = input.shape
    assert coords.shape[0] == n and coords.shape[2] == 2 and coords.ndim == 3
    return input, coords, (n, c, h, w, coords.shape[1]), ("f32" if dt == np.float32 else "f64")


def point_sample_forward(input, coords, mode="bilinear", align_corners=False, library=None):
    """input (N,C,H,W), coords (N,P,2) in [0,1]^2 as (x, y) -> (N,C,P)."""
    input, coords, dims, sfx = _prep_points(input, coords)
    n, c, _, _, p = dims
    out = np.empty((n, c, p), dtype=input.dtype)
    fn = getattr(library or lib(), f"point_sample_cpu_forward_{sfx}")
    fn.restype = None
    fn(_ptr(input), _ptr(coords), _ptr(out), *map(ctypes.c_int, dims),
       ctypes.c_int(_SAMPLE_MODES[mode]), ctypes.c_int(int(bool(align_corners))))
    return out


def point_sample_backward(grad_out, input, coords, mode="bilinear", align_corners=False, library=None):
    """-> (grad_input (N,C,H,W), grad_coords (N,P,2))."""
    input, coords, dims, sfx = _prep_points(input, coords)
    grad_out = np.ascontiguousarray(grad_out, dtype=input.dtype)
    gi = np.empty_like(input)
    gc = np.empty_like(coords)
    fn = getattr(library or lib(), f"point_sample_cpu_backward_{sfx}")
    fn.restype = None
    fn(_ptr(grad_out), _ptr(input), _ptr(coords), _ptr(gi), _ptr(gc), *map(ctypes.c_int, dims),
       ctypes.c_int(_SAMPLE_MODES[mode]), ctypes.c_int(int(bool(align_corners))))
    return gi, gc
