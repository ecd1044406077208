"""ctypes binding of ``libmsda.so`` (the C ABI in ``include/msda.h``)."""

from __future__ import annotations

import ctypes
import os
import threading
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("MSDA_LIBRARY", _PKG / "libmsda.so"))

ABI_VERSION = 8

# dtype codes of include/msda.h
F32, F64, BF16, F16 = 0, 1, 2, 3
FLAG_DETERMINISTIC = 1
FLAG_GENERIC = 2
FLAG_ATOMIC_SCATTER = 4
FLAG_VALUE_HEAD_MAJOR = 8
FLAG_PACKED_POINTS = 16
FLAG_TILE_1D = 32
FLAG_RAW_POINTS = 64
FLAG_BINNED = 128
FLAG_BIN_ONLY = 256
SAMPLE_BILINEAR, SAMPLE_NEAREST = 0, 1

# every symbol include/msda.h declares
EXPORTS = (
    "msda_version",
    "msda_error_string",
    "msda_check_im2col_step",
    "msda_forward",
    "msda_head_major_supported",
    "msda_packed_points_supported",
    "msda_raw_points_supported",
    "msda_value_relayout",
    "msda_backward_workspace_bytes",
    "msda_backward",
    "msda_backward_noalloc",
    "msda_points_forward",
    "msda_points_backward",
    "msda_point_sample_forward",
    "msda_point_sample_backward",
    "msda_point_sample_workspace_bytes",
)

_lock = threading.Lock()
_lib = None


class MsdaLibraryError(RuntimeError):
    pass


def _declare(lib: ctypes.CDLL) -> None:
    c_int, c_vp, c_sz = ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t
    lib.msda_version.restype = c_int
    lib.msda_version.argtypes = []
    lib.msda_error_string.restype = ctypes.c_char_p
    lib.msda_error_string.argtypes = [c_int]
    lib.msda_check_im2col_step.restype = c_int
    lib.msda_check_im2col_step.argtypes = [c_int, c_int]
    lib.msda_forward.restype = c_int
    lib.msda_forward.argtypes = [c_vp] * 7 + [c_int] * 7 + [c_int, c_int, c_int, c_vp]
    lib.msda_head_major_supported.restype = c_int
    lib.msda_head_major_supported.argtypes = [c_int] * 9
    lib.msda_packed_points_supported.restype = c_int
    lib.msda_packed_points_supported.argtypes = [c_int] * 9
    lib.msda_raw_points_supported.restype = c_int
    lib.msda_raw_points_supported.argtypes = [c_int] * 9
    lib.msda_value_relayout.restype = c_int
    lib.msda_value_relayout.argtypes = [c_vp, c_vp, c_vp] + [c_int] * 6 + [c_vp]
    lib.msda_backward_workspace_bytes.restype = c_sz
    lib.msda_backward_workspace_bytes.argtypes = [c_int] * 7 + [c_int, c_int, c_int, c_vp]
    lib.msda_backward.restype = c_int
    lib.msda_backward.argtypes = (
        [c_vp] * 11 + [c_sz] + [c_int] * 7 + [c_int, c_int, c_int, c_vp, c_vp]
    )
    lib.msda_backward_noalloc.restype = c_int
    lib.msda_backward_noalloc.argtypes = (
        [c_vp] * 10 + [c_int] * 7 + [c_int, c_int, c_int, c_vp, c_vp]
    )

    lib.msda_points_forward.restype = c_int
    lib.msda_points_forward.argtypes = [c_vp] * 4 + [c_int] * 7 + [c_vp]
    lib.msda_points_backward.restype = c_int
    lib.msda_points_backward.argtypes = [c_vp] * 4 + [c_int] * 7 + [c_vp]
    lib.msda_point_sample_forward.restype = c_int
    lib.msda_point_sample_forward.argtypes = [c_vp] * 3 + [c_int] * 8 + [c_vp]
    lib.msda_point_sample_backward.restype = c_int
    lib.msda_point_sample_backward.argtypes = [c_vp] * 6 + [c_sz] + [c_int] * 8 + [c_vp]
    lib.msda_point_sample_workspace_bytes.restype = c_sz
    lib.msda_point_sample_workspace_bytes.argtypes = [c_int] * 6


def load() -> ctypes.CDLL:
    """Loads the HIP library; raises if it has not been built (no fallback exists)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not LIB_PATH.exists():
                    raise MsdaLibraryError(
                        f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                        "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback."
                    )
                lib = ctypes.CDLL(str(LIB_PATH))
                missing = [s for s in EXPORTS if not hasattr(lib, s)]
                if missing:
                    raise MsdaLibraryError(f"{LIB_PATH} lacks symbols {missing}")
                _declare(lib)
                if lib.msda_version() != ABI_VERSION:
                    raise MsdaLibraryError(
                        f"{LIB_PATH} has ABI version {lib.msda_version()}, expected {ABI_VERSION}"
                    )
                _lib = lib
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().msda_error_string(code).decode()
        raise RuntimeError(f"{what} failed ({code}): {msg}")
