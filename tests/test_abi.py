"""The C-ABI library: loads on a GPU-less host, exports what include/msda.h declares, and
rejects bad arguments before anything is launched (so these run without a GPU)."""

from __future__ import annotations

import ctypes
import re
from pathlib import Path

import pytest

from unipercept_amd import _lib

ROOT = Path(__file__).resolve().parent.parent
HEADER = (ROOT / "include" / "msda.h").read_text()


def declared_functions():
    body = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    return sorted(set(re.findall(r"\b(msda_[a-z0-9_]+)\s*\(", body)))


def test_header_and_binding_agree():
    assert declared_functions() == sorted(_lib.EXPORTS)


@pytest.mark.parametrize("symbol", declared_functions())
def test_symbol_exported(symbol):
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    assert hasattr(lib, symbol)


def test_version_and_error_strings():
    lib = _lib.load()
    assert lib.msda_version() == _lib.ABI_VERSION
    assert lib.msda_error_string(0) == b"success"
    for code in range(-7, 0):
        assert lib.msda_error_string(code).startswith(b"msda:")


def test_enum_values_match_header():
    enums = dict(re.findall(r"\b(MSDA_[A-Z0-9_]+)\s*=\s*(-?\d+)", HEADER))
    assert int(enums["MSDA_F32"]) == _lib.F32 and int(enums["MSDA_F64"]) == _lib.F64
    assert int(enums["MSDA_BF16"]) == _lib.BF16 and int(enums["MSDA_F16"]) == _lib.F16
    assert int(enums["MSDA_FLAG_DETERMINISTIC"]) == _lib.FLAG_DETERMINISTIC
    assert int(enums["MSDA_FLAG_GENERIC"]) == _lib.FLAG_GENERIC
    assert int(enums["MSDA_FLAG_ATOMIC_SCATTER"]) == _lib.FLAG_ATOMIC_SCATTER
    for name in ("VALUE_HEAD_MAJOR", "PACKED_POINTS", "TILE_1D", "RAW_POINTS", "BINNED", "BIN_ONLY"):
        assert int(enums[f"MSDA_FLAG_{name}"]) == getattr(_lib, f"FLAG_{name}")


@pytest.mark.parametrize(
    "n,step,ok",
    # the reference's test passes im2col_step=8 and 2 with N=1
    # (test_nn_layers_deform_attn.py:12,34,177): step = min(N, im2col_step) must divide N
    [(1, 8, True), (1, 2, True), (8, 8, True), (8, 4, True), (6, 4, False), (8, 64, True),
     (3, 2, False), (4, 0, False), (4, -1, False), (0, 1, False)],
)
def test_im2col_step(n, step, ok):
    rc = _lib.load().msda_check_im2col_step(n, step)
    assert (rc == 0) == ok
    if not ok:
        assert rc == -6


FAKE = 0x1000  # never dereferenced: every call below fails validation first


def _fwd(value=FAKE, shapes=FAKE, lsi=FAKE, loc=FAKE, attn=FAKE, out=FAKE,
         dims=(1, 4, 1, 4, 2, 1, 1), dv=0, dl=0, flags=0, ref=None):
    return _lib.load().msda_forward(value, shapes, lsi, loc, attn, ref, out, *dims, dv, dl, flags, None)


def test_forward_argument_errors():
    assert _fwd(value=None) == -1
    assert _fwd(out=None) == -1
    assert _fwd(dims=(0, 4, 1, 4, 2, 1, 1)) == -2
    assert _fwd(dims=(1, 4, 1, 4, 2, 1, -3)) == -2
    assert _fwd(dims=(1, 4, 1, 4, 2, 17, 1)) == -7
    assert _fwd(dims=(1, 4, 1, 4, 2, 8, 32)) == -7
    assert _fwd(dv=9) == -3
    assert _fwd(dv=0, dl=2) == -3  # fp32 value with bf16 locations is not a supported pair
    assert _fwd(dv=1, dl=0) == -3
    assert _fwd(value=FAKE + 2) == -4  # fp32 pointer not 4-byte aligned
    assert _fwd(shapes=FAKE + 4) == -4
    assert _fwd(dims=(70000, 4, 1, 4, 70000, 1, 1)) == -2  # N*Lq*M over 2^31-1


def _bwd(workspace=None, wbytes=0, dims=(2, 64, 2, 32, 64, 1, 2), dv=2, dl=2, flags=0, **kw):
    p = dict(grad_out=FAKE, value=FAKE, shapes=FAKE, lsi=FAKE, loc=FAKE, attn=FAKE,
             grad_value=FAKE, grad_loc=FAKE, grad_attn=FAKE)
    p.update(kw)
    return _lib.load().msda_backward(
        p["grad_out"], p["value"], p["shapes"], p["lsi"], p["loc"], p["attn"], p.get("ref"),
        p["grad_value"], p["grad_loc"], p["grad_attn"], workspace, wbytes, *dims, dv, dl, flags,
        None, None,
    )


def test_backward_argument_errors():
    assert _bwd(grad_attn=None) == -1
    assert _bwd(dims=(2, 64, 2, 32, 64, 0, 2)) == -2
    assert _bwd(dv=5) == -3
    need = _lib.load().msda_backward_workspace_bytes(2, 64, 2, 32, 64, 1, 2, 2, 2, 0, None)
    assert need > 0
    assert _bwd(workspace=None, wbytes=0) == -5
    assert _bwd(workspace=FAKE * 16, wbytes=need - 1) == -5
    assert _bwd(workspace=FAKE * 16 + 16, wbytes=need) == -5  # not 256-byte aligned


def test_workspace_bytes():
    ws_h = _lib.load().msda_backward_workspace_bytes

    def ws(*a):
        return ws_h(*a, None)

    # generic fp32 / fp64 paths accumulate straight into grad_value
    assert ws(1, 640, 2, 30, 2, 2, 2, _lib.F32, _lib.F32, 0) == 0
    assert ws(1, 640, 2, 30, 2, 2, 2, _lib.F64, _lib.F64, 0) == 0
    # generic 16-bit path: fp32 image of grad_value
    assert ws(1, 640, 2, 30, 2, 2, 2, _lib.BF16, _lib.BF16, 0) == 640 * 2 * 30 * 4
    # reproducible mode: the binned path's scratch where the shape has one (also when the atomic
    # scatter is asked for as well), none for the per-destination gather of the other shapes
    det = _lib.FLAG_DETERMINISTIC
    assert ws(1, 640, 2, 32, 2, 2, 2, _lib.BF16, _lib.BF16, det) == ws(1, 640, 2, 32, 2, 2, 2, _lib.BF16, _lib.BF16, 0) > 0
    assert ws(1, 640, 2, 32, 2, 2, 2, _lib.BF16, _lib.BF16, det | _lib.FLAG_ATOMIC_SCATTER) == ws(1, 640, 2, 32, 2, 2, 2, _lib.BF16, _lib.BF16, 0)
    assert ws(1, 640, 2, 30, 2, 2, 2, _lib.BF16, _lib.BF16, det) == 0
    assert ws(1, 640, 2, 32, 2, 2, 2, _lib.F64, _lib.F64, det) == 0
    # binned path (Cityscapes shapes, batch 8): directory + 64 B per sampled point + grad_out copy
    n, s, m, d, l, p = 8, 43008, 8, 32, 3, 4
    tiles = -(-s // 64)  # 64 positions per destination tile
    qb = 192  # queries per query-side workgroup
    qblocks = -(-s // qb)
    dir_bytes = n * m * qblocks * tiles * 8
    rec_bytes = n * m * qblocks * qb * l * p * 4 * 16
    g_copy_bytes = n * m * s * d * 4  # per-head copy of grad_out for the destination-side gather
    assert ws(n, s, m, d, s, l, p, _lib.F32, _lib.F32, 0) == (
        ((dir_bytes + 255) // 256) * 256 + rec_bytes + g_copy_bytes
    )
    # atomic scatter on request: nothing for fp32, the fp32 image for bf16
    assert ws(n, s, m, d, s, l, p, _lib.F32, _lib.F32, _lib.FLAG_ATOMIC_SCATTER) == 0
    assert ws(n, s, m, d, s, l, p, _lib.BF16, _lib.F32, _lib.FLAG_ATOMIC_SCATTER) == n * s * m * d * 4
    # invalid arguments give 0
    assert ws(0, s, m, d, s, l, p, _lib.F32, _lib.F32, 0) == 0
    # host level shapes select 2-D patches: 7 x 7 (tile kernel by footprint origin) for rows of up
    # to 32 channels, 8 x 8 for the reproducible backward and for wider rows — the directory grows
    # with the tile count, nothing else changes
    host = (ctypes.c_int64 * 6)(128, 256, 64, 128, 32, 64)
    r256 = lambda x: -(-x // 256) * 256
    tiles7 = 19 * 37 + 10 * 19 + 5 * 10
    base8 = ws(n, s, m, d, s, l, p, _lib.F32, _lib.F32, 0)  # 672 runs of 64 = as many 8 x 8 patches
    # (the origin kernel reads grad_out as given: no per-head copy in the workspace)
    assert ws_h(n, s, m, d, s, l, p, _lib.F32, _lib.F32, 0, host) - base8 == (
        r256(n * m * qblocks * tiles7 * 8) - r256(dir_bytes) - g_copy_bytes)
    assert ws_h(n, s, m, d, s, l, p, _lib.F32, _lib.F32, det, host) == base8
    assert ws_h(n, s, m, 64, s, l, p, _lib.F32, _lib.F32, 0, host) == ws(n, s, m, 64, s, l, p, _lib.F32, _lib.F32, 0)
    # ragged extents (3 levels of 20x24, 10x12, 5x6: 12+4+1 patches of 7 x 7, 9+4+1 of 8 x 8,
    # against ceil(630/64) = 10 runs), and the 1-D plan when the table does not cover S
    s2 = 20 * 24 + 10 * 12 + 5 * 6
    ragged = (ctypes.c_int64 * 6)(20, 24, 10, 12, 5, 6)
    base = ws(1, s2, m, d, s2, l, p, _lib.F32, _lib.F32, 0)
    qb2 = -(-s2 // 192)
    assert (ws_h(1, s2, m, d, s2, l, p, _lib.F32, _lib.F32, 0, ragged) - base
            == r256(m * qb2 * 17 * 8) - r256(m * qb2 * 10 * 8) - m * s2 * d * 4)
    assert (ws_h(1, s2, m, d, s2, l, p, _lib.F32, _lib.F32, det, ragged) - base
            == r256(m * qb2 * 14 * 8) - r256(m * qb2 * 10 * 8))
    wrong = (ctypes.c_int64 * 6)(20, 24, 10, 12, 5, 7)
    assert ws_h(1, s2, m, d, s2, l, p, _lib.F32, _lib.F32, 0, wrong) == base


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "libmsda.so")
    with pytest.raises(_lib.MsdaLibraryError):
        _lib.load()


def test_packed_points_flag_arguments():
    lib = _lib.load()
    # with the packed flag `attn` may be NULL; without it that is an error
    dims = (1, 4, 1, 4, 2, 1, 0)  # P = 0 -> the shape check fires after the NULL check passed
    assert lib.msda_forward(FAKE, FAKE, FAKE, FAKE, None, None, FAKE, *dims, 0, 0, _lib.FLAG_PACKED_POINTS, None) == -2
    assert lib.msda_forward(FAKE, FAKE, FAKE, FAKE, None, None, FAKE, *dims, 0, 0, 0, None) == -1
    ok = lib.msda_packed_points_supported
    assert ok(8, 43008, 8, 32, 43008, 3, 4, _lib.F32, _lib.F32) == 1
    assert ok(8, 43008, 8, 32, 43008, 3, 4, _lib.BF16, _lib.F32) == 1
    assert ok(1, 630, 5, 16, 515, 3, 3, _lib.F32, _lib.F32) == 0  # odd L*P
    assert ok(1, 640, 2, 30, 5, 2, 2, _lib.F32, _lib.F32) == 0  # no tuned kernel for D = 30
    assert ok(1, 640, 2, 32, 5, 2, 2, _lib.F64, _lib.F64) == 0
    assert ok(0, 640, 2, 32, 5, 2, 2, _lib.F32, _lib.F32) == 0


def test_no_alloc_backward_limit_is_checked_before_any_device_call():
    """msda_backward_noalloc refuses a problem whose scratch exceeds the static 128 MiB array
    (no GPU needed: the size check comes first), and validates like msda_backward otherwise."""
    lib = _lib.load()
    args = [FAKE] * 6 + [None] + [FAKE] * 3
    headline = (8, 43008, 8, 32, 43008, 3, 4)
    assert lib.msda_backward_workspace_bytes(*headline, 0, 0, 0, None) > 128 << 20
    assert lib.msda_backward_noalloc(*args, *headline, 0, 0, 0, None, None) == -9
    assert b"128 MiB" in lib.msda_error_string(-9)
    cpuref = (1, 10752, 8, 32, 10752, 3, 4)
    assert lib.msda_backward_workspace_bytes(*cpuref, 0, 0, 0, None) <= 128 << 20


def test_raw_points_flag_validation():
    """MSDA_FLAG_RAW_POINTS: needs the reference points, an 8-byte aligned pointer and a shape the
    register-resident backward covers; the support query says which."""
    lib = _lib.load()
    ok = lib.msda_raw_points_supported
    assert ok(8, 43008, 8, 32, 43008, 3, 4, _lib.F32, _lib.F32) == 1
    assert ok(16, 21760, 8, 32, 21760, 4, 4, _lib.BF16, _lib.BF16) == 1
    assert ok(8, 43520, 8, 32, 43520, 4, 8, _lib.F32, _lib.F32) == 1  # 32 points: two half rows
    assert ok(8, 43520, 8, 32, 43520, 4, 16, _lib.F32, _lib.F32) == 0  # 64 points per query
    assert ok(1, 630, 5, 16, 515, 3, 3, _lib.F32, _lib.F32) == 0  # odd L*P
    assert ok(1, 640, 2, 30, 5, 2, 2, _lib.F32, _lib.F32) == 0  # no tuned kernel
    raw = _lib.FLAG_RAW_POINTS
    big = (8, 43008, 8, 32, 43008, 3, 4)
    assert _fwd(dims=big, flags=raw, attn=None, ref=None) == -1
    assert _fwd(dims=big, flags=raw, attn=None, ref=FAKE + 4) == -4
    assert _fwd(dims=(8, 43520, 8, 32, 43520, 4, 16), flags=raw, attn=None, ref=FAKE) == -8
    assert _bwd(dims=big, dv=0, dl=0, flags=raw, attn=None, grad_attn=None, ref=None) == -1
    assert _bwd(dims=big, dv=0, dl=0, flags=raw | _lib.FLAG_ATOMIC_SCATTER, attn=None, grad_attn=None, ref=FAKE) == -8


def test_points_epilogue_argument_errors():
    lib = _lib.load()

    def fwd(raw=FAKE, ref=FAKE, shapes=FAKE, packed=FAKE, dims=(1, 5, 2, 2, 2), dtype=0, pdtype=None):
        return lib.msda_points_forward(raw, ref, shapes, packed, *dims, dtype, dtype if pdtype is None else pdtype, None)

    def bwd(gp=FAKE, packed=FAKE, shapes=FAKE, graw=FAKE, dims=(1, 5, 2, 2, 2), dtype=0, pdtype=None):
        return lib.msda_points_backward(gp, packed, shapes, graw, *dims, dtype, dtype if pdtype is None else pdtype, None)

    assert fwd(raw=None) == -1 and fwd(ref=None) == -1 and bwd(graw=None) == -1
    assert fwd(dims=(0, 5, 2, 2, 2)) == -2 and bwd(dims=(1, 0, 2, 2, 2)) == -2
    assert fwd(dims=(1, 5, 2, 17, 2)) == -7 and bwd(dims=(1, 5, 2, 8, 32)) == -7
    assert fwd(dtype=_lib.F64) == -3 and bwd(dtype=9) == -3
    # the packed side is raw's type or fp32, nothing else
    assert fwd(dtype=_lib.F32, pdtype=_lib.BF16) == -3 and bwd(dtype=_lib.BF16, pdtype=_lib.F16) == -3
    assert fwd(raw=FAKE + 8) == -4 and bwd(graw=FAKE + 4) == -4  # 16-byte alignment
    assert fwd(shapes=FAKE + 4) == -4


def test_point_sample_argument_errors():
    lib = _lib.load()

    def fwd(inp=FAKE, coords=FAKE, out=FAKE, dims=(1, 1, 2, 2, 3), dtype=0, mode=0):
        return lib.msda_point_sample_forward(inp, coords, out, *dims, dtype, mode, 0, None)

    def bwd(go=FAKE, inp=FAKE, coords=FAKE, gi=FAKE, gc=FAKE, dims=(1, 1, 2, 2, 3), dtype=0, mode=0):
        return lib.msda_point_sample_backward(go, inp, coords, gi, gc, None, 0, *dims, dtype, mode, 0, None)

    assert fwd(inp=None) == -1 and fwd(out=None) == -1 and bwd(gc=None) == -1
    assert fwd(dims=(0, 1, 2, 2, 3)) == -2 and bwd(dims=(1, 1, 2, 2, 0)) == -2
    assert fwd(dims=(1, 1, 70000, 70000, 3)) == -2  # H * W over 2^31-1
    assert fwd(dtype=9) == -3 and fwd(mode=2) == -3
    assert bwd(dtype=_lib.BF16) == -3 and bwd(dtype=_lib.F16) == -3  # 16-bit callers convert
    assert fwd(inp=FAKE + 2) == -4 and bwd(gi=FAKE + 2) == -4
    assert fwd(dtype=_lib.F64, coords=FAKE + 4) == -4
    # optional scratch of the backward: 2 * P point indices + 65 offsets per sample, as int32
    ws = lib.msda_point_sample_workspace_bytes
    assert ws(200, 1, 256, 512, 12544, _lib.F32) == (200 * 2 * 12544 + 200 * 65) * 4
    assert ws(200, 1, 256, 512, 12544, _lib.BF16) == 0  # the backward takes fp32 / fp64
    assert ws(1, 2, 3, 9000, 700, _lib.F32) == 0  # too wide for LDS bands: atomic path, no lists
    assert ws(0, 1, 2, 2, 3, _lib.F32) == 0


def test_tuned_kernels_refuse_what_their_24_bit_offsets_cannot_hold():
    """The tuned kernels form pixel, row and query offsets with 24-bit multiplies: 2^24 queries
    (or positions) are left to the generic kernels, which the support query reports."""
    lib = _lib.load()
    hm = lib.msda_head_major_supported
    assert hm(1, 1 << 18, 8, 32, 1 << 18, 3, 4, _lib.F32, _lib.F32) == 1
    assert hm(1, 1024, 1, 32, (1 << 18) - 1, 1, 4, _lib.BF16, _lib.BF16) == 1
    assert hm(1, 1024, 1, 32, 1 << 24, 1, 4, _lib.BF16, _lib.BF16) == 0
    assert hm(1, 1 << 24, 1, 32, 1024, 1, 4, _lib.BF16, _lib.BF16) == 0


def test_workspace_reflects_queries_per_workgroup():
    """Query blocks are 192 queries, or 128 where 16 points per query would otherwise leave room
    for two workgroups per CU only; the workspace size is the visible side of that choice."""
    lib = _lib.load()

    def expect(n, m, d, esize, levels, p, qb):
        s = sum(h * w for h, w in levels)
        lq, l = s, len(levels)
        tiles = sum(-(-h // 7) * -(-w // 7) for h, w in levels)  # 7 x 7 patches (rows of 32 channels)
        blocks = -(-lq // qb)
        dir_b = (n * m * blocks * tiles * 8 + 255) // 256 * 256
        rec_b = n * m * blocks * (qb * l * p * 4) * 16
        return dir_b + rec_b  # (7 x 7 patches: no per-head copy of grad_out)

    def got(n, m, d, dtype, levels, p):
        s = sum(h * w for h, w in levels)
        host = (ctypes.c_int64 * (2 * len(levels)))(*[x for hw in levels for x in hw])
        return lib.msda_backward_workspace_bytes(n, s, m, d, s, len(levels), p, dtype, dtype, 0, host)

    city = [(128, 256), (64, 128), (32, 64)]
    coco = [(128, 128), (64, 64), (32, 32), (16, 16)]
    assert got(8, 8, 32, _lib.F32, city, 4) == expect(8, 8, 32, 4, city, 4, 192)
    assert got(8, 8, 32, _lib.BF16, city, 4) == expect(8, 8, 32, 2, city, 4, 192)
    assert got(16, 8, 32, _lib.BF16, coco, 4) == expect(16, 8, 32, 2, coco, 4, 128)
