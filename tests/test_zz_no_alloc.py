"""msda_backward_noalloc (scratch in a static device array inside the library, no allocation in
the call) against msda_backward on the same inputs, through the C ABI: two calls in a row on one
GPU, the second with a smaller problem that reuses the array the first one left dirty."""

from __future__ import annotations

import ctypes

import pytest
import torch

from conftest import make_inputs

pytestmark = pytest.mark.gpu

_CODE = {torch.float32: 0, torch.bfloat16: 2}

_hip = None


def _device_free_bytes() -> int:
    """hipMemGetInfo's free bytes, asked of the HIP runtime itself: unlike PyTorch's allocator
    statistics it moves with every hipMalloc / hipFree of the process, the library's included."""
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipMemGetInfo.restype = ctypes.c_int
        _hip.hipMemGetInfo.argtypes = [ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert _hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


def _both(lib, case, dtype, flags=0):
    n, m, d, lq, levels, p = case
    value, shapes, lsi, loc, attn, grad_out = make_inputs(
        n, m, d, lq, levels, p, dtype=torch.float32, seed=31, device="cuda"
    )
    value, grad_out = (value * 100).to(dtype), grad_out.to(dtype)  # locations / weights stay fp32
    s, lq_, l = value.shape[1], loc.shape[1], len(levels)
    host = (ctypes.c_int64 * (2 * l))(*[int(x) for x in shapes.flatten().tolist()])
    dv, dl = _CODE[dtype], 0
    dims = (n, s, m, d, lq_, l, p)
    st = torch.cuda.current_stream().cuda_stream
    need = lib.msda_backward_workspace_bytes(*dims, dv, dl, flags, host)
    assert 0 < need <= 128 << 20
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")

    def outputs():
        return torch.full_like(value, 7.0), torch.full_like(loc, 7.0), torch.full_like(attn, 7.0)

    gv, gl, ga = outputs()
    rc = lib.msda_backward(
        grad_out.data_ptr(), value.data_ptr(), shapes.data_ptr(), lsi.data_ptr(), loc.data_ptr(),
        attn.data_ptr(), None, gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), ws.data_ptr(), need, *dims,
        dv, dl, flags, host, st,
    )
    assert rc == 0, rc
    gv2, gl2, ga2 = outputs()
    # one call first, so that what is measured below is a call with the array's address already
    # looked up and the kernels' code already on the device
    rc = lib.msda_backward_noalloc(
        grad_out.data_ptr(), value.data_ptr(), shapes.data_ptr(), lsi.data_ptr(), loc.data_ptr(),
        attn.data_ptr(), None, gv2.data_ptr(), gl2.data_ptr(), ga2.data_ptr(), *dims, dv, dl, flags,
        host, st,
    )
    assert rc == 0, rc
    gv2, gl2, ga2 = outputs()
    torch.cuda.synchronize()
    before, free_before = torch.cuda.memory_allocated(), _device_free_bytes()
    rc = lib.msda_backward_noalloc(
        grad_out.data_ptr(), value.data_ptr(), shapes.data_ptr(), lsi.data_ptr(), loc.data_ptr(),
        attn.data_ptr(), None, gv2.data_ptr(), gl2.data_ptr(), ga2.data_ptr(), *dims, dv, dl, flags,
        host, st,
    )
    assert rc == 0, rc
    torch.cuda.synchronize()
    # neither PyTorch's allocator nor the device's free memory moved: no hipMalloc / hipFree
    assert torch.cuda.memory_allocated() == before
    assert _device_free_bytes() == free_before
    # same kernels on the same inputs: grad_loc / grad_attn bit for bit; grad_value up to the
    # order of its fp32 adds (fp32: within the backward tolerance 1e-3 max-abs, in fact far
    # below; bf16: one rounding step of the result)
    assert torch.equal(gl, gl2) and torch.equal(ga, ga2)
    tol = 1e-5 if dtype == torch.float32 else 2.0**-7
    assert (gv.float() - gv2.float()).abs().max() <= tol * gv.float().abs().max()
    assert (gv.float() - gv2.float()).abs().max() <= 1e-3 or dtype != torch.float32
    return gv2


def test_no_alloc_backward_matches_backward_twice_in_a_row():
    from unipercept_amd import _lib

    lib = _lib.load()
    big = (2, 8, 32, None, [(32, 64), (16, 32), (8, 16)], 4)   # 36 MiB of scratch
    small = (1, 8, 32, 300, [(16, 32), (8, 16), (4, 8)], 4)    # decoder-style, fewer queries
    a = _both(lib, big, torch.float32)
    b = _both(lib, small, torch.float32)
    assert a.abs().max() > 0 and b.abs().max() > 0
    # the other users of the scratch: 1-D tiles, the fp32 accumulation image of a 16-bit
    # atomic scatter (cleared by the call itself), both on the array the calls above left dirty
    _both(lib, small, torch.float32, flags=_lib.FLAG_TILE_1D)
    _both(lib, small, torch.bfloat16, flags=_lib.FLAG_ATOMIC_SCATTER)
    _both(lib, big, torch.bfloat16)


def test_no_alloc_backward_refuses_what_does_not_fit():
    from unipercept_amd import _lib

    lib = _lib.load()
    fake = 1 << 20
    # headline shapes: 2.5 GB of scratch
    rc = lib.msda_backward_noalloc(
        fake, fake, fake, fake, fake, fake, None, fake, fake, fake, 8, 43008, 8, 32, 43008, 3, 4,
        0, 0, 0, None, None,
    )
    assert rc == -9
    assert b"128 MiB" in lib.msda_error_string(rc)
