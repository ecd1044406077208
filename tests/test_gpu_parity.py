"""GPU parity: the HIP path (through the C ABI) against the golden vectors and the CPU oracle."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from conftest import golden_cases, level_start_index, load_golden, make_inputs

pytestmark = pytest.mark.gpu

# Stated tolerances for fp32: max-abs 1e-4 forward, 1e-3 backward.  The golden inputs have
# |value| <= 0.01 and |grad_out| ~ 1, so the observed errors sit orders of magnitude below.
FWD_ATOL_F32 = 1e-4
BWD_ATOL_F32 = 1e-3


def _run(case, dtype, env=None):
    from unipercept_amd.deform_attn import MultiScaleDeformAttnFunction

    z = load_golden(case)
    dev = "cuda"
    value = torch.from_numpy(z["value"]).to(dev, dtype).requires_grad_(True)
    loc = torch.from_numpy(z["loc"]).to(dev, dtype).requires_grad_(True)
    attn = torch.from_numpy(z["attn"]).to(dev, dtype).requires_grad_(True)
    shapes = torch.from_numpy(z["shapes"]).to(dev)
    lsi = level_start_index(shapes)
    out = MultiScaleDeformAttnFunction.apply(value, shapes, lsi, loc, attn, 64)
    out.backward(torch.from_numpy(z["grad_out"]).to(dev, dtype))
    return z, out.detach().cpu(), value.grad.cpu(), loc.grad.cpu(), attn.grad.cpu()


@pytest.mark.parametrize("generic", [False, True], ids=["tuned", "generic"])
@pytest.mark.parametrize("case", golden_cases())
def test_golden(case, generic, monkeypatch):
    if generic:
        monkeypatch.setenv("MSDA_FORCE_GENERIC", "1")
    dtype = torch.float32 if case.endswith("f32") else torch.float64
    z, out, gv, gl, ga = _run(case, dtype)
    if dtype == torch.float32:
        fa, ba = FWD_ATOL_F32, BWD_ATOL_F32
    else:
        fa, ba = 1e-12, 1e-10
    assert np.abs(out.numpy() - z["out"]).max() <= fa
    assert np.abs(gv.numpy() - z["grad_value"]).max() <= ba
    assert np.abs(gl.numpy() - z["grad_loc"]).max() <= ba
    assert np.abs(ga.numpy() - z["grad_attn"]).max() <= ba
    # tighter, relative check: fp32 results should agree with the fp32 golden to ~1e-5 relative
    for got, want in ((out, z["out"]), (gv, z["grad_value"]), (gl, z["grad_loc"]), (ga, z["grad_attn"])):
        scale = np.abs(want).max() + 1e-30
        rel = np.abs(got.numpy() - want).max() / scale
        assert rel <= (2e-5 if dtype == torch.float32 else 1e-12), rel


# ---------------------------------------------------------------------------------------------
# Mid-size cases against the scalar C oracle (fp64 arithmetic on the same, already rounded inputs)
# ---------------------------------------------------------------------------------------------
def _oracle64(value, shapes, loc, attn, grad_out):
    from oracle import cpu as oracle_cpu

    v = value.detach().cpu().double().numpy()
    lo = loc.detach().cpu().double().numpy()
    a = attn.detach().cpu().double().numpy()
    go = grad_out.detach().cpu().double().numpy()
    sh = shapes.cpu().numpy()
    out = oracle_cpu.forward(v, sh, lo, a)
    gv, gl, ga = oracle_cpu.backward(go, v, sh, lo, a)
    return out, gv, gl, ga


def _hip(value, shapes, lsi, loc, attn, grad_out):
    from unipercept_amd.deform_attn import MultiScaleDeformAttnFunction

    v = value.clone().requires_grad_(True)
    lo = loc.clone().requires_grad_(True)
    a = attn.clone().requires_grad_(True)
    out = MultiScaleDeformAttnFunction.apply(v, shapes, lsi, lo, a, 64)
    out.backward(grad_out)
    return tuple(t.detach().cpu().double().numpy() for t in (out, v.grad, lo.grad, a.grad))


MID = {
    # name: (n, m, d, lq, levels, p)
    "enc_L3P4": (2, 8, 32, None, [(32, 64), (16, 32), (8, 16)], 4),
    "enc_L4P4": (1, 8, 32, None, [(32, 32), (16, 16), (8, 8), (4, 4)], 4),
    "enc_L4P8": (1, 8, 32, None, [(16, 32), (8, 16), (4, 8), (2, 4)], 8),
    "dec_q300": (2, 8, 32, 300, [(32, 64), (16, 32), (8, 16)], 4),
    "heads3_d64": (2, 3, 64, 777, [(20, 30), (10, 15), (5, 8)], 4),
    "d16_ragged": (1, 5, 16, 515, [(17, 23), (9, 12), (1, 7)], 3),
}


@pytest.mark.parametrize("scatter", ["binned", "atomic"])
@pytest.mark.parametrize("case", sorted(MID))
def test_tuned_f32_vs_oracle(case, scatter, monkeypatch):
    if scatter == "atomic":
        monkeypatch.setenv("MSDA_ATOMIC_SCATTER", "1")
    n, m, d, lq, levels, p = MID[case]
    value, shapes, lsi, loc, attn, grad_out = make_inputs(
        n, m, d, lq, levels, p, dtype=torch.float32, seed=11, device="cuda"
    )
    value = value * 100  # O(1) values: a harder test of the absolute tolerances
    want = _oracle64(value, shapes, loc, attn, grad_out)
    got = _hip(value, shapes, lsi, loc, attn, grad_out)
    assert np.abs(got[0] - want[0]).max() <= FWD_ATOL_F32
    for g, w_ in zip(got[1:], want[1:]):
        assert np.abs(g - w_).max() <= BWD_ATOL_F32


# bf16: 8 significand bits.  Inputs are rounded to bf16 first and the fp64 oracle is fed those
# rounded values, so what is measured is the kernel's own error: fp32 accumulation plus one
# rounding of each result to bf16 (relative 2^-9).  Stated tolerance: max-abs error <= 2^-7 of
# the tensor's max-abs value (grad_loc: 2^-6, it multiplies two rounded quantities by W or H).
@pytest.mark.parametrize("loc_f32", [False, True], ids=["loc_bf16", "loc_f32"])
@pytest.mark.parametrize("case", ["enc_L3P4", "enc_L4P8", "heads3_d64"])
def test_tuned_bf16_vs_oracle(case, loc_f32):
    n, m, d, lq, levels, p = MID[case]
    value, shapes, lsi, loc, attn, grad_out = make_inputs(
        n, m, d, lq, levels, p, dtype=torch.float32, seed=5, device="cuda"
    )
    value = (value * 100).bfloat16()
    grad_out = grad_out.bfloat16()
    if not loc_f32:
        loc, attn = loc.bfloat16(), attn.bfloat16()
    want = _oracle64(value, shapes, loc, attn, grad_out)
    got = _hip(value, shapes, lsi, loc, attn, grad_out)
    for name, g, w_, tol in zip(
        ("out", "grad_value", "grad_loc", "grad_attn"), got, want, (2**-7, 2**-7, 2**-6, 2**-7)
    ):
        err = np.abs(g - w_).max() / (np.abs(w_).max() + 1e-30)
        assert err <= tol, (name, err)


def test_fp16_generic_and_tuned():
    n, m, d, lq, levels, p = MID["dec_q300"]
    value, shapes, lsi, loc, attn, grad_out = make_inputs(
        n, m, d, lq, levels, p, dtype=torch.float32, seed=3, device="cuda"
    )
    value, grad_out = (value * 100).half(), grad_out.half()
    want = _oracle64(value, shapes, loc, attn, grad_out)
    got = _hip(value, shapes, lsi, loc, attn, grad_out)  # fp16 value, fp32 loc/attn
    for g, w_ in zip(got, want):
        assert np.abs(g - w_).max() / (np.abs(w_).max() + 1e-30) <= 2**-9


@pytest.mark.parametrize("case", ["enc_L3P4", "heads3_d64", "d16_ragged"])
def test_scatter_tile_geometries_agree(case, monkeypatch):
    """With the level shapes known on the host the binned scatter cuts grad_value into 2-D
    patches, without them into runs of the flattened sequence: same result, both within the
    oracle tolerance."""
    n, m, d, lq, levels, p = MID[case]
    value, shapes, lsi, loc, attn, grad_out = make_inputs(
        n, m, d, lq, levels, p, dtype=torch.float32, seed=17, device="cuda"
    )
    want = _oracle64(value, shapes, loc, attn, grad_out)
    bwd = torch.ops.unipercept.ms_deform_attn_bwd
    patches = bwd(grad_out, value, shapes, lsi, loc, attn, 0)
    monkeypatch.setenv("MSDA_SKIP_LEVEL_CHECK", "1")  # no host copy of the table -> 1-D runs
    runs = bwd(grad_out, value, shapes, lsi, loc, attn, 0)
    for a, b, w_ in zip(patches, runs, want[1:]):
        assert (a - b).abs().max() <= 1e-5
        assert np.abs(a.cpu().double().numpy() - w_).max() <= BWD_ATOL_F32
        assert np.abs(b.cpu().double().numpy() - w_).max() <= BWD_ATOL_F32


def _flash(value, shapes, lsi, loc, attn, grad_out):
    from unipercept_amd.deform_attn import MultiScaleFlashAttnFunction

    v = value.clone().requires_grad_(True)
    packed = torch.cat([loc.flatten(-3), attn.flatten(-2)], dim=-1).requires_grad_(True)
    out = MultiScaleFlashAttnFunction.apply(v, shapes, lsi, packed, 64)
    out.backward(grad_out)
    lp2 = loc[0, 0, 0].numel()
    g_loc = packed.grad[..., :lp2].reshape(loc.shape)
    g_attn = packed.grad[..., lp2:].reshape(attn.shape)
    return tuple(t.detach().cpu().double().numpy() for t in (out, v.grad, g_loc, g_attn))


@pytest.mark.parametrize("dtype", ["f32", "bf16", "bf16_locf32"])
@pytest.mark.parametrize("case", ["enc_L3P4", "enc_L4P8", "heads3_d64"])
def test_flash_packed_points_direct(case, dtype, monkeypatch):
    """The packed locations+weights tensor is read by the kernels as it is (even L*P, tuned head
    dims): same results as the oracle, and as the unpacking route."""
    from unipercept_amd.deform_attn import _ops

    n, m, d, lq, levels, p = MID[case]
    value, shapes, lsi, loc, attn, grad_out = make_inputs(
        n, m, d, lq, levels, p, dtype=torch.float32, seed=23, device="cuda"
    )
    value = value * 100
    if dtype != "f32":
        value, grad_out = value.bfloat16(), grad_out.bfloat16()
        if dtype == "bf16":
            loc, attn = loc.bfloat16(), attn.bfloat16()
    packed = torch.cat([loc.flatten(-3), attn.flatten(-2)], dim=-1)
    assert _ops.packed_points_supported(value, packed, len(levels))
    want = _oracle64(value, shapes, loc, attn, grad_out)
    got = _flash(value, shapes, lsi, loc, attn, grad_out)
    monkeypatch.setenv("MSDA_NO_PACKED_POINTS", "1")
    unpacked = _flash(value, shapes, lsi, loc, attn, grad_out)
    tols = (FWD_ATOL_F32, BWD_ATOL_F32, BWD_ATOL_F32, BWD_ATOL_F32)
    for name, g, u, w_, atol, rel in zip(
        ("out", "grad_value", "grad_loc", "grad_attn"), got, unpacked, want, tols,
        (2**-7, 2**-7, 2**-6, 2**-7),
    ):
        if dtype == "f32":
            assert np.abs(g - w_).max() <= atol, name
            assert np.abs(g - u).max() <= 1e-5, name
        else:
            scale = np.abs(w_).max() + 1e-30
            assert np.abs(g - w_).max() / scale <= rel, name
            assert np.abs(g - u).max() / scale <= rel, name


def test_flash_packed_points_fallbacks():
    """Odd L*P and head dims without a tuned kernel take the unpacking route; results agree
    with the oracle either way."""
    from unipercept_amd.deform_attn import _ops

    for case in ("d16_ragged",):  # L = 3, P = 3
        n, m, d, lq, levels, p = MID[case]
        value, shapes, lsi, loc, attn, grad_out = make_inputs(
            n, m, d, lq, levels, p, dtype=torch.float32, seed=29, device="cuda"
        )
        packed = torch.cat([loc.flatten(-3), attn.flatten(-2)], dim=-1)
        assert not _ops.packed_points_supported(value, packed, len(levels))
        want = _oracle64(value, shapes, loc, attn, grad_out)
        got = _flash(value, shapes, lsi, loc, attn, grad_out)
        assert np.abs(got[0] - want[0]).max() <= FWD_ATOL_F32
        for g, w_ in zip(got[1:], want[1:]):
            assert np.abs(g - w_).max() <= BWD_ATOL_F32
    v30 = torch.rand(1, 640, 2, 30, device="cuda")
    assert not _ops.packed_points_supported(v30, torch.rand(1, 5, 2, 3 * 2 * 2, device="cuda"), 2)


def test_head_major_layout_matches_position_major(monkeypatch):
    """The per-head value copy is a pure re-layout: identical results either way."""
    from unipercept_amd import _lib
    from unipercept_amd.deform_attn import _ops

    n, m, d, lq, levels, p = MID["enc_L3P4"]
    value, shapes, lsi, loc, attn, grad_out = make_inputs(
        n, m, d, lq, levels, p, dtype=torch.float32, seed=21, device="cuda"
    )
    assert _ops.head_major_supported(value, loc)
    vt = torch.ops.unipercept.ms_deform_attn_value_head_major(value)
    assert vt.shape == (n, m, value.shape[1], d)
    assert torch.equal(vt, value.permute(0, 2, 1, 3).contiguous())
    fwd, bwd = torch.ops.unipercept.ms_deform_attn_fwd, torch.ops.unipercept.ms_deform_attn_bwd
    hm = _lib.FLAG_VALUE_HEAD_MAJOR
    assert torch.equal(fwd(value, shapes, lsi, loc, attn, 0), fwd(vt, shapes, lsi, loc, attn, hm))
    a = bwd(grad_out, value, shapes, lsi, loc, attn, 0)
    b = bwd(grad_out, vt, shapes, lsi, loc, attn, hm)
    assert a[0].shape == b[0].shape == value.shape
    for x, y in zip(a, b):
        assert (x - y).abs().max() <= 1e-6
    # a shape without a tuned path refuses the flag instead of misreading the tensor
    v30 = torch.rand(1, 640, 2, 30, device="cuda")
    sh = torch.tensor([[32, 16], [16, 8]], device="cuda")
    lo = torch.rand(1, 2, 2, 2, 2, 2, device="cuda")
    at = torch.rand(1, 2, 2, 2, 2, device="cuda")
    assert not _ops.head_major_supported(v30, lo)
    with pytest.raises(RuntimeError, match="head-major"):
        fwd(v30.permute(0, 2, 1, 3).contiguous(), sh, level_start_index(sh), lo, at, hm)
