"""Full-size (benchmark shapes) checks: every benchmark shape family against the scalar C oracle
on one whole sample (the oracle needs a few seconds per sample at these sizes), the launches the
benchmark times at their batch sizes against their own batch-1 slices and the oracle, and size-independent identities
(adjoint, linearity, constant field) on top."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from unipercept_amd import harness

pytestmark = pytest.mark.gpu


def _ops():
    from unipercept_amd.deform_attn import _ops  # noqa: F401  (registers the ops)

    return torch.ops.unipercept.ms_deform_attn_fwd, torch.ops.unipercept.ms_deform_attn_bwd


FULL = [
    ("cityscapes_1024x2048_fp32_b8", 2),
    ("dvps_1024x2048_L4_P8_fp32_b8", 1),
]


@pytest.mark.parametrize("name,batch", FULL)
@pytest.mark.parametrize("dist", ["R", "U"])
def test_adjoint_identities(name, batch, dist):
    """out is linear in value and in attn, so <out, g> = <value, grad_value> = <attn, grad_attn>."""
    fwd, bwd = _ops()
    w = harness.WORKLOADS[name]
    value, shapes, lsi, loc, attn, g = harness.synth_inputs(
        w, "cuda", batch=batch, seed=99, dist=dist, oob_fraction=0.01
    )
    out = fwd(value, shapes, lsi, loc, attn, 0)
    gv, gl, ga = bwd(g, value, shapes, lsi, loc, attn, 0)
    lhs = (out.double() * g.double()).sum()
    via_value = (value.double() * gv.double()).sum()
    via_attn = (attn.double() * ga.double()).sum()
    scale = (out.double().abs() * g.double().abs()).sum()
    assert abs(lhs - via_value) / scale < 1e-6
    assert abs(lhs - via_attn) / scale < 1e-6
    assert torch.isfinite(gl).all()


@pytest.mark.parametrize("name,batch", FULL[:1])
def test_linearity_in_value(name, batch):
    fwd, _ = _ops()
    w = harness.WORKLOADS[name]
    value, shapes, lsi, loc, attn, _ = harness.synth_inputs(w, "cuda", batch=batch, seed=3)
    other = torch.randn_like(value)
    a, b = 0.75, -1.5
    lhs = fwd(a * value + b * other, shapes, lsi, loc, attn, 0)
    rhs = a * fwd(value, shapes, lsi, loc, attn, 0) + b * fwd(other, shapes, lsi, loc, attn, 0)
    assert (lhs - rhs).abs().max() <= 1e-4


def test_constant_field_is_reproduced():
    """In-range points of a per-head constant field sample that constant: out = c_m * sum(attn) = c_m."""
    fwd, bwd = _ops()
    w = harness.WORKLOADS["cityscapes_1024x2048_fp32_b8"]
    value, shapes, lsi, loc, attn, g = harness.synth_inputs(w, "cuda", batch=1, seed=4, dist="U")
    # keep every bilinear tap inside its level: loc in [0.5/W, 1 - 0.5/W] on the smallest level
    loc = loc * (1 - 2 * 0.5 / 32) + 0.5 / 32
    c = torch.arange(1, w.heads + 1, device="cuda", dtype=torch.float32)
    value = c.view(1, 1, -1, 1).expand_as(value).contiguous()
    out = fwd(value, shapes, lsi, loc, attn, 0).view(1, -1, w.heads, w.head_dim)
    assert (out - c.view(1, 1, -1, 1)).abs().max() <= 1e-4
    # and the gradient w.r.t. the locations of a constant field vanishes
    _, gl, _ = bwd(g, value, shapes, lsi, loc, attn, 0)
    assert gl.abs().max() <= 1e-2


@pytest.mark.parametrize("dist", ["R", "U"])
def test_binned_scatter_equals_atomic_scatter_and_generic(dist):
    from unipercept_amd import _lib

    _, bwd = _ops()
    fwd, _ = _ops()
    w = harness.WORKLOADS["cityscapes_1024x2048_fp32_b8"]
    value, shapes, lsi, loc, attn, g = harness.synth_inputs(
        w, "cuda", batch=1, seed=17, dist=dist, oob_fraction=0.01
    )
    binned = bwd(g, value, shapes, lsi, loc, attn, 0)
    atomic = bwd(g, value, shapes, lsi, loc, attn, _lib.FLAG_ATOMIC_SCATTER)
    generic = bwd(g, value, shapes, lsi, loc, attn, _lib.FLAG_GENERIC)
    for x, y, z in zip(binned, atomic, generic):
        assert (x - y).abs().max() <= 1e-3
        assert (x - z).abs().max() <= 1e-3
    assert (fwd(value, shapes, lsi, loc, attn, 0) - fwd(value, shapes, lsi, loc, attn, _lib.FLAG_GENERIC)).abs().max() <= 1e-4


# ---------------------------------------------------------------------------------------------
# One whole sample of every benchmark shape family against the C oracle
# ---------------------------------------------------------------------------------------------
def _oracle64(value, shapes, loc, attn, grad_out):
    """fp64 scalar oracle on the inputs as the kernels see them (already rounded to their dtype)."""
    from oracle import cpu as oracle_cpu

    v, lo, a, go = (t.detach().cpu().double().numpy() for t in (value, loc, attn, grad_out))
    sh = shapes.cpu().numpy()
    out = oracle_cpu.forward(v, sh, lo, a)
    gv, gl, ga = oracle_cpu.backward(go, v, sh, lo, a)
    return out, gv, gl, ga


def _away_from_pixel_edges(loc, levels):
    """grad_loc is discontinuous where a pixel coordinate crosses an integer (the footprint moves
    to other pixels), so points within 1e-4 of such an edge are left out of its comparison.
    Coordinates exactly ON an edge stay in (common with 16-bit locations; there fp32 and fp64
    agree exactly: W and H are powers of two, so loc * W - 0.5 is computed without rounding)."""
    lo = loc.detach().cpu().double().numpy()
    w = np.array([s[1] for s in levels], dtype=np.float64).reshape(1, 1, 1, -1, 1)
    h = np.array([s[0] for s in levels], dtype=np.float64).reshape(1, 1, 1, -1, 1)
    x, y = lo[..., 0] * w - 0.5, lo[..., 1] * h - 0.5
    dx, dy = np.abs(x - np.round(x)), np.abs(y - np.round(y))
    near = ((dx > 0) & (dx < 1e-4)) | ((dy > 0) & (dy < 1e-4))
    assert near.mean() < 5e-3
    return ~near


def _check_against_oracle(got, want, dtype: str, loc, levels):
    out, gv, gl, ga = (t.detach().cpu().double().numpy() for t in got)
    w_out, w_gv, w_gl, w_ga = want
    keep = _away_from_pixel_edges(loc, levels)[..., None]
    if dtype == "f32":
        # BASELINE tolerances (max-abs): 1e-4 forward, 1e-3 backward.  grad_loc carries the factor
        # W_l or H_l (up to 256 here against 32 in the reference's test), so with O(1) values its
        # magnitude reaches thousands: its bound is 1e-3 or 2e-6 of its largest element.
        assert np.abs(out - w_out).max() <= 1e-4
        assert np.abs(gv - w_gv).max() <= 1e-3
        assert np.abs(ga - w_ga).max() <= 1e-3
        assert (np.abs(gl - w_gl) * keep).max() <= max(1e-3, 2e-6 * np.abs(w_gl).max())
    else:
        # 16-bit value: fp32 accumulation and one rounding of each result; relative to the
        # tensor's largest element 2^-7 (grad_loc 2^-6), against fp64 on the same rounded inputs
        for name, g, w_, tol in (("out", out, w_out, 2.0**-7), ("grad_value", gv, w_gv, 2.0**-7),
                                 ("grad_attn", ga, w_ga, 2.0**-7)):
            assert np.abs(g - w_).max() <= tol * np.abs(w_).max(), name
        assert (np.abs(gl - w_gl) * keep).max() <= 2.0**-6 * np.abs(w_gl).max()


ORACLE_FULL = [
    # workload, location distribution, value scale (None: O(1) values)
    ("cityscapes_1024x2048_fp32_b8", "R", None),
    ("cityscapes_1024x2048_fp32_b8", "U", None),
    ("cityscapes_1024x2048_fp32_b8", "R", 0.01),  # the reference test's value scale
    ("cityscapes_1024x2048_bf16_b8", "R", None),
    ("cityscapes_1024x2048_bf16_locf32_b8", "R", None),
    ("coco_1024x1024_L4_bf16_b16", "R", None),
    ("dvps_1024x2048_L4_P8_fp32_b8", "R", None),
    ("dvps_1024x2048_L4_P8_bf16_b8", "U", None),
    ("cpuref_512x1024_fp32_b1", "R", None),
]


@pytest.mark.parametrize("name,dist,scale", ORACLE_FULL)
def test_full_size_sample_against_oracle(name, dist, scale):
    """HIP forward + three gradients of one whole sample at the benchmark shapes against the C
    oracle, about 1 % of the locations outside [0, 1]."""
    from unipercept_amd.deform_attn import MultiScaleDeformAttnFunction

    w = harness.WORKLOADS[name]
    value, shapes, lsi, loc, attn, g = harness.synth_inputs(
        w, "cuda", batch=1, seed=77, dist=dist, oob_fraction=0.01, value_scale=scale
    )
    v, lo, a = (t.clone().requires_grad_(True) for t in (value, loc, attn))
    out = MultiScaleDeformAttnFunction.apply(v, shapes, lsi, lo, a, 64)
    out.backward(g)
    want = _oracle64(value, shapes, loc, attn, g)
    _check_against_oracle((out, v.grad, lo.grad, a.grad), want, "f32" if w.dtype == "f32" else "16", loc, w.levels)


BENCH_BATCH = [
    # workload at its bench batch size, sample checked against the oracle
    ("cityscapes_1024x2048_fp32_b8", 5),  # the headline launch: 2.5 GB workspace
    ("cityscapes_1024x2048_bf16_b8", 3),
    ("coco_1024x1024_L4_bf16_b16", 11),   # 128-query blocks of the query kernel
    ("dvps_1024x2048_L4_P8_fp32_b8", 6),  # 32 points per query
]


@pytest.mark.parametrize("name,probe", BENCH_BATCH)
def test_bench_batch_launch(name, probe):
    """The launches bench.py times, at their batch sizes (blockIdx.y > 0 in every kernel
    template the benchmark runs, the workspace strides of a whole batch): adjoint identities over
    the batch, every sample equal to its own batch-1 launch, one sample against the oracle."""
    fwd, bwd = _ops()
    from unipercept_amd import _lib

    w = harness.WORKLOADS[name]
    value, shapes, lsi, loc, attn, g = harness.synth_inputs(
        w, "cuda", batch=w.batch, seed=1234, dist="R", oob_fraction=0.01
    )
    f32 = w.dtype == "f32"
    hm = _lib.FLAG_VALUE_HEAD_MAJOR
    vt = torch.ops.unipercept.ms_deform_attn_value_head_major(value)
    out = fwd(vt, shapes, lsi, loc, attn, hm)
    gv, gl, ga = bwd(g, vt, shapes, lsi, loc, attn, hm)
    lhs = (out.double() * g.double()).sum()
    scale = (out.double().abs() * g.double().abs()).sum()
    # 16-bit results carry one rounding each (2^-9 relative, random sign): far below 1e-3 of the sum
    tol = 1e-6 if f32 else 1e-3
    assert abs(lhs - (value.double() * gv.double()).sum()) / scale < tol
    assert abs(lhs - (attn.double() * ga.double()).sum()) / scale < tol
    for b in range(w.batch):
        sl = slice(b, b + 1)
        vb = vt[sl].contiguous()
        out_b = fwd(vb, shapes, lsi, loc[sl].contiguous(), attn[sl].contiguous(), hm)
        gv_b, gl_b, ga_b = bwd(g[sl].contiguous(), vb, shapes, lsi, loc[sl].contiguous(), attn[sl].contiguous(), hm)
        assert torch.equal(out[sl], out_b)
        assert torch.equal(gl[sl], gl_b)
        assert torch.equal(ga[sl], ga_b)
        # the scatter's order of fp32 adds is not fixed from launch to launch
        rel = 1e-5 if f32 else 2.0**-7
        assert (gv[sl].float() - gv_b.float()).abs().max() <= rel * gv_b.float().abs().max()
    sl = slice(probe, probe + 1)
    want = _oracle64(value[sl], shapes, loc[sl], attn[sl], g[sl])
    _check_against_oracle((out[sl], gv[sl], gl[sl], ga[sl]), want, "f32" if f32 else "16", loc[sl], w.levels)


@pytest.mark.parametrize("name,batch", FULL)
def test_packed_points_equal_separate_tensors(name, batch):
    """The packed locations+weights entry point reads the same numbers from another layout:
    identical forward, gradients equal up to the order of fp32 adds in grad_value."""
    fwd, bwd = _ops()
    fwd_p, bwd_p = torch.ops.unipercept.ms_flash_attn_fwd, torch.ops.unipercept.ms_flash_attn_bwd
    w = harness.WORKLOADS[name]
    value, shapes, lsi, loc, attn, g = harness.synth_inputs(
        w, "cuda", batch=batch, seed=41, oob_fraction=0.01
    )
    packed = torch.cat([loc.flatten(-3), attn.flatten(-2)], dim=-1)
    lp2 = 2 * w.n_levels * w.points
    assert torch.equal(fwd_p(value, shapes, lsi, packed, 0), fwd(value, shapes, lsi, loc, attn, 0))
    gv, gl, ga = bwd(g, value, shapes, lsi, loc, attn, 0)
    gv_p, gp = bwd_p(g, value, shapes, lsi, packed, 0)
    assert torch.equal(gp[..., :lp2].reshape(gl.shape), gl)
    assert torch.equal(gp[..., lp2:].reshape(ga.shape), ga)
    assert (gv_p - gv).abs().max() <= 1e-4 * gv.abs().max()


def test_points_epilogue_full_size():
    """The sampling-point epilogue at the headline shapes against the PyTorch sequence."""
    from unipercept_amd.deform_attn import _ops as ops  # noqa: F401

    w = harness.WORKLOADS["cityscapes_1024x2048_fp32_b8"]
    shapes, _ = harness.level_tensors(w, "cuda")
    n, s, m, l, p = 2, w.positions, w.heads, w.n_levels, w.points
    gen = torch.Generator(device="cuda").manual_seed(5)
    raw = torch.randn(n, s, m, 3 * l * p, generator=gen, device="cuda") * 2
    ref = torch.rand(n, s, l, 2, generator=gen, device="cuda")
    grad = torch.randn(n, s, m, 3 * l * p, generator=gen, device="cuda")
    packed = torch.ops.unipercept.ms_deform_points_fwd(raw, ref, shapes)
    grad_raw = torch.ops.unipercept.ms_deform_points_bwd(grad, packed, shapes, raw.dtype)
    raw_t = raw.clone().requires_grad_(True)
    lp = l * p
    off = raw_t[..., : 2 * lp].reshape(n, s, m, l, p, 2)
    wh = torch.stack((shapes[:, 1], shapes[:, 0]), -1).float()
    loc = ref[:, :, None, :, None, :] + off / wh[None, None, None, :, None, :]
    want = torch.cat((loc.flatten(-3), raw_t[..., 2 * lp :].softmax(-1)), dim=-1)
    want.backward(grad)
    assert (packed - want).abs().max() <= 2e-6 * want.abs().max()
    assert (grad_raw - raw_t.grad).abs().max() <= 1e-5 * raw_t.grad.abs().max()


def test_raw_points_full_size_against_oracle():
    """The raw-points route (offsets + logits + reference points in, softmax and location
    arithmetic inside the kernels) on one whole headline-shaped sample: forward and grad_value
    against the C oracle run on the derived locations / weights, the gradient of the raw tensor
    against the oracle's grad_loc / grad_attn pushed through the chain rule in fp64."""
    from unipercept_amd.deform_attn import _function

    w = harness.WORKLOADS["cityscapes_1024x2048_fp32_b8"]
    value, shapes, lsi, _, _, g = harness.synth_inputs(w, "cuda", batch=1, seed=55)
    n, s, m, l, p = 1, w.positions, w.heads, w.n_levels, w.points
    lp = l * p
    gen = torch.Generator(device="cuda").manual_seed(56)
    raw = torch.randn(n, s, m, 3 * lp, generator=gen, device="cuda")
    raw[..., : 2 * lp] *= 4.0  # offsets of a few pixels
    ref = torch.rand(n, s, l, 2, generator=gen, device="cuda") * 1.02 - 0.01
    assert _function._ops.raw_points_supported(value, raw, l)
    v, r, rp = (t.clone().requires_grad_(True) for t in (value, raw, ref))
    vhm = _function.HeadMajorValueFunction.apply(v)
    out = _function.MultiScaleRawPointsAttnFunction.apply(vhm, shapes, lsi, r, rp)
    out.backward(g)

    wh = torch.stack((shapes[:, 1], shapes[:, 0]), -1).double()[None, None, None, :, None, :]
    off = raw.double()[..., : 2 * lp].reshape(n, s, m, l, p, 2)
    # the kernels form each location with one fp32 rounding (1 / W is exact for these levels)
    loc = (ref.double()[:, :, None, :, None, :] + off / wh).float()
    attn = raw.double()[..., 2 * lp :].softmax(-1).reshape(n, s, m, l, p)
    w_out, w_gv, w_gl, w_ga = _oracle64(value, shapes, loc, attn, g)
    assert np.abs(out.detach().cpu().double().numpy() - w_out).max() <= 1e-4
    assert np.abs(v.grad.cpu().double().numpy() - w_gv).max() <= 1e-3
    gl, ga = torch.from_numpy(w_gl), torch.from_numpy(w_ga)
    a = attn.cpu()
    want_logit = a * (ga - (a * ga).sum(dim=(3, 4), keepdim=True))
    want_off = gl / wh.cpu()
    got = r.grad.cpu().double()
    assert (got[..., 2 * lp :].reshape(n, s, m, l, p) - want_logit).abs().max() <= 1e-3
    keep = torch.from_numpy(_away_from_pixel_edges(loc, w.levels))[..., None]
    err = (got[..., : 2 * lp].reshape(n, s, m, l, p, 2) - want_off).abs() * keep
    assert err.max() <= max(1e-3, 2e-6 * want_off.abs().max().item())
    # grad of the reference points = sum over heads and points of the location gradients
    g_loc = got[..., : 2 * lp].reshape(n, s, m, l, p, 2) * wh.cpu()
    sum_ref = g_loc.sum(dim=(2, 4))
    assert (rp.grad.cpu().double() - sum_ref).abs().max() <= 1e-4 * sum_ref.abs().max()
