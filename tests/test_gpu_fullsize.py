"""Full-size (benchmark shapes) checks through size-independent properties; the CPU oracle would
take minutes at these sizes, so parity at full size rests on identities the op must satisfy."""

from __future__ import annotations

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


def test_bf16_full_size_close_to_fp32():
    fwd, bwd = _ops()
    w = harness.WORKLOADS["cityscapes_1024x2048_bf16_b8"]
    value, shapes, lsi, loc, attn, g = harness.synth_inputs(w, "cuda", batch=1, seed=8)
    out = fwd(value, shapes, lsi, loc, attn, 0)
    gv, gl, ga = bwd(g, value, shapes, lsi, loc, attn, 0)
    # same (already rounded) inputs evaluated by the fp32 kernels
    out32 = fwd(value.float(), shapes, lsi, loc.float(), attn.float(), 0)
    gv32, gl32, ga32 = bwd(g.float(), value.float(), shapes, lsi, loc.float(), attn.float(), 0)
    for got, want, tol in ((out, out32, 2**-7), (gv, gv32, 2**-7), (ga, ga32, 2**-7), (gl, gl32, 2**-6)):
        assert (got.float() - want).abs().max() <= tol * want.abs().max()


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
    grad_raw = torch.ops.unipercept.ms_deform_points_bwd(grad, packed, shapes)
    raw_t = raw.clone().requires_grad_(True)
    lp = l * p
    off = raw_t[..., : 2 * lp].reshape(n, s, m, l, p, 2)
    wh = torch.stack((shapes[:, 1], shapes[:, 0]), -1).float()
    loc = ref[:, :, None, :, None, :] + off / wh[None, None, None, :, None, :]
    want = torch.cat((loc.flatten(-3), raw_t[..., 2 * lp :].softmax(-1)), dim=-1)
    want.backward(grad)
    assert (packed - want).abs().max() <= 2e-6 * want.abs().max()
    assert (grad_raw - raw_t.grad).abs().max() <= 1e-5 * raw_t.grad.abs().max()
