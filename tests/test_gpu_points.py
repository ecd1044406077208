"""Sampling-point epilogue (raw offsets + logits -> packed locations + softmax weights) and the
layer path built on it, against the plain PyTorch formulation of the canonical layer."""

from __future__ import annotations

import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _torch_points(raw, ref, shapes):
    """The unfused sequence: softmax over the logits, offsets / (W, H) + reference point, concat."""
    n, lq, m, width = raw.shape
    levels = shapes.shape[0]
    lp = width // 3
    off = raw[..., : 2 * lp].reshape(n, lq, m, levels, lp // levels, 2)
    wh = torch.stack((shapes[:, 1], shapes[:, 0]), -1).to(raw.dtype)
    loc = ref[:, :, None, :, None, :] + off / wh[None, None, None, :, None, :]
    attn = raw[..., 2 * lp :].softmax(-1)
    return torch.cat((loc.flatten(-3), attn), dim=-1)


CASES = {
    # name: (n, lq, m, levels, p)
    "enc_L3P4": (2, 700, 8, [(32, 64), (16, 32), (8, 16)], 4),
    "L4P8": (1, 333, 8, [(16, 32), (8, 16), (4, 8), (2, 4)], 8),
    "odd_L3P3": (1, 515, 5, [(17, 23), (9, 12), (1, 7)], 3),
    "one_row": (1, 1, 1, [(4, 4)], 1),
}


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", ["enc_L3P4", "odd_L3P3"])
def test_points_16bit_raw_with_fp32_reference_points(case, dtype):
    """A 16-bit projection output next to fp32 reference points: locations and weights come out
    in fp32 (computed in fp32 from the rounded offsets and logits), the gradient of `raw` in the
    16-bit type."""
    from unipercept_amd.deform_attn import SamplingPointsFunction

    n, lq, m, levels, p = CASES[case]
    g = torch.Generator(device="cuda").manual_seed(9)
    shapes = torch.as_tensor(levels, dtype=torch.long, device="cuda")
    l = len(levels)
    raw = (torch.randn(n, lq, m, 3 * l * p, generator=g, device="cuda") * 3).to(dtype)
    ref = torch.rand(n, lq, l, 2, generator=g, device="cuda")
    grad = torch.randn(n, lq, m, 3 * l * p, generator=g, device="cuda")
    raw_a = raw.clone().requires_grad_(True)
    ref_a = ref.clone().requires_grad_(True)
    got = SamplingPointsFunction.apply(raw_a, ref_a, shapes)
    assert got.dtype == torch.float32 and raw_a.dtype == dtype
    got.backward(grad)
    assert raw_a.grad.dtype == dtype and ref_a.grad.dtype == torch.float32
    raw_b = raw.double().requires_grad_(True)
    ref_b = ref.double().requires_grad_(True)
    want = _torch_points(raw_b, ref_b, shapes)
    want.backward(grad.double())
    rel16 = {torch.bfloat16: 2**-8, torch.float16: 2**-10}[dtype]
    for name, a, b, rel in (
        ("packed", got, want, 2e-6), ("grad_raw", raw_a.grad, raw_b.grad, rel16),
        ("grad_ref", ref_a.grad, ref_b.grad, 2e-6),
    ):
        err = (a.double() - b).abs().max().item()
        assert err / (b.abs().max().item() + 1e-30) <= rel, (name, err)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", sorted(CASES))
def test_points_forward_backward_match_torch(case, dtype):
    from unipercept_amd.deform_attn import SamplingPointsFunction

    n, lq, m, levels, p = CASES[case]
    g = torch.Generator(device="cuda").manual_seed(7)
    shapes = torch.as_tensor(levels, dtype=torch.long, device="cuda")
    l = len(levels)
    raw = (torch.randn(n, lq, m, 3 * l * p, generator=g, device="cuda") * 3).to(dtype)
    ref = torch.rand(n, lq, l, 2, generator=g, device="cuda").to(dtype)
    grad = torch.randn(n, lq, m, 3 * l * p, generator=g, device="cuda").to(dtype)

    raw_a = raw.clone().requires_grad_(True)
    ref_a = ref.clone().requires_grad_(True)
    got = SamplingPointsFunction.apply(raw_a, ref_a, shapes)
    got.backward(grad)

    # reference in fp64 from the same (rounded) inputs
    raw_b = raw.double().requires_grad_(True)
    ref_b = ref.double().requires_grad_(True)
    want = _torch_points(raw_b, ref_b, shapes)
    want.backward(grad.double())

    # fp32: a few ulp (1/W by multiplication, expf); 16-bit: one rounding of the result
    rel = {torch.float32: 2e-6, torch.bfloat16: 2**-8, torch.float16: 2**-10}[dtype]
    for name, a, b in (
        ("packed", got, want), ("grad_raw", raw_a.grad, raw_b.grad), ("grad_ref", ref_a.grad, ref_b.grad)
    ):
        err = (a.double() - b).abs().max().item()
        scale = b.abs().max().item() + 1e-30
        if name == "grad_ref" and dtype != torch.float32:
            # a sum of M*P rounded terms
            assert err / scale <= rel * 8, (name, err, scale)
        elif name == "grad_raw" and dtype != torch.float32:
            # the backward reads the softmax weights as the forward stored them (rounded to the
            # 16-bit type) and rounds its own result: two roundings
            assert err / scale <= rel * 2, (name, err, scale)
        else:
            assert err / scale <= rel, (name, err, scale)


@pytest.mark.parametrize("route", ["raw_points", "point_epilogue"])
def test_fused_layer_matches_unfused(route, monkeypatch):
    """MSDeformAttn with the single projection and either the in-kernel softmax / location
    arithmetic or the point epilogue + packed kernels gives the same output and parameter
    gradients as the separate-projection route."""
    from unipercept_amd.deform_attn import MSDeformAttn

    if route == "point_epilogue":
        monkeypatch.setenv("MSDA_NO_RAW_POINTS", "1")

    torch.manual_seed(3)
    levels = [(32, 64), (16, 32), (8, 16)]
    shapes = torch.as_tensor(levels, dtype=torch.long, device="cuda")
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    s = int(shapes.prod(1).sum())
    layer = MSDeformAttn(d_model=256, n_levels=3, n_heads=8, n_points=4).cuda()
    torch.nn.init.normal_(layer.sampling_offsets.weight, std=0.02)
    torch.nn.init.normal_(layer.attention_weights.weight, std=0.05)
    feats = torch.randn(2, s, 256, device="cuda")
    ref = torch.rand(2, s, 3, 2, device="cuda")
    target = torch.randn(2, s, 256, device="cuda")

    def run(fused: bool):
        layer.fused_points = fused
        layer.zero_grad()
        x = feats.clone().requires_grad_(True)
        assert layer.can_fuse_points(x, ref, layer.project_value(x)) == fused
        out = layer(x, ref, x, shapes, lsi)
        (out * target).sum().backward()
        return [out.detach(), x.grad] + [p.grad.clone() for p in layer.parameters()]

    a, b = run(True), run(False)
    for x, y in zip(a, b):
        scale = y.abs().max().item() + 1e-30
        assert (x - y).abs().max().item() / scale <= 2e-4


RAW_CASES = {
    # name: (n, lq, m, d, levels, p): all take MSDA_FLAG_RAW_POINTS (tuned head dim, even L*P <= 32;
    # rows of more than 16 points run as two half rows per query)
    "enc_L4P8": (1, None, 8, 32, [(32, 32), (16, 16), (8, 8), (4, 4)], 8),
    "dec_q301_L3P8": (2, 301, 4, 32, [(32, 64), (16, 32), (8, 16)], 8),
    "enc_L3P4": (2, None, 8, 32, [(32, 64), (16, 32), (8, 16)], 4),
    "enc_L4P4": (1, None, 8, 32, [(32, 32), (16, 16), (8, 8), (4, 4)], 4),
    "dec_q301": (2, 301, 8, 32, [(32, 64), (16, 32), (8, 16)], 4),
    "heads3_d64": (2, 777, 3, 64, [(20, 30), (10, 15), (5, 8)], 2),
}


@pytest.mark.parametrize("dtype", ["f32", "bf16", "bf16_rawf32"])
@pytest.mark.parametrize("case", sorted(RAW_CASES))
def test_raw_points_op(case, dtype):
    """Softmax and reference-point arithmetic inside the attention kernels (raw offsets + logits in,
    gradient of the raw tensor out) against the fp64 PyTorch formulation of the same thing, and
    against the two-step route (point epilogue pass, then the packed-points kernels)."""
    from oracle import grid_sample_ref
    from unipercept_amd.deform_attn import _function

    n, lq, m, d, levels, p = RAW_CASES[case]
    g = torch.Generator(device="cuda").manual_seed(13)
    shapes = torch.as_tensor(levels, dtype=torch.long, device="cuda")
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    s = int(shapes.prod(1).sum())
    lq = s if lq is None else lq
    l = len(levels)
    vdt = torch.float32 if dtype == "f32" else torch.bfloat16
    rdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    value = torch.randn(n, s, m, d, generator=g, device="cuda").to(vdt)
    raw = torch.randn(n, lq, m, 3 * l * p, generator=g, device="cuda")
    raw[..., : 2 * l * p] *= 6.0  # offsets of a few pixels; some points leave the levels
    raw = raw.to(rdt)
    ref = torch.rand(n, lq, l, 2, generator=g, device="cuda") * 1.1 - 0.05
    grad_out = torch.randn(n, lq, m * d, generator=g, device="cuda").to(vdt)
    assert _function._ops.raw_points_supported(value, raw, l)

    def fused():
        v, r, rp = (t.clone().requires_grad_(True) for t in (value, raw, ref))
        vhm = _function.HeadMajorValueFunction.apply(v)
        out = _function.MultiScaleRawPointsAttnFunction.apply(vhm, shapes, lsi, r, rp)
        out.backward(grad_out)
        return out.detach(), v.grad, r.grad, rp.grad

    def two_step():
        v, r, rp = (t.clone().requires_grad_(True) for t in (value, raw, ref))
        packed = _function.SamplingPointsFunction.apply(r, rp, shapes)
        out = _function.MultiScaleFlashAttnFunction.apply(v, shapes, lsi, packed, 64)
        out.backward(grad_out)
        return out.detach(), v.grad, r.grad, rp.grad

    def reference64():
        v, r, rp = (t.double().clone().requires_grad_(True) for t in (value, raw, ref))
        packed = _torch_points(r, rp, shapes)
        lp = l * p
        loc = packed[..., : 2 * lp].reshape(n, lq, m, l, p, 2)
        attn = packed[..., 2 * lp :].reshape(n, lq, m, l, p)
        out = grid_sample_ref.deform_attn(v, shapes.tolist(), loc, attn)
        out.backward(grad_out.double())
        return out.detach(), v.grad, r.grad, rp.grad

    got, two, want = fused(), two_step(), reference64()
    # the gradient of an offset is discontinuous where the pixel coordinate crosses an integer
    # (the footprint moves on); fp32 and fp64 disagree about the side for a few of the millions of
    # coordinates here, so those points are left out of the fp64 comparison of grad_raw, and
    # grad_ref (a sum over points) is compared with the two-step route, which does the same fp32
    # arithmetic, and with the sum it must equal
    lp = l * p
    wh = torch.stack((shapes[:, 1], shapes[:, 0]), -1).double()  # (L, 2) as (W, H)
    off = raw.double()[..., : 2 * lp].reshape(n, lq, m, l, p, 2)
    pix = (ref.double()[:, :, None, :, None, :] + off / wh[None, None, None, :, None, :]) * wh[None, None, None, :, None, :] - 0.5
    dist = (pix - pix.round()).abs()
    near = ((dist > 0) & (dist < 1e-3)).any(-1, keepdim=True).expand_as(off).reshape(n, lq, m, 2 * lp)
    keep = torch.cat((~near, torch.ones(n, lq, m, lp, dtype=torch.bool, device="cuda")), -1)
    assert near.double().mean().item() < 0.02
    if dtype == "f32":
        tols = {"out": 2e-5, "grad_value": 2e-5, "grad_raw": 2e-4, "grad_ref": 2e-4}
    else:  # 16-bit value: one rounding of out / grad_value; grad_raw is rounded when raw is 16-bit
        tols = {"out": 2.0**-7, "grad_value": 2.0**-7, "grad_raw": 2.0**-6, "grad_ref": 2.0**-6}
    for name, a, b, w_ in zip(("out", "grad_value", "grad_raw", "grad_ref"), got, two, want):
        scale = w_.abs().max().item() + 1e-30
        tol = tols[name]
        if name == "grad_raw":
            assert ((a.double() - w_).abs() * keep).max().item() / scale <= tol, (name, "vs fp64")
        elif name != "grad_ref":
            assert (a.double() - w_).abs().max().item() / scale <= tol, (name, "vs fp64")
        assert (a.double() - b.double()).abs().max().item() / scale <= tol, (name, "vs two-step")
    g_off = got[2].double()[..., : 2 * lp].reshape(n, lq, m, l, p, 2)
    sum_ref = (g_off * wh[None, None, None, :, None, :]).sum(dim=(2, 4))
    scale = sum_ref.abs().max().item() + 1e-30
    assert (got[3].double() - sum_ref).abs().max().item() / scale <= (1e-5 if dtype != "bf16" else 2.0**-6)


def test_value_epilogue_head_major_with_padding_mask():
    """The value projection's epilogue pass: per-head layout and padding mask together."""
    from unipercept_amd.deform_attn import _function

    g = torch.Generator(device="cuda").manual_seed(11)
    for dtype in (torch.float32, torch.bfloat16):
        value = torch.randn(2, 777, 8, 32, generator=g, device="cuda").to(dtype)
        mask = torch.rand(2, 777, generator=g, device="cuda") < 0.3
        got = torch.ops.unipercept.ms_deform_attn_value_head_major(value, mask)
        want = value.masked_fill(mask[..., None, None], 0.0).permute(0, 2, 1, 3).contiguous()
        assert torch.equal(got, want)
        assert torch.equal(torch.ops.unipercept.ms_deform_attn_value_head_major(value), value.permute(0, 2, 1, 3).contiguous())
    v = torch.randn(2, 50, 8, 32, generator=g, device="cuda", requires_grad=True)
    mask = torch.rand(2, 50, generator=g, device="cuda") < 0.3
    out = _function.HeadMajorValueFunction.apply(v, mask)
    grad = torch.randn(2, 50, 8, 32, generator=g, device="cuda")
    out.backward(grad.permute(0, 2, 1, 3))  # the op hands its position-major gradient on as a view
    assert torch.equal(v.grad, grad.masked_fill(mask[..., None, None], 0.0))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_layer_with_padding_mask_matches_unfused(dtype):
    """Layer with a padding mask: the fused route (one projection for offsets and logits, point
    epilogue, value epilogue with the mask, packed kernels on the caller-made per-head copy)
    against the plain route (masked_fill, separate projections, op-made copy)."""
    from unipercept_amd.deform_attn import MSDeformAttn

    torch.manual_seed(7)
    levels = [(32, 64), (16, 32), (8, 16)]
    shapes = torch.as_tensor(levels, dtype=torch.long, device="cuda")
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    s = int(shapes.prod(1).sum())
    layer = MSDeformAttn(d_model=256, n_levels=3, n_heads=8, n_points=4).cuda().to(dtype)
    torch.nn.init.normal_(layer.sampling_offsets.weight, std=0.02)
    torch.nn.init.normal_(layer.attention_weights.weight, std=0.05)
    feats = torch.randn(2, s, 256, device="cuda").to(dtype)
    ref = torch.rand(2, s, 3, 2, device="cuda")
    target = torch.randn(2, s, 256, device="cuda").to(dtype)
    mask = torch.zeros(2, s, dtype=torch.bool, device="cuda")
    mask[1, s // 3 :] = True

    def run(fused: bool):
        layer.fused_points = fused
        layer.zero_grad()
        x = feats.clone().requires_grad_(True)
        out = layer(x, ref, x, shapes, lsi, mask)
        (out * target).sum().backward()
        return [out.detach(), x.grad] + [p.grad.clone() for p in layer.parameters()]

    a, b = run(True), run(False)
    tol = 2e-4 if dtype == torch.float32 else 2.0**-5
    for x, y in zip(a, b):
        scale = y.float().abs().max().item() + 1e-30
        assert (x.float() - y.float()).abs().max().item() / scale <= tol


def test_fused_layer_keeps_fp32_locations_under_autocast():
    """Under bf16 autocast the projections run in bf16; the fused path must still hand the
    kernels fp32 locations and weights (a bf16 location near 1.0 has a resolution of 1/256, a
    pixel on the 256-wide level used here).  Both routes against an fp32 run of the same layer:
    the fused one may not be further off than the unfused one (plus rounding noise)."""
    from unipercept_amd.deform_attn import MSDeformAttn, _function

    torch.manual_seed(5)
    levels = [(64, 256), (32, 128), (16, 64)]
    shapes = torch.as_tensor(levels, dtype=torch.long, device="cuda")
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    s = int(shapes.prod(1).sum())
    layer = MSDeformAttn(d_model=256, n_levels=3, n_heads=8, n_points=4).cuda()
    torch.nn.init.normal_(layer.sampling_offsets.weight, std=0.02)
    feats = torch.randn(1, s, 256, device="cuda")
    ref = torch.rand(1, s, 3, 2, device="cuda") * 0.2 + 0.8  # near 1.0: coarsest bf16 steps
    seen = {}
    orig_raw, orig_flash = _function._ops.ms_deform_raw_attn_fwd, _function._ops.ms_flash_attn_fwd

    def spy_raw(value, shapes_, lsi_, raw, ref_points, flags):
        # the kernels compute the locations in fp32 from the (possibly 16-bit) offsets and these
        seen["raw"] = (ref_points.dtype, value.dtype)
        return orig_raw(value, shapes_, lsi_, raw, ref_points, flags)

    def spy_flash(value, shapes_, lsi_, packed, flags):
        seen["flash"] = (packed.dtype, value.dtype)
        return orig_flash(value, shapes_, lsi_, packed, flags)

    want = layer(feats, ref, feats, shapes, lsi)
    outs = {}
    _function._ops.ms_deform_raw_attn_fwd, _function._ops.ms_flash_attn_fwd = spy_raw, spy_flash
    try:
        for route in ("raw", "epilogue", "unfused"):
            layer.fused_points = route != "unfused"
            if route == "epilogue":
                os.environ["MSDA_NO_RAW_POINTS"] = "1"
            try:
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    outs[route] = layer(feats, ref, feats, shapes, lsi).float()
            finally:
                os.environ.pop("MSDA_NO_RAW_POINTS", None)
    finally:
        _function._ops.ms_deform_raw_attn_fwd, _function._ops.ms_flash_attn_fwd = orig_raw, orig_flash
    assert seen == {"raw": (torch.float32, torch.bfloat16), "flash": (torch.float32, torch.bfloat16)}
    scale = want.abs().max().item()
    err = {k: (v - want).abs().max().item() / scale for k, v in outs.items()}
    for route in ("raw", "epilogue"):
        assert err[route] <= 2.0**-5 and err[route] <= err["unfused"] * 1.5 + 2.0**-8, err


@pytest.mark.parametrize("mode", ["f32", "bf16", "autocast"])
def test_token_linear_matches_linear(mode):
    """The projections' sliced weight-gradient GEMM against nn.functional.linear."""
    from unipercept_amd.deform_attn.layer import _split_k, token_linear

    torch.manual_seed(1)
    tokens, c, o = 2 * 43008, 256, 288
    assert _split_k(tokens) == 64 and _split_k(1000) == 0
    dt = torch.bfloat16 if mode == "bf16" else torch.float32
    x = torch.randn(2, 43008, c, device="cuda", dtype=dt)
    w = (torch.randn(o, c, device="cuda") * 0.05).to(dt)
    b = torch.randn(o, device="cuda").to(dt)
    g = torch.randn(2, 43008, o, device="cuda", dtype=dt)

    def run(fn):
        xx, ww, bb = (t.clone().requires_grad_(True) for t in (x, w, b))
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode == "autocast"):
            out = fn(xx, ww, bb)
        out.backward(g.to(out.dtype))
        return out, xx.grad, ww.grad, bb.grad

    got, want = run(token_linear), run(torch.nn.functional.linear)
    tol = 1e-5 if mode == "f32" else 2.0**-6
    for a, r in zip(got, want):
        assert a.dtype == r.dtype and a.shape == r.shape
        assert (a.float() - r.float()).abs().max().item() <= tol * (r.float().abs().max().item() + 1e-30)
