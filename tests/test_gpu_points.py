"""Sampling-point epilogue (raw offsets + logits -> packed locations + softmax weights) and the
layer path built on it, against the plain PyTorch formulation of the canonical layer."""

from __future__ import annotations

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


def test_fused_layer_matches_unfused():
    """MSDeformAttn with the single projection + point epilogue + packed kernels gives the same
    output and parameter gradients as the separate-projection route."""
    from unipercept_amd.deform_attn import MSDeformAttn

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
    orig = _function._ops.ms_flash_attn_fwd

    def spy(value, shapes_, lsi_, packed, flags):
        seen["packed_dtype"], seen["value_dtype"] = packed.dtype, value.dtype
        return orig(value, shapes_, lsi_, packed, flags)

    want = layer(feats, ref, feats, shapes, lsi)
    outs = {}
    for fused in (True, False):
        layer.fused_points = fused
        with torch.autocast("cuda", dtype=torch.bfloat16):
            if fused:
                _function._ops.ms_flash_attn_fwd = spy
            try:
                outs[fused] = layer(feats, ref, feats, shapes, lsi).float()
            finally:
                _function._ops.ms_flash_attn_fwd = orig
    assert seen == {"packed_dtype": torch.float32, "value_dtype": torch.bfloat16}
    scale = want.abs().max().item()
    err_fused = (outs[True] - want).abs().max().item() / scale
    err_unfused = (outs[False] - want).abs().max().item() / scale
    assert err_fused <= 2.0**-5 and err_fused <= err_unfused * 1.5 + 2.0**-8, (err_fused, err_unfused)


def test_encoder_stack_hip_matches_oracle_injected():
    """Two encoder layers with the HIP ops (fused point path) against the same weights run on
    the CPU with the oracle as the sampling op: outputs and all gradients."""
    import copy

    from oracle import grid_sample_ref
    from unipercept_amd.deform_attn import MSDeformAttnEncoder, flatten_levels

    torch.manual_seed(5)
    enc = MSDeformAttnEncoder(2, d_model=64, d_ffn=128, n_levels=3, n_heads=2, n_points=4)
    for layer in enc.layers:
        torch.nn.init.normal_(layer.self_attn.sampling_offsets.weight, std=0.02)
        torch.nn.init.normal_(layer.self_attn.attention_weights.weight, std=0.1)
    cpu = copy.deepcopy(enc)
    for layer in cpu.layers:
        layer.self_attn.op = lambda v, sh, li, lo, a, step: grid_sample_ref.deform_attn(v, sh, lo, a)
    enc = enc.cuda()
    feats = [torch.randn(2, 64, 16, 24), torch.randn(2, 64, 8, 12), torch.randn(2, 64, 4, 6)]
    target = torch.randn(2, 16 * 24 + 8 * 12 + 4 * 6, 64)

    def run(model, dev):
        model.zero_grad()
        src, shapes, lsi, shape_list, pos = flatten_levels([f.to(dev) for f in feats], position_feats=32)
        src.requires_grad_(True)
        out = model(src, shapes, lsi, pos=pos, spatial_shapes_list=shape_list)
        (out * target.to(dev)).sum().backward()
        return [out.detach().cpu(), src.grad.cpu()] + [p.grad.cpu() for p in model.parameters()]

    got, want = run(enc, "cuda"), run(cpu, "cpu")
    assert all(l.self_attn.can_fuse_points(torch.empty(2, 5, 64, device="cuda"), torch.empty(2, 5, 3, 2),
                                           torch.empty(2, 9, 2, 32, device="cuda")) for l in enc.layers)
    for g, w_ in zip(got, want):
        scale = w_.abs().max().item() + 1e-30
        assert (g - w_).abs().max().item() / scale <= 1e-3
