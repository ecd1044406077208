"""The reference's own test of this path, run against the drop-in module on a ROCm device.

Mirrors tests/unipercept/nn/layers/test_nn_layers_deform_attn.py of the reference: same import
path (through the shim package), same shapes, same input construction, same tolerances
(`allclose(rtol=1e-2, atol=1e-3)` forward, `gradcheck` in fp64 for seven head dims).  Differences,
all stated here: inputs are seeded; fp16/bf16 are supported on this implementation, so they are
tested instead of xfail-ed; gradcheck runs in fast mode for head dims >= 1025 (the full mode
perturbs every one of ~4 million value elements separately); the reference's forward test of
`MultiScaleFlashAttnFunction` is skipped there (`pytest.skip()` at :78) and runs here; its
gradient test concatenates a 6-d and a 5-d tensor (:190) and cannot run as written, so the packed
input is built as in its forward test (:109).
"""

from __future__ import annotations

import pytest
import torch
from torch.autograd import gradcheck

pytestmark = pytest.mark.gpu

N, M, D = 1, 2, 2
Q, L, P = 2, 2, 2


def _inputs(channels, seed):
    from unipercept.nn.layers.deform_attn import MultiScaleDeformAttnFunction, reference  # noqa: F401

    torch.manual_seed(seed)
    with torch.device("cuda"):
        shapes = torch.as_tensor([(32, 16), (16, 8)], dtype=torch.long)
        level_start_index = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
        S = sum([(H * W).item() for H, W in shapes])
        value = torch.rand(N, S, M, channels) * 0.01
        loc = torch.rand(N, Q, M, L, P, 2)
        attn = torch.rand(N, Q, M, L, P) + 1e-5
        attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    return shapes, level_start_index, value, loc, attn


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32, torch.float64])
@torch.no_grad()
def test_deform_attn_reference(dtype):
    from unipercept.nn.layers.deform_attn import MultiScaleDeformAttnFunction, reference

    shapes, level_start_index, value, loc, attn = _inputs(D, seed=0)
    im2col_step = 8
    value, loc, attn = value.to(dtype), loc.to(dtype), attn.to(dtype)
    # the pure-PyTorch evaluation runs in fp32 for the 16-bit cases (on the same rounded inputs)
    ref_dtype = torch.float32 if dtype in (torch.float16, torch.bfloat16) else dtype
    output_pytorch = (
        reference.deform_attn(value.to(ref_dtype), shapes, loc.to(ref_dtype), attn.to(ref_dtype))
        .detach().cpu()
    )
    output_hip = (
        MultiScaleDeformAttnFunction.apply(value, shapes, level_start_index, loc, attn, im2col_step)
        .detach().cpu().to(ref_dtype)
    )
    assert torch.allclose(output_hip, output_pytorch, rtol=1e-2, atol=1e-3)
    if dtype in (torch.float32, torch.float64):
        assert (output_hip - output_pytorch).abs().max() <= 1e-4  # the fp32 forward bound stated for this op (max-abs 1e-4)


@pytest.mark.parametrize("channels", [30, 32, 64, 71, 1025, 2048, 3096])
def test_deform_attn_gradient(channels):
    from unipercept.nn.layers.deform_attn import MultiScaleDeformAttnFunction

    shapes, level_start_index, value, sampling_locations, attention_weights = _inputs(channels, seed=1)
    im2col_step = 8
    func = MultiScaleDeformAttnFunction.apply
    value.requires_grad = True
    sampling_locations.requires_grad = True
    attention_weights.requires_grad = True
    assert gradcheck(
        func,
        (
            value.double(),
            shapes,
            level_start_index,
            sampling_locations.double(),
            attention_weights.double(),
            im2col_step,
        ),
        fast_mode=channels >= 1025,
    )


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32, torch.float64])
@torch.no_grad()
def test_flash_attn_reference(dtype):
    from unipercept.nn.layers.deform_attn import MultiScaleFlashAttnFunction, reference

    shapes, level_idx, value, loc, attn = _inputs(D, seed=2)
    im2col_step = 8
    value, loc, attn = value.to(dtype), loc.to(dtype), attn.to(dtype)
    ref_dtype = torch.float32 if dtype in (torch.float16, torch.bfloat16) else dtype
    output_pytorch = (
        reference.deform_attn(value.to(ref_dtype), shapes, loc.to(ref_dtype), attn.to(ref_dtype))
        .detach().cpu()
    )
    output_hip = (
        MultiScaleFlashAttnFunction.apply(
            value,
            shapes,
            level_idx,
            torch.cat([loc.flatten(-3), attn.flatten(-2)], dim=-1),
            im2col_step,
        )
        .detach().cpu().to(ref_dtype)
    )
    assert torch.allclose(output_hip, output_pytorch, rtol=1e-2, atol=1e-3)


@pytest.mark.parametrize("channels", [30, 32, 64, 71, 1025, 2048, 3096])
def test_flash_attn_gradient(channels):
    from unipercept.nn.layers.deform_attn import MultiScaleFlashAttnFunction

    torch.manual_seed(3)
    with torch.device("cuda"):
        shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
        level_start_index = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
        S = sum([(H * W).item() for H, W in shapes])
        value = torch.rand(N, S, M, channels) * 0.01
        loc = torch.rand(N, Q, M, L, P, 2)
        attn = torch.rand(N, Q, M, L, P) + 1e-5
        attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    im2col_step = 2
    value.requires_grad = True
    packed = torch.cat([loc.flatten(-3), attn.flatten(-2)], dim=-1).double().requires_grad_(True)
    assert gradcheck(
        MultiScaleFlashAttnFunction.apply,
        (value.double(), shapes, level_start_index, packed, im2col_step),
        fast_mode=channels >= 1025,
    )
