"""Host-side logic that needs no GPU: validation, flags, the layer's arithmetic, the harness."""

from __future__ import annotations

import math

import pytest
import torch

from conftest import level_start_index, make_inputs
from unipercept_amd import _lib, harness
from unipercept_amd.deform_attn import MSDeformAttn, MultiScaleDeformAttnFunction, _function, _ops


def _small():
    return make_inputs(1, 2, 4, 3, [(4, 4), (2, 2)], 2, dtype=torch.float32, seed=1)


def test_cpu_tensors_are_refused_not_silently_computed():
    value, shapes, lsi, loc, attn, _ = _small()
    with pytest.raises(RuntimeError, match="only on a ROCm GPU"):
        MultiScaleDeformAttnFunction.apply(value, shapes, lsi, loc, attn, 8)


@pytest.mark.parametrize(
    "mutate,exc",
    [
        (lambda v, s, i, lo, a: (v[0], s, i, lo, a), ValueError),  # value not 4-d
        (lambda v, s, i, lo, a: (v, s, i, lo[..., :1], a), ValueError),  # last dim of loc != 2
        (lambda v, s, i, lo, a: (v, s, i, lo, a[:, :, :1]), ValueError),  # heads differ
        (lambda v, s, i, lo, a: (v, s[:1], i, lo, a), ValueError),  # level table too short
        (lambda v, s, i, lo, a: (v, s, i[:1], lo, a), ValueError),
        (lambda v, s, i, lo, a: (v.to(torch.int32), s, i, lo, a), TypeError),
    ],
)
def test_shape_and_dtype_validation(mutate, exc):
    value, shapes, lsi, loc, attn, _ = _small()
    with pytest.raises(exc):
        _ops._validate(*mutate(value, shapes, lsi, loc, attn))


def test_level_table_check():
    shapes = torch.tensor([[4, 4], [2, 2]])
    lsi = level_start_index(shapes)
    host = _ops._check_levels(shapes, lsi, shapes, lsi, 20)
    assert list(host) == [4, 4, 2, 2]  # host copy of the (H, W) pairs, handed to the backward
    with pytest.raises(ValueError, match="cover"):
        _ops._check_levels(shapes, lsi, shapes, lsi, 21)
    bad = torch.tensor([0, 15])
    with pytest.raises(ValueError, match="prefix sum"):
        _ops._check_levels(shapes, bad, shapes, bad, 20)
    zero, z = torch.tensor([[4, 0], [2, 2]]), torch.tensor([0, 0])
    with pytest.raises(ValueError, match="positive"):
        _ops._check_levels(zero, z, zero, z, 4)


def test_im2col_step_is_validated_before_any_launch():
    value, shapes, lsi, loc, attn, _ = _small()
    with pytest.raises(RuntimeError, match="im2col_step"):
        MultiScaleDeformAttnFunction.apply(value, shapes, lsi, loc, attn, 0)
    v3 = value.expand(3, -1, -1, -1)
    with pytest.raises(RuntimeError, match="im2col_step"):
        MultiScaleDeformAttnFunction.apply(v3, shapes, lsi, loc, attn, 2)


def test_backward_flags(monkeypatch):
    value, shapes, lsi, loc, attn, _ = _small()
    for k in ("MSDA_FORCE_GENERIC", "MSDA_DETERMINISTIC", "MSDA_ATOMIC_SCATTER"):
        monkeypatch.delenv(k, raising=False)
    assert _function.backward_flags(value, loc) == 0
    # fp64 (gradcheck) on small problems takes the reproducible path
    assert _function.backward_flags(value.double(), loc) == _lib.FLAG_DETERMINISTIC
    monkeypatch.setenv("MSDA_DETERMINISTIC", "0")
    assert _function.backward_flags(value.double(), loc) == 0
    monkeypatch.setenv("MSDA_DETERMINISTIC", "1")
    assert _function.backward_flags(value, loc) == _lib.FLAG_DETERMINISTIC
    monkeypatch.delenv("MSDA_DETERMINISTIC")
    monkeypatch.setenv("MSDA_FORCE_GENERIC", "1")
    monkeypatch.setenv("MSDA_ATOMIC_SCATTER", "1")
    assert _function.backward_flags(value, loc) == _lib.FLAG_GENERIC | _lib.FLAG_ATOMIC_SCATTER
    assert _function.forward_flags() == _lib.FLAG_GENERIC


def test_deterministic_mode_has_a_work_limit(monkeypatch):
    """torch.use_deterministic_algorithms(True) routes small problems to the reproducible path;
    at benchmark sizes (where that path would run for hours) it raises like a PyTorch op without
    a deterministic kernel, or warns and keeps the binned path under warn_only."""
    for k in ("MSDA_FORCE_GENERIC", "MSDA_DETERMINISTIC", "MSDA_ATOMIC_SCATTER", "MSDA_TILE_1D"):
        monkeypatch.delenv(k, raising=False)
    value, shapes, lsi, loc, attn, _ = _small()
    big_value = torch.empty((8, 43008, 8, 32), device="meta")
    big_loc = torch.empty((8, 43008, 8, 3, 4, 2), device="meta")
    was, was_warn = torch.are_deterministic_algorithms_enabled(), torch.is_deterministic_algorithms_warn_only_enabled()
    try:
        torch.use_deterministic_algorithms(True)
        assert _function.backward_flags(value, loc) == _lib.FLAG_DETERMINISTIC
        with pytest.raises(RuntimeError, match="deterministic"):
            _function.backward_flags(big_value, big_loc)
        torch.use_deterministic_algorithms(True, warn_only=True)
        with pytest.warns(UserWarning, match="deterministic"):
            assert _function.backward_flags(big_value, big_loc) == 0
        monkeypatch.setenv("MSDA_DETERMINISTIC", "1")  # explicit request: honoured at any size
        assert _function.backward_flags(big_value, big_loc) == _lib.FLAG_DETERMINISTIC
        monkeypatch.setenv("MSDA_DETERMINISTIC", "0")
        assert _function.backward_flags(big_value, big_loc) == 0
    finally:
        torch.use_deterministic_algorithms(was, warn_only=was_warn)
    monkeypatch.delenv("MSDA_DETERMINISTIC")
    monkeypatch.setenv("MSDA_TILE_1D", "1")
    assert _function.backward_flags(value, loc) == _lib.FLAG_TILE_1D


def test_fake_kernels_give_shapes_without_running():
    value, shapes, lsi, loc, attn, grad_out = _small()
    to_meta = lambda t: t.to("meta")
    out = torch.ops.unipercept.ms_deform_attn_fwd(
        to_meta(value), to_meta(shapes), to_meta(lsi), to_meta(loc), to_meta(attn), 0
    )
    assert out.shape == (1, 3, 8) and out.dtype == torch.float32
    gv, gl, ga = torch.ops.unipercept.ms_deform_attn_bwd(
        to_meta(grad_out), to_meta(value), to_meta(shapes), to_meta(lsi), to_meta(loc),
        to_meta(attn), 0,
    )
    assert gv.shape == value.shape and gl.shape == loc.shape and ga.shape == attn.shape


def test_shim_module_path():
    import unipercept.nn.layers.deform_attn as shim

    assert shim.MultiScaleDeformAttnFunction is MultiScaleDeformAttnFunction
    # the reference's test module imports these three names from the package (:6-10)
    from unipercept.nn.layers.deform_attn import (  # noqa: F401
        MultiScaleDeformAttnFunction as _a,
        MultiScaleFlashAttnFunction as _b,
        reference as _c,
    )
    assert callable(shim.reference.deform_attn)


# ---- layer --------------------------------------------------------------------------------------
def test_layer_parameter_count_and_init():
    layer = MSDeformAttn(d_model=256, n_levels=3, n_heads=8, n_points=4)
    n = sum(p.numel() for p in layer.parameters())
    assert n == 205_600 == harness.layer_param_count(256, 8, 3, 4)
    assert harness.layer_param_count(256, 8, 4, 4) == 230_272
    assert harness.layer_param_count(256, 8, 4, 8) == 328_960
    bias = layer.sampling_offsets.bias.view(8, 3, 4, 2)
    # ring of directions, scaled by the point index; head 0 points along +x
    assert torch.allclose(bias[0, 0, :, 0], torch.tensor([1.0, 2.0, 3.0, 4.0]))
    assert torch.allclose(bias[0, 0, :, 1], torch.zeros(4), atol=1e-6)
    assert torch.allclose(bias[:, 0], bias[:, 2])
    assert float(layer.attention_weights.weight.detach().abs().max()) == 0.0


@pytest.mark.parametrize("ref_dim", [2, 4])
def test_layer_sampling_inputs(ref_dim):
    torch.manual_seed(0)
    levels = [(6, 8), (3, 4)]
    shapes = torch.tensor(levels)
    s = 60
    layer = MSDeformAttn(d_model=32, n_levels=2, n_heads=4, n_points=3).double()
    nn_init = torch.nn.init.normal_
    nn_init(layer.sampling_offsets.weight, std=0.1)
    nn_init(layer.attention_weights.weight, std=0.1)
    query = torch.randn(2, 5, 32, dtype=torch.float64)
    feats = torch.randn(2, s, 32, dtype=torch.float64)
    ref = torch.rand(2, 5, 2, ref_dim, dtype=torch.float64)
    mask = torch.zeros(2, s, dtype=torch.bool)
    mask[:, -7:] = True
    value, loc, w = layer.sampling_inputs(query, ref, feats, shapes, mask)
    assert value.shape == (2, s, 4, 8) and loc.shape == (2, 5, 4, 2, 3, 2) and w.shape == (2, 5, 4, 2, 3)
    assert float(value[:, -7:].abs().max()) == 0.0
    assert torch.allclose(w.sum((-1, -2)), torch.ones(2, 5, 4, dtype=torch.float64))
    off = layer.sampling_offsets(query).view(2, 5, 4, 2, 3, 2)
    for lvl, (h, wd) in enumerate(levels):
        for pt in range(3):
            if ref_dim == 2:
                want_x = ref[:, :, lvl, 0][:, :, None] + off[:, :, :, lvl, pt, 0] / wd
                want_y = ref[:, :, lvl, 1][:, :, None] + off[:, :, :, lvl, pt, 1] / h
            else:
                want_x = ref[:, :, lvl, 0][:, :, None] + off[:, :, :, lvl, pt, 0] / 3 * ref[:, :, lvl, 2][:, :, None] * 0.5
                want_y = ref[:, :, lvl, 1][:, :, None] + off[:, :, :, lvl, pt, 1] / 3 * ref[:, :, lvl, 3][:, :, None] * 0.5
            assert torch.allclose(loc[:, :, :, lvl, pt, 0], want_x)
            assert torch.allclose(loc[:, :, :, lvl, pt, 1], want_y)


def test_layer_with_injected_op_matches_reference_formulation():
    """The layer's op slot takes any callable with the Function's signature."""
    from oracle import grid_sample_ref

    torch.manual_seed(1)
    levels = [(4, 6), (2, 3)]
    shapes = torch.tensor(levels)
    lsi = level_start_index(shapes)
    layer = MSDeformAttn(d_model=16, n_levels=2, n_heads=2, n_points=2)
    layer.op = lambda v, sh, li, lo, a, step: grid_sample_ref.deform_attn(v, sh, lo, a)
    query = torch.randn(1, 7, 16)
    feats = torch.randn(1, 30, 16)
    ref = torch.rand(1, 7, 2, 2)
    out = layer(query, ref, feats, shapes, lsi)
    value, loc, w = layer.sampling_inputs(query, ref, feats, shapes)
    want = layer.output_proj(grid_sample_ref.deform_attn(value, shapes, loc, w))
    assert torch.allclose(out, want)


# ---- harness ------------------------------------------------------------------------------------
def test_workload_sizes_and_algorithmic_bytes():
    w = harness.WORKLOADS["cityscapes_1024x2048_fp32_b8"]
    assert w.positions == 43008 and w.levels == ((128, 256), (64, 128), (32, 64))
    assert w.sampled_points(1) == 4_128_768 and w.sampled_points() == 33_030_144
    ab = harness.algorithmic_bytes(w)
    pts = w.sampled_points()
    assert ab["fwd"] / pts == pytest.approx(33.333, abs=1e-3)
    assert ab["fwd_bwd"] / pts == pytest.approx(89.333, abs=1e-3)
    assert ab["fwd"] == pytest.approx(1.101e9, rel=1e-3) and ab["bwd"] == pytest.approx(1.850e9, rel=1e-3)
    bf = harness.algorithmic_bytes(harness.WORKLOADS["cityscapes_1024x2048_bf16_b8"])
    assert bf["fwd_bwd"] / pts == pytest.approx(44.667, abs=1e-3)
    mixed = harness.algorithmic_bytes(harness.WORKLOADS["cityscapes_1024x2048_bf16_locf32_b8"])
    assert mixed["fwd_bwd"] / pts == pytest.approx(62.667, abs=1e-3)
    coco = harness.WORKLOADS["coco_1024x1024_L4_bf16_b16"]
    assert coco.positions == 21760 and coco.sampled_points(1) == 2_785_280
    assert harness.algorithmic_bytes(coco, 1)["fwd_bwd"] / coco.sampled_points(1) == pytest.approx(38.0)
    dvps = harness.WORKLOADS["dvps_1024x2048_L4_P8_fp32_b8"]
    assert dvps.positions == 43520 and dvps.sampled_points(1) == 11_141_120
    assert harness.algorithmic_bytes(dvps, 1)["fwd_bwd"] / dvps.sampled_points(1) == pytest.approx(56.0)
    cpu = harness.WORKLOADS["cpuref_512x1024_fp32_b1"]
    assert cpu.positions == 10752 and cpu.sampled_points() == 1_032_192


@pytest.mark.parametrize("dist", ["R", "U"])
def test_synthetic_inputs(dist):
    w = harness.Workload("tiny", ((8, 16), (4, 8)), 2, 8, 3, 2, "f32")
    value, shapes, lsi, loc, attn, grad_out = harness.synth_inputs(w, "cpu", seed=5, dist=dist)
    s = 160
    assert value.shape == (2, s, 2, 8) and loc.shape == (2, s, 2, 2, 3, 2)
    assert attn.shape == (2, s, 2, 2, 3) and grad_out.shape == (2, s, 16)
    assert shapes.tolist() == [[8, 16], [4, 8]] and lsi.tolist() == [0, 128]
    assert torch.allclose(attn.sum((-1, -2)), torch.ones(2, s, 2), atol=1e-5)
    again = harness.synth_inputs(w, "cpu", seed=5, dist=dist)
    assert torch.equal(loc, again[3]) and torch.equal(value, again[0])
    other = harness.synth_inputs(w, "cpu", seed=6, dist=dist)
    assert not torch.equal(loc, other[3])
    if dist == "U":
        assert 0.0 <= float(loc.min()) and float(loc.max()) <= 1.0
    else:
        # offsets are a few pixels around the query's own position
        ref = harness.encoder_reference_points(w, "cpu")
        d = (loc - ref.view(1, s, 1, 1, 1, 2)).abs()
        assert float(d[..., 0].mean()) < 6.0 / 8  # sigma <= 6 px on a level >= 8 px wide


def test_reference_points_are_pixel_centres():
    w = harness.Workload("tiny", ((2, 4), (1, 2)), 1, 4, 1, 1, "f32")
    ref = harness.encoder_reference_points(w, "cpu")
    assert ref.shape == (10, 2)
    assert ref[0].tolist() == pytest.approx([0.125, 0.25])
    assert ref[5].tolist() == pytest.approx([0.375, 0.75])
    assert ref[8].tolist() == pytest.approx([0.25, 0.5])


def test_shard_batch():
    assert list(harness.shard_batch(64, 8, 0)) == list(range(8))
    assert list(harness.shard_batch(64, 8, 7)) == list(range(56, 64))
    seen = sorted(i for r in range(4) for i in harness.shard_batch(16, 4, r))
    assert seen == list(range(16))
    with pytest.raises(ValueError, match="divisible"):
        harness.shard_batch(10, 4, 0)
    with pytest.raises(ValueError):
        harness.shard_batch(8, 4, 4)


def test_flat_bucket_views_parameters():
    layer = MSDeformAttn(d_model=16, n_levels=2, n_heads=2, n_points=2)
    bucket = harness.FlatGradBucket.for_parameters(layer.parameters())
    assert bucket.flat.numel() == sum(p.numel() for p in layer.parameters())
    layer.value_proj.weight.grad.fill_(2.0)
    assert float(bucket.flat.sum()) == 2.0 * layer.value_proj.weight.numel()
    assert math.isfinite(harness.roofline_fraction(1e9, 1e-3))


def test_packaging_build_hook_is_current():
    """`setup.py build_hip` (the packaging entry of the HIP build) compiles the same source with
    the same flags as __graft_entry__.build_library and sees an up-to-date library as such."""
    import importlib.util
    import subprocess
    import sys
    from pathlib import Path

    import __graft_entry__ as entry

    root = Path(entry.__file__).resolve().parent
    spec = importlib.util.spec_from_file_location("_setup_under_test", root / "setup.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = root / "unipercept_amd" / "libmsda.so"
    cmd = mod.hip_command(out)
    assert cmd[1:] == [*entry.HIP_FLAGS, "-o", str(out), str(root / "unipercept_amd" / "csrc" / "msda.hip")]
    entry.build_library()
    r = subprocess.run([sys.executable, "setup.py", "-q", "build_hip"], cwd=root, capture_output=True, text=True)
    assert r.returncode == 0 and "up to date" in r.stdout, r.stdout + r.stderr
