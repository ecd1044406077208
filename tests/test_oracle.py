"""The CPU oracles against the golden vectors (and against each other)."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from conftest import golden_cases, load_golden, make_inputs
from oracle import cpu as oracle_cpu
from oracle import grid_sample_ref


def _tol(case):
    return (1e-6, 1e-5) if case.endswith("f32") else (1e-12, 1e-11)


@pytest.mark.parametrize("case", golden_cases())
def test_c_oracle_matches_golden(case):
    z = load_golden(case)
    fa, ba = _tol(case)
    out = oracle_cpu.forward(z["value"], z["shapes"], z["loc"], z["attn"])
    gv, gl, ga = oracle_cpu.backward(z["grad_out"], z["value"], z["shapes"], z["loc"], z["attn"])
    assert np.abs(out - z["out"]).max() <= fa
    assert np.abs(gv - z["grad_value"]).max() <= ba
    assert np.abs(gl - z["grad_loc"]).max() <= ba
    assert np.abs(ga - z["grad_attn"]).max() <= ba


@pytest.mark.parametrize("case", golden_cases())
def test_torch_oracle_matches_golden(case):
    z = load_golden(case)
    fa, ba = _tol(case)
    t = {k: torch.from_numpy(v) for k, v in z.items()}
    out, gv, gl, ga = grid_sample_ref.deform_attn_with_grads(
        t["value"], t["shapes"], t["loc"], t["attn"], t["grad_out"]
    )
    assert (out - t["out"]).abs().max() <= fa
    assert (gv - t["grad_value"]).abs().max() <= ba
    assert (gl - t["grad_loc"]).abs().max() <= ba
    assert (ga - t["grad_attn"]).abs().max() <= ba


@pytest.mark.parametrize("case", golden_cases())
def test_product_reference_module_matches_golden(case):
    """`reference.deform_attn` of the drop-in package is the same function as the oracle."""
    from unipercept_amd.deform_attn import reference

    z = load_golden(case)
    fa, ba = _tol(case)
    v = torch.from_numpy(z["value"]).requires_grad_(True)
    lo = torch.from_numpy(z["loc"]).requires_grad_(True)
    a = torch.from_numpy(z["attn"]).requires_grad_(True)
    out = reference.deform_attn(v, torch.from_numpy(z["shapes"]), lo, a)
    gv, gl, ga = torch.autograd.grad(out, (v, lo, a), torch.from_numpy(z["grad_out"]))
    assert np.abs(out.detach().numpy() - z["out"]).max() <= fa
    assert np.abs(gv.numpy() - z["grad_value"]).max() <= ba
    assert np.abs(gl.numpy() - z["grad_loc"]).max() <= ba
    assert np.abs(ga.numpy() - z["grad_attn"]).max() <= ba


def test_c_oracle_matches_torch_oracle_mid_size():
    value, shapes, lsi, loc, attn, grad_out = make_inputs(
        2, 4, 16, 97, [(13, 17), (7, 9), (3, 5)], 4, dtype=torch.float64, seed=3
    )
    out, gv, gl, ga = grid_sample_ref.deform_attn_with_grads(value, shapes, loc, attn, grad_out)
    c_out = oracle_cpu.forward(value.numpy(), shapes.numpy(), loc.numpy(), attn.numpy())
    c_gv, c_gl, c_ga = oracle_cpu.backward(
        grad_out.numpy(), value.numpy(), shapes.numpy(), loc.numpy(), attn.numpy()
    )
    assert np.abs(c_out - out.numpy()).max() <= 1e-13
    assert np.abs(c_gv - gv.numpy()).max() <= 1e-12
    assert np.abs(c_gl - gl.numpy()).max() <= 1e-12
    assert np.abs(c_ga - ga.numpy()).max() <= 1e-12


def test_boundary_locations():
    """Pixel centres, level edges and the 1-pixel skirt: zeros padding, align_corners=False."""
    shapes = np.array([[2, 3]], dtype=np.int64)
    value = np.arange(6, dtype=np.float64).reshape(1, 6, 1, 1) + 1.0  # [[1,2,3],[4,5,6]]
    attn = np.ones((1, 1, 1, 1, 1))

    def at(x, y):
        loc = np.array([x, y], dtype=np.float64).reshape(1, 1, 1, 1, 1, 2)
        return float(oracle_cpu.forward(value, shapes, loc, attn)[0, 0, 0])

    assert at(0.5 / 3, 0.5 / 2) == pytest.approx(1.0)  # centre of pixel (0,0)
    assert at(2.5 / 3, 1.5 / 2) == pytest.approx(6.0)  # centre of pixel (1,2)
    assert at(1.0 / 3, 0.5 / 2) == pytest.approx(1.5)  # halfway between (0,0) and (0,1)
    assert at(0.0, 0.0) == pytest.approx(0.25)  # corner of the image: 3 of 4 taps are padding
    assert at(-0.5 / 3, 0.25) == 0.0  # x = -1 px: outside the skirt
    assert at(1.0 + 0.49 / 3, 0.75) == pytest.approx(6.0 * 0.01, abs=1e-9)  # x = W - 0.01 px
    assert at(1.0 + 0.5 / 3, 0.75) == 0.0  # x = W px: outside
    assert at(float("nan"), 0.5) == 0.0  # comparisons with NaN are false: skipped


def test_oracle_rejects_half():
    with pytest.raises(TypeError):
        oracle_cpu.forward(
            np.zeros((1, 1, 1, 1), np.float16), np.array([[1, 1]]),
            np.zeros((1, 1, 1, 1, 1, 2), np.float16), np.zeros((1, 1, 1, 1, 1), np.float16),
        )


def _pointrend_cases():
    from pathlib import Path

    # (the reference's known answer, pointrend_reftest_f32, holds a forward result only)
    return sorted(
        p.stem for p in (Path(__file__).parent / "golden").glob("pointrend_*.npz") if "reftest" not in p.stem
    )


@pytest.mark.parametrize("case", golden_cases() + _pointrend_cases())
def test_oracle_under_sanitizers(case, tmp_path):
    """The C oracles built with AddressSanitizer + UndefinedBehaviorSanitizer (host code only),
    run on every golden vector through a file-driven program whose buffers are heap blocks of
    exactly the case's sizes: no report, and the same numbers as the golden file."""
    import subprocess
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    built = subprocess.run(["make", "-C", str(root / "oracle"), "sanitize_driver"], capture_output=True, text=True)
    assert built.returncode == 0, built.stderr[-2000:]
    z = np.load(root / "tests" / "golden" / f"{case}.npz")
    src, dst = tmp_path / "case.bin", tmp_path / "result.bin"
    if case.startswith("pointrend_"):
        x = z["input"]
        f64 = x.dtype == np.float64
        n, c, h, w = x.shape
        p = z["coords"].shape[1]
        mode = {"bilinear": 0, "nearest": 1}[str(z["mode"])]
        head = np.array([1, int(f64), n, c, h, w, p, mode, int(bool(z["align_corners"]))], dtype=np.int32)
        blocks = [x, z["coords"].astype(x.dtype), z["grad_out"].astype(x.dtype)]
        wants = [z["out"], z["grad_input"], z["grad_coords"]]
        fa, ba = (2e-5, 2e-5) if not f64 else (1e-12, 1e-11)
        scale = [max(1.0, float(np.abs(t).max())) for t in wants]
    else:
        v = z["value"]
        f64 = v.dtype == np.float64
        n, s, m, d = v.shape
        _, q, _, l, p, _ = z["loc"].shape
        shapes = z["shapes"].astype(np.int64)
        hw = shapes[:, 0] * shapes[:, 1]
        lsi = np.concatenate([[0], np.cumsum(hw)[:-1]]).astype(np.int64)
        head = np.array([0, int(f64), n, s, m, d, q, l, p], dtype=np.int32)
        blocks = [shapes, lsi, v, z["loc"], z["attn"], z["grad_out"]]
        wants = [z["out"], z["grad_value"], z["grad_loc"], z["grad_attn"]]
        fa, ba = _tol(case)
        scale = [1.0] * 4
    with open(src, "wb") as fh:
        fh.write(head.tobytes())
        for b in blocks:
            fh.write(np.ascontiguousarray(b).tobytes())
    run = subprocess.run(
        [str(root / "oracle" / "sanitize_driver"), str(src), str(dst)], capture_output=True, text=True,
        env={"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"},
    )
    assert run.returncode == 0 and "ERROR" not in run.stderr and "runtime error" not in run.stderr, run.stderr[-3000:]
    raw = np.fromfile(dst, dtype=wants[0].dtype)
    at = 0
    for i, want in enumerate(wants):
        got = raw[at : at + want.size].reshape(want.shape)
        at += want.size
        assert np.abs(got - want).max() <= (fa if i == 0 else ba) * scale[i], (case, i)
    assert at == raw.size
