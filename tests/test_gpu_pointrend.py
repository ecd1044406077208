"""GPU parity of the point-sampling path: the HIP kernels (through the C ABI and the Python
drop-in) against the golden vectors, the CPU oracle and the reference's own tests
(tests/unipercept/vision/test_pointrend.py of the reference)."""

from __future__ import annotations

import ctypes

import numpy as np
import pytest
import torch

from conftest import load_golden, pointrend_cases

pytestmark = pytest.mark.gpu

# fp32: max-abs 1e-4 forward / 1e-3 backward as for the attention op; observed errors are at the
# 1e-6 level (grad_coords, which is scaled by the map size, at 1e-5)
FWD_ATOL_F32, BWD_ATOL_F32 = 1e-4, 1e-3


def _abi(inp, coords, grad_out, mode, align):
    """forward + backward through the C ABI on cuda:0 (numpy in, numpy out)."""
    from unipercept_amd import _lib

    lib = _lib.load()
    code = {np.dtype("float32"): _lib.F32, np.dtype("float64"): _lib.F64}[inp.dtype]
    mcode = {"bilinear": _lib.SAMPLE_BILINEAR, "nearest": _lib.SAMPLE_NEAREST}[mode]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    x, pc, go = t(inp), t(coords), t(grad_out)
    n, c, h, w = x.shape
    p = pc.shape[1]
    out = torch.empty((n, c, p), dtype=x.dtype, device="cuda")
    gi = torch.full_like(x, float("nan"))
    gc = torch.full_like(pc, float("nan"))
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.msda_point_sample_forward(x.data_ptr(), pc.data_ptr(), out.data_ptr(), n, c, h, w, p,
                                       code, mcode, int(align), st)
    assert rc == 0, rc
    # with the optional point-list workspace, and without it: the same gradients
    nbytes = lib.msda_point_sample_workspace_bytes(n, c, h, w, p, code)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device="cuda")
    rc = lib.msda_point_sample_backward(go.data_ptr(), x.data_ptr(), pc.data_ptr(), gi.data_ptr(),
                                        gc.data_ptr(), ws.data_ptr(), nbytes, n, c, h, w, p, code,
                                        mcode, int(align), st)
    assert rc == 0, rc
    gi2 = torch.full_like(x, float("nan"))
    gc2 = torch.full_like(pc, float("nan"))
    rc = lib.msda_point_sample_backward(go.data_ptr(), x.data_ptr(), pc.data_ptr(), gi2.data_ptr(),
                                        gc2.data_ptr(), None, 0, n, c, h, w, p, code, mcode,
                                        int(align), st)
    assert rc == 0, rc
    torch.cuda.synchronize()
    tol = 1e-12 if x.dtype == torch.float64 else 1e-5
    assert (gi - gi2).abs().max() <= tol * max(1.0, float(gi2.abs().max()))
    # (channel slices add their parts of grad_coords with atomics: last-bit differences)
    assert (gc - gc2).abs().max() <= tol * max(1.0, float(gc2.abs().max()))
    return out.cpu().numpy(), gi.cpu().numpy(), gc.cpu().numpy()


@pytest.mark.parametrize("case", [c for c in pointrend_cases() if "reftest" not in c])
def test_golden_through_abi(case):
    z = load_golden(case)
    mode, align = str(z["mode"]), bool(z["align_corners"])
    out, gi, gc = _abi(z["input"], z["coords"], z["grad_out"], mode, align)
    f64 = z["input"].dtype == np.float64
    fa, ba = (1e-12, 1e-11) if f64 else (FWD_ATOL_F32, BWD_ATOL_F32)
    assert np.abs(out - z["out"]).max() <= fa
    assert np.abs(gi - z["grad_input"]).max() <= ba
    assert np.abs(gc - z["grad_coords"]).max() <= ba
    if not f64:  # tighter: agreement with the fp32 vectors to ~1e-5 of the tensor's scale
        for got, want in ((out, z["out"]), (gi, z["grad_input"]), (gc, z["grad_coords"])):
            assert np.abs(got - want).max() <= 2e-5 * (np.abs(want).max() + 1e-30)


def test_reference_test_sample():
    # tests/unipercept/vision/test_pointrend.py:6-13 of the reference, on the GPU
    from unipercept.vision import pointrend

    src = torch.arange(4).view(1, 1, 2, 2).float().cuda()
    points = torch.tensor([[0.25, 0.25], [0.75, 0.75]]).unsqueeze(0).float().cuda()
    samples = pointrend.sample(src, points, mode="nearest", align_corners=False)
    assert samples.shape == (1, 1, points.shape[1])
    assert samples[0, 0, 0] == 0 and samples[0, 0, 1] == 3
    z = load_golden("pointrend_reftest_f32")
    assert np.array_equal(samples.cpu().numpy(), z["out"])


def test_reference_test_random_points_with_mask():
    # tests/unipercept/vision/test_pointrend.py:27-48 of the reference, on the GPU
    from unipercept.vision import pointrend

    torch.manual_seed(1)
    src = torch.rand(1, 1, 100, 100, device="cuda")
    for mask in (
        torch.tensor([[False, True, False], [False, False, False], [False, False, False]])[None, None],
        torch.arange(20).view(1, 1, 2, 10) == 15,
        torch.rand(1, 1, 100, 100) < 0.5,
        torch.rand(1, 1, 100, 100) < 0.1,
    ):
        mask = mask.cuda()
        points = pointrend.random_points(src, 5, 2, mask)
        assert points.shape == (1, 5, 2) and points.min() >= 0 and points.max() <= 1
        samples = pointrend.sample(mask.float(), points, mode="nearest", align_corners=False)
        assert samples.shape == (1, 1, 5)
        assert torch.all(samples.bool()), samples.tolist()


def test_reference_test_importance():
    # tests/unipercept/vision/test_pointrend.py:51-74 of the reference, on the GPU
    from unipercept.vision import pointrend

    src = torch.arange(6).view(1, 1, 2, 3).float().cuda()
    points = pointrend.random_points_with_importance(
        src, 3, 2, oversample_ratio=300, importance_sample_ratio=1.0, importance_fn=lambda t: t,
        mode="nearest", align_corners=False,
    )
    samples = pointrend.sample(src, points, mode="nearest", align_corners=False)
    assert points.shape == (1, 3, 2) and samples.shape == (1, 1, 3)
    assert samples.min() >= 3.0 and samples.max() <= 5.0


@pytest.mark.parametrize("mode", ["bilinear", "nearest"])
@pytest.mark.parametrize("align", [False, True])
# the last shape is too wide for the LDS bands of the grad_input kernel: global-atomic fallback
@pytest.mark.parametrize(
    "shape", [(3, 1, 37, 53, 1000), (2, 19, 16, 24, 300), (1, 70, 9, 5, 33), (1, 2, 3, 9000, 700)]
)
def test_against_oracle_f32_inputs(shape, align, mode):
    """Mid-size cases against the scalar C oracle in fp64 on the same (fp32-representable)
    inputs, points up to 20 % outside the map on every side; channel counts that split into
    several slices (19, 70) and one that does not."""
    from oracle import cpu as oracle_cpu

    n, c, h, w, p = shape
    g = torch.Generator().manual_seed(n * 1000 + c)
    inp = torch.randn(n, c, h, w, generator=g)
    coords = torch.rand(n, p, 2, generator=g) * 1.4 - 0.2
    grad_out = torch.randn(n, c, p, generator=g)
    out, gi, gc = _abi(inp.numpy(), coords.numpy(), grad_out.numpy(), mode, align)
    d = lambda t: t.double().numpy()
    want = oracle_cpu.point_sample_forward(d(inp), d(coords), mode, align)
    wgi, wgc = oracle_cpu.point_sample_backward(d(grad_out), d(inp), d(coords), mode, align)
    if mode == "nearest":
        # a coordinate within an ulp of a half-pixel boundary may round to the other pixel in
        # fp32: compare where fp32 and fp64 pick the same pixel (all but a handful of points)
        same = (out == want.astype(np.float32)).all(axis=1)
        assert same.mean() > 0.995
        assert not gc.any()
        return
    # an fp32 pixel coordinate resolves 2^-24 of the map size: on the 9000-pixel-wide map the
    # interpolation fraction itself is only good to ~5e-4, whatever the implementation
    coarse = max(1.0, max(h, w) / 512.0)
    assert np.abs(out - want).max() <= FWD_ATOL_F32 * coarse
    assert np.abs(gi - wgi).max() <= BWD_ATOL_F32 * coarse
    # the clamp decides on the fp32 coordinate: compare away from the exact border
    inside = (np.abs(gc - wgc) <= BWD_ATOL_F32 * max(h, w) * coarse) | (gc == 0) | (wgc == 0)
    assert inside.mean() > 0.999
    assert np.abs(gc - wgc)[(gc != 0) & (wgc != 0)].max() <= BWD_ATOL_F32 * max(h, w) * coarse


def test_gradcheck_fp64():
    from unipercept.vision import pointrend

    g = torch.Generator().manual_seed(5)
    inp = torch.randn(2, 3, 6, 7, generator=g, dtype=torch.float64).cuda().requires_grad_(True)
    # keep clear of pixel edges (the interpolant has a kink there) and of the border clamp
    coords = ((torch.randint(0, 5, (2, 9, 2), generator=g) + 0.5 + 0.3 * torch.rand(2, 9, 2, generator=g) + 0.1)
              / torch.tensor([7.0, 6.0])).double().cuda().requires_grad_(True)
    assert torch.autograd.gradcheck(
        lambda a, b: pointrend.sample(a, b), (inp, coords), eps=1e-6, atol=1e-6, rtol=1e-5, nondet_tol=1e-12
    )
    assert torch.autograd.gradcheck(
        lambda a, b: pointrend.sample(a, b, align_corners=True), (inp, coords), eps=1e-6, atol=1e-6,
        rtol=1e-5, nondet_tol=1e-12,
    )


def test_grid_shaped_coordinates_and_16bit():
    from oracle import cpu as oracle_cpu
    from unipercept.vision import pointrend

    g = torch.Generator().manual_seed(11)
    inp = torch.randn(2, 4, 12, 10, generator=g)
    coords = torch.rand(2, 5, 7, 2, generator=g)
    out = pointrend.sample(inp.cuda(), coords.cuda())
    assert out.shape == (2, 4, 5, 7)
    want = oracle_cpu.point_sample_forward(inp.numpy(), coords.flatten(1, 2).numpy())
    assert np.abs(out.flatten(2).cpu().numpy() - want).max() <= 1e-5
    for dt in (torch.bfloat16, torch.float16):
        xi = inp.to(dt).cuda().requires_grad_(True)
        pc = coords.flatten(1, 2).to(dt).cuda().requires_grad_(True)
        o = pointrend.sample(xi, pc)
        assert o.dtype == dt
        o.float().sum().backward()
        assert xi.grad.dtype == dt and pc.grad.dtype == dt
        # fp64 oracle on the same rounded inputs; one rounding of the result: 2^-8 relative
        w64 = oracle_cpu.point_sample_forward(xi.detach().double().cpu().numpy(), pc.detach().double().cpu().numpy())
        assert np.abs(o.detach().double().cpu().numpy() - w64).max() <= 2.0 ** -7 * np.abs(w64).max()
        wgi, _ = oracle_cpu.point_sample_backward(
            np.ones_like(w64), xi.detach().double().cpu().numpy(), pc.detach().double().cpu().numpy())
        assert np.abs(xi.grad.double().cpu().numpy() - wgi).max() <= 2.0 ** -7 * np.abs(wgi).max()


def test_full_size_properties():
    """Mask-loss sized problem (200 masks of 256x512, 12544 points each): properties that need no
    oracle run — a constant map samples to the constant, sampling is linear in the map,
    <out, g> = <input, grad_input> (adjoint identity), and grad_input sums to sum(g) per plane
    (bilinear weights sum to 1 under border padding)."""
    from unipercept.vision import pointrend

    g = torch.Generator(device="cuda").manual_seed(3)
    n, h, w, p = 200, 256, 512, 112 * 112
    a = torch.randn(n, 1, h, w, generator=g, device="cuda")
    b = torch.randn(n, 1, h, w, generator=g, device="cuda")
    pts = torch.rand(n, p, 2, generator=g, device="cuda") * 1.1 - 0.05
    const = pointrend.sample(torch.full_like(a, 2.5), pts)
    assert (const - 2.5).abs().max() <= 1e-5
    sa, sb = pointrend.sample(a, pts), pointrend.sample(b, pts)
    sab = pointrend.sample(a + 3 * b, pts)
    assert (sab - (sa + 3 * sb)).abs().max() <= 1e-4
    go = torch.randn(n, 1, p, generator=g, device="cuda")
    x = a.clone().requires_grad_(True)
    out = pointrend.sample(x, pts)
    out.backward(go)
    lhs = (out.detach().double() * go.double()).sum()
    rhs = (a.double() * x.grad.double()).sum()
    assert abs(lhs - rhs) <= 1e-6 * (out.detach().double() * go.double()).abs().sum()
    assert (x.grad.double().sum((1, 2, 3)) - go.double().sum((1, 2))).abs().max() <= 1e-2


def test_abi_rejects_16bit_backward_and_bad_mode():
    from unipercept_amd import _lib

    lib = _lib.load()
    x = torch.zeros(1, 1, 2, 2, device="cuda")
    pc = torch.zeros(1, 1, 2, device="cuda")
    o = torch.zeros(1, 1, 1, device="cuda")
    args = (x.data_ptr(), pc.data_ptr(), o.data_ptr(), 1, 1, 2, 2, 1)
    assert lib.msda_point_sample_forward(*args, _lib.F32, 7, 0, None) == -3
    assert lib.msda_point_sample_backward(o.data_ptr(), x.data_ptr(), pc.data_ptr(), x.data_ptr(),
                                          pc.data_ptr(), None, 0, 1, 1, 2, 2, 1, _lib.BF16, 0, 0,
                                          None) == -3
