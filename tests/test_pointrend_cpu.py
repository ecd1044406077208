"""CPU side of the point-sampling path: the C oracle against the golden vectors (torch's
grid_sample called as the reference's ``pointrend.sample`` calls it) and the reference test's
known answer; the plain-PyTorch point generators, with the reference's own test cases
(tests/unipercept/vision/test_pointrend.py:16-74) and the oracle standing in for the GPU
sampler; the refusal to run off the GPU."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from conftest import load_golden, pointrend_cases
from oracle import cpu as oracle_cpu
from unipercept_amd.vision import pointrend


def _tol(dtype, grad=False):
    # fp32: the producer's vectorised CPU kernel writes the coordinate transform as
    # (g + 1) * (size / 2) - 0.5, the scalar algorithm as ((g + 1) * size - 1) / 2: pixel
    # coordinates differ by an ulp, the coordinate gradient (scaled by the map size and the values)
    # by a few 1e-5
    if dtype == np.float64:
        return 1e-12
    return 1e-4 if grad else 1e-5


@pytest.mark.parametrize("case", pointrend_cases())
def test_oracle_matches_golden(case):
    z = load_golden(case)
    mode, align = str(z["mode"]), bool(z["align_corners"])
    out = oracle_cpu.point_sample_forward(z["input"], z["coords"], mode, align)
    assert out.dtype == z["out"].dtype
    assert np.abs(out - z["out"]).max() <= _tol(out.dtype.type)
    if "grad_out" not in z:
        return
    gi, gc = oracle_cpu.point_sample_backward(z["grad_out"], z["input"], z["coords"], mode, align)
    assert np.abs(gi - z["grad_input"]).max() <= _tol(out.dtype.type)
    assert np.abs(gc - z["grad_coords"]).max() <= _tol(out.dtype.type, grad=True)
    if mode == "nearest":
        assert not gc.any()


def test_reference_known_answer():
    # tests/unipercept/vision/test_pointrend.py:6-13 of the reference
    src = np.arange(4, dtype=np.float32).reshape(1, 1, 2, 2)
    points = np.array([[[0.25, 0.25], [0.75, 0.75]]], dtype=np.float32)
    out = oracle_cpu.point_sample_forward(src, points, "nearest", False)
    assert out.shape == (1, 1, 2) and out[0, 0, 0] == 0 and out[0, 0, 1] == 3


def test_oracle_border_and_gradient_mask():
    # outside the unit square the value is the border pixel's and the coordinate gradient is 0
    inp = np.arange(12, dtype=np.float64).reshape(1, 1, 3, 4)
    pts = np.array([[[-0.3, -0.2], [1.4, 0.5], [0.5, 1.7], [0.5, 0.5]]])
    out = oracle_cpu.point_sample_forward(inp, pts)
    assert out[0, 0, 0] == 0.0 and out[0, 0, 1] == 7.0
    assert out[0, 0, 2] == pytest.approx(9.5) and out[0, 0, 3] == pytest.approx(5.5)
    _, gc = oracle_cpu.point_sample_backward(np.ones((1, 1, 4)), inp, pts)
    assert gc[0, 0].tolist() == [0.0, 0.0]
    assert gc[0, 1, 0] == 0.0 and gc[0, 2, 1] == 0.0
    # inside: d/dx = W * (right - left) = 4 * 1, d/dy = H * (below - above) = 3 * 4
    assert gc[0, 3].tolist() == pytest.approx([4.0, 12.0])


def _nearest(src: torch.Tensor, pts: torch.Tensor) -> torch.Tensor:
    return torch.from_numpy(
        oracle_cpu.point_sample_forward(src.float().numpy(), pts.float().numpy(), "nearest", False)
    )


def test_random_points():
    src = torch.rand(1, 1, 10, 10)
    points = pointrend.random_points(src, 10, 2)
    assert points.shape == (1, 10, 2) and points.min() >= 0 and points.max() <= 1


def test_random_points_with_mask():
    torch.manual_seed(0)
    src = torch.rand(1, 1, 100, 100)
    for mask in (
        torch.tensor([[False, True, False], [False, False, False], [False, False, False]])[None, None],
        torch.arange(20).view(1, 1, 2, 10) == 15,
        torch.rand_like(src) < 0.5,
        torch.rand_like(src) < 0.1,
    ):
        points = pointrend.random_points(src, 5, 2, mask)
        assert points.shape == (1, 5, 2) and points.min() >= 0 and points.max() <= 1
        assert torch.all(_nearest(mask, points).bool())


def test_random_points_mask_needs_2d():
    with pytest.raises(NotImplementedError):
        pointrend.random_points(torch.rand(1, 1, 4, 4, 4), 3, 3, torch.ones(1, 1, 4, 4, 4, dtype=torch.bool))


def test_logit_uncertainty():
    logits = torch.tensor([[-1.0, 0.5], [0.2, -0.3]], requires_grad=True)
    u = pointrend.logit_uncertainty(logits)
    assert u.tolist() == [[-1.0, -0.5], [-0.20000000298023224, -0.30000001192092896]]
    assert not u.requires_grad


def test_sample_refuses_cpu_and_bad_arguments():
    src = torch.rand(1, 1, 4, 4)
    pts = torch.rand(1, 3, 2)
    with pytest.raises(RuntimeError, match="ROCm GPU"):
        pointrend.sample(src, pts)
    with pytest.raises(NotImplementedError):
        pointrend.sample(src, pts, mode="bicubic")
    with pytest.raises(TypeError):
        pointrend.sample(src, pts, padding_mode="zeros")
    with pytest.raises(ValueError):
        pointrend.sample(src[0], pts)
    with pytest.raises(ValueError):
        pointrend.sample(src, pts[..., :1])
    with pytest.raises(TypeError):
        pointrend.sample(src, pts.double())


@pytest.mark.parametrize("n,c,p", [(0, 3, 5), (2, 0, 5), (2, 3, 0), (0, 0, 0)])
def test_sample_of_nothing_is_empty_like_grid_sample(n, c, p):
    """No samples, channels or points (a mask loss on a batch without instances): an empty
    (N, C, P) result and zero-sized / zero gradients, exactly as grid_sample gives; no launch, so
    this also holds for CPU tensors."""
    src = torch.rand(n, c, 4, 6, requires_grad=True)
    pts = torch.rand(n, p, 2, requires_grad=True)
    out = pointrend.sample(src, pts)
    want = torch.nn.functional.grid_sample(
        src, 2 * pts.unsqueeze(2) - 1, padding_mode="border", align_corners=False
    ).squeeze(3)
    assert out.shape == want.shape == (n, c, p) and out.dtype == want.dtype
    out.sum().backward()
    assert src.grad.shape == src.shape and pts.grad.shape == pts.shape
    assert float(src.grad.abs().sum()) == 0.0 and float(pts.grad.abs().sum()) == 0.0
    grid = pointrend.sample(src, torch.rand(n, 0, 7, 2))
    assert grid.shape == (n, c, 0, 7)
    with pytest.raises(ValueError):
        pointrend.sample(torch.rand(1, 1, 0, 4), torch.rand(1, 3, 2))


def test_importance_sampling_host_logic(monkeypatch):
    # tests/unipercept/vision/test_pointrend.py:51-74 of the reference, with the sampler replaced
    # by the oracle (the HIP sampler needs a GPU; the GPU suite runs the same case through it)
    monkeypatch.setattr(pointrend, "sample", lambda s, p, **kw: _nearest(s, p))
    src = torch.arange(6).view(1, 1, 2, 3).float()
    points = pointrend.random_points_with_importance(
        src, 3, 2, oversample_ratio=300, importance_sample_ratio=1.0, importance_fn=lambda t: t,
        mode="nearest", align_corners=False,
    )
    samples = _nearest(src, points)
    assert points.shape == (1, 3, 2) and samples.shape == (1, 1, 3)
    assert samples.min() >= 3.0 and samples.max() <= 5.0
    # a quarter of the points is drawn afresh when the ratio says so
    points = pointrend.random_points_with_importance(src, 8, 2, importance_sample_ratio=0.75)
    assert points.shape == (1, 8, 2)


# ---- value-bin helpers: tests/unipercept/vision/test_pointrend.py:77-207 of the reference ----
def _masked_mean(t, m):
    return torch.nanmean(t.masked_fill(~m, torch.nan), dim=1, keepdim=True)


def _count(t, m):
    return m.int().sum(dim=1, keepdim=True)


@pytest.mark.parametrize("use_batch", [False, True])
@pytest.mark.parametrize("mode", ["linear", "quantile"])
def test_bins_by_values(use_batch, mode):
    values = torch.rand(4, 30)
    bins = pointrend.bins_by_values(values, 10, use_batch=use_batch, mode=mode)
    assert bins.shape == (4, 10) and bins.min() >= 0 and bins.max() <= 1 + 1e-5
    assert torch.all(bins[:, 1:] >= bins[:, :-1])
    if not use_batch:  # one set of boundaries for the whole batch
        assert torch.equal(bins[0], bins[1])
    # the last boundary lies just above the largest value that counts
    top = values.max(dim=1).values if use_batch else values.max().expand(4)
    assert torch.all(bins[:, -1] > top) and torch.all(bins[:, -1] - top <= 1e-5)


@pytest.mark.parametrize("use_batch", [False, True])
@pytest.mark.parametrize("mode", ["linear", "quantile"])
def test_map_in_bins(use_batch, mode):
    values = torch.rand(2, 1000)
    mask = values > 0.1
    bins = pointrend.bins_by_values(values, 10, mask=mask, use_batch=use_batch, mode=mode)
    assert pointrend.map_in_bins(values, bins, _masked_mean).shape == (2, 10)
    sources = torch.rand(2, 1000)
    logv = sources.log1p()
    bins = pointrend.bins_by_values(logv, 10, use_batch=use_batch, mode=mode)
    assert pointrend.map_in_bins(sources, bins, _masked_mean, values=logv).shape == (2, 10)
    counts = pointrend.map_in_bins(sources, bins, _count, values=logv)
    assert counts.shape == (2, 10) and counts.sum(dim=1).tolist() == [1000, 1000]
    if mode == "quantile" and use_batch:
        assert counts.min() >= 99 and counts.max() <= 101  # equally populated
    mask = sources > 0.1
    bins = pointrend.bins_by_values(logv, 10, mask=mask, use_batch=use_batch, mode=mode)
    counts = pointrend.map_in_bins(sources, bins, _count, values=logv, mask=mask)
    assert counts.shape == (2, 10) and counts.sum(dim=1).tolist() == mask.sum(dim=1).tolist()


@pytest.mark.parametrize("use_batch", [False, True])
@pytest.mark.parametrize("mode", ["linear", "quantile"])
def test_random_points_with_bins(use_batch, mode):
    sources = torch.rand(2, 1, 16, 16)
    points = pointrend.random_points_with_bins(
        sources, 8, use_batch=use_batch, mode=mode, n_bins=10, values=sources.log1p()
    )
    assert points.shape == (2, 8, 2) and points.min() >= 0 and points.max() <= 1


def test_bins_mode_strings():
    assert pointrend.BinsMode("linear") is pointrend.BinsMode.LINEAR
    assert f"{pointrend.BinsMode.QUANTILE}" == "quantile"
    with pytest.raises(ValueError):
        pointrend.bins_by_values(torch.rand(1, 5), 2, mode="cubic")
