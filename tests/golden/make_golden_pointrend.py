"""Generates the point-sampling golden vectors (``pointrend_*.npz``).  Run ONLY in the build
container:

    python tests/golden/make_golden_pointrend.py

The producer is ``torch.nn.functional.grid_sample`` called the way the body of the reference's
``unipercept.vision.pointrend.sample`` calls it (sources/unipercept/vision/pointrend.py:63-76 of
the reference: a (N, P, 2) coordinate tensor gets a unit dimension, the grid is
``2 * point_coords - 1``, ``padding_mode="border"``), on the CPU, with autograd gradients.  The
reference package itself cannot be imported here (it needs Python >= 3.12); torch is the
third-party library in which the arithmetic of that function lives.  The case ``reftest`` is the
known-answer vector of the reference's own test (tests/unipercept/vision/test_pointrend.py:6-13).
Only the ``.npz`` data files written here travel with the repository.
"""

from __future__ import annotations

from pathlib import Path

import numpy as np
import torch
from torch.nn.functional import grid_sample

HERE = Path(__file__).resolve().parent

# name -> (N, C, H, W, P, mode, align_corners, coordinate range)
CASES = {
    "masks": (3, 1, 16, 24, 200, "bilinear", False, (-0.15, 1.15)),
    "masks_nearest": (3, 1, 16, 24, 200, "nearest", False, (-0.15, 1.15)),
    "feat": (2, 5, 7, 9, 64, "bilinear", False, (0.0, 1.0)),
    "feat_align": (2, 5, 7, 9, 64, "bilinear", True, (-0.1, 1.1)),
    "feat_align_nearest": (2, 3, 7, 9, 64, "nearest", True, (-0.1, 1.1)),
    "thin": (2, 2, 1, 6, 40, "bilinear", False, (-0.2, 1.2)),
    "column": (1, 2, 5, 1, 40, "bilinear", False, (-0.2, 1.2)),
}


def make_inputs(name, dtype):
    n, c, h, w, p, mode, align, (lo, hi) = CASES[name]
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    inp = torch.randn(n, c, h, w, generator=g, dtype=torch.float64)
    coords = torch.rand(n, p, 2, generator=g, dtype=torch.float64) * (hi - lo) + lo
    # exact corner cases: map corners, map centre, pixel centres and pixel edges
    special = torch.tensor(
        [[0.0, 0.0], [1.0, 1.0], [0.5, 0.5], [0.5 / w, 0.5 / h], [1.0 / w, 1.0 / h], [1.0, 0.0]],
        dtype=torch.float64,
    )
    coords[:, : special.shape[0]] = special
    grad_out = torch.randn(n, c, p, generator=g, dtype=torch.float64)
    return [t.to(dtype) for t in (inp, coords, grad_out)], mode, align


def run(inp, coords, grad_out, mode, align):
    x = inp.clone().requires_grad_(True)
    pc = coords.clone().requires_grad_(True)
    out = grid_sample(x, 2.0 * pc.unsqueeze(2) - 1.0, mode=mode, padding_mode="border",
                      align_corners=align).squeeze(3)
    gi, gc = torch.autograd.grad(out, (x, pc), grad_out, allow_unused=True)
    if gc is None:
        gc = torch.zeros_like(coords)
    return out.detach(), gi, gc


def main():
    # the reference's own known answer
    src = torch.arange(4).view(1, 1, 2, 2).float()
    points = torch.tensor([[0.25, 0.25], [0.75, 0.75]]).unsqueeze(0).float()
    out, _, _ = run(src, points, torch.ones(1, 1, 2), "nearest", False)
    assert out.tolist() == [[[0.0, 3.0]]]
    np.savez_compressed(HERE / "pointrend_reftest_f32.npz", input=src.numpy(), coords=points.numpy(),
                        out=out.numpy(), mode="nearest", align_corners=False)
    for name in CASES:
        for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            (inp, coords, grad_out), mode, align = make_inputs(name, dtype)
            out, gi, gc = run(inp, coords, grad_out, mode, align)
            np.savez_compressed(
                HERE / f"pointrend_{name}_{tag}.npz",
                input=inp.numpy(), coords=coords.numpy(), grad_out=grad_out.numpy(),
                out=out.numpy(), grad_input=gi.numpy(), grad_coords=gc.numpy(),
                mode=mode, align_corners=align,
            )
            print(name, tag, tuple(out.shape), float(out.abs().max()))


if __name__ == "__main__":
    main()
