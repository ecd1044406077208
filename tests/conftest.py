from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "shim"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def golden_cases():
    """MSDeformAttn vectors (tests/golden/make_golden.py)."""
    return sorted(p.stem for p in GOLDEN.glob("*.npz") if not p.stem.startswith("pointrend_"))


def pointrend_cases():
    """Point-sampling vectors (tests/golden/make_golden_pointrend.py)."""
    return sorted(p.stem for p in GOLDEN.glob("pointrend_*.npz"))


def load_golden(name: str) -> dict[str, np.ndarray]:
    with np.load(GOLDEN / f"{name}.npz", allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def level_start_index(shapes: torch.Tensor) -> torch.Tensor:
    return torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))


def make_inputs(n, m, d, lq, levels, p, *, dtype=torch.float32, seed=0, wide=True, device="cpu"):
    """Seeded synthetic inputs in the style of the reference's test (value = rand * 0.01,
    normalised attention); `wide` stretches locations to [-0.25, 1.25] for out-of-range cover."""
    g = torch.Generator().manual_seed(seed)
    shapes = torch.as_tensor(levels, dtype=torch.long)
    s = int(shapes.prod(1).sum())
    lq = s if lq is None else lq
    l = len(levels)
    value = torch.rand(n, s, m, d, generator=g, dtype=torch.float64) * 0.01
    loc = torch.rand(n, lq, m, l, p, 2, generator=g, dtype=torch.float64)
    if wide:
        loc = loc * 1.5 - 0.25
    attn = torch.rand(n, lq, m, l, p, generator=g, dtype=torch.float64) + 1e-5
    attn = attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    grad_out = torch.randn(n, lq, m * d, generator=g, dtype=torch.float64)
    to = lambda t: t.to(dtype).to(device)
    return to(value), shapes.to(device), level_start_index(shapes).to(device), to(loc), to(attn), to(grad_out)
