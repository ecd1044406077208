"""GPU parity sweep: seeded random problem shapes around the boundaries of the kernels' work
decomposition (query counts next to the 64-query passes and 192-query blocks of the backward,
head dims with 4 / 8 / 16 lane chunks, point counts on both sides of the staging and pairing
limits, queries that are / are not the pixels), fp32 and bf16, against the fp64 C oracle on the
same inputs."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from conftest import make_inputs
from test_gpu_parity import BWD_ATOL_F32, FWD_ATOL_F32, _hip, _oracle64

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(20240607)
    lqs = [1, 63, 64, 65, 191, 192, 193, 383, 385, None]
    out = []
    for k in range(20):
        d = int(rng.choice([16, 32, 64]))
        m = int(rng.choice([1, 3, 8]))
        l = int(rng.integers(1, 5))
        p = int(rng.choice([1, 2, 3, 4, 8]))
        levels = [(int(rng.integers(1, 20)), int(rng.integers(1, 24))) for _ in range(l)]
        lq = lqs[k % len(lqs)]
        n = int(rng.integers(1, 4))
        out.append((n, m, d, lq, levels, p))
    return out


CASES = _cases()


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_sweep_f32(idx):
    n, m, d, lq, levels, p = CASES[idx]
    value, shapes, lsi, loc, attn, grad_out = make_inputs(
        n, m, d, lq, levels, p, dtype=torch.float32, seed=100 + idx, device="cuda"
    )
    value = value * 100
    want = _oracle64(value, shapes, loc, attn, grad_out)
    got = _hip(value, shapes, lsi, loc, attn, grad_out)
    assert np.abs(got[0] - want[0]).max() <= FWD_ATOL_F32, CASES[idx]
    for g, w_ in zip(got[1:], want[1:]):
        assert np.abs(g - w_).max() <= BWD_ATOL_F32, CASES[idx]


@pytest.mark.parametrize("idx", range(0, len(CASES), 2))
def test_sweep_bf16(idx):
    n, m, d, lq, levels, p = CASES[idx]
    value, shapes, lsi, loc, attn, grad_out = make_inputs(
        n, m, d, lq, levels, p, dtype=torch.float32, seed=200 + idx, device="cuda"
    )
    value = (value * 100).bfloat16()
    grad_out = grad_out.bfloat16()
    want = _oracle64(value, shapes, loc, attn, grad_out)  # fp32 locations and weights
    got = _hip(value, shapes, lsi, loc, attn, grad_out)
    for name, g, w_, tol in zip(
        ("out", "grad_value", "grad_loc", "grad_attn"), got, want, (2**-7, 2**-7, 2**-6, 2**-7)
    ):
        # same stated tolerance as the fixed bf16 cases: relative to the tensor's max-abs
        err = np.abs(g - w_).max() / (np.abs(w_).max() + 1e-30)
        assert err <= tol, (name, err, CASES[idx])
