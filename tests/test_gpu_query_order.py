"""Self-attention over the pyramid (the queries are the positions of the levels): the tuned
kernels then hand the queries to their workgroups patch by patch (64 x 8 pixel patches of every
level, cut at the bottom and right edges) instead of in index order.  Which workgroup takes which
query must not show in any result: level extents on both sides of the patch size, levels smaller
than a patch, single-row and single-column levels, against the fp64 C oracle; and the same
launch with the queries NOT being the pyramid (one query more), which keeps the index order."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from conftest import make_inputs
from test_gpu_parity import BWD_ATOL_F32, FWD_ATOL_F32, _hip, _oracle64

pytestmark = pytest.mark.gpu

# (levels, points per level, heads)
CASES = [
    ([(70, 13), (33, 9), (5, 3)], 4, 2),      # cut patches in both directions, a level below a patch
    ([(130, 20)], 4, 1),                      # three patch rows, the last 2 pixels tall
    ([(64, 8)], 4, 3),                        # exactly one patch
    ([(65, 9), (1, 1)], 4, 2),                # one pixel more than a patch each way; a 1 x 1 level
    ([(128, 7), (3, 200)], 4, 2),             # narrower than a patch; a flat, wide level
    ([(66, 10), (30, 7), (9, 9), (2, 5)], 8, 2),  # 32 points per query: two half rows per query
    ([(66, 10), (30, 7), (9, 9), (2, 5)], 4, 8),  # 16 points per query: 128-query blocks
    ([(1, 300)], 2, 2),                       # a single row
    ([(300, 1)], 2, 2),                       # a single column
]


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_patch_order_against_oracle(idx):
    levels, p, m = CASES[idx]
    value, shapes, lsi, loc, attn, grad_out = make_inputs(
        2, m, 32, None, levels, p, dtype=torch.float32, seed=900 + idx, device="cuda"
    )
    value = value * 100
    want = _oracle64(value, shapes, loc, attn, grad_out)
    got = _hip(value, shapes, lsi, loc, attn, grad_out)
    assert np.abs(got[0] - want[0]).max() <= FWD_ATOL_F32, CASES[idx]
    for g, w_ in zip(got[1:], want[1:]):
        assert np.abs(g - w_).max() <= BWD_ATOL_F32, CASES[idx]


def test_patch_order_bf16_and_reproducible_mode():
    """bf16 rows (the forward's 8-lanes-per-query form) in patch order; and the reproducible
    backward, which keeps the index order, on the same inputs: same results."""
    from unipercept_amd import _lib
    from unipercept_amd.deform_attn import _ops  # noqa: F401

    levels, p, m = CASES[0]
    value, shapes, lsi, loc, attn, grad_out = make_inputs(
        2, m, 32, None, levels, p, dtype=torch.float32, seed=77, device="cuda"
    )
    vb, gb = value.bfloat16(), grad_out.bfloat16()
    want = _oracle64(vb, shapes, loc, attn, gb)
    got = _hip(vb, shapes, lsi, loc, attn, gb)
    for g, w_, tol in zip(got, want, (2**-7, 2**-7, 2**-6, 2**-7)):
        assert np.abs(g - w_).max() / (np.abs(w_).max() + 1e-30) <= tol
    bwd = torch.ops.unipercept.ms_deform_attn_bwd
    fast = bwd(grad_out, value, shapes, lsi, loc, attn, 0)
    det = bwd(grad_out, value, shapes, lsi, loc, attn, _lib.FLAG_DETERMINISTIC)
    assert torch.equal(fast[1], det[1]) and torch.equal(fast[2], det[2])
    assert (fast[0] - det[0]).abs().max() <= 1e-5 * det[0].abs().max()


def test_one_query_more_keeps_index_order():
    """Lq = S + 1 (not the pyramid): index order; the first S queries get the same results as in
    the Lq = S launch, bit for bit where no add order is involved."""
    levels, p, m = CASES[0]
    value, shapes, lsi, loc, attn, grad_out = make_inputs(
        1, m, 32, None, levels, p, dtype=torch.float32, seed=31, device="cuda"
    )
    s = value.shape[1]
    a = _hip(value, shapes, lsi, loc, attn, grad_out)
    loc2 = torch.cat((loc, loc[:, :1]), 1).contiguous()
    attn2 = torch.cat((attn, attn[:, :1]), 1).contiguous()
    g2 = torch.cat((grad_out, torch.zeros_like(grad_out[:, :1])), 1).contiguous()
    b = _hip(value, shapes, lsi, loc2, attn2, g2)
    assert np.array_equal(a[0], b[0][:, :s])
    assert np.array_equal(a[2], b[2][:, :s]) and np.array_equal(a[3], b[3][:, :s])
    assert np.abs(a[1] - b[1]).max() <= 1e-5 * np.abs(a[1]).max()
