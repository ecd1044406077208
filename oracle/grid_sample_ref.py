"""TEST INFRASTRUCTURE: pure-PyTorch oracle (``grid_sample`` formulation).

Restates the algorithm of the reference's ``reference.deform_attn(value, shapes, loc, attn)``
(call site: tests/unipercept/nn/layers/test_nn_layers_deform_attn.py:40-49 of the reference; the
module body is absent from the reference snapshot).  The arithmetic lives in
``torch.nn.functional.grid_sample`` (bilinear, zero padding, ``align_corners=False``), a
third-party dependency of the reference (torch >= 2.5.1, pyproject.toml:26).

PARITY UNPINNED against the reference's own outputs (it holds no golden vectors for this path and
its implementation is absent).  Pinned instead by ``tests/test_oracle.py`` against the vectors in
``tests/golden/`` that ``tests/golden/make_golden.py`` produced in the build container from the
canonical formulation.
"""

from __future__ import annotations

import torch
import torch.nn.functional as F


def deform_attn(
    value: torch.Tensor,
    shapes,
    loc: torch.Tensor,
    attn: torch.Tensor,
) -> torch.Tensor:
    """value (N,S,M,D); shapes (L,2) as (H,W); loc (N,Q,M,L,P,2) as (x,y); attn (N,Q,M,L,P)."""
    n, _, m, d = value.shape
    _, q, _, l, p, _ = loc.shape
    hw = [(int(h), int(w)) for h, w in (shapes.tolist() if torch.is_tensor(shapes) else shapes)]
    per_level = value.split([h * w for h, w in hw], dim=1)
    grids = loc * 2 - 1
    sampled = []
    for lvl, (h, w) in enumerate(hw):
        # (N, H*W, M, D) -> (N*M, D, H, W)
        v = per_level[lvl].permute(0, 2, 3, 1).reshape(n * m, d, h, w)
        # (N, Q, M, P, 2) -> (N*M, Q, P, 2)
        g = grids[:, :, :, lvl].permute(0, 2, 1, 3, 4).reshape(n * m, q, p, 2)
        sampled.append(
            F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False)
        )
    # (N*M, D, Q, L, P)
    stacked = torch.stack(sampled, dim=3)
    a = attn.permute(0, 2, 1, 3, 4).reshape(n * m, 1, q, l, p)
    out = (stacked * a).sum(dim=(3, 4))  # (N*M, D, Q)
    return out.view(n, m * d, q).transpose(1, 2).contiguous()


def deform_attn_with_grads(value, shapes, loc, attn, grad_out):
    """Forward + autograd backward; returns (out, grad_value, grad_loc, grad_attn)."""
    value = value.detach().clone().requires_grad_(True)
    loc = loc.detach().clone().requires_grad_(True)
    attn = attn.detach().clone().requires_grad_(True)
    out = deform_attn(value, shapes, loc, attn)
    gv, gl, ga = torch.autograd.grad(out, (value, loc, attn), grad_out)
    return out.detach(), gv, gl, ga
