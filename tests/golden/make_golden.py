"""Generates the golden vectors in this directory.  Run ONLY in the build container:

    HF_HUB_OFFLINE=1 python tests/golden/make_golden.py

The producer is the canonical ``grid_sample`` formulation of multi-scale deformable attention
that ships inside the container's installed ``transformers`` package
(``transformers.models.mask2former.modeling_mask2former.multi_scale_deformable_attention``),
i.e. the same algorithm the reference's ``reference.deform_attn`` names
(tests/unipercept/nn/layers/test_nn_layers_deform_attn.py:40-49 of the reference; the reference
holds no golden files of its own for this path).  Gradients come from torch autograd through that
function.  Nothing from ``transformers`` or from the reference travels with the repository; only
the ``.npz`` data files written here do.
"""

from __future__ import annotations

import os
from pathlib import Path

import numpy as np
import torch

os.environ.setdefault("HF_HUB_OFFLINE", "1")
from transformers.models.mask2former.modeling_mask2former import (  # noqa: E402
    multi_scale_deformable_attention,
)

HERE = Path(__file__).resolve().parent

# name -> (N, M, D, Lq (None = S), levels [(H, W)...], P, loc_mode)
CASES = {
    # shapes of the reference's own test (test_nn_layers_deform_attn.py:12-13,24)
    "reftest": (1, 2, 2, 2, [(32, 16), (16, 8)], 2, "unit"),
    # gradcheck head dims (test_nn_layers_deform_attn.py:125), small ones
    "d30": (1, 2, 30, 2, [(32, 16), (16, 8)], 2, "unit"),
    "d71": (1, 2, 71, 3, [(6, 4), (3, 2)], 2, "wide"),
    # shrunken benchmark configurations (encoder style, Lq = S)
    "cfg2_small": (2, 4, 32, None, [(4, 8), (2, 4), (1, 2)], 4, "wide"),
    "cfg4_small": (1, 8, 32, None, [(8, 8), (4, 4), (2, 2), (1, 1)], 4, "wide"),
    "cfg5_small": (2, 2, 32, None, [(4, 8), (2, 4), (1, 2), (1, 1)], 8, "wide"),
    # decoder style, odd sizes, one level of width 1
    "ragged": (3, 3, 5, 7, [(5, 3), (1, 7), (4, 1)], 3, "wide"),
}


def make_inputs(name, dtype):
    n, m, d, lq, levels, p, mode = CASES[name]
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    s = sum(h * w for h, w in levels)
    lq = s if lq is None else lq
    l = len(levels)
    value = torch.rand(n, s, m, d, generator=g, dtype=torch.float64) * 0.01
    loc = torch.rand(n, lq, m, l, p, 2, generator=g, dtype=torch.float64)
    if mode == "wide":  # some locations out of range on every side, incl. beyond the -1 px skirt
        loc = loc * 1.5 - 0.25
    attn = torch.rand(n, lq, m, l, p, generator=g, dtype=torch.float64) + 1e-5
    attn = attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    grad_out = torch.randn(n, lq, m * d, generator=g, dtype=torch.float64)
    return [t.to(dtype) for t in (value, loc, attn, grad_out)], levels


def main():
    for name in CASES:
        for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            if tag == "f64" and name.startswith("cfg"):
                continue  # keep the fixture directory small
            (value, loc, attn, grad_out), levels = make_inputs(name, dtype)
            v = value.clone().requires_grad_(True)
            lo = loc.clone().requires_grad_(True)
            a = attn.clone().requires_grad_(True)
            out = multi_scale_deformable_attention(v, levels, lo, a)
            gv, gl, ga = torch.autograd.grad(out, (v, lo, a), grad_out)
            np.savez_compressed(
                HERE / f"{name}_{tag}.npz",
                shapes=np.asarray(levels, dtype=np.int64),
                value=value.numpy(), loc=loc.numpy(), attn=attn.numpy(), grad_out=grad_out.numpy(),
                out=out.detach().numpy(), grad_value=gv.numpy(), grad_loc=gl.numpy(),
                grad_attn=ga.numpy(),
            )
            print(name, tag, tuple(out.shape), float(out.detach().abs().max()))


if __name__ == "__main__":
    main()
