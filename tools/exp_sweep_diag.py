"""Where a parity-sweep case disagrees in grad_loc: prints the pixel coordinates of the worst points
(a coordinate within rounding of an integer flips the bilinear cell between fp32 and fp64)."""
import os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from conftest import make_inputs
from test_gpu_parity import _hip, _oracle64
rng = np.random.default_rng(777)
lqs = [1, 2, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 257, 383, 385, 577, None, None, None]
for k in range(150):
    d = int(rng.choice([16, 32, 64])); m = int(rng.choice([1, 2, 3, 8])); l = int(rng.integers(1, 5))
    p = int(rng.choice([1, 2, 3, 4, 6, 8]))
    levels = [(int(rng.integers(1, 40)), int(rng.integers(1, 48))) for _ in range(l)]
    lq = lqs[k % len(lqs)]; n = int(rng.integers(1, 4))
    value, shapes, lsi, loc, attn, grad_out = make_inputs(n, m, d, lq, levels, p, dtype=torch.float32, seed=1000 + k, device="cuda")
    value = value * 100
    want = _oracle64(value, shapes, loc, attn, grad_out); got = _hip(value, shapes, lsi, loc, attn, grad_out)
    err = np.abs(got[2] - want[2])
    if err.max() > 1e-3:
        print("case", k, (n, m, d, lq, levels, p), "max err", err.max())
        idx = np.argwhere(err > 1e-3)
        locn = loc.cpu().numpy()
        for b, q, h, lv, pt, c in idx[:8]:
            H, W = levels[lv]
            x32 = np.float32(locn[b, q, h, lv, pt, 0]) * np.float32(W) - np.float32(0.5)
            y32 = np.float32(locn[b, q, h, lv, pt, 1]) * np.float32(H) - np.float32(0.5)
            x64 = float(locn[b, q, h, lv, pt, 0]) * W - 0.5; y64 = float(locn[b, q, h, lv, pt, 1]) * H - 0.5
            print("   point", (b, q, h, lv, pt, c), "x32 %.9g x64 %.12g  y32 %.9g y64 %.12g" % (x32, x64, y32, y64), "err", err[b, q, h, lv, pt, c])
