"""Throughput of the point-sampling kernels on MI355X at mask-loss sizes:

    python tools/bench_point_sample.py [--masks 200] [--height 256] [--width 512] [--points 12544] [--channels 1]

Prints one JSON line: sampled points/s forward and forward+backward, the fraction of the HBM
roofline on algorithmic bytes (map read once, coordinates, result; the backward additionally
grad_out, the map again, grad_input and grad_coords), PyTorch-ROCm's own grid_sample on the same
GPU for scale, and the CPU oracle on a bounded sample.
"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unipercept_amd.vision import pointrend

ap = argparse.ArgumentParser()
ap.add_argument("--masks", type=int, default=200)
ap.add_argument("--channels", type=int, default=1)
ap.add_argument("--height", type=int, default=256)
ap.add_argument("--width", type=int, default=512)
ap.add_argument("--points", type=int, default=112 * 112)
ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
n, c, h, w, p = a.masks, a.channels, a.height, a.width, a.points
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(n, c, h, w, generator=g, device="cuda")
pts = torch.rand(n, p, 2, generator=g, device="cuda")
go = torch.randn(n, c, p, generator=g, device="cuda")


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters


fwd = torch.ops.unipercept.point_sample_fwd
bwd = torch.ops.unipercept.point_sample_bwd
t_f = timed(lambda: fwd(x, pts, 0, False))
t_b = timed(lambda: bwd(go, x, pts, 0, False))


def torch_fwd_bwd():
    xx = x.detach().requires_grad_(True)
    pp = pts.detach().requires_grad_(True)
    o = torch.nn.functional.grid_sample(xx, 2 * pp.unsqueeze(2) - 1, padding_mode="border", align_corners=False)
    o.backward(go.unsqueeze(3))


t_tf = timed(lambda: torch.nn.functional.grid_sample(x, 2 * pts.unsqueeze(2) - 1, padding_mode="border", align_corners=False))
t_tfb = timed(torch_fwd_bwd)
e = 4
fwd_bytes = (n * c * h * w + n * p * 2 + n * c * p) * e
bwd_bytes = (n * c * p + n * c * h * w + n * p * 2 + n * c * h * w + n * p * 2) * e
pts_total = n * c * p

from oracle import cpu as oracle_cpu
ns = min(n, 20)
xc, pc, gc = x[:ns].cpu().numpy(), pts[:ns].cpu().numpy(), go[:ns].cpu().numpy()
t0 = time.perf_counter()
oracle_cpu.point_sample_forward(xc, pc)
oracle_cpu.point_sample_backward(gc, xc, pc)
t_cpu = time.perf_counter() - t0
print(json.dumps({
    "workload": f"point_sample masks={n} C={c} {h}x{w} P={p} fp32 bilinear",
    "fwd_ms": t_f, "bwd_ms": t_b,
    "fwd_points_per_s": pts_total / (t_f * 1e-3),
    "fwd_bwd_points_per_s": pts_total / ((t_f + t_b) * 1e-3),
    "fwd_hbm_frac": fwd_bytes / (t_f * 1e-3) / 8e12,
    "bwd_hbm_frac": bwd_bytes / (t_b * 1e-3) / 8e12,
    "torch_rocm_grid_sample": {"fwd_ms": t_tf, "fwd_bwd_ms": t_tfb},
    "cpu_baseline": {"value": ns * c * p / t_cpu, "unit": "points/s", "cores": 1, "kind": "port",
                     "sample": f"{ns} masks fwd+bwd, scalar C oracle"},
}))
