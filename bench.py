#!/usr/bin/env python3
"""MSDeformAttn fwd+bwd throughput on MI355X: sampled points/s and fraction of the HBM roofline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--dist R|U]

One step = one forward + one backward of the op over one batch of synthetic inputs that already
sit in HBM.  N > 1 shards the batch, one rank per GPU: every rank runs the same per-GPU workload
on its own samples (weak scaling); the only collective is the all-reduce of a flat fp32 bucket
the size of one MSDeformAttn layer's parameter gradients (RCCL), issued behind the backward on
the compute stream (--comm inline, the default) or on a side stream overlapped with the next
step (--comm overlap).  The ranks come from torch.distributed.run (RANK / WORLD_SIZE in the
environment) or, for a plain `python bench.py --gpus N`, from this script itself: it starts N
children before anything touches the GPU and relays rank 0's line.

Rank 0 prints ONE JSON line.
"""

from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time
from pathlib import Path

# RCCL over xGMI between the ranks of one node needs dmabuf IPC on this driver stack
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))


def maybe_launch() -> None:
    """The launcher branch, taken BEFORE torch (or anything else that could load the HIP runtime)
    is imported: a plain `python bench.py --gpus N` (N > 1, or any N with --spawn) outside
    torch.distributed.run starts N ranks of this script and exits with their status.  The parent
    counts GPUs from sysfs only and never touches one."""
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument("--gpus", type=int, default=1)
    pre.add_argument("--spawn", action="store_true")
    known, _ = pre.parse_known_args()
    if "WORLD_SIZE" in os.environ or not (known.gpus > 1 or known.spawn):
        return
    from unipercept_amd import launch  # torch-free

    visible = launch.visible_gpu_count()
    if visible is not None and visible < known.gpus:
        raise SystemExit(f"--gpus {known.gpus} but only {visible} GPU(s) are visible")
    raise SystemExit(launch.launch_ranks(str(Path(__file__).resolve()), sys.argv[1:], known.gpus))


if __name__ == "__main__":
    maybe_launch()

import torch  # noqa: E402

from unipercept_amd import harness  # noqa: E402

HBM_PEAK = 8.0e12  # bytes/s, MI355X spec


def cpu_baseline(threads: int, dist_name: str = "R") -> dict:
    """The pure-PyTorch grid_sample oracle, timed on this host's cores on a bounded sample:
    the CPU-reference workload (batch 1, 512x1024, 3 levels, 8 heads, 4 points, fp32)."""
    from oracle import grid_sample_ref

    w = harness.WORKLOADS["cpuref_512x1024_fp32_b1"]
    torch.set_num_threads(threads)
    value, shapes, lsi, loc, attn, grad_out = harness.synth_inputs(w, "cpu", seed=1234, dist=dist_name)
    levels = shapes.tolist()

    def run():
        v = value.requires_grad_(True)
        lo = loc.requires_grad_(True)
        a = attn.requires_grad_(True)
        out = grid_sample_ref.deform_attn(v, levels, lo, a)
        torch.autograd.grad(out, (v, lo, a), grad_out)

    for _ in range(2):
        run()
    times = []
    t_end = time.perf_counter() + 20.0
    while len(times) < 10 and (time.perf_counter() < t_end or len(times) < 3):
        t0 = time.perf_counter()
        run()
        times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    pts = w.sampled_points()
    return {
        "value": pts / med,
        "unit": "points/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{w.name}: {pts} sampled points fwd+bwd, median of {len(times)} runs "
        f"({med * 1e3:.1f} ms each), torch grid_sample formulation, location distribution "
        f"{dist_name}, {threads} threads (os.cpu_count() = {os.cpu_count()}, OMP_NUM_THREADS = "
        f"{os.environ.get('OMP_NUM_THREADS', 'unset')})",
    }


POINT_SAMPLE_WORKLOADS = {
    # mask loss at Cityscapes 1/4 resolution: 200 mask logit maps, 112 x 112 points each
    "point_sample_masks_256x512_p12544": dict(n=200, c=1, h=256, w=512, p=112 * 112),
    # feature sampling: 16 maps of 256 channels
    "point_sample_feat_c256_128x256_p4096": dict(n=16, c=256, h=128, w=256, p=4096),
}


def point_sample_cpu_baseline(x, pts, go) -> dict:
    """The scalar C oracle of the sampling op on one host core, on a bounded sample (the first
    maps of the workload, forward + backward)."""
    from oracle import cpu as oracle_cpu

    per_map = x.shape[1] * pts.shape[1]
    ns = max(1, min(x.shape[0], 4_000_000 // max(per_map, 1)))
    xc, pc, gc = x[:ns].cpu().numpy(), pts[:ns].cpu().numpy(), go[:ns].cpu().numpy()
    t0 = time.perf_counter()
    oracle_cpu.point_sample_forward(xc, pc)
    oracle_cpu.point_sample_backward(gc, xc, pc)
    dt = time.perf_counter() - t0
    return {
        "value": ns * per_map / dt,
        "unit": "points/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {ns} maps of the workload, forward + backward, scalar C oracle ({dt * 1e3:.0f} ms)",
    }


def bench_point_sample(args, device, rank: int, world: int, distributed: bool) -> None:
    """`--op point_sample`: PointRend point sampling (grid_sample with border padding at
    normalised points), forward + backward per step, maps and points resident in HBM.  The maps
    are sharded over the ranks with no exchange (the op has no parameters): weak scaling."""
    import torch.distributed as dist

    from unipercept_amd.vision import pointrend  # noqa: F401  (registers the ops, loads libmsda.so)

    name = args.workload if args.workload in POINT_SAMPLE_WORKLOADS else "point_sample_masks_256x512_p12544"
    cfg = POINT_SAMPLE_WORKLOADS[name]
    n, c, h, w, p = cfg["n"], cfg["c"], cfg["h"], cfg["w"], cfg["p"]
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    x = torch.randn(n, c, h, w, generator=g, device=device)
    pts = torch.rand(n, p, 2, generator=g, device=device)
    go = torch.randn(n, c, p, generator=g, device=device)
    fwd = torch.ops.unipercept.point_sample_fwd
    bwd = torch.ops.unipercept.point_sample_bwd
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]

    def step(i):
        if i is not None:
            ev[i][0].record()
        fwd(x, pts, 0, False)
        if i is not None:
            ev[i][1].record()
        bwd(go, x, pts, 0, False)
        if i is not None:
            ev[i][2].record()

    def fence():
        torch.cuda.synchronize(device)
        if distributed:
            dist.barrier(device_ids=[device.index])
            torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step(None)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        return
    fwd_ms = statistics.mean(e[0].elapsed_time(e[1]) for e in ev)
    bwd_ms = statistics.mean(e[1].elapsed_time(e[2]) for e in ev)
    e4 = 4
    fwd_bytes = (n * c * h * w + n * p * 2 + n * c * p) * e4
    bwd_bytes = (n * c * p + 2 * n * c * h * w + 2 * n * p * 2) * e4
    points = n * c * p
    traffic = None
    try:
        rec = json.loads((ROOT / "profiles" / "hbm_traffic.json").read_text()).get(name)
        if rec is not None:
            traffic = rec["backward"]
    except (OSError, ValueError, KeyError):
        traffic = None
    line = {
        "metric": "point_sample_fwd_bwd_points_per_sec",
        "value": points * world * args.steps / elapsed,
        "unit": "points/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": name, "maps_per_gpu": n, "channels": c, "height": h, "width": w,
            "points_per_map": p, "mode": "bilinear", "align_corners": False,
            "sampled_points_per_gpu_step": points, "parallelism": f"dp{world}", "collective": None,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "msda_point_sample_backward (grad_coords gather + grad_input band kernel, HIP events in the timed loop)",
            "achieved": bwd_bytes / (bwd_ms * 1e-3) / 1e9,
            "peak": HBM_PEAK / 1e9,
            "unit": "GB/s",
            "frac": bwd_bytes / (bwd_ms * 1e-3) / HBM_PEAK,
            "traffic": traffic,
            "algorithmic_bytes_per_launch": bwd_bytes,
            "avg_launch_ms": bwd_ms,
        },
        "phases": {
            "fwd_ms": fwd_ms, "bwd_ms": bwd_ms,
            "fwd_points_per_s": points / (fwd_ms * 1e-3),
            "fwd_hbm_frac": fwd_bytes / (fwd_ms * 1e-3) / HBM_PEAK,
            "bwd_hbm_frac": bwd_bytes / (bwd_ms * 1e-3) / HBM_PEAK,
        },
    }
    if args.torch_compare:
        def t_fwd():
            return torch.nn.functional.grid_sample(x, 2 * pts.unsqueeze(2) - 1, padding_mode="border", align_corners=False)

        def t_fwd_bwd():
            xx = x.detach().requires_grad_(True)
            pp = pts.detach().requires_grad_(True)
            torch.nn.functional.grid_sample(xx, 2 * pp.unsqueeze(2) - 1, padding_mode="border",
                                            align_corners=False).backward(go.unsqueeze(3))

        res = {}
        for label, fn in (("fwd_ms", t_fwd), ("fwd_bwd_ms", t_fwd_bwd)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize(device)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.steps):
                fn()
            b.record()
            torch.cuda.synchronize(device)
            res[label] = a.elapsed_time(b) / args.steps
        line["torch_rocm_grid_sample"] = res
    if not args.no_cpu_baseline and world == 1:
        line["cpu_baseline"] = point_sample_cpu_baseline(x, pts, go)
    emit_result(line)


_RESULT_OUT = None


def _claim_stdout() -> None:
    """Keeps the process's real stdout for the ONE JSON line and sends everything else that
    writes to file descriptor 1 (RCCL prints a version banner there when a communicator is
    created) to stderr."""
    global _RESULT_OUT
    if _RESULT_OUT is None:
        sys.stdout.flush()
        _RESULT_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit_result(line: dict) -> None:
    out = _RESULT_OUT or sys.stdout
    out.write(json.dumps(line) + "\n")
    out.flush()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--op", default="msda", choices=["msda", "point_sample"],
                    help="msda: the MSDeformAttn hot path (default, the headline metric); "
                    "point_sample: PointRend point sampling")
    ap.add_argument("--torch-compare", action="store_true",
                    help="point_sample: also time PyTorch-ROCm's own grid_sample on the same GPU")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=None,
                    choices=sorted(harness.WORKLOADS) + sorted(POINT_SAMPLE_WORKLOADS))
    ap.add_argument("--dist", default="R", choices=["R", "U"], help="sampling-location distribution")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch override")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fwd-only", action="store_true")
    ap.add_argument("--packed", action="store_true",
                    help="time the packed locations+weights entry point (MultiScaleFlashAttnFunction's "
                    "kernels) instead of the separate-tensor one")
    ap.add_argument("--layer", action="store_true",
                    help="time the whole MSDeformAttn layer (the four projections on PyTorch-ROCm GEMMs "
                    "around the op) with its real parameter gradients in the all-reduced bucket; "
                    "sampling locations are then the layer's own (ring-initialised offsets)")
    ap.add_argument("--comm", default="inline", choices=["inline", "overlap"],
                    help="multi-rank runs: issue the gradient all-reduce on the compute stream right "
                    "after the backward (inline, default) or on a side stream overlapped with the "
                    "next step (overlap)")
    ap.add_argument("--no-head-major", action="store_true",
                    help="gather from the (N,S,M,D) value as given instead of a per-head copy")
    ap.add_argument("--spawn", action="store_true",
                    help="start the ranks from this script even for --gpus 1 (the self-launcher path "
                    "of a plain `--gpus N` run, with the RCCL process group, rehearsed on one GPU)")
    args = ap.parse_args()
    _claim_stdout()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # (a plain `--gpus N` run never gets here: maybe_launch() turned it into N ranks)
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no GPU visible); there is no CPU fallback")

    import torch.distributed as dist

    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # under torch.distributed.run the process group is set up even for one rank, so that the
    # collective path can be rehearsed on a single GPU
    distributed = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", device_id=device)

    if args.op == "point_sample":
        bench_point_sample(args, device, rank, world, distributed)
        if distributed:
            dist.barrier(device_ids=[local_rank])
            dist.destroy_process_group()
        return
    if args.workload is None:
        args.workload = harness.HEADLINE
    if args.workload not in harness.WORKLOADS:
        raise SystemExit(f"{args.workload} is a point-sampling workload: add --op point_sample")

    from unipercept_amd.deform_attn import _ops  # loads libmsda.so (raises if missing)

    w = harness.WORKLOADS[args.workload]
    batch = args.batch or w.batch
    value, shapes, lsi, loc, attn, grad_out = harness.synth_inputs(
        w, device, batch=batch, seed=1234 + rank, dist=args.dist
    )
    fwd = torch.ops.unipercept.ms_deform_attn_fwd
    bwd = torch.ops.unipercept.ms_deform_attn_bwd
    to_head_major = torch.ops.unipercept.ms_deform_attn_value_head_major
    from unipercept_amd import _lib

    # same sequence as MultiScaleDeformAttnFunction: one per-head re-layout of value per step,
    # shared by the forward and the backward (it is inside the timed region)
    head_major = not args.no_head_major and _ops.head_major_supported(value, loc)
    flags = _lib.FLAG_VALUE_HEAD_MAJOR if head_major else 0

    packed = None
    if args.packed:
        packed = torch.cat([loc.flatten(-3), attn.flatten(-2)], dim=-1).contiguous()
        if not _ops.packed_points_supported(value, packed, w.n_levels):
            raise SystemExit(f"{w.name}: the kernels do not take the packed form for this shape")
        fwd_p = torch.ops.unipercept.ms_flash_attn_fwd
        bwd_p = torch.ops.unipercept.ms_flash_attn_bwd
        del loc, attn

    bucket = None
    comm_stream = None
    if distributed:
        n_params = harness.layer_param_count(w.heads * w.head_dim, w.heads, w.n_levels, w.points)
        bucket = harness.FlatGradBucket([n_params], device)
        bucket.flat.normal_()
        comm_stream = torch.cuda.Stream(device)

    n_ev = args.steps
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n_ev)]

    layer = None
    if args.layer:
        from unipercept_amd.deform_attn import MSDeformAttn

        if args.packed or args.fwd_only or args.no_head_major:
            raise SystemExit("--layer does not combine with --packed / --fwd-only / --no-head-major")
        d_model = w.heads * w.head_dim
        torch.manual_seed(1234)  # identical parameters on every rank, as after DDP's broadcast
        layer = MSDeformAttn(d_model, w.n_levels, w.heads, w.points).to(device, value.dtype)
        gq = torch.Generator(device=device).manual_seed(4321 + rank)
        query = torch.randn(batch, w.positions, d_model, generator=gq, device=device).to(value.dtype)
        src = value.reshape(batch, w.positions, d_model)
        ref = harness.encoder_reference_points(w, device).to(value.dtype)
        ref = ref.view(1, w.positions, 1, 2).expand(batch, w.positions, w.n_levels, 2).contiguous()
        bucket = harness.FlatGradBucket.for_parameters(layer.parameters())  # p.grad are views of it
        if comm_stream is None:
            comm_stream = torch.cuda.Stream(device)
        del loc, attn

    def layer_step(i: int | None) -> None:
        if i is not None:
            ev[i][0].record()
        out = layer(query, ref, src, shapes, lsi)
        if i is not None:
            ev[i][1].record()
        # the previous step's bucket must have been reduced before this backward adds into it
        torch.cuda.current_stream(device).wait_stream(comm_stream)
        bucket.flat.zero_()
        out.backward(grad_out)
        if i is not None:
            ev[i][2].record()
        if distributed:
            reduce_bucket()

    def reduce_bucket() -> None:
        if args.comm == "overlap":
            # the gradient bucket of this step is reduced while the next step computes
            comm_stream.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(comm_stream):
                bucket.all_reduce()
        else:
            bucket.all_reduce()  # behind the backward, on the compute stream

    def step(i: int | None) -> None:
        if layer is not None:
            return layer_step(i)
        if i is not None:
            ev[i][0].record()
        v = to_head_major(value) if head_major else value
        out = fwd_p(v, shapes, lsi, packed, flags) if args.packed else fwd(v, shapes, lsi, loc, attn, flags)
        if i is not None:
            ev[i][1].record()
        if not args.fwd_only:
            if args.packed:
                bwd_p(grad_out, v, shapes, lsi, packed, flags)
            else:
                bwd(grad_out, v, shapes, lsi, loc, attn, flags)
        if i is not None:
            ev[i][2].record()
        del out, v
        if bucket is not None:
            reduce_bucket()

    def fence() -> None:
        torch.cuda.synchronize(device)
        if distributed:
            dist.barrier(device_ids=[local_rank])
            torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step(None)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    fwd_ms = statistics.mean(e[0].elapsed_time(e[1]) for e in ev)
    bwd_ms = statistics.mean(e[1].elapsed_time(e[2]) for e in ev) if not args.fwd_only else 0.0

    if rank == 0:
        pts_per_gpu = w.sampled_points(batch)
        ab = harness.algorithmic_bytes(w, batch)
        ms_per_step = elapsed / args.steps * 1e3
        value_pts = pts_per_gpu * world * args.steps / elapsed
        dom, dom_ms, dom_bytes = (
            ("forward", fwd_ms, ab["fwd"]) if args.fwd_only or fwd_ms > bwd_ms else ("backward", bwd_ms, ab["bwd"])
        )
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters are collected offline (rocprofv3 cannot run
        # inside this process); tools/collect_profiles.sh leaves them in profiles/hbm_traffic.json
        traffic = None
        try:
            rec = json.loads((ROOT / "profiles" / "hbm_traffic.json").read_text()).get(f"{w.name}:{args.dist}")
            if rec is not None and batch == w.batch and not args.packed and not args.layer:
                traffic = rec[dom]
        except (OSError, ValueError, KeyError):
            traffic = None
        step_bytes = ab["fwd"] if args.fwd_only else ab["fwd_bwd"]
        line = {
            "metric": "msdeformattn_fwd_bwd_sampled_points_per_sec"
            if not args.fwd_only
            else "msdeformattn_fwd_sampled_points_per_sec",
            "value": value_pts,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": w.dtype,
            "data": "synthetic",
            "config": {
                "workload": w.name,
                "entry_point": "MSDeformAttn layer (projections on PyTorch-ROCm GEMMs, per-head value epilogue, raw-points attention kernels)"
                if args.layer
                else "MultiScaleFlashAttnFunction kernels (packed locations+weights, read in place)"
                if args.packed
                else "MultiScaleDeformAttnFunction kernels (separate locations / weights)",
                "levels": [list(x) for x in w.levels],
                "heads": w.heads,
                "head_dim": w.head_dim,
                "points": w.points,
                "per_gpu_batch": batch,
                "global_batch": batch * world,
                "queries": w.positions,
                "loc_dtype": w.loc_dtype or w.dtype,
                "location_distribution": "layer (ring-initialised offsets around the pixel centres)" if args.layer else args.dist,
                "value_gather_layout": "per-head copy (N,M,S,D), re-laid out inside the step"
                if head_major
                else "as given (N,S,M,D)",
                "sampled_points_per_gpu_step": pts_per_gpu,
                "parallelism": f"dp{world}",
                "collective": None
                if not distributed
                else f"all_reduce(sum)/world of one flat fp32 bucket of {bucket.flat.numel()} "
                "layer-parameter gradients per step, RCCL, "
                + ("side stream, overlapped with the next step" if args.comm == "overlap"
                   else "on the compute stream behind the backward"),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": f"MSDeformAttn layer {dom} (projection GEMMs included in the time; algorithmic bytes of the op alone)"
                if args.layer
                else f"msda_{dom} (all launches of the op, HIP events in the timed loop)",
                "achieved": achieved,
                "peak": HBM_PEAK / 1e9,
                "unit": "GB/s",
                "frac": achieved * 1e9 / HBM_PEAK,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": dom_bytes,
                "avg_launch_ms": dom_ms,
            },
            "phases": {
                "fwd_ms": fwd_ms,
                "bwd_ms": bwd_ms,
                "fwd_points_per_s": pts_per_gpu / (fwd_ms * 1e-3),
                "fwd_hbm_frac": ab["fwd"] / (fwd_ms * 1e-3) / HBM_PEAK,
                "bwd_hbm_frac": ab["bwd"] / (bwd_ms * 1e-3) / HBM_PEAK if bwd_ms else None,
                "step_hbm_frac": step_bytes / (ms_per_step * 1e-3) / HBM_PEAK,
            },
        }
        if not args.no_cpu_baseline and world == 1:
            # every core this process may use: the box caps a job's share through OMP_NUM_THREADS
            # (os.cpu_count() shows the whole machine there); without the cap, all of them
            threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
            line["cpu_baseline"] = cpu_baseline(threads, args.dist)
        emit_result(line)

    if distributed:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
