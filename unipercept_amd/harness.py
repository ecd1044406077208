"""Batch-sharded benchmark harness for the MSDeformAttn hot path: named workloads, seeded
synthetic inputs, algorithmic-byte accounting, batch sharding and the flat gradient bucket.

The reference has no benchmark directory (its ``make benchmark`` target, Makefile:27-28, points
at nothing); the data-parallel pattern mirrored here is the reference's only one: per-rank sample
sharding with a batch that must divide the world size (sources/unipercept/data/_loader.py:49-57)
and a summed-then-averaged gradient all-reduce (DDP, sources/unipercept/engine/_engine.py:848).
"""

from __future__ import annotations

import dataclasses
import math
import os

import torch


@dataclasses.dataclass(frozen=True)
class Workload:
    name: str
    levels: tuple[tuple[int, int], ...]  # (H, W) per level
    heads: int
    head_dim: int
    points: int
    batch: int  # per GPU
    dtype: str  # "f32" | "bf16" | "f16" | "f64"
    loc_dtype: str | None = None  # None = same as dtype

    @property
    def positions(self) -> int:
        return sum(h * w for h, w in self.levels)

    @property
    def n_levels(self) -> int:
        return len(self.levels)

    def sampled_points(self, batch: int | None = None, queries: int | None = None) -> int:
        b = self.batch if batch is None else batch
        q = self.positions if queries is None else queries
        return b * q * self.heads * self.n_levels * self.points


_TORCH_DTYPE = {
    "f32": torch.float32,
    "f64": torch.float64,
    "bf16": torch.bfloat16,
    "f16": torch.float16,
}
_ESIZE = {"f32": 4, "f64": 8, "bf16": 2, "f16": 2}


def torch_dtype(tag: str) -> torch.dtype:
    return _TORCH_DTYPE[tag]


def _pyramid(h: int, w: int, strides) -> tuple[tuple[int, int], ...]:
    return tuple((h // s, w // s) for s in strides)


# Workloads of the benchmark contract.  Encoder style: one query per position (Lq = S).
WORKLOADS = {
    # CPU-reference case: 512x1024 input, strides 8/16/32
    "cpuref_512x1024_fp32_b1": Workload(
        "cpuref_512x1024_fp32_b1", _pyramid(512, 1024, (8, 16, 32)), 8, 32, 4, 1, "f32"
    ),
    # Cityscapes 1024x2048 pixel-decoder encoder, strides 8/16/32, fp32, batch 8 on one GPU
    "cityscapes_1024x2048_fp32_b8": Workload(
        "cityscapes_1024x2048_fp32_b8", _pyramid(1024, 2048, (8, 16, 32)), 8, 32, 4, 8, "f32"
    ),
    # same shapes in bf16 (loc/attn bf16 too); batch 8 per GPU = 64 over 8 GPUs
    "cityscapes_1024x2048_bf16_b8": Workload(
        "cityscapes_1024x2048_bf16_b8", _pyramid(1024, 2048, (8, 16, 32)), 8, 32, 4, 8, "bf16"
    ),
    # bf16 value/out/grads with fp32 locations and weights
    "cityscapes_1024x2048_bf16_locf32_b8": Workload(
        "cityscapes_1024x2048_bf16_locf32_b8", _pyramid(1024, 2048, (8, 16, 32)), 8, 32, 4, 8,
        "bf16", "f32",
    ),
    # COCO 1024x1024 Mask2Former-style, strides 8/16/32/64
    "coco_1024x1024_L4_bf16_b16": Workload(
        "coco_1024x1024_L4_bf16_b16", _pyramid(1024, 1024, (8, 16, 32, 64)), 8, 32, 4, 16, "bf16"
    ),
    # DVPS clip: frames folded into the batch, 4 levels, 8 points
    "dvps_1024x2048_L4_P8_fp32_b8": Workload(
        "dvps_1024x2048_L4_P8_fp32_b8", _pyramid(1024, 2048, (8, 16, 32, 64)), 8, 32, 8, 8, "f32"
    ),
    "dvps_1024x2048_L4_P8_bf16_b8": Workload(
        "dvps_1024x2048_L4_P8_bf16_b8", _pyramid(1024, 2048, (8, 16, 32, 64)), 8, 32, 8, 8, "bf16"
    ),
}

HEADLINE = "cityscapes_1024x2048_fp32_b8"


def algorithmic_bytes(w: Workload, batch: int | None = None) -> dict[str, int]:
    """Bytes if every tensor of the op is touched exactly once (Lq = S).

    forward : read value, loc, attn; write out            -> per (b,q,m): 2*D*e_v + 3*L*P*e_l
    backward: read grad_out, value, loc, attn; write grad_value, grad_loc, grad_attn
                                                          -> per (b,q,m): 3*D*e_v + 6*L*P*e_l
    """
    b = w.batch if batch is None else batch
    ev = _ESIZE[w.dtype]
    el = _ESIZE[w.loc_dtype or w.dtype]
    triples = b * w.positions * w.heads
    lp = w.n_levels * w.points
    fwd = triples * (2 * w.head_dim * ev + 3 * lp * el)
    bwd = triples * (3 * w.head_dim * ev + 6 * lp * el)
    return {"fwd": fwd, "bwd": bwd, "fwd_bwd": fwd + bwd}


def level_tensors(w: Workload, device) -> tuple[torch.Tensor, torch.Tensor]:
    shapes = torch.as_tensor(w.levels, dtype=torch.long, device=device)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    return shapes, lsi


def encoder_reference_points(w: Workload, device, dtype=torch.float32) -> torch.Tensor:
    """(S, 2) centre (x, y) of every position in its own level, normalised to [0, 1]."""
    refs = []
    for h, wd in w.levels:
        ys = (torch.arange(h, device=device, dtype=dtype) + 0.5) / h
        xs = (torch.arange(wd, device=device, dtype=dtype) + 0.5) / wd
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        refs.append(torch.stack((xx.reshape(-1), yy.reshape(-1)), -1))
    return torch.cat(refs, 0)


def synth_inputs(
    w: Workload,
    device,
    *,
    batch: int | None = None,
    seed: int = 1234,
    dist: str = "R",
    value_scale: float | None = None,
    oob_fraction: float = 0.0,
):
    """Seeded synthetic (value, shapes, lsi, loc, attn, grad_out) resident on `device`.

    dist "U": locations uniform in [0, 1] (the reference test's distribution, worst locality).
    dist "R": encoder pattern — query q sits at a pixel centre of its own level, that reference
    point is shared by all levels, and point k = 1..P on level l is displaced by a gaussian of
    sigma 2k pixels of level l (nearby, growing with k).
    """
    b = w.batch if batch is None else batch
    s, m, d, l, p = w.positions, w.heads, w.head_dim, w.n_levels, w.points
    vdt = torch_dtype(w.dtype)
    ldt = torch_dtype(w.loc_dtype or w.dtype)
    gen = torch.Generator(device=device).manual_seed(seed)
    shapes, lsi = level_tensors(w, device)
    value = torch.randn(b, s, m, d, generator=gen, device=device, dtype=torch.float32)
    if value_scale is not None:
        value = value * value_scale
    attn = torch.randn(b, s, m, l * p, generator=gen, device=device, dtype=torch.float32)
    attn = attn.softmax(-1).view(b, s, m, l, p)
    if dist == "U":
        loc = torch.rand(b, s, m, l, p, 2, generator=gen, device=device, dtype=torch.float32)
    elif dist == "R":
        ref = encoder_reference_points(w, device)  # (S, 2)
        wh = torch.stack((shapes[:, 1], shapes[:, 0]), -1).to(torch.float32)  # (L, 2) as (W, H)
        sigma = 2.0 * torch.arange(1, p + 1, device=device, dtype=torch.float32)  # (P,)
        off = torch.randn(b, s, m, l, p, 2, generator=gen, device=device, dtype=torch.float32)
        off = off * sigma.view(1, 1, 1, 1, p, 1) / wh.view(1, 1, 1, l, 1, 2)
        loc = ref.view(1, s, 1, 1, 1, 2) + off
        del off
    else:
        raise ValueError(f"unknown location distribution {dist!r}")
    if oob_fraction > 0:
        mask = torch.rand(b, s, m, l, p, 1, generator=gen, device=device) < oob_fraction
        loc = torch.where(mask, loc * 3.0 - 1.0, loc)
    grad_out = torch.randn(b, s, m * d, generator=gen, device=device, dtype=torch.float32)
    return (
        value.to(vdt), shapes, lsi, loc.to(ldt).contiguous(), attn.to(ldt).contiguous(),
        grad_out.to(vdt),
    )


def shard_batch(global_batch: int, world_size: int, rank: int) -> range:
    """Sample indices of `rank`: contiguous equal shards; the batch must divide the world size
    (same rule as the reference's loader, sources/unipercept/data/_loader.py:49-57)."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError(f"bad rank {rank} for world size {world_size}")
    if global_batch % world_size != 0:
        raise ValueError(
            f"global batch {global_batch} is not divisible by the world size {world_size}"
        )
    per = global_batch // world_size
    return range(rank * per, (rank + 1) * per)


def layer_param_count(d_model: int, heads: int, levels: int, points: int) -> int:
    """Parameters of one MSDeformAttn layer: sampling_offsets, attention_weights, value_proj,
    output_proj (weights + biases)."""
    off = d_model * heads * levels * points * 2 + heads * levels * points * 2
    att = d_model * heads * levels * points + heads * levels * points
    proj = 2 * (d_model * d_model + d_model)
    return off + att + proj


class FlatGradBucket:
    """One flat bucket (fp32 unless built from parameters of another dtype) for the layer's parameter gradients, all-reduced (sum, then divided
    by the world size: DDP semantics) in a single collective.  The tensors handed in become views
    into the bucket, so no copy happens per step."""

    def __init__(self, numels: list[int], device, dtype=torch.float32):
        self.numels = list(numels)
        self.flat = torch.zeros(sum(self.numels), device=device, dtype=dtype)
        self.views = []
        o = 0
        for n in self.numels:
            self.views.append(self.flat[o : o + n])
            o += n

    @classmethod
    def for_parameters(cls, params) -> "FlatGradBucket":
        params = [p for p in params if p.requires_grad]
        bucket = cls([p.numel() for p in params], params[0].device, params[0].dtype)
        for p, v in zip(params, bucket.views):
            p.grad = v.view_as(p)
        return bucket

    def all_reduce(self, group=None, async_op: bool = False):
        import torch.distributed as dist

        world = dist.get_world_size(group)
        self.flat.div_(world)
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


from .launch import launch_ranks  # noqa: E402,F401  (torch-free; re-exported for callers of the harness)


def roofline_fraction(bytes_moved: float, seconds: float, peak_bytes_per_s: float = 8.0e12) -> float:
    return bytes_moved / seconds / peak_bytes_per_s if seconds > 0 else math.nan
