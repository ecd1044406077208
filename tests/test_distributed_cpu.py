"""The data-parallel path on CPU: two gloo ranks, batch sharded, one flat gradient bucket.

The ranks run the MSDeformAttn layer with the CPU oracle injected as the sampling op (the HIP op
needs a GPU); what is under test is the harness: sharding, the bucket, and that the averaged
gradients equal those of the whole batch in one process.
"""

from __future__ import annotations

import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent

LEVELS = [(6, 8), (3, 4)]
GLOBAL_BATCH = 4


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_layer():
    from oracle import grid_sample_ref
    from unipercept_amd.deform_attn import MSDeformAttn

    torch.manual_seed(42)
    layer = MSDeformAttn(d_model=32, n_levels=2, n_heads=4, n_points=2).double()
    torch.nn.init.normal_(layer.sampling_offsets.weight, std=0.05)
    torch.nn.init.normal_(layer.attention_weights.weight, std=0.05)
    layer.op = lambda v, sh, li, lo, a, step: grid_sample_ref.deform_attn(v, sh, lo, a)
    return layer


def _data():
    g = torch.Generator().manual_seed(7)
    s = sum(h * w for h, w in LEVELS)
    feats = torch.randn(GLOBAL_BATCH, s, 32, generator=g, dtype=torch.float64)
    ref = torch.rand(GLOBAL_BATCH, s, len(LEVELS), 2, generator=g, dtype=torch.float64)
    target = torch.randn(GLOBAL_BATCH, s, 32, generator=g, dtype=torch.float64)
    return feats, ref, target


def _loss(layer, feats, ref, target):
    shapes = torch.tensor(LEVELS)
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    out = layer(feats, ref, feats, shapes, lsi)
    return ((out - target) ** 2).mean()


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    for p in (str(ROOT), str(ROOT / "shim")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from unipercept_amd import harness

        layer = _make_layer()
        bucket = harness.FlatGradBucket.for_parameters(layer.parameters())
        feats, ref, target = _data()
        mine = list(harness.shard_batch(GLOBAL_BATCH, world, rank))
        loss = _loss(layer, feats[mine], ref[mine], target[mine])
        loss.backward()
        # grads were accumulated into the bucket's views (p.grad aliases the bucket)
        for p, v in zip(layer.parameters(), bucket.views):
            assert p.grad.data_ptr() == v.data_ptr()
        work = bucket.all_reduce(async_op=True)
        work.wait()
        torch.save(bucket.flat.clone(), os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_gradient_average_equals_full_batch(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(tmp_path / f"rank{r}.pt", weights_only=True) for r in range(world)]
    assert torch.equal(got[0], got[1])

    layer = _make_layer()
    feats, ref, target = _data()
    loss = _loss(layer, feats, ref, target)  # mean over the whole batch
    loss.backward()
    want = torch.cat([p.grad.reshape(-1) for p in layer.parameters()])
    assert torch.allclose(got[0].double(), want, rtol=1e-5, atol=1e-8)


@pytest.mark.timeout(180)
def test_launcher_spawns_one_rank_per_process(tmp_path):
    """`python bench.py --gpus N` invoked plainly becomes a launcher (harness.launch_ranks): N
    children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rank 0's line on the inherited
    stdout, a failing rank fails the launch.  Rehearsed here with two gloo ranks."""
    import json
    import subprocess

    probe = str(ROOT / "tests" / "_rank_probe.py")
    code = (
        "import sys; sys.path.insert(0, %r); from unipercept_amd import harness; "
        "sys.exit(harness.launch_ranks(%r, sys.argv[1:], 2, timeout=120))" % (str(ROOT), probe)
    )
    ok = subprocess.run([sys.executable, "-c", code, "--steps", "3"], capture_output=True, text=True, timeout=170)
    assert ok.returncode == 0, ok.stderr[-2000:]
    lines = [ln for ln in ok.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec == {"world": 2, "sum": 3.0, "argv": ["--steps", "3"]}
    bad = subprocess.run([sys.executable, "-c", code, "--fail-rank", "1"], capture_output=True, text=True, timeout=170)
    assert bad.returncode == 3


def test_bench_refuses_more_ranks_than_gpus():
    """Without GPUs the plain `--gpus 2` invocation ends with a clear message: from the launcher
    when sysfs shows fewer GPUs, else from the ranks it started (there is no CPU fallback)."""
    import subprocess

    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=170)
    assert r.returncode != 0
    assert "GPU(s) are visible" in r.stderr or "no GPU visible" in r.stderr


_LAUNCH_SPY = r"""
import json, os, runpy, subprocess, sys
seen = []
class Spy:
    def __init__(self, cmd, env=None, **kw):
        # what the parent has loaded at the moment it starts a rank
        seen.append({
            "torch_loaded": any(m == "torch" or m.startswith("torch.") for m in sys.modules),
            "ctypes_hip": any("amdhip" in (getattr(m, "__file__", "") or "") for m in list(sys.modules.values())),
            "cmd": cmd[1:], "rank": env["RANK"], "world": env["WORLD_SIZE"],
            "addr": env["MASTER_ADDR"], "ipc": env.get("HSA_ENABLE_IPC_MODE_LEGACY"),
        })
        self.returncode = 0
    def poll(self): return 0
    def wait(self, timeout=None): return 0
    def terminate(self): pass
    def kill(self): pass
subprocess.Popen = Spy
sys.argv = [sys.argv[1]] + sys.argv[2:]
try:
    runpy.run_path(sys.argv[0], run_name="__main__")
except SystemExit as e:
    print(json.dumps({"exit": e.code, "seen": seen}))
"""


@pytest.mark.parametrize("argv", [["--gpus", "2", "--steps", "3"], ["--gpus", "1", "--spawn"]])
def test_launcher_parent_never_loads_torch(argv):
    """The self-launcher of `python bench.py --gpus N` starts its ranks before torch (and with it
    anything that could initialise the HIP runtime) has been imported into the parent; `--spawn`
    takes the same path with one rank.  The ranks get the torch.distributed.run environment and
    the launcher's own arguments."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run(
        [sys.executable, "-c", _LAUNCH_SPY, str(ROOT / "bench.py"), *argv],
        capture_output=True, text=True, timeout=120, env=env,
    )
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    n = int(argv[1])
    assert rec["exit"] == 0 and len(rec["seen"]) == n
    for rank, s in enumerate(rec["seen"]):
        assert not s["torch_loaded"] and not s["ctypes_hip"]
        assert s["cmd"] == [str(ROOT / "bench.py"), *argv]
        assert (s["rank"], s["world"], s["addr"], s["ipc"]) == (str(rank), str(n), "127.0.0.1", "0")


def test_visible_gpu_count_reads_no_runtime(monkeypatch):
    from unipercept_amd import launch

    got = launch.visible_gpu_count()
    assert got is None or got >= 0
    import glob as _glob

    monkeypatch.setattr(_glob, "glob", lambda pattern: [])
    assert launch.visible_gpu_count() is None
