"""One process per GPU, started by a parent that never touches the GPU.

This module imports nothing that can load the HIP runtime (no torch, no ctypes binding of the
library): a process that has initialised the GPU must not start other programs on this kind of
machine, so the launcher decides and launches before any such import happens.  The pattern is the
reference's only parallelism — one DDP rank per device
(sources/unipercept/engine/_engine.py:848, batch split over the ranks as in
sources/unipercept/data/_loader.py:49-57).
"""

from __future__ import annotations

import glob
import os
import socket
import subprocess
import sys
import time


def visible_gpu_count() -> int | None:
    """GPUs this process would see, found WITHOUT the HIP runtime: the KFD topology in sysfs
    (nodes with a non-zero simd_count are GPUs), narrowed by ROCR_VISIBLE_DEVICES /
    HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when one of them is set.  None when sysfs has no
    KFD topology (the ranks then find out for themselves)."""
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    gpus = 0
    for path in nodes:
        try:
            with open(path) as fh:
                for line in fh:
                    key, _, val = line.partition(" ")
                    if key == "simd_count":
                        gpus += int(val) > 0
                        break
        except (OSError, ValueError):
            continue
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        listed = os.environ.get(var)
        if listed is not None:
            gpus = min(gpus, len([x for x in listed.split(",") if x.strip()]))
    return gpus


def launch_ranks(script: str, argv: list[str], nproc: int, *, env: dict | None = None,
                 timeout: float | None = None) -> int:
    """Starts `nproc` copies of `python script *argv`, one rank per process (RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR, MASTER_PORT in their environment, as torch.distributed.run sets
    them), waits for all of them and returns 0 when every rank exited with 0, else the first
    non-zero exit code (the other ranks are then terminated).  The children inherit stdout and
    stderr, so whatever rank 0 prints is the launcher's own output.

    The parent must not have touched the GPU: it only starts children and waits."""
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    base = dict(os.environ if env is None else env)
    base.update(
        WORLD_SIZE=str(nproc), LOCAL_WORLD_SIZE=str(nproc), MASTER_ADDR="127.0.0.1",
        MASTER_PORT=str(port),
    )
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(nproc):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, script, *argv], env=e))
    rc = 0
    try:
        pending = set(range(nproc))
        deadline = None if timeout is None else (time.monotonic() + timeout)
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code
            if rc != 0 or (deadline is not None and time.monotonic() > deadline):
                if rc == 0:
                    rc = 124
                break
            if pending:
                try:
                    procs[min(pending)].wait(timeout=0.2)
                except subprocess.TimeoutExpired:
                    pass
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.terminate()
        for pr in procs:
            try:
                pr.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pr.kill()
    return rc
