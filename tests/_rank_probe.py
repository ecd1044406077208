"""Helper of test_distributed_cpu.py (not a test): one rank of a gloo group started by
`harness.launch_ranks`.  All-reduces (rank + 1) and lets rank 0 print one JSON line, the way
bench.py's ranks do."""

import json
import os
import sys

import torch
import torch.distributed as dist


def main() -> None:
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert int(os.environ["LOCAL_RANK"]) == rank
    if "--fail-rank" in sys.argv and rank == int(sys.argv[sys.argv.index("--fail-rank") + 1]):
        raise SystemExit(3)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"world": world, "sum": float(t.item()), "argv": sys.argv[1:]}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
