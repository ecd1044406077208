#!/bin/bash
# first GPU pass of the round: GPU tests, plain bench, 1-rank torchrun rehearsal (inline / overlap)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_tests1.log 2>&1
rc=$?; tail -5 gpurun_out/r2_tests1.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python bench.py --steps 30 --warmup 5 > gpurun_out/r2_bench_plain.json 2> gpurun_out/r2_bench_plain.err || exit 1
for mode in inline overlap; do
  timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --comm $mode > gpurun_out/r2_bench_tr1_$mode.json 2> gpurun_out/r2_bench_tr1_$mode.err || exit 1
done
python - <<'PY'
import json
for n in ("plain", "tr1_inline", "tr1_overlap"):
    d = json.load(open(f"gpurun_out/r2_bench_{n}.json"))
    print(n, round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["phases"].items() if k.endswith("_ms")}, d.get("cpu_baseline", {}).get("cores"))
PY
