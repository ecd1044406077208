#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r2_layer_kt; rm -rf $out; mkdir -p $out
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --layer > $out/bench.json 2> $out/kt.err || { tail -5 $out/kt.err; exit 1; }
python - <<PY
import csv,glob
rows=list(csv.DictReader(open(glob.glob("$out/kt/*/*_kernel_stats.csv")[0])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:28]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e6:7.4f} ms  per-step {float(r["TotalDurationNs"])/13/1e6:7.4f} ms')
print("sum per step", tot/13/1e6)
PY
