#!/bin/bash
# usage: tools/ktime.sh <outdir> [bench args...]  -- per-kernel average times via rocprofv3
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p $out
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $out/bench.json 2> $out/kt.err
python - <<PY
import csv,glob,json
try:
    print({k: round(v,3) if isinstance(v,float) else v for k,v in json.load(open("$out/bench.json"))["phases"].items()})
except Exception as e: print("bench failed", e)
for r in csv.DictReader(open(glob.glob("$out/kt/*/*_kernel_stats.csv")[0])):
    if "msda" in r["Name"] or "rocclr" in r["Name"]: print(r["Name"][:60], r["Calls"], round(float(r["AverageNs"])/1e6,4),"ms")
PY
