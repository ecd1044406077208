#!/bin/bash
# rocprofv3 evidence for the point-sampling kernels (GPU box): kernel trace + three PMC passes.
#   usage: tools/profile_point_sample.sh <tag> [bench.py --op point_sample args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/prof_$tag
mkdir -p "$out"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python bench.py --op point_sample --steps 20 --warmup 5 --no-cpu-baseline "$@" > "$out/bench_under_trace.json" 2> "$out/kt.err" || exit 1
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo "$pass" | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d "$out/pmc_$name" -- python bench.py --op point_sample --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> "$out/pmc_$name.err" || exit 1
done
timeout -k 10 200 python bench.py --op point_sample --torch-compare "$@" > "$out/bench.json" 2> "$out/bench.err" || exit 1
python - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {"bench": json.load(open(f"{out}/bench.json")), "kernels": {}}
for r in csv.DictReader(open(glob.glob(f"{out}/kt/*/*_kernel_stats.csv")[0])):
    if "msda::point_" in r["Name"]:
        res["kernels"][r["Name"]] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6}
for name in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/pmc_{name}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "msda::point_" in r["Kernel_Name"]:
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            res["kernels"].setdefault(k, {})[c + "_per_launch"] = sum(v) / len(v)
for k, d in res["kernels"].items():
    if "FETCH_SIZE_per_launch" in d:
        d["hbm_read_bytes"] = d["FETCH_SIZE_per_launch"] * 1024 * 2  # KiB; gfx950 tallies 64 B per 128-B request
    if "WRITE_SIZE_per_launch" in d:
        d["hbm_write_bytes"] = d["WRITE_SIZE_per_launch"] * 1024
    if "TCC_HIT_sum_per_launch" in d:
        d["l2_hit_rate"] = d["TCC_HIT_sum_per_launch"] / (d["TCC_HIT_sum_per_launch"] + d["TCC_MISS_sum_per_launch"])
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
