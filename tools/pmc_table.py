"""Markdown table of every rocprofv3 --pmc pass found under a directory: one row per kernel, one
column per counter (mean over the launches of the kernel).

    python tools/pmc_table.py gpurun_out/r3_lds_pmc
"""
import collections, csv, glob, sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(path)):
        if "msda" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:64]
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
counters = sorted({c for v in agg.values() for c in v})
print("| kernel | " + " | ".join(counters) + " |")
print("|---|" + "---|" * len(counters))
for k, v in agg.items():
    cells = [f"{sum(v[c]) / len(v[c]):.4g}" if c in v else "" for c in counters]
    print(f"| `{k}` | " + " | ".join(cells) + " |")
