"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel."""
import collections, csv, glob, sys
path = glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
    if "msda" in r["Kernel_Name"]:
        agg[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"    {c:28s} mean {sum(x)/len(x):16.1f}  (n={len(x)})")
