"""Copies the summaries of tools/collect_profiles.sh from gpurun_out/ into profiles/ and rebuilds
profiles/hbm_traffic.json (what bench.py reports as roofline.traffic)."""
import glob, json, os, shutil

os.makedirs("profiles", exist_ok=True)
traffic = {}
for d in sorted(glob.glob("gpurun_out/prof_*")):
    tag = os.path.basename(d)[len("prof_"):]
    js = f"{d}/summary_{tag}.json"
    if not os.path.exists(js):
        continue
    shutil.copy(js, f"profiles/summary_{tag}.json")
    shutil.copy(f"{d}/summary_{tag}.md", f"profiles/summary_{tag}.md")
    for f in glob.glob(f"{d}/kt/**/*_kernel_stats.csv", recursive=True):
        shutil.copy(f, f"profiles/{tag}_kernel_stats.csv")
    s = json.load(open(js))
    cfg = s["bench"]["config"]
    key = f"{cfg['workload']}:{cfg['location_distribution']}"
    fwd = bwd = 0.0
    for k, v in s["pmc_per_launch"].items():
        b = v.get("hbm_read_bytes_x2", 0.0) + v.get("hbm_write_bytes", 0.0)
        if "bwd_" in k:
            bwd += b
        else:  # forward kernel and the value re-layout that precedes it
            fwd += b
    traffic[key] = {
        "source": f"profiles/summary_{tag}.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate "
        "passes; KiB -> bytes; FETCH_SIZE doubled: gfx950 counts 64 B per 128-B read request)",
        "forward": fwd,
        "backward": bwd,
    }
# point-sampling profiles (tools/profile_point_sample.sh): one summary.json per directory
for d in sorted(glob.glob("gpurun_out/prof_*point_sample*")):
    tag = os.path.basename(d)[len("prof_"):]
    js = f"{d}/summary.json"
    if not os.path.exists(js):
        continue
    shutil.copy(js, f"profiles/summary_{tag}.json")
    for f in glob.glob(f"{d}/kt/**/*_kernel_stats.csv", recursive=True):
        shutil.copy(f, f"profiles/{tag}_kernel_stats.csv")
    s = json.load(open(js))
    fwd = bwd = 0.0
    for k, v in s["kernels"].items():
        b = v.get("hbm_read_bytes", 0.0) + v.get("hbm_write_bytes", 0.0)
        if "_bwd_" in k:
            bwd += b
        else:
            fwd += b
    traffic[s["bench"]["config"]["workload"]] = {
        "source": f"profiles/summary_{tag}.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate "
        "passes; KiB -> bytes; FETCH_SIZE doubled)",
        "forward": fwd,
        "backward": bwd,
    }
json.dump(traffic, open("profiles/hbm_traffic.json", "w"), indent=1)
print(json.dumps(traffic, indent=1))
