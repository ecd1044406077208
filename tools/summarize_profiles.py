"""Turns the raw rocprofv3 CSVs of tools/collect_profiles.sh into a summary (markdown + json)."""
import collections, csv, glob, json, sys

out, tag = sys.argv[1], sys.argv[2]


def kernel_stats():
    rows = []
    for r in csv.DictReader(open(glob.glob(f"{out}/kt/**/*_kernel_stats.csv", recursive=True)[0])):
        if "msda" in r["Name"] or "rocclr" in r["Name"]:
            rows.append({"kernel": r["Name"].split("(")[0], "calls": int(r["Calls"]),
                         "avg_ms": float(r["AverageNs"]) / 1e6, "min_ms": float(r["MinNs"]) / 1e6,
                         "max_ms": float(r["MaxNs"]) / 1e6})
    return rows


def counters(name):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(f"{out}/pmc_{name}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if "msda" in r["Kernel_Name"]:
                agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


stats = kernel_stats()
pmc = {}
for name in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "GRBM_GUI_ACTIVE", "SQ_WAVES", "SQ_LDS_BANK_CONFLICT"):
    for k, d in counters(name).items():
        pmc.setdefault(k, {}).update(d)
bench = json.loads(open(f"{out}/bench.json").read().strip().splitlines()[-1])
summary = {"tag": tag, "bench": bench, "kernels": stats, "pmc_per_launch": pmc}
for k, d in pmc.items():
    # FETCH_SIZE / WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE counts 64 B per 128-B
    # request of a wide coalesced read, i.e. half the bytes; both raw and doubled are kept.
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_read_bytes_raw"] = d["FETCH_SIZE"] * 1024
        d["hbm_read_bytes_x2"] = d["FETCH_SIZE"] * 2048
        d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
    if "TCC_HIT_sum" in d:
        d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
    if "TA_TA_BUSY_sum" in d and d.get("GRBM_GUI_ACTIVE"):
        # TA_TA_BUSY is summed over the 256 CUs, GRBM_GUI_ACTIVE over the 8 XCDs
        cyc = d["GRBM_GUI_ACTIVE"] / 8
        d["ta_busy_frac"] = d["TA_TA_BUSY_sum"] / 256 / cyc
        d["l1_accesses_per_clk_per_cu"] = d["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256 / cyc
        if "SQ_INSTS_VALU" in d:
            # a wave64 VALU instruction occupies its SIMD for 4 cycles; 1024 SIMDs
            d["valu_busy_frac"] = d["SQ_INSTS_VALU"] * 4 / (1024 * cyc)
json.dump(summary, open(f"{out}/summary_{tag}.json", "w"), indent=1)
with open(f"{out}/summary_{tag}.md", "w") as f:
    f.write(f"# rocprofv3 summary `{tag}`\n\nbench line (un-profiled run, same command):\n\n```\n{json.dumps(bench)}\n```\n\n")
    f.write("## kernel trace (`rocprofv3 --kernel-trace --stats`)\n\n| kernel | calls | avg ms | min ms | max ms |\n|---|---|---|---|---|\n")
    for r in stats:
        f.write(f"| `{r['kernel']}` | {r['calls']} | {r['avg_ms']:.4f} | {r['min_ms']:.4f} | {r['max_ms']:.4f} |\n")
    f.write("\n## counters per launch (`rocprofv3 --pmc`, one pass per group)\n\n")
    f.write("| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM read B (x2 corrected) | HBM write B | L2 hit rate | TCC_EA0_RDREQ | TCC_EA0_WRREQ |\n|---|---|---|---|---|---|---|---|\n")
    for k, d in pmc.items():
        f.write(f"| `{k}` | {d.get('FETCH_SIZE', 0):.0f} | {d.get('WRITE_SIZE', 0):.0f} | {d.get('hbm_read_bytes_x2', 0):.3e} | {d.get('hbm_write_bytes', 0):.3e} | {d.get('l2_hit_rate', 0):.3f} | {d.get('TCC_EA0_RDREQ_sum', 0):.3e} | {d.get('TCC_EA0_WRREQ_sum', 0):.3e} |\n")
    f.write("\n## pipeline counters per launch\n\n| kernel | cycles (GRBM_GUI_ACTIVE / 8 XCDs) | TA busy | L1 accesses / clk / CU | VALU busy (4 clk per wave instr, 1024 SIMDs) | VALU instr | SALU instr | LDS instr | VMEM rd | VMEM wr | waves |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
    for k, d in pmc.items():
        if "ta_busy_frac" not in d:
            continue
        f.write(f"| `{k}` | {d['GRBM_GUI_ACTIVE'] / 8:.3e} | {d['ta_busy_frac']:.2f} | {d['l1_accesses_per_clk_per_cu']:.2f} | {d.get('valu_busy_frac', 0):.2f} | {d.get('SQ_INSTS_VALU', 0):.3e} | {d.get('SQ_INSTS_SALU', 0):.3e} | {d.get('SQ_INSTS_LDS', 0):.3e} | {d.get('SQ_INSTS_VMEM_RD', 0):.3e} | {d.get('SQ_INSTS_VMEM_WR', 0):.3e} | {d.get('SQ_WAVES', 0):.0f} |\n")
    f.write("\n## wave state and LDS per launch (sums over the chip)\n\nWAIT_ANY: wave parked at s_waitcnt or a barrier; WAIT_INST_ANY: wave ready but its instruction cannot issue (the pipeline it needs is busy); LDS bank conflict = conflict cycles / LDS-array cycles; LDS busy = LDS-array cycles per CU / kernel cycles.\n\n| kernel | wave cycles | WAIT_ANY | WAIT_INST_ANY | WAIT_INST_LDS | issuing (ACTIVE_INST_ANY) | LDS bank conflict | LDS busy |\n|---|---|---|---|---|---|---|---|\n")
    for k, d in pmc.items():
        if "SQ_WAIT_ANY" not in d or not d.get("SQ_WAVE_CYCLES"):
            continue
        wc = d["SQ_WAVE_CYCLES"]
        cyc = d.get("GRBM_GUI_ACTIVE", 0) / 8
        d["lds_bank_conflict_frac"] = d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1)
        d["lds_busy_frac"] = d["SQ_LDS_IDX_ACTIVE"] / 256 / cyc if cyc else 0
        f.write(f"| `{k}` | {wc:.3e} | {d['SQ_WAIT_ANY'] / wc:.2f} | {d['SQ_WAIT_INST_ANY'] / wc:.2f} | {d['SQ_WAIT_INST_LDS'] / wc:.3f} | {d['SQ_ACTIVE_INST_ANY'] / wc:.2f} | {d['lds_bank_conflict_frac']:.2f} | {d['lds_busy_frac']:.2f} |\n")
json.dump(summary, open(f"{out}/summary_{tag}.json", "w"), indent=1)
print(open(f"{out}/summary_{tag}.md").read())
