#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof_*) into the small tracked files under profiles/.

    python scripts/summarize_prof.py r01_c1 gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_tcc [kernel-substring]

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary, verbatim),
profiles/<tag>_pmc.json (per-launch counter means for the dominant kernel, with the gfx950
corrections of MI355X_MICROARCH.md section HBM applied: FETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE
reports 1/2 of the bytes of 16-B-per-lane coalesced reads -> doubled) and profiles/traffic_latest.json.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(d, kernel):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {"mean": sum(v) / len(v), "n": len(v)} for k, v in acc.items()}


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    pmc_dirs = [a for a in sys.argv[3:] if os.path.isdir(a)]
    kernel = next((a for a in sys.argv[3:] if not os.path.isdir(a)), "spmm_rows")
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    ks = glob.glob(os.path.join(stats_dir, "**", "*_kernel_stats.csv"), recursive=True)
    summary = {"tag": tag, "kernel_filter": kernel}
    if ks:
        shutil.copy(ks[0], os.path.join(out, f"{tag}_kernel_stats.csv"))
        for r in csv.DictReader(open(ks[0])):
            if kernel in r["Name"]:
                summary["kernel"] = r["Name"]
                summary["calls"] = int(r["Calls"])
                summary["avg_ns"] = float(r["AverageNs"])
                summary["min_ns"] = float(r["MinNs"])
                summary["max_ns"] = float(r["MaxNs"])
                summary["pct_of_gpu_time"] = float(r["Percentage"])
                break
    c = {}
    for d in pmc_dirs:
        c.update(counters(d, kernel))
    summary["counters_per_launch"] = c
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        fetch = c["FETCH_SIZE"]["mean"] * 1024.0
        write = c["WRITE_SIZE"]["mean"] * 1024.0
        summary["fetch_bytes_raw"] = fetch
        summary["fetch_bytes_corrected_x2"] = 2.0 * fetch
        summary["write_bytes"] = write
        summary["hbm_bytes_per_launch"] = 2.0 * fetch + write
        summary["note"] = ("FETCH_SIZE/WRITE_SIZE in KiB; FETCH_SIZE doubled (gfx950: 128-B requests tallied at 64 B for "
                           "16-B-per-lane coalesced reads, MI355X_MICROARCH.md HBM section); counters sit on the L2's "
                           "memory side, so Infinity-Cache hits are included")
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        h, m = c["TCC_HIT_sum"]["mean"], c["TCC_MISS_sum"]["mean"]
        summary["l2_hit_rate"] = h / (h + m)
    json.dump(summary, open(os.path.join(out, f"{tag}_pmc.json"), "w"), indent=1)
    if "hbm_bytes_per_launch" in summary:
        json.dump({"source": f"profiles/{tag}_pmc.json", "hbm_bytes_per_launch": summary["hbm_bytes_per_launch"]},
                  open(os.path.join(out, "traffic_latest.json"), "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
