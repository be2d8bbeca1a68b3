#!/usr/bin/env python3
"""Condense the passes of scripts/prof.sh into profiles/<tag>_profile.json (+ the verbatim kernel_stats.csv).

    python scripts/summarize_prof.py <tag> <steps> <kernel-substring> [--traffic-key C1 --feat 128]

Reads gpurun_out/<tag>_{stats,fetch,write,tcc,sq}/ (rocprofv3 --output-format csv).  For every kernel whose name
contains the substring: calls, average duration, per-launch counter means; and PER STEP (= one run() of the
operator, which may be several launches of several kernels): time, traffic (FETCH_SIZE/WRITE_SIZE are KiB;
FETCH_SIZE doubled -- gfx950 tallies 128-byte requests of 16-byte-per-lane reads at 64 bytes,
MI355X_MICROARCH.md, HBM section; the counters sit on the L2's memory side, so Infinity-Cache hits are
included), L2 hit rate, MFMA busy share.  `steps` = timed + warm-up launches of the whole step in the profiled
command (every launch of a matching kernel is assumed to belong to a step).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(files):
    """gpurun MERGES what a call wrote into gpurun_out/: a directory profiled twice holds both passes' files (<pid>_*.csv).  Only the newest pass counts --
    two passes summed would double every per-step figure."""
    return [max(files, key=os.path.getmtime)] if files else []


def counters(d, sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in newest(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    tag, steps, sub = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    base = os.path.join(ROOT, "gpurun_out", tag)
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    summary = {"tag": tag, "kernel_filter": sub, "steps": steps, "kernels": {}}
    ks = newest(glob.glob(os.path.join(base + "_stats", "**", "*kernel_stats.csv"), recursive=True))
    if ks:
        shutil.copy(ks[0], os.path.join(out, f"{tag}_kernel_stats.csv"))
        for r in csv.DictReader(open(ks[0])):
            if sub in r["Name"]:
                summary["kernels"][r["Name"]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                                                 "max_ns": float(r["MaxNs"]), "pct_of_gpu_time": float(r["Percentage"])}
    tot = collections.defaultdict(float)
    for p in ("fetch", "write", "tcc", "sq"):
        for name, cs in counters(base + "_" + p, sub).items():
            k = summary["kernels"].setdefault(name, {})
            for c, v in cs.items():
                k.setdefault("counters_per_launch", {})[c] = {"mean": sum(v) / len(v), "n": len(v)}
                tot[c] += sum(v) / steps          # all launches of all matching kernels, per step
    step = {"ns": sum(k.get("avg_ns", 0.0) * k.get("calls", 0) for k in summary["kernels"].values()) / steps,
            "launches": sum(k.get("calls", 0) for k in summary["kernels"].values()) / steps}
    if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
        step["fetch_bytes_x2"] = 2.0 * tot["FETCH_SIZE"] * 1024.0
        step["write_bytes"] = tot["WRITE_SIZE"] * 1024.0
        step["traffic_bytes"] = step["fetch_bytes_x2"] + step["write_bytes"]
    if "TCC_HIT_sum" in tot:
        step["l2_hit_rate"] = tot["TCC_HIT_sum"] / max(1.0, tot["TCC_HIT_sum"] + tot["TCC_MISS_sum"])
    if "SQ_VALU_MFMA_BUSY_CYCLES" in tot:
        step["mfma_mops_f32"] = tot.get("SQ_INSTS_VALU_MFMA_MOPS_F32")
        # busy cycles are summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE over the 8 XCDs
        step["mfma_busy_cycles_per_simd"] = tot["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0
        if tot.get("GRBM_GUI_ACTIVE"):
            step["gui_active_cycles"] = tot["GRBM_GUI_ACTIVE"] / 8.0
            step["mfma_busy_frac"] = step["mfma_busy_cycles_per_simd"] / step["gui_active_cycles"]
    summary["per_step"] = step
    json.dump(summary, open(os.path.join(out, f"{tag}_profile.json"), "w"), indent=1)
    # --traffic-key K [--feat N]: also record the step's counters under key K of profiles/traffic_latest.json
    # (bench.py quotes them as `traffic`, with this file and commit as `traffic_source`)
    if "--traffic-key" in sys.argv and "traffic_bytes" in step:
        import subprocess
        key = sys.argv[sys.argv.index("--traffic-key") + 1]
        tp = os.path.join(out, "traffic_latest.json")
        tl = json.load(open(tp)) if os.path.exists(tp) else {}
        # The commit is only recorded when it describes the sources that were profiled: a dirty tree (round 4 stamped the parent commit of a re-tuned
        # rule: 11 strips by its plan.hpp, 13 in the profile) gets "dirty-tree" and is identified by kernel_sources_sha256 alone.
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "unknown"
        dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "hpc_amd/csrc", "include"], capture_output=True, text=True).stdout.strip()
        ent = {"source": f"profiles/{tag}_profile.json",
               "commit": head if not dirty else f"dirty-tree (parent {head}): see kernel_sources_sha256",
               "hbm_bytes_per_launch": step["traffic_bytes"], "fetch_bytes_x2": step["fetch_bytes_x2"], "write_bytes": step["write_bytes"]}
        sys.path.insert(0, ROOT)
        from hpc_amd._lib import kernel_sources_sha256
        ent["kernel_sources_sha256"] = kernel_sources_sha256()      # bench.py: "traffic_stale" when the tree's differs
        if "--feat" in sys.argv:
            ent["N"] = int(sys.argv[sys.argv.index("--feat") + 1])
        if "--rows" in sys.argv:                                    # rows of the workload when it is not 2^20 (bench.py matches on it)
            ent["M"] = int(sys.argv[sys.argv.index("--rows") + 1])
        for k in ("l2_hit_rate", "mfma_busy_frac"):
            if k in step:
                ent[k] = step[k]
        tl[key] = ent
        json.dump(tl, open(tp, "w"), indent=1)
    print(json.dumps(summary["per_step"], indent=1))
    for n, k in summary["kernels"].items():
        print(n[:100], k.get("calls"), k.get("avg_ns"))


if __name__ == "__main__":
    main()
