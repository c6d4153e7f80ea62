#!/usr/bin/env python3
"""Condense rocprofv3 output under gpurun_out/ into the tracked summaries in profiles/.

    gpurun_out/prof_stats/*kernel_stats.csv         (rocprofv3 --kernel-trace --stats)
    gpurun_out/prof_fetch|prof_write/*counter_collection.csv   (separate --pmc passes)

HBM bytes per launch follow /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x, so the read
side is doubled (upper bound for our mostly narrow accesses)."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

out = {}
for f in glob.glob(os.path.join(src, "prof_stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats.csv"))
    for row in csv.DictReader(open(f)):
        if "pipeline_kernel" in row["Name"]:
            out["kernel"] = row["Name"]
            out["calls"] = int(row["Calls"])
            out["avg_ms"] = float(row["AverageNs"]) / 1e6
            out["min_ms"] = float(row["MinNs"]) / 1e6
            out["max_ms"] = float(row["MaxNs"]) / 1e6

for name, d in (("FETCH_SIZE", "prof_fetch"), ("WRITE_SIZE", "prof_write")):
    vals, meta = [], {}
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "pipeline_kernel" in row["Kernel_Name"] and row["Counter_Name"] == name:
                vals.append(float(row["Counter_Value"]))
                meta = {k: row[k] for k in ("Grid_Size", "Workgroup_Size", "VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size")}
    if vals:
        out[name + "_KiB_per_launch"] = sum(vals) / len(vals)
        out["dispatch"] = meta
if "FETCH_SIZE_KiB_per_launch" in out and "WRITE_SIZE_KiB_per_launch" in out:
    out["hbm_bytes_per_launch"] = (2.0 * out["FETCH_SIZE_KiB_per_launch"] + out["WRITE_SIZE_KiB_per_launch"]) * 1024.0
    out["hbm_bytes_note"] = "(2 x FETCH_SIZE + WRITE_SIZE) x 1024, separate --pmc passes, gfx950 read-side correction"
json.dump(out, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
if "hbm_bytes_per_launch" in out:
    json.dump({"hbm_bytes_per_launch": out["hbm_bytes_per_launch"], "source": f"profiles/{tag}_summary.json"},
              open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
bj = os.path.join(src, "bench.json")
if os.path.exists(bj):
    shutil.copy(bj, os.path.join(dst, f"{tag}_bench.json"))
print(json.dumps(out, indent=1))
