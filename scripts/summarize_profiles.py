#!/usr/bin/env python3
"""Condense the rocprofv3 output of scripts/profile_all.sh <tag> <workload> (gpurun_out/<tag>_<workload>/) into the tracked
summaries profiles/<tag>_<workload>_{kernel_stats.csv, summary.json, sq_counters.json}.

    stats/   rocprofv3 --kernel-trace --stats            -> average / min / max duration of the workload's kernel
    fetch/, write/   separate --pmc passes               -> HBM bytes per launch
    sq1/, sq2/       two SQ --pmc passes                 -> instruction mix, LDS activity, bank conflicts, wave cycles

HBM bytes per launch follow /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE are in KiB; on
gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x, so the read side is doubled (upper bound for our mostly narrow
accesses).  The summary records the git revision it was made at: bench.py flags a live kernel time that has moved away
from the profiled one (`profile_stale`)."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
wl = sys.argv[2] if len(sys.argv) > 2 else "headline"
KERNEL = {"bp4": "bp4_kernel"}.get(wl, "pipeline_kernel")
src = os.path.join(ROOT, "gpurun_out", f"{tag}_{wl}")
dst = os.path.join(ROOT, "profiles")
stem = os.path.join(dst, f"{tag}_{wl}")
os.makedirs(dst, exist_ok=True)


def newest(pattern):
    """files of the most recent run only: gpurun merges every call's output into the same local directory, and rocprofv3 names
    its files by process id, so a re-profiled workload leaves the older run's files next to the new ones"""
    files = glob.glob(pattern, recursive=True)
    if not files:
        return []
    t = max(os.path.getmtime(f) for f in files)
    return [f for f in files if t - os.path.getmtime(f) < 600]


def git_rev():
    try:
        rev = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
        dirty = subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "slidingwindowdecoder_amd", "bench.py"], text=True).strip()
        return rev + ("+uncommitted" if dirty else "")
    except Exception:
        return None


out = {"git": git_rev(), "workload": wl}
for f in newest(os.path.join(src, "stats", "**", "*kernel_stats.csv")):
    shutil.copy(f, stem + "_kernel_stats.csv")
    for row in csv.DictReader(open(f)):
        if KERNEL in row["Name"]:
            out["kernel"] = row["Name"]
            out["calls"] = int(row["Calls"])
            out["avg_ms"] = float(row["AverageNs"]) / 1e6
            out["min_ms"] = float(row["MinNs"]) / 1e6
            out["max_ms"] = float(row["MaxNs"]) / 1e6
        elif wl == "bp4" and "bp4_osd_kernel" in row["Name"]:  # the second launch of a bp4 batch: OSD on the queue of unconverged decodes
            out["companion_kernel"] = row["Name"]
            out["companion_avg_ms"] = float(row["AverageNs"]) / 1e6
        elif wl == "bp4" and ("bp4_weight_kernel" in row["Name"] or "shot_order_kernel" in row["Name"]):  # the start order of the batch (round 6)
            out.setdefault("helper_kernels", {})[row["Name"].split("(")[0].replace("void ", "")] = float(row["AverageNs"]) / 1e6
    if "avg_ms" in out and "companion_avg_ms" in out:
        # what HIP events around one batch see (+ the gaps between the launches)
        out["avg_ms_with_companion"] = out["avg_ms"] + out["companion_avg_ms"] + sum(out.get("helper_kernels", {}).values())

for name, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    vals, meta = [], {}
    for f in newest(os.path.join(src, d, "**", "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if KERNEL in row["Kernel_Name"] and row["Counter_Name"] == name:
                vals.append(float(row["Counter_Value"]))
                meta = {k: row[k] for k in ("Grid_Size", "Workgroup_Size", "VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size") if k in row}
    if vals:
        out[name + "_KiB_per_launch"] = sum(vals) / len(vals)
        out["dispatch"] = meta
if "FETCH_SIZE_KiB_per_launch" in out and "WRITE_SIZE_KiB_per_launch" in out:
    out["hbm_bytes_per_launch"] = (2.0 * out["FETCH_SIZE_KiB_per_launch"] + out["WRITE_SIZE_KiB_per_launch"]) * 1024.0
    out["hbm_bytes_note"] = "(2 x FETCH_SIZE + WRITE_SIZE) x 1024, separate --pmc passes, gfx950 read-side correction"
for log in glob.glob(os.path.join(src, "stats.log")):
    for ln in open(log):
        if ln.startswith("{"):
            out["bench_line_under_the_profiler"] = json.loads(ln)
json.dump(out, open(stem + "_summary.json", "w"), indent=1)

vals = {}
for d in ("sq1", "sq2", "sq3", "sq4"):
    for f in newest(os.path.join(src, d, "**", "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if KERNEL in row["Kernel_Name"]:
                vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in vals.items()}
sq = {"git": out["git"], "workload": wl, "per_launch_mean": c,
      "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles summed over all waves; SQ_INSTS_* in wave instructions"}
if "SQ_WAVE_CYCLES" in c:
    w = c["SQ_WAVE_CYCLES"]
    sq["fractions_of_wave_cycles"] = {k: c[k] / w for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                                                            "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS") if k in c}
if "SQ_LDS_IDX_ACTIVE" in c and "SQ_LDS_BANK_CONFLICT" in c:
    sq["lds_bank_conflict_share_of_lds_active"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
if c:
    json.dump(sq, open(stem + "_sq_counters.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "bench_line_under_the_profiler"}, indent=1))
print(json.dumps(sq.get("fractions_of_wave_cycles", {}), indent=1))
