#!/usr/bin/env python3
"""scripts/lds_conflict_attribution.sh <tag> -> profiles/<tag>_lds_conflicts.json: per-launch LDS counters of the headline kernel
with the full iteration caps, with the post phase capped at 4 iterations and with both phases capped at 4; the differences
belong to the removed iterations (per live-edge-iteration figures use the kernel's own statistics from the bench lines)."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", f"{tag}_conflicts")
runs = {}
for name in ("full", "post4", "pre4post4"):
    vals = {}
    for f in glob.glob(os.path.join(src, name, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "pipeline_kernel" in row["Kernel_Name"]:
                vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    line = None
    for ln in open(os.path.join(src, name + ".log")):
        if ln.startswith("{"):
            line = json.loads(ln)
    runs[name] = {"counters": {k: sum(v) / len(v) for k, v in vals.items()},
                  "bp_iterations_pre_post": line["config"].get("bp_iterations_pre_post_rank0") if line else None,
                  "live_edge_iterations_pre_post": line["config"].get("live_edge_iterations_pre_post_rank0") if line else None,
                  "exit_classes_pre_post_osd": line["config"].get("exit_classes_pre_post_osd_rank0") if line else None,
                  "kernel": next((row for row in []), None)}
out = {"runs": runs}


def diff(a, b, what):
    ca, cb = runs[a]["counters"], runs[b]["counters"]
    d = {k: ca[k] - cb[k] for k in ca if k in cb}
    ea, eb = runs[a]["live_edge_iterations_pre_post"], runs[b]["live_edge_iterations_pre_post"]
    idx = 1 if what == "post" else 0
    edges = (ea[idx] - eb[idx]) if ea and eb else None
    r = {"counters": d, "live_edge_iterations_removed": edges}
    if d.get("SQ_LDS_IDX_ACTIVE"):
        r["bank_conflict_share"] = d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"]
    if edges:
        r["lds_cycles_per_live_edge_iteration"] = d["SQ_LDS_IDX_ACTIVE"] / edges
        r["conflict_cycles_per_live_edge_iteration"] = d["SQ_LDS_BANK_CONFLICT"] / edges
        r["lds_instructions_x64_per_live_edge_iteration"] = d["SQ_INSTS_LDS"] * 64.0 / edges
    return r


if all(runs[k]["counters"] for k in ("full", "post4")):
    out["post_phase_iterations_5_to_200"] = diff("full", "post4", "post")
if all(runs[k]["counters"] for k in ("post4", "pre4post4")):
    out["pre_phase_iterations_5_to_8_(plus_what_changes_downstream)"] = diff("post4", "pre4post4", "pre")
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_lds_conflicts.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "runs"}, indent=1))
