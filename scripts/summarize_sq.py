#!/usr/bin/env python3
"""SQ counter summary of the pipeline kernel from two rocprofv3 --pmc passes under gpurun_out/pmc_sq1|2
 -> profiles/<tag>_sq_counters.json"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
vals = {}
for d in ("pmc_sq1", "pmc_sq2"):
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "pipeline_kernel" in row["Kernel_Name"]:
                vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in vals.items()}
out = {"per_launch_mean": c, "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles summed over all waves; SQ_INSTS_* in wave instructions"}
if "SQ_WAVE_CYCLES" in c:
    w = c["SQ_WAVE_CYCLES"]
    out["fractions_of_wave_cycles"] = {k: c[k] / w for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                                                                 "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS") if k in c}
if "SQ_LDS_IDX_ACTIVE" in c and "SQ_LDS_BANK_CONFLICT" in c:
    out["lds_bank_conflict_share_of_lds_active"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_sq_counters.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
