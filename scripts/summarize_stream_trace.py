#!/usr/bin/env python3
"""Condense the kernel trace of scripts/profile_stream.sh <tag> <workload> (gpurun_out/<tag>_<workload>_stream/) into
profiles/<tag>_<workload>_stream_summary.json: what the launches of the STREAMED timed region look like on the device.

A launch counts as streamed when its [start, end] interval on the device overlaps another launch of the same kernel (the timed region
and its warm-up; the single launches that bench.py times afterwards for `roofline` overlap nothing).  Reported:
  avg_kernel_ms_under_overlap   mean duration of a streamed launch (longer than a single launch: it shares the device)
  makespan_ms_per_step          (last end - first start of the longest run of overlapping launches) / launches in it
  overlap_fraction              1 - union of the intervals / sum of the durations
  avg_concurrent_launches       sum of the durations / union of the intervals
  single_launch_avg_ms          mean duration of the launches that overlap nothing (the roofline's kernel time)"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
wl = sys.argv[2] if len(sys.argv) > 2 else "headline"
KERNEL = {"bp4": "bp4_kernel"}.get(wl, "pipeline_kernel")
src = os.path.join(ROOT, "gpurun_out", f"{tag}_{wl}_stream")
files = sorted(glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
if not files:
    raise SystemExit(f"no kernel trace under {src}")
rows = []
for row in csv.DictReader(open(files[-1])):
    if KERNEL in row["Kernel_Name"]:
        rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"]))
rows.sort()
# the workload's own kernel = the most frequent instantiation
names = {}
for _, _, n in rows:
    names[n] = names.get(n, 0) + 1
kname = max(names, key=names.get)
iv = [(a, b) for a, b, n in rows if n == kname]
# runs of overlapping launches
runs, cur = [], [iv[0]]
for a, b in iv[1:]:
    if a < max(e for _, e in cur):
        cur.append((a, b))
    else:
        runs.append(cur)
        cur = [(a, b)]
runs.append(cur)
streamed = max(runs, key=len)
singles = [r[0] for r in runs if len(r) == 1]
dur = [b - a for a, b in streamed]
union, end = 0, streamed[0][0]
for a, b in streamed:
    if b > end:
        union += b - max(a, end)
        end = b
span = max(b for _, b in streamed) - streamed[0][0]


def git_rev():
    try:
        rev = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
        dirty = subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "slidingwindowdecoder_amd", "bench.py"], text=True).strip()
        return rev + ("+uncommitted" if dirty else "")
    except Exception:
        return None


out = {"git": git_rev(), "workload": wl, "kernel": kname, "launches_in_trace": len(iv), "launches": len(streamed),
       "avg_kernel_ms_under_overlap": sum(dur) / len(dur) / 1e6, "min_ms": min(dur) / 1e6, "max_ms": max(dur) / 1e6,
       "makespan_ms_per_step": span / len(streamed) / 1e6, "overlap_fraction": 1.0 - union / sum(dur),
       "avg_concurrent_launches": sum(dur) / union,
       "single_launches": len(singles), "single_launch_avg_ms": (sum(b - a for a, b in singles) / len(singles) / 1e6) if singles else None,
       "source": os.path.relpath(files[-1], ROOT)}
bench_log = os.path.join(src, "bench.log")
if os.path.exists(bench_log):
    for line in open(bench_log):
        if line.startswith("{"):
            j = json.loads(line)
            out["bench_line_under_the_profiler"] = {k: j.get(k) for k in ("value", "ms_per_step", "steps")}
            out["bench_line_under_the_profiler"]["avg_kernel_ms"] = j["roofline"]["avg_kernel_ms"]
dst = os.path.join(ROOT, "profiles", f"{tag}_{wl}_stream_summary.json")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
