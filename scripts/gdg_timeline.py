#!/usr/bin/env python3
"""Timeline of one launch of the guessing-decoder pipeline (diagnostic build -DSWD_TSPROF): when shots are admitted and finished,
how long a shot's chain of windows takes, how long a committed window waits for its successor to start.
python scripts/gdg_timeline.py [shots]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from slidingwindowdecoder_amd import SlidingWindowDecoder
from slidingwindowdecoder_amd.windows import sample_dem
shots = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
plan = bench.build_problem()
dec = SlidingWindowDecoder(plan, **bench.GDG_KW)
det, _, _ = sample_dem(plan.chk, plan.obs, plan.priors, shots, seed=1)
d = torch.from_numpy(det).cuda()
dec.decode_device(d); torch.cuda.synchronize()
dec.set_profiling(True)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record(); _, stats, _ = dec.decode_device(d); ev1.record(); torch.cuda.synchronize()
prof = dec.get_profile(shots).astype(np.float64) / 100.0  # us
start, commit = prof[..., 4], prof[..., 6]
base = start.min()
start -= base; commit -= base
span = commit.max()
print(f"{shots} shots x {dec.W} windows: launch {ev0.elapsed_time(ev1):.2f} ms, first start to last commit {span / 1e3:.2f} ms")
adm, fin = start[:, 0], commit[:, -1]
print("last admission at %.2f ms; shot latency (admission -> last commit) mean %.2f ms p50 %.2f p90 %.2f p99 %.2f max %.2f" % (
    adm.max() / 1e3, (fin - adm).mean() / 1e3, *(np.percentile(fin - adm, [50, 90, 99]) / 1e3), (fin - adm).max() / 1e3))
lat = commit - start
gap = start[:, 1:] - commit[:, :-1]
print("window latency (start -> commit) mean %.0f us p50 %.0f p90 %.0f p99 %.0f max %.0f; wait of the next window (commit -> start) mean %.0f us p90 %.0f max %.0f" % (
    lat.mean(), *np.percentile(lat, [50, 90, 99]), lat.max(), gap.mean(), np.percentile(gap, 90), gap.max()))
edges = np.linspace(0, span, 21)
for a, b in zip(edges[:-1], edges[1:]):
    infl = (np.minimum(fin, b) - np.maximum(adm, a)).clip(min=0).sum() / (b - a)
    print("  %6.2f-%6.2f ms: %7.1f shots in flight, %5d admitted, %5d finished, %6d windows committed" % (
        a / 1e3, b / 1e3, infl, ((adm >= a) & (adm < b)).sum(), ((fin >= a) & (fin < b)).sum(), ((commit >= a) & (commit < b)).sum()))
