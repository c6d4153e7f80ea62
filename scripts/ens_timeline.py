#!/usr/bin/env python3
"""Timeline of one launch of the threaded-ensemble pipeline (diagnostic build -DSWD_TSPROF, SWD_LIB=...): how busy the persistent
grid is, how long units wait for their predecessor window.  python scripts/ens_timeline.py [shots] [D] [S]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from slidingwindowdecoder_amd import SlidingWindowDecoder
from slidingwindowdecoder_amd.windows import sample_dem
shots = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D = int(sys.argv[2]) if len(sys.argv) > 2 else 5
S = int(sys.argv[3]) if len(sys.argv) > 3 else 6
plan = bench.build_problem()
dec = SlidingWindowDecoder(plan, **dict(bench.GDG_KW, multi_thread=True, max_tree_depth=D, max_side_depth=S))
det, _, _ = sample_dem(plan.chk, plan.obs, plan.priors, shots, seed=1)
d = torch.from_numpy(det).cuda()
dec.decode_device(d); torch.cuda.synchronize()
dec.set_profiling(True)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record(); _, stats, _ = dec.decode_device(d); ev1.record(); torch.cuda.synchronize()
prof = dec.get_profile(shots).astype(np.float64) / 100.0  # us
t_iter, t_start, t_commit = prof[..., 5], prof[..., 4], prof[..., 6]
base = t_iter.min()
t_iter -= base; t_start -= base; t_commit -= base
span = t_commit.max()
work, wait = t_commit - t_start, t_start - t_iter
print(f"D={D} S={S}: {shots} shots x {dec.W} windows: launch {ev0.elapsed_time(ev1):.2f} ms, span {span / 1e3:.2f} ms")
print("sum of unit work %.1f ms per workgroup slot (512), sum of ticket + predecessor waits %.1f ms per slot; unit work mean %.0f us p50 %.0f p90 %.0f p99 %.0f max %.0f; wait mean %.0f us p90 %.0f p99 %.0f max %.0f" % (
    work.sum() / 512e3, wait.sum() / 512e3, work.mean(), *np.percentile(work, [50, 90, 99]), work.max(), wait.mean(), *np.percentile(wait, [90, 99]), wait.max()))
for wi in range(dec.W):
    print("  window %2d: first start %7.2f ms, last commit %7.2f ms, wait mean %6.0f us, work mean %6.0f us" % (wi, t_start[:, wi].min() / 1e3, t_commit[:, wi].max() / 1e3, wait[:, wi].mean(), work[:, wi].mean()))
edges = np.linspace(0, span, 21)
for a, b in zip(edges[:-1], edges[1:]):
    busy = (np.minimum(t_commit, b) - np.maximum(t_start, a)).clip(min=0).sum() / (b - a)
    waiting = (np.minimum(t_start, b) - np.maximum(t_iter, a)).clip(min=0).sum() / (b - a)
    print("  %6.2f-%6.2f ms: %6.1f units working, %6.1f waiting" % (a / 1e3, b / 1e3, busy, waiting))
