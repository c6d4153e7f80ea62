#!/usr/bin/env python3
"""Concurrency over time of the work units of one launch (diagnostic build with -DSWD_TSPROF)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from slidingwindowdecoder_amd import _lib
if os.environ.get("SWD_LIB"): _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["SWD_LIB"])
from slidingwindowdecoder_amd import SlidingWindowDecoder
from slidingwindowdecoder_amd.windows import sample_dem
shots = 4096
plan = bench.build_problem()
dec = SlidingWindowDecoder(plan, **dict(bench.DECODER_KW, osd_order=0))
det, obs, _ = sample_dem(plan.chk, plan.obs, plan.priors, shots, seed=1)
d = torch.from_numpy(det).cuda()
dec.decode_device(d); torch.cuda.synchronize()
dec.set_profiling(True)
dec.decode_device(d); torch.cuda.synchronize()
prof = dec.get_profile(shots)
t0, t1 = prof[..., 5].ravel().astype(np.float64) / 100.0, prof[..., 6].ravel().astype(np.float64) / 100.0  # us
base = t0.min()
t0 -= base; t1 -= base
print("launch span %.2f ms, units %d, mean unit %.1f us, sum of unit time / span = %.1f units in flight on average" % (
    t1.max() / 1e3, t0.size, (t1 - t0).mean(), (t1 - t0).sum() / t1.max()))
edges = np.linspace(0, t1.max(), 21)
for a, b in zip(edges[:-1], edges[1:]):
    infl = (np.minimum(t1, b) - np.maximum(t0, a)).clip(min=0).sum() / (b - a)
    started = ((t0 >= a) & (t0 < b)).sum()
    print("  %6.2f-%6.2f ms: %6.1f units in flight, %5d started" % (a / 1e3, b / 1e3, infl, started))
