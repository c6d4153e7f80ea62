#!/usr/bin/env python3
"""Guessing-decoder windows, serial tree walk: the sum of the units' own device time over the workgroup slots of the grid against the
launch time -- how much of a launch is waiting (for a predecessor window, for the last shots) rather than decoding.
python scripts/gdg_ideal_makespan.py [shots]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SWD_GDG_SERIAL", "1")
import bench
from slidingwindowdecoder_amd import SlidingWindowDecoder
from slidingwindowdecoder_amd.windows import sample_dem
shots = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
plan = bench.build_problem()
dec = SlidingWindowDecoder(plan, **bench.GDG_KW)
det, _, _ = sample_dem(plan.chk, plan.obs, plan.priors, shots, seed=1)
d = torch.from_numpy(det).cuda()
dec.decode_device(d); torch.cuda.synchronize()
t = time.time(); dec.decode_device(d); torch.cuda.synchronize(); dt0 = time.time() - t
dec.set_profiling(True)
t = time.time(); _, stats, _ = dec.decode_device(d); torch.cuda.synchronize(); dt = time.time() - t
prof = dec.get_profile(shots).astype(np.float64) / 100.0  # us
unit = prof.sum(axis=2)
slots = 512
print(f"{shots} shots x {dec.W} windows: launch {dt0 * 1e3:.1f} ms (profiled {dt * 1e3:.1f}); units' own time: mean {unit.mean():.1f} us, p50 {np.percentile(unit, 50):.0f}, "
      f"p90 {np.percentile(unit, 90):.0f}, p99 {np.percentile(unit, 99):.0f}, max {unit.max():.0f}")
print(f"sum of unit time / {slots} slots = {unit.sum() / slots / 1e3:.2f} ms; longest shot (its 11 windows in sequence) {unit.sum(axis=1).max() / 1e3:.2f} ms")
for i in range(prof.shape[2]):
    print(f"  timer {i}: share {100 * prof[..., i].sum() / prof.sum():5.1f} %")
