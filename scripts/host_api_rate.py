#!/usr/bin/env python3
"""PCIe-inclusive rate of the headline workload through the host-buffer API (SlidingWindowDecoder.decode:
pageable numpy arrays in and out), next to the device-resident rate bench.py reports."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from slidingwindowdecoder_amd import SlidingWindowDecoder
from slidingwindowdecoder_amd.windows import sample_dem

plan = bench.build_problem()
dec = SlidingWindowDecoder(plan, **dict(bench.DECODER_KW, osd_order=0))
det, _, _ = sample_dem(plan.chk, plan.obs, plan.priors, 4096, seed=11)
dec.decode(det)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); dec.decode(det); ts.append(time.perf_counter() - t0)
t = float(np.median(ts))
print(f"host-buffer API: {t * 1e3:.1f} ms per 4096-shot batch -> {4096 * dec.W / t / 1e6:.2f} M windows/s (detectors in: {det.nbytes / 1e6:.1f} MB, "
      f"faults + statistics out: {(4096 * plan.chk.shape[1] + 4096 * dec.W * 40) / 1e6:.1f} MB, pageable memory)")
