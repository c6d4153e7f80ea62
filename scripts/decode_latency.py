#!/usr/bin/env python3
"""Latency of the reference-style one-syndrome-per-call API: osd_window.decode() on the [[144,12,12]] mid window
(host buffers in and out, one kernel launch per call), next to the oracle on one host core."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import oracle as O
from slidingwindowdecoder_amd import osd_window
from slidingwindowdecoder_amd.windows import sample_dem

plan = bench.build_problem()
w = plan.windows[5]
kw = dict(bench.DECODER_KW, osd_order=0)
det, _, _ = sample_dem(plan.chk, plan.obs, plan.priors, 400, seed=3)
synd = det[:, w.row0:w.row1]
dev = osd_window(w.mat, channel_probs=w.prior, **kw)
ora = O.osd_window(w.mat, channel_probs=w.prior, **kw)
for s in synd[:20]:
    dev.decode(s)
t = []
for s in synd:
    t0 = time.perf_counter(); dev.decode(s); t.append(time.perf_counter() - t0)
t = np.array(t) * 1e6
t0 = time.perf_counter()
for s in synd:
    ora.decode(s)
tc = (time.perf_counter() - t0) / len(synd) * 1e6
print(f"osd_window.decode() on the device: median {np.median(t):.0f} us, mean {t.mean():.0f} us, p99 {np.percentile(t, 99):.0f} us per call; "
      f"oracle on one host core: mean {tc:.0f} us per call")
z = np.zeros(w.mat.shape[0], np.uint8)
for _ in range(20):
    dev.decode(z)
t0 = time.perf_counter()
for _ in range(300):
    dev.decode(z)
print(f"all-zero syndrome (one BP iteration): {(time.perf_counter() - t0) / 300 * 1e6:.0f} us per call = fixed cost of a call")
cls = []
for s in synd[:200]:
    t0 = time.perf_counter(); dev.decode(s); cls.append((dev.exit_class, (time.perf_counter() - t0) * 1e6))
cls = np.array(cls)
for c in (0, 1, 2):
    sel = cls[:, 0] == c
    if sel.any():
        print(f"exit class {c}: {sel.sum()} calls, mean {cls[sel, 1].mean():.0f} us")
