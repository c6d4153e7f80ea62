#!/usr/bin/env python3
"""Decodes/s of the general form (csrc/swd_huge.hip) on the un-windowed [[288,12,18]] detector error model of an 18-round memory
experiment (2736 x ~26 k, IBM.ipynb's global decode with N = 288): python scripts/huge_rate.py [shots] [p]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from slidingwindowdecoder_amd import osd_window
from slidingwindowdecoder_amd.windows import sample_dem
shots = int(sys.argv[1]) if len(sys.argv) > 1 else 512
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.003
plan = bench.build_problem(N=288, p=p, rounds=18, W=19, F=1)
w = plan.windows[0]
det, _, _ = sample_dem(plan.chk, plan.obs, plan.priors, shots, seed=3)
t0 = time.perf_counter()
dec = osd_window(w.mat, channel_probs=w.prior, pre_max_iter=16, post_max_iter=1000, ms_scaling_factor=1.0, osd_method="osd_cs", osd_order=10)
t_create = time.perf_counter() - t0
synd = np.ascontiguousarray(det[:, w.row0:w.row1])
dec.decode_batch(synd[:8])
t0 = time.perf_counter()
dec.decode_batch(synd)
el = time.perf_counter() - t0
cls = np.bincount(dec.last_status & 0xFF, minlength=6)
print(json.dumps({"workload": f"osd_window(pre 16, post 1000, osd_cs 10) on the un-windowed [[288,12,18]] DEM, p = {p}, 18 rounds", "shape": list(w.mat.shape),
                  "edges": int(w.mat.nnz), "new_n": dec.new_n, "rank": dec.rank, "shots": shots, "seconds": round(el, 3), "decodes_per_s": round(shots / el, 1),
                  "exit_classes_pre_post_osd": cls[:3].tolist(), "mean_iterations": float(dec.last_iterations.mean()), "create_s": round(t_create, 2)}))
