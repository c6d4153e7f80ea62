#!/usr/bin/env python3
"""Raw per-window timer columns of a diagnostic build (-DSWD_INITPROF: set-up of a unit; -DSWD_SHPROF: the shortening step):
SWD_LIB=<dev build> python scripts/subphase_profile.py init|shorten [shots] [order]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from slidingwindowdecoder_amd import SlidingWindowDecoder
from slidingwindowdecoder_amd.windows import sample_dem
kind = sys.argv[1]
shots = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
order = int(sys.argv[3]) if len(sys.argv) > 3 else 10
plan = bench.build_problem()
dec = SlidingWindowDecoder(plan, **dict(bench.DECODER_KW, osd_order=order))
det, _, _ = sample_dem(plan.chk, plan.obs, plan.priors, shots, seed=1)
d = torch.from_numpy(det).cuda()
dec.decode_device(d); torch.cuda.synchronize()
dec.set_profiling(True)
_, stats, _ = dec.decode_device(d); torch.cuda.synchronize()
prof = dec.get_profile(shots).astype(np.float64) / 100.0  # us
cls = stats.cpu().numpy()[..., 0] & 0xFF
names = {"init": ["ticket + wait + state load", "reset loops", "cache pack / loads", "barrier", "bp_init", "check caches", "rest of the unit after set-up", "epilogue (after the window's decode)"],
         "shorten": ["(std) init", "livemask rebuild + contradiction test", "backup + peel rounds", "compaction, slot lists, degree histogram", "cn_assign + caches + bp_init", "(std)", "(std)", "(std)"]}[kind]
sel = np.ones_like(cls, bool) if kind == "init" else (cls != 0)
for i, nme in enumerate(names):
    print(f"{nme:48s} mean {prof[..., i][sel].mean():8.2f} us over {int(sel.sum())} windows")
