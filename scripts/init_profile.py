#!/usr/bin/env python3
"""Where the set-up of a window goes (diagnostic build -DSWD_INITPROF): python scripts/init_profile.py [shots]"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from slidingwindowdecoder_amd import _lib
if os.environ.get("SWD_LIB"): _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["SWD_LIB"])
from slidingwindowdecoder_amd import SlidingWindowDecoder
from slidingwindowdecoder_amd.windows import sample_dem
shots = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
plan = bench.build_problem(N=288, W=4, F=1) if os.environ.get('SWD_CONFIG') == '288' else bench.build_problem()
dec = SlidingWindowDecoder(plan, **dict(bench.DECODER_KW, osd_order=0))
det, obs, _ = sample_dem(plan.chk, plan.obs, plan.priors, shots, seed=1)
d = torch.from_numpy(det).cuda()
dec.decode_device(d); torch.cuda.synchronize()
dec.set_profiling(True)
dec.decode_device(d); torch.cuda.synchronize()
prof = dec.get_profile(shots).astype(np.float64) / 100.0  # us
names = ["ticket + wait + state load", "reset loops", "variable-node cache loads", "barrier", "bp_init", "check caches + barrier", "decode (after set-up) + commit", "commit"]
for i, nme in enumerate(names):
    ph = prof[..., i]
    print(f"{nme:34s} mean {ph.mean():8.2f} us   first window {ph[:, 0].mean():8.2f}   later windows {ph[:, 1:].mean():8.2f}")
