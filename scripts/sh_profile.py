#!/usr/bin/env python3
"""Sub-phase split of the shortening step (diagnostic build: SWD_DEV_OUT=libswd_hip_dev.so scripts/devbuild.sh -DSWD_SHPROF,
SWD_LIB=libswd_hip_dev.so)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slidingwindowdecoder_amd import _lib
if os.environ.get("SWD_LIB"): _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["SWD_LIB"])
import bench
from slidingwindowdecoder_amd import SlidingWindowDecoder
from slidingwindowdecoder_amd.windows import sample_dem
shots = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
plan = bench.build_problem()
dec = SlidingWindowDecoder(plan, **dict(bench.DECODER_KW, osd_order=0))
det, obs, _ = sample_dem(plan.chk, plan.obs, plan.priors, shots, seed=1)
d = torch.from_numpy(det).cuda()
dec.decode_device(d); torch.cuda.synchronize()
dec.set_profiling(True)
total, stats, pm = dec.decode_device(d); torch.cuda.synchronize()
prof = dec.get_profile(shots).astype(np.float64) / 100.0
st = stats.cpu().numpy()
sel = st[..., 3] > 0
for name, k in (("livemask + contradiction test", 1), ("hard reset + peel", 2), ("compaction + slot lists", 3), ("cache loads + bp_init", 4)):
    print(f"{name:32s} {prof[..., k][sel].mean():8.2f} us")
