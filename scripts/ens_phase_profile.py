#!/usr/bin/env python3
"""Where the threaded ensemble's prefix-tree walk spends its time (diagnostic build: SWD_DEV_OUT=libswd_hip_prof.so scripts/devbuild.sh
-DSWD_GDGPROF; SWD_LIB=libswd_hip_prof.so): python scripts/ens_phase_profile.py [shots] [D] [S]"""
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
_, stats, _ = dec.decode_device(d); torch.cuda.synchronize()
prof = dec.get_profile(shots).astype(np.float64) / 100.0  # us
st = stats.cpu().numpy()
names = ["init + pre-processing BP", "sort + reset", "cache refresh (+ bp_init)", "BP blocks", "scan (select_vn core)", "restore + decimation + peel", "offers, saves, bookkeeping", "whole unit"]
tot = prof[..., 7].sum()
post = (st[..., 0] & 0xFF) == 1
print(f"D={D} S={S}: {shots} shots x {dec.W} windows; ensembles: {100 * post.mean():.1f} % of the windows; mean unit {prof[..., 7].mean():.1f} us, mean ensemble unit {prof[..., 7][post].mean():.1f} us")
for i, n in enumerate(names[:7]):
    print(f"{n:34s} share of all unit time {100 * prof[..., i].sum() / tot:5.1f} %   mean over ensemble windows {prof[..., i][post].mean():8.1f} us")
print("BP blocks counted per ensemble (once per thread that would run them): mean %.1f; iterations: mean %.1f" % (st[..., 5][post].mean(), st[..., 3][post].mean()))
