#!/usr/bin/env python3
"""Where a guessing decoder's window spends its time (diagnostic build: scripts/devbuild.sh -DSWD_GDGPROF, SWD_LIB=libswd_hip_dev.so,
serial tree walk): python scripts/gdg_phase_profile.py [shots]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SWD_GDG_SERIAL", "1")
import bench
from slidingwindowdecoder_amd import SlidingWindowDecoder
from slidingwindowdecoder_amd.windows import sample_dem
shots = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
plan = bench.build_problem()
dec = SlidingWindowDecoder(plan, **bench.GDG_KW)
det, _, _ = sample_dem(plan.chk, plan.obs, plan.priors, shots, seed=1)
d = torch.from_numpy(det).cuda()
dec.decode_device(d); torch.cuda.synchronize()
dec.set_profiling(True)
_, stats, _ = dec.decode_device(d); torch.cuda.synchronize()
prof = dec.get_profile(shots).astype(np.float64) / 100.0  # us
st = stats.cpu().numpy()
names = ["init + pre-processing BP", "sort + reset", "cache rebuilds", "BP blocks", "select_vn / decimation", "branch starts (snapshot load, set, peel)", "path metric + copies", "whole unit"]
tot = prof[..., 7].sum()
post = (st[..., 0] & 0xFF) == 1
print(f"{shots} shots x {dec.W} windows; windows that went into the tree walk: {100 * post.mean():.1f} %; mean unit {prof[..., 7].mean():.1f} us, mean tree-walk unit {prof[..., 7][post].mean():.1f} us")
for i, n in enumerate(names[:7]):
    print(f"{n:44s} share of all unit time {100 * prof[..., i].sum() / tot:5.1f} %   mean over tree-walk windows {prof[..., i][post].mean():8.1f} us")
print("BP blocks per tree-walk window: mean %.1f; snapshots pushed: mean %.1f; iterations inside blocks: mean %.1f" % (st[..., 5][post].mean(), st[..., 4][post].mean(), st[..., 3][post].mean()))
print("us per BP block %.2f ; us per cache rebuild %.2f ; us per select %.2f" % (prof[..., 3].sum() / st[..., 5][post].sum(), prof[..., 2].sum() / st[..., 5][post].sum(), prof[..., 4].sum() / max(st[..., 5][post].sum(), 1)))
