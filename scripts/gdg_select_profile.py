#!/usr/bin/env python3
"""Where select_vn and the cache rebuild of the guessing decoders spend their time (diagnostic build -DSWD_SELPROF, serial tree walk):
python scripts/gdg_select_profile.py [shots]"""
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
post = (st[..., 0] & 0xFF) == 1
blocks = st[..., 5][post].sum()
names = ["select: classification", "select: two arg-minima", "select: decimations + peeling (wave 0)", "select: snapshot save", "select: favoured value + peeling",
         "rebuild: compaction of the live nodes", "rebuild: variable-node caches", "rebuild: check caches"]
for i, n in enumerate(names):
    print(f"{n:44s} {prof[..., i][post].sum() / blocks:6.2f} us per BP block   ({prof[..., i][post].mean():7.1f} us per tree-walk window)")
