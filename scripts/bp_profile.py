#!/usr/bin/env python3
"""Cycle split of the post-phase BP iteration (diagnostic build libswd_hip_bpprof.so, -DSWD_BPPROF)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slidingwindowdecoder_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ.get("SWD_LIB", "libswd_hip_bpprof.so"))
import bench
from slidingwindowdecoder_amd import SlidingWindowDecoder
from slidingwindowdecoder_amd.windows import sample_dem
shots = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
plan = bench.build_problem(N=288, W=4, F=1) if os.environ.get("SWD_CONFIG") == "288" else bench.build_problem()
dec = SlidingWindowDecoder(plan, **dict(bench.DECODER_KW, osd_order=0))
det, obs, _ = sample_dem(plan.chk, plan.obs, plan.priors, shots, seed=1)
d = torch.from_numpy(det).cuda()
dec.decode_device(d); torch.cuda.synchronize()
dec.set_profiling(True)
total, stats, pm = dec.decode_device(d); torch.cuda.synchronize()
prof = dec.get_profile(shots).astype(np.float64)
st = stats.cpu().numpy()
post = st[..., 3].astype(np.float64)
sel = post > 0
for name, k in (("CN pass", 0), ("block_any(+barrier)", 5), ("VN pass", 6), ("end barrier", 7)):
    print(f"{name:22s} cycles per post iteration: {prof[..., k][sel].sum() / post[sel].sum():8.0f}")
osd = (st[..., 0] & 0xFF) == 2
if osd.any():
    print("OSD shots: steps mean %.0f max %.0f ; cycles: all step evaluations %.0f ; first fence %.0f ; T updates %.0f" % (
        prof[..., 1][osd].mean(), prof[..., 1][osd].max(), 16 * prof[..., 2][osd].mean(), 16 * prof[..., 4][osd].mean(), 16 * prof[..., 3][osd].mean()))
print("post ticks(100MHz)/iter: %.1f us" % (prof[..., 4][sel].sum() / post[sel].sum() / 100.0))
w = st[..., 7][sel].astype(np.uint32)
wm = np.stack([(w >> (8 * i)) & 0xFF for i in range(4)], -1)
print("post-phase positions walked per wave (mean):", wm.mean(0).round(1).tolist(), " max of waves mean %.1f" % wm.max(1).mean(),
      " live edges per live check mean %.1f" % (st[..., 6][sel] / np.maximum(st[..., 5][sel], 1)).mean(),
      " live checks %.0f live vns %.0f" % (st[..., 5][sel].mean(), st[..., 4][sel].mean()))
