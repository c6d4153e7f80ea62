#!/usr/bin/env python3
"""Per-phase device time of the pipeline kernel (diagnostics): python scripts/phase_profile.py [shots]"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from slidingwindowdecoder_amd import _lib
if os.environ.get("SWD_LIB"): _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["SWD_LIB"])
from slidingwindowdecoder_amd import SlidingWindowDecoder
from slidingwindowdecoder_amd.windows import sample_dem
shots = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
order = int(sys.argv[2]) if len(sys.argv) > 2 else 0
plan = bench.build_problem()
dec = SlidingWindowDecoder(plan, **dict(bench.DECODER_KW, osd_order=order))
det, obs, _ = sample_dem(plan.chk, plan.obs, plan.priors, shots, seed=1)
d = torch.from_numpy(det).cuda()
dec.decode_device(d); torch.cuda.synchronize()
dec.set_profiling(True)
t = time.time(); total, stats, pm = dec.decode_device(d); torch.cuda.synchronize(); dt = time.time() - t
prof = dec.get_profile(shots).astype(np.float64) / 100.0  # us
st = stats.cpu().numpy()
cls = st[..., 0] & 0xFF
names = ["init", "preBP", "sort1", "shorten", "postBP", "osd_sort", "osd_elim", "osd_sweep+epi"]
print(f"launch {dt*1e3:.1f} ms, {shots*dec.W/dt:.0f} windows/s; lds {dec.lds_bytes} B, threads {dec.threads}")
print("sum of phase time per shot (us): mean %.0f" % prof.sum(axis=(1, 2)).mean())
tot = prof.sum()
for i, nme in enumerate(names):
    ph = prof[..., i]
    nz = ph > 0
    print(f"{nme:14s} share {100*ph.sum()/tot:5.1f}%  mean over windows where run {ph[nz].mean() if nz.any() else 0:8.1f} us  (run in {100*nz.mean():4.1f}% of windows)")
pre = st[..., 2]; post = st[..., 3]
print("pre iters/window mean %.2f ; post iters mean %.2f ; us per pre-iter %.2f ; us per post-iter %.2f" % (
    pre.mean(), post.mean(), prof[..., 1].sum() / max(pre.sum(), 1), prof[..., 4].sum() / max(post.sum(), 1)))
print("classes", np.bincount(cls.ravel(), minlength=6).tolist())
