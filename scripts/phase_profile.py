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
if os.environ.get('SWD_CONFIG') == 'shyps12':  # SHYPS r=3, twelve-round windows (252 x 2240)
    from slidingwindowdecoder_amd import shyps
    from slidingwindowdecoder_amd.windows import plan_windows
    _dem = shyps.shyps_dem(3, 0.001, 14)
    plan = plan_windows(_dem.chk, _dem.obs, _dem.priors, 21, 12, 1, method=1)
elif os.environ.get('SWD_CONFIG') == 'global144':  # the un-windowed 936 x 8784 DEM on the large-graph kernels (IBM.ipynb:119-135)
    plan = bench.build_problem(**bench.WORKLOADS["global144"]["problem"])
elif os.environ.get('SWD_CONFIG') == '288w3':  # [[288,12,18]] (3,1) windows: 432 x 3456
    plan = bench.build_problem(N=288, W=3, F=1)
else:
    plan = bench.build_problem(N=288, W=4, F=1) if os.environ.get('SWD_CONFIG') == '288' else bench.build_problem()
dec = SlidingWindowDecoder(plan, **dict(bench.DECODER_KW, osd_order=order, **(bench.WORKLOADS["global144"]["decoder_kw"] if os.environ.get('SWD_CONFIG') == 'global144' else {})))
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
# per-shot latency distribution and what a greedy dispatch of the shots onto the 512 workgroup slots gives
tot_shot = prof.sum(axis=(1, 2))
print("per-shot device time (us): mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f" % (
    tot_shot.mean(), *np.percentile(tot_shot, [50, 90, 99]), tot_shot.max()))
import heapq
slots = [0.0] * 512
heapq.heapify(slots)
for t in tot_shot:
    heapq.heappush(slots, heapq.heappop(slots) + t)
print("greedy makespan over 512 slots: %.2f ms (sum/512 = %.2f ms)" % (max(slots) / 1e3, tot_shot.sum() / 512 / 1e3))
wt = det.sum(axis=1)
print("correlation(detector weight, shot time) = %.3f" % np.corrcoef(wt, tot_shot)[0, 1])
for name, order_ in (("heaviest syndrome first", np.argsort(-wt, kind="stable")), ("oracle: longest first", np.argsort(-tot_shot))):
    slots = [0.0] * 512
    heapq.heapify(slots)
    for t in tot_shot[order_]:
        heapq.heappush(slots, heapq.heappop(slots) + t)
    print("greedy makespan, %s: %.2f ms" % (name, max(slots) / 1e3))
