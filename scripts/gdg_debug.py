#!/usr/bin/env python3
"""Diagnostic: the [[144]] GDG pipeline in the parallel form against the serial form on the same shots
(SWD_LIB=libswd_hip_dev.so built with -DSWD_GDG_DEBUG counts commits per unit in statistics word 7)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from slidingwindowdecoder_amd import SlidingWindowDecoder
from slidingwindowdecoder_amd.windows import sample_dem
shots = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kw = dict(decoder="bpgdg_decoder", max_iter=8, max_iter_per_step=6, max_step=25, max_tree_depth=3, max_side_depth=10,
          max_tree_branch_step=10, max_side_branch_step=10)
plan = bench.build_problem()
det, obs, _ = sample_dem(plan.chk, plan.obs, plan.priors, shots, seed=7)
d = torch.from_numpy(np.ascontiguousarray(det)).cuda()
os.environ["SWD_GDG_SERIAL"] = "1"
ser = SlidingWindowDecoder(plan, **kw)
del os.environ["SWD_GDG_SERIAL"]
par = SlidingWindowDecoder(plan, **kw)
def run(dec):
    stats = torch.zeros((shots, dec.W, 8), dtype=torch.int32, device="cuda")
    total = torch.zeros((shots, plan.chk.shape[1]), dtype=torch.uint8, device="cuda")
    shot = torch.zeros((shots, 2), dtype=torch.int32, device="cuda")
    dec.decode_device(d, total=total, stats=stats, shot_result=shot)
    torch.cuda.synchronize()
    import ctypes as C
    from slidingwindowdecoder_amd import _lib
    L = _lib.lib()
    if hasattr(L, "swd_pipeline_debug_counters"):
        buf = (C.c_uint32 * 16)()
        L.swd_pipeline_debug_counters.argtypes = [C.c_void_p, C.c_void_p]
        L.swd_pipeline_debug_counters(dec._h, buf)
        c = list(buf)
        if c[1]:
            print(f"  tasks {c[1]} (pruned at start {c[5]}), steps {c[4]}, task ms total {c[2] / 1e5:.1f} ({c[2] / 1e2 / max(c[1], 1):.0f} us each), "
                  f"scheduler ms total {c[3] / 1e5:.1f} ({c[3] / 1e2 / max(c[1], 1):.0f} us each); loop: empty polls {c[6]}, items {c[7]}, "
                  f"get-work ms {c[8] / 1e5:.1f}; until state loaded {c[13] / 1e2 / max(c[14], 1):.0f} us per unit; units parked {c[9]} ({c[11] / 1e2 / max(c[9], 1):.0f} us each), inline {c[10]} ({c[12] / 1e2 / max(c[10], 1):.0f} us each); "
                  f"waits for an item longer than 50 us: {c[6]}, {c[15] / 1e5:.1f} ms in total")
    try:
        dec.check_status()
    except RuntimeError as e:
        print("  status:", e)
    return total.cpu().numpy(), stats.cpu().numpy(), shot.cpu().numpy()
t0, s0, r0 = run(ser)
print("serial: commits per unit", np.bincount(s0[..., 7].ravel())[:4].tolist())
for rep in range(reps):
    t1, s1, r1 = run(par)
    dt = np.flatnonzero((t0 != t1).any(axis=1))
    ds = np.argwhere((s0[..., :7] != s1[..., :7]).any(axis=2))
    firsts = {}
    for (bb, ww) in ds:
        firsts.setdefault(int(bb), int(ww))
    for bb, ww in list(firsts.items())[:8]:
        same_in = s0[bb, ww, 5] == s1[bb, ww, 5]
        prev_same_out = ww == 0 or s0[bb, ww - 1, 6] == s1[bb, ww - 1, 6]
        print(f"   shot {bb}: first differing window {ww}: input checksum equal {same_in}, previous window's output equal {prev_same_out}; "
              f"serial {s0[bb, ww, :5].tolist()} parallel {s1[bb, ww, :5].tolist()}")
    print(f"rep {rep}: shots with differing corrections {dt.size}, (shot, window) with differing stats {len(ds)} first {ds[:6].tolist()}, "
          f"commits per unit histogram {np.bincount(s1[..., 7].ravel())[:4].tolist()}, flagged differ {(r0[:, 1] != r1[:, 1]).sum()}")

# phase timers of pre-converged units (exit class 0): serial form vs parallel form
for name, dec in (("serial", ser), ("parallel", par)):
    dec.set_profiling(True)
    t, s_, r = run(dec)
    prof = dec.get_profile(shots)  # [B, W, 8] ticks of 10 ns
    cls = s_[..., 0] & 0xFF
    pre = (cls == 0) & ((s_[..., 0] & 0x100) != 0)
    print(name, "pre-converged units:", int(pre.sum()), "mean us per phase [get+load, init, preBP, ...]:", (prof[pre].mean(axis=0) / 100).round(1).tolist())
