#!/usr/bin/env python3
"""Where a bp4 bench step's wall time goes: host time per asynchronous call, device time per launch pair (HIP events), wall time of K back-to-back calls."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
class A: pass
a = A(); a.distinct_batches = 4; a.steps = 10; a.warmup = 2
eng = bench.Bp4Engine(a, 0, 0, 0, 65536) if hasattr(bench, "Bp4Engine") else None
if eng is None:
    cls = [v for k, v in vars(bench).items() if isinstance(v, type) and "bp4" in (v.__doc__ or "").lower()]
    eng = cls[0](a, 0, 0, 0, 65536)
for i in range(3): eng.step(i)
torch.cuda.synchronize()
for K in (1, 4, 10, 20):
    t0 = time.perf_counter(); hs = []
    for i in range(K):
        h0 = time.perf_counter(); eng.step(i); hs.append(time.perf_counter() - h0)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"K={K:2d}: wall {1e3 * (t2 - t0) / K:.3f} ms per step; host time per call mean {1e3 * np.mean(hs):.3f} max {1e3 * np.max(hs):.3f} ms; final sync {1e3 * (t2 - t1):.3f} ms")
eng.set_timing(True)
for i in range(6): eng.step(i); torch.cuda.synchronize()
ms, k = eng.get_timing()
print(f"device time per launch pair (HIP events, one at a time): {ms / k:.3f} ms")
