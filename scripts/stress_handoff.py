#!/usr/bin/env python3
"""Determinism stress of the window hand-over: the same batch decoded repeatedly must give identical results."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from slidingwindowdecoder_amd import SlidingWindowDecoder
from slidingwindowdecoder_amd.windows import sample_dem
for name, plan, shots in (("bb144", bench.build_problem(), 4096), ("bb288", bench.build_problem(N=288, W=4, F=1), 1024)):
    dec = SlidingWindowDecoder(plan, **dict(bench.DECODER_KW, osd_order=0))
    det, obs, _ = sample_dem(plan.chk, plan.obs, plan.priors, shots, seed=3)
    d = torch.from_numpy(det).cuda()
    ref = None
    bad = 0
    for rep in range(12):
        total, stats, pm = dec.decode_device(d)
        torch.cuda.synchronize()
        t = total.cpu().numpy(); st = stats.cpu().numpy()
        if ref is None:
            ref = (t, st)
        else:
            nd = int((t != ref[0]).any(axis=1).sum()); ns = int((st[..., :4] != ref[1][..., :4]).any(axis=(1, 2)).sum())
            if nd or ns:
                bad += 1
                w = np.argwhere((st[..., :4] != ref[1][..., :4]).any(axis=2))
                print(f"  {name} rep {rep}: {nd} shots differ in total, {ns} in stats; first (shot, window): {w[:3].tolist()}")
    print(name, "threads", dec.threads, "lds", dec.lds_bytes, "->", "OK" if not bad else f"{bad} bad repetitions")
