#!/usr/bin/env python3
"""Many small launches of the sliding-window pipeline on two streams of ONE decoder handle: every launch has to
reproduce the reference result of its batch (hand-over between workgroups on different XCDs, launch-slot ring,
scheduling flags).      python scripts/stress_launches.py [launches=100000] [shots per launch=97]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from slidingwindowdecoder_amd import SlidingWindowDecoder
from slidingwindowdecoder_amd.windows import sample_dem
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
shots = int(sys.argv[2]) if len(sys.argv) > 2 else 97
plan = bench.build_problem()
dec = SlidingWindowDecoder(plan, **dict(bench.DECODER_KW, osd_order=0))
dets, want = [], []
for k in range(4):
    det, _, _ = sample_dem(plan.chk, plan.obs, plan.priors, shots, seed=100 + k)
    d = torch.from_numpy(np.ascontiguousarray(det)).cuda()
    total, stats, _ = dec.decode_device(d)
    torch.cuda.synchronize()
    dets.append(d); want.append((total.clone(), stats[..., :4].clone()))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
outs = [[torch.empty_like(want[0][0]), torch.empty((shots, dec.W, 8), dtype=torch.int32, device="cuda")] for _ in range(8)]
bad = 0
t0 = time.time()
for i in range(0, launches, 8):
    for j in range(8):  # eight launches in flight on two streams: twice the launch-slot ring
        k = (i + j) % 4
        with torch.cuda.stream(streams[j % 2]):
            dec.decode_device(dets[k], total=outs[j][0], stats=outs[j][1], stream=streams[j % 2])
    torch.cuda.synchronize()
    for j in range(8):
        k = (i + j) % 4
        if not (torch.equal(outs[j][0], want[k][0]) and torch.equal(outs[j][1][..., :4], want[k][1])):
            bad += 1
            print(f"launch {i + j}: result differs from the reference result of its batch")
dec.check_status()
print(f"{launches} launches of {shots} shots x {dec.W} windows on two streams: {bad} differing, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
