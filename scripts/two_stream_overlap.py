#!/usr/bin/env python3
"""Does overlapping consecutive launches (two streams, two sets of output buffers) hide the tail of the persistent grid?
python scripts/two_stream_overlap.py [order]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from slidingwindowdecoder_amd import DemSampler, SlidingWindowDecoder
order = int(sys.argv[1]) if len(sys.argv) > 1 else 10
shots = 4096
plan = bench.build_problem()
dec = SlidingWindowDecoder(plan, **dict(bench.DECODER_KW, osd_order=order))
sampler = DemSampler(plan.chk, plan.obs, plan.priors)
dets = [sampler.sample_device(shots, seed=20240318, first_shot=i << 24)[0] for i in range(4)]
def bufs():
    return dict(total=torch.empty((shots, plan.chk.shape[1]), dtype=torch.uint8, device="cuda"),
                stats=torch.empty((shots, dec.W, 8), dtype=torch.int32, device="cuda"),
                shot_result=torch.empty((shots, 2), dtype=torch.int32, device="cuda"))
for nstreams in (1, 2, 3, 1, 2):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    B = [bufs() for _ in range(nstreams)]
    for i in range(4):
        with torch.cuda.stream(streams[i % nstreams]):
            dec.decode_device(dets[i % 4], min_pm=None, **B[i % nstreams])
    torch.cuda.synchronize()
    K = 60
    t0 = time.perf_counter()
    for i in range(K):
        with torch.cuda.stream(streams[i % nstreams]):
            dec.decode_device(dets[i % 4], min_pm=None, **B[i % nstreams])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dec.check_status()
    print(f"{nstreams} stream(s): {dt / K * 1e3:.3f} ms per step, {shots * dec.W * K / dt / 1e6:.3f} M windows/s")
# results identical whatever the overlap
ref = bufs(); dec.decode_device(dets[1], min_pm=None, **ref); torch.cuda.synchronize()
s2 = [torch.cuda.Stream() for _ in range(2)]; B = [bufs(), bufs()]
for i in range(6):
    with torch.cuda.stream(s2[i % 2]):
        dec.decode_device(dets[(i + 1) % 4] if i != 4 else dets[1], min_pm=None, **B[i % 2])
torch.cuda.synchronize()
print("overlapped launch reproduces the serial result:", bool(torch.equal(B[0]["total"], ref["total"])) and bool(torch.equal(B[0]["shot_result"], ref["shot_result"])))
