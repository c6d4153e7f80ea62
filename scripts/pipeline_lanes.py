#!/usr/bin/env python3
"""Sliding-window pipeline rate with K launches in flight on K caller-owned streams (swd_pipeline_decode_dev takes the stream; the handle has four
launch slots): python scripts/pipeline_lanes.py <workload> [shots]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from slidingwindowdecoder_amd import DemSampler, SlidingWindowDecoder
wl = sys.argv[1] if len(sys.argv) > 1 else "gdg64"
shots = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
plan = bench.build_problem(**bench.WORKLOADS[wl]["problem"])
kw = dict(bench.GDG_KW) if wl == "gdg" else (dict(bench.GDG64_KW) if wl == "gdg64" else dict(bench.DECODER_KW, osd_order=10, **bench.WORKLOADS[wl].get("decoder_kw", {})))
dec = SlidingWindowDecoder(plan, **kw)
dev = torch.device("cuda", 0)
sampler = DemSampler(plan.chk, plan.obs, plan.priors)
dets = [sampler.sample_device(shots, seed=20240318, first_shot=i * (1 << 24))[0] for i in range(4)]
W = len(plan.windows)
outs = [dict(total=torch.empty((shots, plan.chk.shape[1]), dtype=torch.uint8, device=dev), stats=torch.empty((shots, W, 8), dtype=torch.int32, device=dev),
             shot_result=torch.empty((shots, 2), dtype=torch.int32, device=dev)) for _ in range(4)]
lanes = [torch.cuda.Stream(dev, priority=-(i & 1)) for i in range(4)]
torch.cuda.synchronize()
steps = int(os.environ.get("STEPS", "16"))
for K in (1, 2, 3, 4):
    for ln in lanes: ln.wait_stream(torch.cuda.current_stream(dev))
    for k in range(2 * K): dec.decode_device(dets[k % 4], min_pm=None, stream=lanes[k % K], **outs[k % K])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps): dec.decode_device(dets[k % 4], min_pm=None, stream=lanes[k % K], **outs[k % K])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    dec.check_status()
    print(f"{wl}: {K} launches in flight: {ms:.2f} ms per step of {shots} shots = {shots * W / ms / 1e3:.3f} M windows/s")
