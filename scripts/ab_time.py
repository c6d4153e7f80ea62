#!/usr/bin/env python3
"""Launch time of the headline pipeline for one build of the library: SWD_LIB=libswd_hip_devA.so python scripts/ab_time.py"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from slidingwindowdecoder_amd import _lib
if os.environ.get("SWD_LIB"): _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["SWD_LIB"])
from slidingwindowdecoder_amd import SlidingWindowDecoder
from slidingwindowdecoder_amd.windows import sample_dem
shots = 4096
plan = bench.build_problem(N=288, W=4, F=1) if os.environ.get("SWD_CONFIG") == "288" else bench.build_problem()  # SWD_CONFIG=288: configs[3]
dec = SlidingWindowDecoder(plan, **dict(bench.DECODER_KW, osd_order=int(os.environ.get("SWD_ORDER", "0"))))
ds = [torch.from_numpy(sample_dem(plan.chk, plan.obs, plan.priors, shots, seed=s)[0]).cuda() for s in (1, 2)]
for d in ds: dec.decode_device(d)
torch.cuda.synchronize()
ts = []
for r in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); dec.decode_device(ds[r % 2]); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts = np.array(ts)
print(f"{os.environ.get('SWD_LIB', 'libswd_hip.so'):28s} ms per launch: min {ts.min():.2f} median {np.median(ts):.2f}  -> {shots * dec.W / np.median(ts) / 1e3:.3f} M windows/s; lds {dec.lds_bytes} threads {dec.threads}")
