#!/usr/bin/env python3
"""Do two HIP streams overlap when other live streams exist?  K extra streams are made (and used once) before the pair that alternates bp4_osd
launches; ms per launch with the pair against one launch at a time.  python scripts/hwq_probe.py  (GPU_MAX_HW_QUEUES in the environment is the
runtime's own knob: hardware queues per device, default 4)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slidingwindowdecoder_amd import bp4_osd
from slidingwindowdecoder_amd.codes import bb_code
B = 65536
c, _, _ = bb_code(144)
hx, hz = np.asarray(c.hx), np.asarray(c.hz)
n = hx.shape[1]; p = 0.02
pr = np.full(n, p / 3)
dec = bp4_osd(hx, hz, channel_probs_x=pr, channel_probs_y=pr, channel_probs_z=pr, max_iter=100, ms_scaling_factor=0.625, osd_method="osd_cs", osd_order=10)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(5)
pauli = rng.choice(4, size=(B, n), p=[1 - p, p / 3, p / 3, p / 3])
ex, ez = ((pauli == 1) | (pauli == 2)).astype(np.uint8), ((pauli == 3) | (pauli == 2)).astype(np.uint8)
sx = torch.from_numpy(np.ascontiguousarray((ez @ hx.T % 2).astype(np.uint8))).to(dev)
sz = torch.from_numpy(np.ascontiguousarray((ex @ hz.T % 2).astype(np.uint8))).to(dev)
outs = [(torch.empty((B, 2, n), dtype=torch.uint8, device=dev), torch.empty((B, 8), dtype=torch.int32, device=dev)) for _ in range(2)]
keep = []
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
for K in range(0, 9):
    while len(keep) < K:
        s = torch.cuda.Stream(dev)
        with torch.cuda.stream(s):
            torch.zeros(16, device=dev).add_(1)
        keep.append(s)
    torch.cuda.synchronize()
    lanes = [torch.cuda.Stream(dev), torch.cuda.Stream(dev, priority=-1 if os.environ.get("PROBE_PRIO") else 0)]
    for ln in lanes: ln.wait_stream(torch.cuda.current_stream(dev))
    res = []
    for rep in range(2):
        for k in range(4): dec.decode_batch_device(sx, sz, out=outs[k & 1][0], stats=outs[k & 1][1], stream=lanes[k & 1])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(30): dec.decode_batch_device(sx, sz, out=outs[k & 1][0], stats=outs[k & 1][1], stream=lanes[k & 1])
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 30 * 1e3)
    print(f"{K} other live streams: {min(res):.3f} ms per launch on the pair")
    del lanes
