#!/usr/bin/env python3
"""bp4_osd decodes/s with K launches in flight on K streams (the handle has four launch slots): python scripts/bp4_lanes.py [decodes per launch]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slidingwindowdecoder_amd import bp4_osd
from slidingwindowdecoder_amd.codes import bb_code
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
c, _, _ = bb_code(144)
hx, hz = np.asarray(c.hx), np.asarray(c.hz)
n = hx.shape[1]; p = 0.02
pr = np.full(n, p / 3)
dec = bp4_osd(hx, hz, channel_probs_x=pr, channel_probs_y=pr, channel_probs_z=pr, max_iter=100, ms_scaling_factor=0.625, osd_method="osd_cs", osd_order=10)
dev = torch.device("cuda", 0)
data = []
for i in range(4):
    rng = np.random.default_rng([20240318, i, 0])
    pauli = rng.choice(4, size=(B, n), p=[1 - p, p / 3, p / 3, p / 3])
    ex, ez = ((pauli == 1) | (pauli == 2)).astype(np.uint8), ((pauli == 3) | (pauli == 2)).astype(np.uint8)
    data.append((torch.from_numpy(np.ascontiguousarray((ez @ hx.T % 2).astype(np.uint8))).to(dev), torch.from_numpy(np.ascontiguousarray((ex @ hz.T % 2).astype(np.uint8))).to(dev)))
outs = [(torch.empty((B, 2, n), dtype=torch.uint8, device=dev), torch.empty((B, 8), dtype=torch.int32, device=dev)) for _ in range(4)]
lanes = [torch.cuda.Stream(dev), torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev), torch.cuda.Stream(dev, priority=-1)]
torch.cuda.synchronize()
for K in (1, 2, 3, 4):
    best = 1e9
    for rep in range(3):
        for k in range(2 * K): dec.decode_batch_device(*data[k % 4], out=outs[k % K][0], stats=outs[k % K][1], stream=lanes[k % K])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(48): dec.decode_batch_device(*data[k % 4], out=outs[k % K][0], stats=outs[k % K][1], stream=lanes[k % K])
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 48 * 1e3)
    print(f"{K} launches in flight: {best:.3f} ms per launch of {B} decodes = {B / best / 1e3:.1f} M decodes/s")
