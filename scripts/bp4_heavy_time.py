#!/usr/bin/env python3
"""How long does a bp4_osd decode that runs all max_iter iterations take on its own?  (the critical path of a ticket-scheduled launch)
python scripts/bp4_heavy_time.py -- [[144,12,12]] depolarizing p = 0.02: the non-converging decodes of a 65 536-decode batch, alone on the device"""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slidingwindowdecoder_amd import bp4_osd
from slidingwindowdecoder_amd.codes import bb_code
code, _, _ = bb_code(144)
n, p = 144, 0.02
pr = np.full(n, p / 3)
dec = bp4_osd(code.hx, code.hz, channel_probs_x=pr, channel_probs_y=pr, channel_probs_z=pr, max_iter=100, ms_scaling_factor=0.625, osd_method="osd_cs", osd_order=10)
rng = np.random.default_rng([20240318, 0, 0])
B = 65536
pauli = rng.choice(4, size=(B, n), p=[1 - p, p / 3, p / 3, p / 3])
ex, ez = ((pauli == 1) | (pauli == 2)).astype(np.uint8), ((pauli == 3) | (pauli == 2)).astype(np.uint8)
sx = np.ascontiguousarray((ez @ code.hx.T % 2).astype(np.uint8)); sz = np.ascontiguousarray((ex @ code.hz.T % 2).astype(np.uint8))
dev = torch.device("cuda", 0)
def run(sx, sz, reps=5):
    tx, tz = torch.from_numpy(sx).to(dev), torch.from_numpy(sz).to(dev)
    out = torch.empty((len(sx), 2, n), dtype=torch.uint8, device=dev); st = torch.empty((len(sx), 8), dtype=torch.int32, device=dev)
    for _ in range(2): dec.decode_batch_device(tx, tz, out=out, stats=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): dec.decode_batch_device(tx, tz, out=out, stats=st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, st.cpu().numpy()
ms_all, st = run(sx, sz)
heavy = np.flatnonzero(st[:, 1] >= 100)
light = np.flatnonzero(st[:, 1] < 100)
ms_heavy, _ = run(sx[heavy], sz[heavy])
ms_one, _ = run(sx[heavy[:1]], sz[heavy[:1]])
ms_light, _ = run(sx[light], sz[light])
print(json.dumps({"decodes": B, "ms_all": round(ms_all, 3), "heavy_decodes": int(len(heavy)), "ms_heavy_only": round(ms_heavy, 3), "ms_one_heavy": round(ms_one, 3),
                  "ms_light_only": round(ms_light, 3), "light_decodes": int(len(light))}))
