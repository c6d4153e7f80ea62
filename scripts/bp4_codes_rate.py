#!/usr/bin/env python3
"""bp4_osd decodes/s on device-resident syndromes for the codes of the reference's notebooks (diagnostics; SWD_LIB selects a build):
python scripts/bp4_codes_rate.py [decodes per launch]"""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slidingwindowdecoder_amd import bp4_osd
from slidingwindowdecoder_amd.codes import bb_code
from slidingwindowdecoder_amd import shyps
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cases = []
for N in (72, 144, 288, 360, 756):
    c, _, _ = bb_code(N)
    cases.append((f"bb{N}", np.asarray(c.hx), np.asarray(c.hz), 0.02))
hx, hz = shyps.shyps_stabilizers(3)
cases.append(("shyps_r3", np.asarray(hx), np.asarray(hz), 0.01))
dev = torch.device("cuda", 0)
# (ONE pair of streams for every code: HIP deals streams onto a few hardware queues in creation order, and a pair that lands on one
#  queue does not overlap at all -- seen with a fresh pair per code: the second pair ran its launches back to back)
lanes = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
for name, hx, hz, p in cases:
    n = hx.shape[1]
    pr = np.full(n, p / 3)
    dec = bp4_osd(hx, hz, channel_probs_x=pr, channel_probs_y=pr, channel_probs_z=pr, max_iter=100, ms_scaling_factor=0.625, osd_method="osd_cs", osd_order=min(10, n - max(hx.shape[0], hz.shape[0])), device=0)
    rng = np.random.default_rng(5)
    pauli = rng.choice(4, size=(B, n), p=[1 - p, p / 3, p / 3, p / 3])
    ex, ez = ((pauli == 1) | (pauli == 2)).astype(np.uint8), ((pauli == 3) | (pauli == 2)).astype(np.uint8)
    sx = torch.from_numpy(np.ascontiguousarray((ez @ hx.T % 2).astype(np.uint8))).to(dev)
    sz = torch.from_numpy(np.ascontiguousarray((ex @ hz.T % 2).astype(np.uint8))).to(dev)
    out = torch.empty((B, 2, n), dtype=torch.uint8, device=dev); stats = torch.empty((B, 8), dtype=torch.int32, device=dev)
    for _ in range(2): dec.decode_batch_device(sx, sz, out=out, stats=stats)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): dec.decode_batch_device(sx, sz, out=out, stats=stats)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    # consecutive launches on two streams in turn (two sets of output buffers), wall clock over 20 launches
    import time
    outs = [(out, stats), (torch.empty_like(out), torch.empty_like(stats))]
    for ln in lanes: ln.wait_stream(torch.cuda.current_stream(dev))
    for k in range(4): dec.decode_batch_device(sx, sz, out=outs[k & 1][0], stats=outs[k & 1][1], stream=lanes[k & 1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(20): dec.decode_batch_device(sx, sz, out=outs[k & 1][0], stats=outs[k & 1][1], stream=lanes[k & 1])
    torch.cuda.synchronize()
    ms2 = (time.perf_counter() - t0) / 20 * 1e3
    st = stats.cpu().numpy()
    print(json.dumps({"code": name, "n": n, "decodes_per_launch": B, "ms_per_launch": round(ms, 3), "decodes_per_s": round(B / ms * 1e3), "two_streams_ms_per_launch": round(ms2, 3), "two_streams_decodes_per_s": round(B / ms2 * 1e3), "osd_share": float(((st[:, 0] & 0xFF) == 2).mean()), "mean_iters": float(st[:, 1].mean()), "checksum": int(out.sum().item())}))
