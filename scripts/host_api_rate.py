#!/usr/bin/env python3
"""PCIe-inclusive rate of the headline workload through the host-buffer API -- what a notebook that switches to this package
calls: numpy arrays in and out (pageable memory).  Three forms, OSD-CS order 10 (and order 0 beside it):
  decode(det)                one synchronous call per 4096-shot batch (the library cuts it in two halves on its two lanes)
  decode(det, packed=True)   the same, total_e_hat returned bit-packed as it travels
  decode_stream(batches)     consecutive batches, two in flight (copies / unpacking of batch k overlap the launch of k + 1)
next to the device-resident rate bench.py reports.  python scripts/host_api_rate.py [shots] [batches]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from slidingwindowdecoder_amd import SlidingWindowDecoder
from slidingwindowdecoder_amd.windows import sample_dem

shots = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 12
plan = bench.build_problem()
dets = [sample_dem(plan.chk, plan.obs, plan.priors, shots, seed=11 + i)[0] for i in range(4)]
out = {"shots_per_batch": shots, "windows_per_shot": len(plan.windows), "bytes_in_per_batch": int(dets[0].nbytes),
       "bytes_out_unpacked": int(shots * plan.chk.shape[1]), "bytes_out_packed": int(shots * ((plan.chk.shape[1] + 7) // 8))}
for order in (10, 0):
    dec = SlidingWindowDecoder(plan, **dict(bench.DECODER_KW, osd_order=order))
    W = dec.W
    ref = dec.decode(dets[0])

    def med(fn, k=7):
        ts = []
        for i in range(k):
            t0 = time.perf_counter(); fn(i); ts.append(time.perf_counter() - t0)
        return float(np.median(ts))
    t_sync = med(lambda i: dec.decode(dets[i % 4]))
    t_packed = med(lambda i: dec.decode(dets[i % 4], packed=True))
    bits = dec.decode(dets[0], packed=True)
    assert np.array_equal(np.unpackbits(bits, axis=1, count=plan.chk.shape[1], bitorder="little"), ref)
    res = {}
    for packed in (False, True):
        list(dec.decode_stream([dets[i % 4] for i in range(3)], packed=packed))  # warm-up (allocates the lanes)
        t0 = time.perf_counter()
        n = 0
        for tot, st, pm, flips, flagged in dec.decode_stream((dets[i % 4] for i in range(nb)), packed=packed):
            n += tot.shape[0]
        res[packed] = (time.perf_counter() - t0) / nb
        assert n == nb * shots
    first = next(iter(dec.decode_stream([dets[0]])))[0]
    assert np.array_equal(first, ref)
    out[f"osd_cs_{order}"] = {
        "decode_ms_per_batch": t_sync * 1e3, "decode_windows_per_s": shots * W / t_sync,
        "decode_packed_ms_per_batch": t_packed * 1e3, "decode_packed_windows_per_s": shots * W / t_packed,
        "stream_ms_per_batch": res[False] * 1e3, "stream_windows_per_s": shots * W / res[False],
        "stream_packed_ms_per_batch": res[True] * 1e3, "stream_packed_windows_per_s": shots * W / res[True]}
print(json.dumps(out))
