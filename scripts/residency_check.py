#!/usr/bin/env python3
"""Diagnostic (library built with -DSWD_RESIDENCY): how many workgroups of a pipeline launch are resident at once?
    SWD_LIB=libswd_hip_devR.so python scripts/residency_check.py [N W] ...   (pairs of code size and window width)"""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from slidingwindowdecoder_amd import SlidingWindowDecoder, _lib
from slidingwindowdecoder_amd.windows import sample_dem
args = [int(x) for x in sys.argv[1:]] or [144, 3]
L = _lib.lib(); buf = (C.c_uint32 * 16)()
L.swd_pipeline_debug_counters.argtypes = [C.c_void_p, C.c_void_p]
for N, W in zip(args[0::2], args[1::2]):
    try:
        plan = bench.build_problem(N=N, W=W)
        det, obs, _ = sample_dem(plan.chk, plan.obs, plan.priors, 4096, seed=7)
        dec = SlidingWindowDecoder(plan, **dict(bench.DECODER_KW, osd_order=0))
        dec.decode(det)
        L.swd_pipeline_debug_counters(dec._h, buf)
        dec.decode(det)
        L.swd_pipeline_debug_counters(dec._h, buf)
        w = list(buf)
        print(f"[[{N}]] (W={W}): threads {dec.threads}, LDS {dec.lds_bytes} B | resident at once (max): {w[5]}, workgroups that decoded >= 1 unit: {w[6]}", flush=True)
    except Exception as e:
        print(f"[[{N}]] (W={W}): {type(e).__name__}: {str(e)[:100]}", flush=True)
