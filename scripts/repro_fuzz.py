#!/usr/bin/env python3
"""Re-create one trial of tests/fuzz_vs_oracle.py (osdw mode) and compare histories: python scripts/repro_fuzz.py <seed> <mlo> <mhi> <trial>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from slidingwindowdecoder_amd import osd_window
seed, mlo, mhi, want = (int(x) for x in sys.argv[1:5])
rng = np.random.default_rng(seed)
for t in range(want + 1):
    m = int(rng.integers(mlo, mhi)); n = int(rng.integers(max(m + 4, 0), max(min(6 * m, 2000), max(m + 4, 0) + 1)))
    dens = rng.uniform(1.5, 4.0) / m
    H = (rng.random((m, n)) < dens).astype(np.uint8)
    for c in range(n):
        if H[:, c].sum() == 0: H[rng.integers(m), c] = 1
    for r in range(m):
        if H[r].sum() == 0: H[r, rng.integers(n)] = 1
    if H.sum(axis=0).max() > 8 or H.sum(axis=1).max() > 60: continue
    p = rng.uniform(0.002, 0.1, size=n)
    if rng.random() < 0.3: p[:] = rng.uniform(0.005, 0.05)
    method = ["osd_0", "osd_cs", "osd_e"][int(rng.integers(3))]
    order = 0 if method == "osd_0" else int(rng.integers(0, 6))
    kw = dict(channel_probs=p, pre_max_iter=int(rng.integers(1, 10)), post_max_iter=int(rng.integers(1, 60)),
              ms_scaling_factor=float(rng.choice([1.0, 0.9, 0.75, 0.625])), osd_method=method, osd_order=order, new_n=int(rng.integers(m, n + 1)))
    try:
        ora = O.osd_window(H, **kw)
    except ValueError:
        continue
    B = 256
    e = (rng.random((B, n)) < p * rng.uniform(0.5, 3.0)).astype(np.uint8)
    synd = (e @ H.T) % 2
    synd[B // 2:] = (rng.random((B - B // 2, m)) < 0.3).astype(np.uint8)
    if t < want: continue
    dev = osd_window(H, **kw)
    print("trial", t, "m n", m, n, {k: v for k, v in kw.items() if k != "channel_probs"}, "col weight max", H.sum(axis=0).max(), "row", H.sum(axis=1).max())
    out = dev.decode_batch(synd, return_history=True, return_osd0=True)
    hist = dev.last_history
    want_out, res = ora.decode_batch(synd)
    bad = np.flatnonzero((out != want_out).any(axis=1))
    print("differing shots", bad.size, "of", B, "classes of differing", np.bincount(res["exit_class"][bad], minlength=4))
    # per-shot oracle history
    k = int(bad[0]) if bad.size else 0
    o2 = O.osd_window(H, **kw); w = o2.decode(synd[k]); oh = o2.log_prob_ratios  # [n,4]
    dh = hist[k].T  # [n,4]
    d = np.abs(dh - oh)
    print("shot", k, "history max abs diff", d.max(), "entries differing", int((dh != oh).sum()), "of", dh.size)
    print("osd0 equal:", np.array_equal(dev.last_osd0[k], o2.osd0_decoding), "out differ bits", int((out[k] != w).sum()))
    # without the history output (kernel kind 0/3 as the fuzz runs it)
    out2 = dev.decode_batch(synd)
    print("plain decode_batch differing shots", int((out2 != want_out).any(axis=1).sum()))
    break
