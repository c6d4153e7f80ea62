#!/usr/bin/env python3
"""Randomised device-vs-oracle comparison (not collected by pytest; run by hand on a GPU box):
    python tests/fuzz_vs_oracle.py [trials] [seed]
Random ragged matrices, priors, iteration counts, scaling factors, OSD methods/orders, shortening lengths;
syndromes from sampled errors and from random bits (inconsistent).  Everything must agree bit for bit."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from slidingwindowdecoder_amd import osd_window  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
mlo, mhi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (6, 120)
mode = sys.argv[5] if len(sys.argv) > 5 else "osdw"  # osdw | gdg | gd | bp | ens (bpgdg_decoder(multi_thread=True), the threaded ensemble)
nlo, nhi = (int(sys.argv[6]), int(sys.argv[7])) if len(sys.argv) > 7 else (0, 2000)  # column range (default: m + 4 .. min(6 m, 2000))
bad = 0
ran, threads_seen, osd_shots = 0, set(), 0
for t in range(trials):
    m = int(rng.integers(mlo, mhi))
    n = int(rng.integers(max(m + 4, nlo), max(min(6 * m, nhi), max(m + 4, nlo) + 1)))
    dens = rng.uniform(1.5, 4.0) / m
    H = (rng.random((m, n)) < dens).astype(np.uint8)
    for c in range(n):
        if H[:, c].sum() == 0:
            H[rng.integers(m), c] = 1
    for r in range(m):
        if H[r].sum() == 0:
            H[r, rng.integers(n)] = 1
    if H.sum(axis=0).max() > 8 or H.sum(axis=1).max() > 60:
        continue
    p = rng.uniform(0.002, 0.1, size=n)
    if rng.random() < 0.3:
        p[:] = rng.uniform(0.005, 0.05)  # equal priors: many exact ties in the orderings
    method = ["osd_0", "osd_cs", "osd_e"][int(rng.integers(3))]
    order = 0 if method == "osd_0" else int(rng.integers(0, 6))
    kw = dict(channel_probs=p, pre_max_iter=int(rng.integers(1, 10)), post_max_iter=int(rng.integers(1, 60)),
              ms_scaling_factor=float(rng.choice([1.0, 0.9, 0.75, 0.625])), osd_method=method, osd_order=order,
              new_n=int(rng.integers(m, n + 1)))
    if mode != "osdw":
        from slidingwindowdecoder_amd import bp_history_decoder, bpgd_decoder, bpgdg_decoder
        okw = dict(channel_probs=p, max_iter=int(rng.integers(1, 12)), ms_scaling_factor=kw["ms_scaling_factor"],
                   max_iter_per_step=int(rng.integers(2, 8)), max_step=int(rng.integers(3, 20)), max_tree_depth=int(rng.integers(1, 4)),
                   max_side_depth=int(rng.integers(4, 10)), max_tree_branch_step=10, max_side_branch_step=int(rng.integers(3, 10)),
                   new_n=kw["new_n"], low_error_mode=bool(rng.integers(2)))
        dc, oc = {"gdg": (bpgdg_decoder, O.bpgdg_decoder), "gd": (bpgd_decoder, O.bpgd_decoder), "bp": (bp_history_decoder, O.bp_history_decoder),
                  "ens": (bpgdg_decoder, O.bpgdg_decoder)}[mode]
        if mode not in ("gdg", "ens"):
            okw.pop("low_error_mode")
        if mode == "ens":  # the reference's threaded ensemble: tree depth 0..4, a few side threads, short tree / side walks
            okw.update(multi_thread=True, max_tree_depth=int(rng.integers(0, 5)), max_tree_branch_step=int(rng.integers(1, 8)))
            okw["max_side_depth"] = okw["max_tree_depth"] + int(rng.integers(0, 6))
        dev, ora = dc(H, **okw), oc(H, **okw)
        B = 48
        e = (rng.random((B, n)) < p * rng.uniform(0.5, 3.0)).astype(np.uint8)
        synd = (e @ H.T) % 2
        out = dev.decode_batch(synd)
        bad_here = 0
        for b in range(B):
            ora = oc(H, **okw)  # a fresh object per shot: the device gives every shot of a batch a fresh decoder's state
            w = ora.decode(synd[b])
            if not np.array_equal(w, out[b]) or bool(ora.converge) != bool(dev.last_status[b] & 0x100):
                bad_here += 1
        if bad_here:
            bad += 1
            print(f"trial {t}: {mode} MISMATCH m={m} n={n} shots {bad_here}/{B} kw={ {k: v for k, v in okw.items() if k != 'channel_probs'} }")
        continue
    try:
        ora = O.osd_window(H, **kw)
    except ValueError:
        continue
    dev = osd_window(H, **kw)
    B = 256 if m * n < 2_000_000 else 16  # (the oracle needs ~0.4 s per decode on the matrices of the large-graph kernels)
    e = (rng.random((B, n)) < p * rng.uniform(0.5, 3.0)).astype(np.uint8)
    synd = (e @ H.T) % 2
    synd[B // 2:] = (rng.random((B - B // 2, m)) < 0.3).astype(np.uint8)  # inconsistent half
    out = dev.decode_batch(synd)
    want, res = ora.decode_batch(synd)
    ran += 1; threads_seen.add(getattr(dev, "threads", None)); osd_shots += int((res["exit_class"] == 2).sum())
    ok = (out == want).all() and np.array_equal(dev.last_iterations, res["bp_iteration"]) and \
        np.array_equal(dev.last_min_pm, res["min_pm"]) and np.array_equal(dev.last_status & 0xFF, res["exit_class"])
    if not ok:
        bad += 1
        d = np.flatnonzero((out != want).any(axis=1) | (dev.last_iterations != res["bp_iteration"]) | (dev.last_min_pm != res["min_pm"])
                           | ((dev.last_status & 0xFF) != res["exit_class"]))
        print(f"trial {t}: MISMATCH m={m} n={n} kw={ {k: v for k, v in kw.items() if k != 'channel_probs'} } shots {d[:5].tolist()} "
              f"classes dev {(dev.last_status[d[:5]] & 0xFF).tolist()} ora {res['exit_class'][d[:5]].tolist()} "
              f"its dev {dev.last_iterations[d[:5]].tolist()} ora {res['bp_iteration'][d[:5]].tolist()} "
              f"vector differs {(out[d[:5]] != want[d[:5]]).sum(axis=1).tolist()} threads {getattr(dev, 'threads', None)}")
print(f"{trials} trials ({ran} osd_window comparisons, threads per shot {sorted(x for x in threads_seen if x)}, {osd_shots} shots through OSD), {bad} mismatching")
sys.exit(1 if bad else 0)
