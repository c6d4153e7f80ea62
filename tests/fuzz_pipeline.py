#!/usr/bin/env python3
"""Randomised whole-pipeline comparison (not collected by pytest; run by hand or from the GPU suite):
    python tests/fuzz_pipeline.py [trials] [seed] [decoder] [max detectors per round]
Random block-banded detector error models (R rounds of h detectors; faults local to a round or
reaching into the next one), random (W, F), priors and decoder parameters.  The device runs all
shots and windows in one launch; the expectation is the host-side window loop of
windows.sliding_window_decode_host (commit rule of osd.py:130-179) driven with the oracle, a fresh
oracle state per decode.  Everything must agree bit for bit."""
import os
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from slidingwindowdecoder_amd import SlidingWindowDecoder  # noqa: E402
from slidingwindowdecoder_amd.windows import Window, WindowPlan, sliding_window_decode_host  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
decoder = sys.argv[3] if len(sys.argv) > 3 else "osd_window"
hmax = int(sys.argv[4]) if len(sys.argv) > 4 else 48


class Fresh:
    """Oracle decoder whose state is that of a newly built object at every decode."""

    def __init__(self, make):
        self.make = make

    def decode(self, s):
        self.d = self.make()
        return self.d.decode(s)


def random_plan():
    h = int(rng.integers(6, hmax))
    R = int(rng.integers(4, 9))
    W = int(rng.integers(2, min(R, 4) + 1))
    F = int(rng.integers(1, W))  # F < W: the committed columns must lie inside the window
    nloc = [int(rng.integers(h, 3 * h)) for _ in range(R)]
    nspan = [int(rng.integers(h // 2, 2 * h)) if r < R - 1 else 0 for r in range(R)]
    rows, cols = [], []
    col = 0
    starts = []
    for r in range(R):
        starts.append(col)
        for kind, cnt in (("loc", nloc[r]), ("span", nspan[r])):
            for i in range(cnt):
                if kind == "loc":
                    k = int(rng.integers(1, 4))
                    rr = r * h + rng.choice(h, size=min(k, h), replace=False)
                    if i < h:  # every detector has a fault local to its own round: no empty window rows
                        rr = np.union1d(rr, [r * h + i])
                else:
                    k0, k1 = int(rng.integers(1, 3)), int(rng.integers(1, 3))
                    rr = np.concatenate((r * h + rng.choice(h, size=k0, replace=False),
                                         (r + 1) * h + rng.choice(h, size=k1, replace=False)))
                rows += rr.tolist()
                cols += [col] * len(rr)
                col += 1
    starts.append(col)
    num_row, num_col = R * h, col
    chk = sp.csr_matrix((np.ones(len(rows), np.uint8), (rows, cols)), shape=(num_row, num_col))
    if np.diff(chk.indptr).max() > 60 or np.diff(chk.indptr).min() == 0:
        return None
    priors = rng.uniform(0.002, 0.06, size=num_col)
    if rng.random() < 0.3:
        priors[:] = rng.uniform(0.005, 0.03)
    nobs = int(rng.integers(1, 13))
    obs = sp.csr_matrix((rng.random((nobs, num_col)) < 0.1).astype(np.uint8))
    noisy_prior = float(rng.uniform(0.01, 0.1))
    anchors = [(r * h, starts[r]) for r in range(R)] + [(num_row, num_col)]
    wins = []
    top = 0
    while True:
        last = top + W >= R
        a = anchors[top]
        b = anchors[min(top + W, R)]
        if not last:
            c1 = starts[top + W - 1] + nloc[top + W - 1]
            sub = chk[a[0]:b[0], a[1]:c1]
            nrow = b[0] - a[0]
            ident = sp.csr_matrix((np.ones(h, np.uint8), (np.arange(nrow - h, nrow), np.arange(h))), shape=(nrow, h))
            mat = sp.hstack((sub, ident), format="csr")
            prior = np.concatenate((priors[a[1]:c1], np.full(h, noisy_prior)))
            ncg = c1 - a[1]
            commit = anchors[top + F][1] - a[1]
        else:
            mat = sp.csr_matrix(chk[a[0]:b[0], a[1]:b[1]])
            prior = priors[a[1]:b[1]].copy()
            ncg = commit = b[1] - a[1]
        mat.sort_indices()
        wins.append(Window(a[0], b[0], a[1], ncg, commit, mat, prior, last))
        if last:
            break
        top += F
    return WindowPlan(chk, obs, priors, np.arange(num_col), anchors, wins, noisy_prior, h), (h, R, W, F)


bad = done = refused = 0
while done < trials:
    r = random_plan()
    if r is None:
        continue
    plan, geo = r
    mmax = max(w.mat.shape[0] for w in plan.windows)
    nmax = max(w.mat.shape[1] for w in plan.windows)
    nmin = min(w.mat.shape[1] for w in plan.windows)
    if decoder == "osd_window":
        method = ["osd_0", "osd_cs", "osd_e"][int(rng.integers(3))]
        kw = dict(pre_max_iter=int(rng.integers(1, 10)), post_max_iter=int(rng.integers(1, 40)),
                  ms_scaling_factor=float(rng.choice([1.0, 0.9, 0.75, 0.625])), osd_method=method,
                  osd_order=0 if method == "osd_0" else int(rng.integers(0, 6)), new_n=int(rng.integers(mmax, nmax + 1)))
        make = lambda w: Fresh(lambda: O.osd_window(w.mat, channel_probs=w.prior, **kw))  # noqa: E731
    else:
        kw = dict(max_iter=int(rng.integers(1, 12)), ms_scaling_factor=float(rng.choice([1.0, 0.9, 0.75, 0.625])),
                  max_iter_per_step=int(rng.integers(1, 8)), max_step=int(rng.integers(3, 20)), max_tree_depth=int(rng.integers(1, 4)),
                  max_side_depth=int(rng.integers(4, 10)), max_tree_branch_step=10, max_side_branch_step=int(rng.integers(3, 10)),
                  new_n=int(rng.integers(mmax, nmax + 1)))
        if decoder == "bpgdg_decoder":
            kw["low_error_mode"] = bool(rng.integers(2))
        if decoder == "ens":  # bpgdg_decoder(multi_thread=True): the threaded ensemble -- in a pipeline its tree and side threads are work items
            kw.update(multi_thread=True, low_error_mode=bool(rng.integers(2)), max_tree_depth=int(rng.integers(0, 5)),
                      max_tree_branch_step=int(rng.integers(0, 6)))
            kw["max_side_depth"] = kw["max_tree_depth"] + int(rng.integers(0, 5))
        oc = getattr(O, "bpgdg_decoder" if decoder == "ens" else decoder)
        make = lambda w: Fresh(lambda: oc(w.mat, channel_probs=w.prior, **kw))  # noqa: E731
    try:
        for w in plan.windows:
            make(w).decode(np.zeros(w.mat.shape[0], np.uint8))
    except ValueError:
        continue  # a window matrix the oracle rejects (rank deficient for OSD)
    done += 1
    B = 64
    e = (rng.random((B, plan.chk.shape[1])) < plan.priors * rng.uniform(0.5, 2.5)).astype(np.uint8)
    det = ((sp.csr_matrix(e) @ plan.chk.T.astype(np.int32)).toarray() % 2).astype(np.uint8)
    det[B - 8:] = (rng.random((8, plan.chk.shape[0])) < 0.2).astype(np.uint8)  # inconsistent tail
    try:
        dev = SlidingWindowDecoder(plan, decoder="bpgdg_decoder" if decoder == "ens" else decoder, **kw)
    except (ValueError, RuntimeError) as ex:
        if decoder != "osd_window" and "bytes of LDS" in str(ex):
            refused += 1  # the guessing decoders have no large-graph form (docs/history/DESIGN_rounds_1-5.md section 7): a documented refusal, not a result
            continue
        print(f"trial {done}: device rejected geo={geo} m<={mmax} n<={nmax}: {ex}")
        bad += 1
        continue
    total = dev.decode(det)
    its = np.zeros((B, len(plan.windows)), np.int64)
    conv = np.zeros((B, len(plan.windows)), bool)

    def tap(wi, j, dec, s, e_hat):
        its[j, wi] = dec.d.bp_iteration if hasattr(dec.d, "bp_iteration") else 0
        conv[j, wi] = bool(dec.d.converge)

    want, _ = sliding_window_decode_host(plan, det, make, on_decode=tap)
    ok = np.array_equal(total, want) and np.array_equal((dev.last_stats[:, :, 0] & 0x100) != 0, conv)
    if decoder == "osd_window":
        ok = ok and np.array_equal(dev.last_stats[:, :, 1], its)
    if ok and done % 3 == 0:  # every third trial: the packed transport and the two-lane stream must give what the one-shot decode gave
        st0, pm0, fl0, fg0 = dev.last_stats.copy(), dev.last_min_pm.copy(), dev.last_obs_flips.copy(), dev.last_flagged.copy()
        bits = dev.decode(det, packed=True)
        ok = ok and np.array_equal(np.unpackbits(bits, axis=1, count=plan.chk.shape[1], bitorder="little"), total) and np.array_equal(dev.last_stats, st0)
        cuts = sorted(set(int(x) for x in rng.integers(1, B, size=3)))
        parts = [det[a:b] for a, b in zip([0] + cuts, cuts + [B])]
        got = list(dev.decode_stream(parts, packed=bool(done % 2)))
        tot_s = np.concatenate([np.unpackbits(g[0], axis=1, count=plan.chk.shape[1], bitorder="little") if done % 2 else g[0] for g in got])
        ok = ok and np.array_equal(tot_s, total) and np.array_equal(np.concatenate([g[1] for g in got]), st0)
        ok = ok and np.array_equal(np.concatenate([g[2] for g in got]), pm0, equal_nan=True)
        ok = ok and np.array_equal(np.concatenate([g[3] for g in got]).astype(np.uint32), fl0) and np.array_equal(np.concatenate([g[4] for g in got]).astype(bool), fg0)
        dev.last_stats, dev.last_min_pm, dev.last_obs_flips, dev.last_flagged = st0, pm0, fl0, fg0
        if not ok:
            print(f"trial {done}: packed / streamed decode differs from the one-shot decode")
    pred = (sp.csr_matrix(want) @ plan.obs.T.astype(np.int32)).toarray() % 2
    mask = (pred.astype(np.uint32) << np.arange(plan.obs.shape[0], dtype=np.uint32)).sum(axis=1).astype(np.uint32)
    resid = ((det + (sp.csr_matrix(want) @ plan.chk.T.astype(np.int32)).toarray()) % 2).any(axis=1)
    ok = ok and np.array_equal(dev.last_obs_flips, mask) and np.array_equal(dev.last_flagged, resid)
    if not ok:
        bad += 1
        d = np.flatnonzero((total != want).any(axis=1))
        print(f"trial {done}: MISMATCH {decoder} geo(h,R,W,F)={geo} windows={len(plan.windows)} m<={mmax} n in [{nmin},{nmax}] "
              f"kw={kw} shots differing {d.size}/{B} first {d[:6].tolist()} threads {dev.threads} "
              f"flips_ok {np.array_equal(dev.last_obs_flips, mask)} flagged_ok {np.array_equal(dev.last_flagged, resid)} "
              f"conv_ok {np.array_equal((dev.last_stats[:, :, 0] & 0x100) != 0, conv)}")
        if os.environ.get("SWD_FUZZ_DEBUG") and decoder == "ens" and d.size:
            # which window, and does the single-window device decoder agree with the oracle on that window's syndrome?
            from slidingwindowdecoder_amd import bpgdg_decoder
            j = int(d[0])
            chk_t = sp.csr_matrix(plan.chk.T.astype(np.int32))
            tot = np.zeros(plan.chk.shape[1], np.uint8)
            cur = det[j].copy()
            for wi, w in enumerate(plan.windows):
                synd = cur[w.row0:w.row1]
                o = oc(w.mat, channel_probs=w.prior, **kw)
                e_o = np.asarray(o.decode(synd))
                dv = bpgdg_decoder(w.mat, channel_probs=w.prior, **kw)
                e_d = np.asarray(dv.decode_batch(synd[None, :]))[0]
                info = o.ensemble_info() if o._res.exit_class != 0 else None
                print(f"  shot {j} window {wi}: oracle exit {o._res.exit_class} conv {o.converge} pm {o.min_pm} info {None if info is None else (info[0].tolist(), info[1], info[2])} blocks {o.ensemble_blocks() if info else None} | "
                      f"single-window device == oracle: {np.array_equal(e_d, e_o)} stats {dv.last_stats[0].tolist()} pm {dv.last_min_pm[0]} | pipeline stats {dev.last_stats[j, wi].tolist()} pm {dev.last_min_pm[j, wi]} "
                      f"committed equal {np.array_equal(total[j, w.col0:w.col0 + w.commit], e_o[:w.commit])}")
                if not np.array_equal(e_d, e_o) and os.environ.get("SWD_FUZZ_DUMP"):
                    import json
                    m_ = sp.csr_matrix(w.mat)
                    np.savez(os.environ["SWD_FUZZ_DUMP"], indptr=m_.indptr, indices=m_.indices, shape=np.array(m_.shape), prior=w.prior, synd=synd,
                             kw=np.array(json.dumps(kw)))
                tot[w.col0:w.col0 + w.commit] = e_o[:w.commit]
                cur = ((det[j] + (sp.csr_matrix(tot[None, :]) @ chk_t).toarray()[0]) % 2).astype(np.uint8)
print(f"{trials} trials, {bad} mismatching" + (f" ({refused} windows beyond one CU's LDS refused by the guessing decoder)" if refused else ""))
sys.exit(1 if bad else 0)
