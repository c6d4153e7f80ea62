#!/usr/bin/env python3
"""Generate the golden parity vectors in tests/golden/ from the REFERENCE ITSELF.

Runs only in the build container (needs /root/reference); the GPU box only ever sees the
.npz files this script wrote.  The reference's compiled Cython extension is the thing
being recorded -- this script never copies reference source into the repo.

Recipe for making the reference importable here (Python 3.10, Cython 3.2, numpy 2.2); the
two sed edits touch declarations only, never arithmetic:

    cp -r /root/reference /tmp/refbuild && chmod -R u+w /tmp/refbuild && cd /tmp/refbuild
    sed -i 's/np\\.int_t/np.int64_t/g' src/*.pyx src/*.pxd
    sed -i 's/= long(/= int(/g' src/bp4_osd.pyx src/osd_window.pyx
    rm -f src/bp4_osd.cpp src/osd_window.cpp src/bp_guessing_decoder.cpp src/mod2sparse.c
    python3 setup.py build_ext --inplace

Inputs (check matrices, priors, syndromes) come from this repo's own host-side code
(slidingwindowdecoder_amd.codes / circuit / windows) and are stored in the fixture, so the
expected outputs are tied to exactly the matrices the tests feed to the oracle and to the
HIP path.

Fixture layout (all .npz, bits packed with np.packbits(axis=-1)):
  graph:   indptr, indices, shape, priors         (CSR of the check matrix)
  params:  json string of the constructor kwargs
  synd, out          packed syndromes / returned vectors, one row per decode, in call order
  converge, bp_iteration, min_pm
  hist_hash          blake2b-64 of the raw bytes of ``log_prob_ratios`` after each decode
                     (the reference object keeps its history between decodes, so this is the
                     sequential, stateful value)
  hist_idx, hist     full n x 4 history for a few decodes (same stateful semantics)
"""
from __future__ import annotations

import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFBUILD = os.environ.get("SWD_REFBUILD", "/tmp/refbuild")
sys.path.insert(0, ROOT)


def ensure_reference():
    if not os.path.exists(os.path.join(REFBUILD, "src")) or not any(
            f.startswith("osd_window.") and f.endswith(".so") for f in os.listdir(os.path.join(REFBUILD, "src"))):
        sh = f"""
        set -e
        rm -rf {REFBUILD} && cp -r /root/reference {REFBUILD} && chmod -R u+w {REFBUILD} && cd {REFBUILD}
        sed -i 's/np\\.int_t/np.int64_t/g' src/*.pyx src/*.pxd
        sed -i 's/= long(/= int(/g' src/bp4_osd.pyx src/osd_window.pyx
        rm -f src/bp4_osd.cpp src/osd_window.cpp src/bp_guessing_decoder.cpp src/mod2sparse.c
        python3 setup.py build_ext --inplace > build.log 2>&1
        """
        subprocess.check_call(["bash", "-c", sh])
    sys.path.insert(0, REFBUILD)


def h64(a: np.ndarray) -> np.uint64:
    return np.frombuffer(hashlib.blake2b(np.ascontiguousarray(a).tobytes(), digest_size=8).digest(),
                         dtype=np.uint64)[0]


def pack(a):
    return np.packbits(np.asarray(a, dtype=np.uint8), axis=-1)


class Recorder:
    """Wraps a reference decoder object and records every decode."""

    def __init__(self, dec, n, hist_every=0, has_hist=True, extra=()):
        self.dec, self.n = dec, n
        self.synd, self.out, self.conv, self.iters, self.pm, self.hh = [], [], [], [], [], []
        self.hist_idx, self.hist = [], []
        self.hist_every, self.has_hist = hist_every, has_hist
        self.osd0 = []

    def decode(self, s):
        out = self.dec.decode(s)
        k = len(self.synd)
        self.synd.append(np.asarray(s, dtype=np.uint8))
        self.out.append(np.asarray(out, dtype=np.uint8))
        self.conv.append(int(self.dec.converge))
        if hasattr(self.dec, "bp_iteration"):
            self.iters.append(int(self.dec.bp_iteration))
            self.pm.append(float(self.dec.min_pm))
        if self.has_hist:
            lp = np.asarray(self.dec.log_prob_ratios)
            self.hh.append(h64(lp))
            if self.hist_every and k % self.hist_every == 0:
                self.hist_idx.append(k)
                self.hist.append(lp.copy())
        if hasattr(self.dec, "osd0_decoding"):
            self.osd0.append(np.asarray(self.dec.osd0_decoding, dtype=np.uint8))
        return out

    def arrays(self, prefix=""):
        d = {prefix + "synd": pack(np.array(self.synd)), prefix + "out": pack(np.array(self.out)),
             prefix + "converge": np.array(self.conv, dtype=np.uint8)}
        if self.iters:
            d[prefix + "bp_iteration"] = np.array(self.iters, dtype=np.int32)
            d[prefix + "min_pm"] = np.array(self.pm, dtype=np.float64)
        if self.hh:
            d[prefix + "hist_hash"] = np.array(self.hh, dtype=np.uint64)
        if self.hist:
            d[prefix + "hist_idx"] = np.array(self.hist_idx, dtype=np.int32)
            d[prefix + "hist"] = np.array(self.hist)
        if self.osd0:
            d[prefix + "osd0"] = pack(np.array(self.osd0))
        return d


def graph_arrays(mat, priors, prefix=""):
    a = sp.csr_matrix(mat)
    a.sort_indices()
    return {prefix + "indptr": a.indptr.astype(np.int32), prefix + "indices": a.indices.astype(np.int32),
            prefix + "shape": np.array(a.shape, dtype=np.int32), prefix + "priors": np.asarray(priors, np.float64)}


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.0f} KiB")


# ------------------------------------------------------------------------------------------
def gen_bb72(ref):
    """Config 1 ([[72,12,6]] code capacity) + harder noise to reach every exit class."""
    from slidingwindowdecoder_amd.codes import bb_code
    from src.codes_q import create_bivariate_bicycle_codes as ref_bb
    code, _, _ = bb_code(72)
    rcode, _, _ = ref_bb(6, 6, [3], [1, 2], [1, 2], [3])
    assert (rcode.hx == code.hx).all() and (rcode.hz == code.hz).all()
    hx = sp.csr_matrix(code.hx.astype(np.uint8))
    arrs = {"hx": code.hx, "hz": code.hz}
    rng = np.random.default_rng(72)
    sets = [
        ("c1", 0.005, dict(pre_max_iter=8, post_max_iter=100, ms_scaling_factor=1.0, osd_method="osd_0", osd_order=0), 500),
        ("osd0", 0.05, dict(pre_max_iter=8, post_max_iter=100, ms_scaling_factor=0.9, osd_method="osd_0", osd_order=0), 600),
        ("cs10", 0.06, dict(pre_max_iter=8, post_max_iter=100, ms_scaling_factor=0.9, osd_method="osd_cs", osd_order=10), 600),
        ("e6", 0.06, dict(pre_max_iter=8, post_max_iter=64, ms_scaling_factor=1.0, osd_method="osd_e", osd_order=6), 400),
        ("short", 0.06, dict(pre_max_iter=2, post_max_iter=3, ms_scaling_factor=1.0, osd_method="osd_cs", osd_order=4, new_n=60), 300),
    ]
    for tag, p, kw, shots in sets:
        priors = np.full(72, p)
        dec = Recorder(ref.osd_window(hx, channel_probs=priors, **kw), 72, hist_every=10)
        for _ in range(shots):
            e = (rng.random(72) < p).astype(np.uint8)
            dec.decode((hx @ e) % 2)
        arrs.update(dec.arrays(tag + "_"))
        arrs.update(graph_arrays(hx, priors, tag + "_"))
        arrs[tag + "_params"] = json.dumps(kw)
        conv = np.array(dec.conv)
        it = np.array(dec.iters)
        print(f"  bb72/{tag}: pre {int(((conv == 1) & (it <= kw['pre_max_iter'])).sum())} "
              f"post {int(((conv == 1) & (it > kw['pre_max_iter'])).sum())} osd {int((conv == 0).sum())}")
    # GDG / GD on the same code (single-thread = deterministic reference path)
    gkw = dict(max_iter=8, ms_scaling_factor=1.0, max_iter_per_step=6, max_step=25, max_tree_depth=3,
               max_side_depth=10, max_tree_branch_step=10, max_side_branch_step=10, gdg_factor=1.0,
               multi_thread=False, low_error_mode=False)
    priors = np.full(72, 0.06)
    for tag, cls, kw in (("gdg", ref.bpgdg_decoder, gkw),
                         ("gdg_low", ref.bpgdg_decoder, dict(gkw, low_error_mode=True, gdg_factor=0.625, ms_scaling_factor=0.625)),
                         ("gd", ref.bpgd_decoder, dict(max_iter=8, ms_scaling_factor=1.0, max_iter_per_step=6, max_step=25, gd_factor=1.0))):
        dec = Recorder(cls(hx, channel_probs=priors, **kw), 72, has_hist=False)
        for _ in range(600):
            e = (rng.random(72) < 0.06).astype(np.uint8)
            dec.decode((hx @ e) % 2)
        arrs.update(dec.arrays(tag + "_"))
        arrs.update(graph_arrays(hx, priors, tag + "_"))
        arrs[tag + "_params"] = json.dumps(kw)
        print(f"  bb72/{tag}: converge {int(np.sum(dec.conv))}/600")
    save("bb72_capacity.npz", **arrs)


def gen_bb144(ref, shots=192):
    """Configs 2/3: [[144,12,12]] circuit level p=0.003, (W,F)=(3,1), 12 rounds: the full
    sliding-window trace of the reference (osd.py:130-179 loop) for OSD-CS order 0 and 10,
    plus single-thread GDG."""
    from slidingwindowdecoder_amd.codes import bb_code
    from slidingwindowdecoder_amd.circuit import bb_dem
    from slidingwindowdecoder_amd.windows import plan_windows, sample_dem, sliding_window_decode_host, logical_error_stats
    code, A, B = bb_code(144)
    dem = bb_dem(code, A, B, 0.003, 12)
    plan = plan_windows(dem.chk, dem.obs, dem.priors, 72, 3, 1, method=1)
    det, obs, faults = sample_dem(plan.chk, plan.obs, plan.priors, shots, seed=20240318)
    arrs = {"det": pack(det), "obs_data": pack(obs), "num_shots": np.int32(shots)}
    arrs.update(graph_arrays(plan.chk, plan.priors, "chk_"))
    arrs.update(graph_arrays(plan.obs, plan.priors, "obs_"))
    arrs["anchors"] = np.array(plan.anchors, dtype=np.int32)
    arrs["noisy_prior"] = np.float64(plan.noisy_prior)
    for wi, w in enumerate(plan.windows):
        arrs.update(graph_arrays(w.mat, w.prior, f"win{wi}_"))
        arrs[f"win{wi}_meta"] = np.array([w.row0, w.row1, w.col0, w.ncols_global, w.commit, int(w.is_last)], dtype=np.int32)

    def run(tag, factory, has_hist):
        recs = []

        def fac(w):
            r = Recorder(factory(w), w.mat.shape[1], hist_every=0, has_hist=has_hist)
            recs.append(r)
            return r
        t = time.time()
        total, flagged = sliding_window_decode_host(plan, det, fac)
        fl, le = logical_error_stats(plan, det, obs, total)
        print(f"  bb144/{tag}: {time.time() - t:.1f}s flagged/window {flagged} final flagged {int(fl.sum())} logical {int(le.sum())}/{shots}")
        for wi, r in enumerate(recs):
            arrs.update(r.arrays(f"{tag}_win{wi}_"))
        arrs[f"{tag}_total"] = pack(total)
        arrs[f"{tag}_flagged_per_window"] = np.array(flagged, dtype=np.int32)
        arrs[f"{tag}_logical"] = le.astype(np.uint8)

    okw = dict(pre_max_iter=8, post_max_iter=200, ms_scaling_factor=1.0, new_n=None, osd_method="osd_cs")
    for order in (0, 10):
        kw = dict(okw, osd_order=order)
        arrs[f"osd{order}_params"] = json.dumps(kw)
        run(f"osd{order}", lambda w: ref.osd_window(w.mat, channel_probs=w.prior, **kw), True)
    gkw = dict(max_iter=8, max_iter_per_step=6, max_step=25, max_tree_depth=3, max_side_depth=10,
               max_tree_branch_step=10, max_side_branch_step=10, multi_thread=False, low_error_mode=False,
               gdg_factor=1.0, ms_scaling_factor=1.0)
    arrs["gdg_params"] = json.dumps(gkw)
    run("gdg", lambda w: ref.bpgdg_decoder(w.mat, channel_probs=w.prior, **gkw), False)

    # full LLR history for a few decodes of a mid window, one fresh reference object each
    w = plan.windows[5]
    cur = det[:, w.row0:w.row1]  # raw detectors of that window (not the committed residual): just inputs
    kw = dict(okw, osd_order=0)
    hs, hi, ho, hc, hit = [], [], [], [], []
    for j in range(24):
        d = ref.osd_window(w.mat, channel_probs=w.prior, **kw)
        out = d.decode(cur[j])
        hs.append(cur[j]); ho.append(np.asarray(out, np.uint8)); hi.append(np.asarray(d.log_prob_ratios))
        hc.append(int(d.converge)); hit.append(int(d.bp_iteration))
    arrs.update({"fresh_win": np.int32(5), "fresh_synd": pack(np.array(hs)), "fresh_out": pack(np.array(ho)),
                 "fresh_hist": np.array(hi), "fresh_converge": np.array(hc, np.uint8),
                 "fresh_bp_iteration": np.array(hit, np.int32)})

    # rank-deficient last window with random (mostly inconsistent) syndromes
    w = plan.windows[-1]
    rng = np.random.default_rng(210)
    for order in (0, 10):
        r = Recorder(ref.osd_window(w.mat, channel_probs=w.prior, **dict(okw, osd_order=order, post_max_iter=20)), w.mat.shape[1])
        for _ in range(60):
            r.decode((rng.random(w.mat.shape[0]) < 0.08).astype(np.uint8))
        arrs.update(r.arrays(f"incons{order}_"))
    arrs["incons_params"] = json.dumps(dict(okw, post_max_iter=20))
    save("bb144_circuit_p003_w3f1.npz", **arrs)


def gen_bb288(ref, shots=24):
    """Config 4: [[288,12,18]], (W,F)=(4,1), p=0.005, 6 rounds (Sliding Window OSD.ipynb cell
    with N=288) -- fewer shots, the reference needs ~16 ms per window here."""
    from slidingwindowdecoder_amd.codes import bb_code
    from slidingwindowdecoder_amd.circuit import bb_dem
    from slidingwindowdecoder_amd.windows import plan_windows, sample_dem, sliding_window_decode_host, logical_error_stats
    from src.codes_q import create_bivariate_bicycle_codes as ref_bb
    code, A, B = bb_code(288)
    rcode, _, _ = ref_bb(12, 12, [3], [2, 7], [1, 2], [3])
    assert (rcode.hx == code.hx).all() and (rcode.hz == code.hz).all()
    dem = bb_dem(code, A, B, 0.005, 6)
    plan = plan_windows(dem.chk, dem.obs, dem.priors, 144, 4, 1, method=1)
    det, obs, faults = sample_dem(plan.chk, plan.obs, plan.priors, shots, seed=288)
    arrs = {"det": pack(det), "obs_data": pack(obs), "num_shots": np.int32(shots), "hx": code.hx}
    arrs.update(graph_arrays(plan.chk, plan.priors, "chk_"))
    arrs["anchors"] = np.array(plan.anchors, dtype=np.int32)
    arrs["noisy_prior"] = np.float64(plan.noisy_prior)
    for wi, w in enumerate(plan.windows):
        arrs.update(graph_arrays(w.mat, w.prior, f"win{wi}_"))
        arrs[f"win{wi}_meta"] = np.array([w.row0, w.row1, w.col0, w.ncols_global, w.commit, int(w.is_last)], dtype=np.int32)
    kw = dict(pre_max_iter=8, post_max_iter=200, ms_scaling_factor=1.0, new_n=None, osd_method="osd_cs", osd_order=10)
    arrs["osd10_params"] = json.dumps(kw)
    recs = []

    def fac(w):
        r = Recorder(ref.osd_window(w.mat, channel_probs=w.prior, **kw), w.mat.shape[1])
        recs.append(r)
        return r
    t = time.time()
    total, flagged = sliding_window_decode_host(plan, det, fac)
    print(f"  bb288: {time.time() - t:.1f}s windows {[w.mat.shape for w in plan.windows]} flagged {flagged}")
    for wi, r in enumerate(recs):
        arrs.update(r.arrays(f"osd10_win{wi}_"))
    arrs["osd10_total"] = pack(total)
    save("bb288_circuit_p005_w4f1.npz", **arrs)


def gen_bb288_gdg(ref, shots=48):
    """The reference's [[288,12,18]] guessing-decoder run (`Sliding Window GDG.ipynb` cell 8 = guessing.py:160-197 with N = 288:
    p = 0.005, 6 rounds, (W,F) = (4,1), max_iter = 16, max_step = 60, max_tree_depth = 4, max_side_depth = 20,
    max_side_branch_step = max_tree_branch_step = 40 -> the 32-thread ensemble shape D4/S20) on the 576 x 4752 / 4896 windows, recorded
    from the deterministic single-thread gdg() (`multi_thread=False`, the parity target), plus the notebooks' default shape D3/S10
    on the same windows.  The matrices are those of bb288_circuit_p005_w4f1.npz (same generator, same seed for the shots)."""
    from slidingwindowdecoder_amd.codes import bb_code
    from slidingwindowdecoder_amd.circuit import bb_dem
    from slidingwindowdecoder_amd.windows import plan_windows, sample_dem, sliding_window_decode_host
    code, A, B = bb_code(288)
    dem = bb_dem(code, A, B, 0.005, 6)
    plan = plan_windows(dem.chk, dem.obs, dem.priors, 144, 4, 1, method=1)
    det, obs, faults = sample_dem(plan.chk, plan.obs, plan.priors, shots, seed=288)
    arrs = {"det": pack(det), "obs_data": pack(obs), "num_shots": np.int32(shots)}
    arrs.update(graph_arrays(plan.chk, plan.priors, "chk_"))
    arrs.update(graph_arrays(plan.obs, plan.priors, "obs_"))
    arrs["anchors"] = np.array(plan.anchors, dtype=np.int32)
    arrs["noisy_prior"] = np.float64(plan.noisy_prior)
    for wi, w in enumerate(plan.windows):
        arrs.update(graph_arrays(w.mat, w.prior, f"win{wi}_"))
        arrs[f"win{wi}_meta"] = np.array([w.row0, w.row1, w.col0, w.ncols_global, w.commit, int(w.is_last)], dtype=np.int32)
    base = dict(max_iter=16, max_iter_per_step=6, ms_scaling_factor=1.0, gdg_factor=1.0, multi_thread=False, low_error_mode=False)
    sets = [("d4s20", dict(base, max_step=60, max_tree_depth=4, max_side_depth=20, max_tree_branch_step=40, max_side_branch_step=40)),
            ("d3s10", dict(base, max_step=25, max_tree_depth=3, max_side_depth=10, max_tree_branch_step=10, max_side_branch_step=10))]
    for tag, kw in sets:
        arrs[tag + "_params"] = json.dumps(kw)
        recs = []

        def fac(w):
            r = Recorder(ref.bpgdg_decoder(w.mat, channel_probs=w.prior, **kw), w.mat.shape[1], has_hist=False)
            recs.append(r)
            return r
        t = time.time()
        total, flagged = sliding_window_decode_host(plan, det, fac)
        print(f"  bb288_gdg/{tag}: {time.time() - t:.1f}s flagged/window {flagged} converge "
              f"{[int(np.sum(r.conv)) for r in recs]} of {shots}")
        for wi, r in enumerate(recs):
            arrs.update(r.arrays(f"{tag}_win{wi}_"))
        arrs[f"{tag}_total"] = pack(total)
    save("bb288_gdg_p005_w4f1.npz", **arrs)


def gen_global144(ref, shots=288):
    """Row J: osd_window on the UN-windowed detector error model, as /root/reference/IBM.ipynb:119-135 configures it
    (`decode(..., shorten=True)`: pre_max_iter=16, post_max_iter=1000, new_n=None -> 2 x 936 columns kept, osd_cs order 10) on the
    [[144,12,12]] circuit of 12 rounds at p = 0.004 (936 x 8784, 30 672 edges: the messages of one shot do not fit a CU's LDS).
    One reference object decodes all shots in sequence, like the notebook's loop."""
    from slidingwindowdecoder_amd.codes import bb_code
    from slidingwindowdecoder_amd.circuit import bb_dem
    from slidingwindowdecoder_amd.windows import sample_dem
    code, A, B = bb_code(144)
    dem = bb_dem(code, A, B, 0.004, 12)
    chk = sp.csr_matrix(dem.chk)
    assert chk.shape == (936, 8784)
    det, obs, _ = sample_dem(chk, dem.obs, dem.priors, shots, seed=20240401)
    kw = dict(pre_max_iter=16, post_max_iter=1000, ms_scaling_factor=1.0, new_n=None, osd_method="osd_cs", osd_order=10)
    arrs = {"params": json.dumps(kw), "obs_data": pack(obs)}
    arrs.update(graph_arrays(chk, dem.priors, "chk_"))
    arrs.update(graph_arrays(sp.csr_matrix(dem.obs), dem.priors, "obs_"))
    t = time.time()
    rec = Recorder(ref.osd_window(chk, channel_probs=dem.priors, **kw), chk.shape[1], hist_every=0, has_hist=False)
    for j in range(shots):
        rec.decode(det[j])
    conv, it = np.array(rec.conv), np.array(rec.iters)
    print(f"  global144: {time.time() - t:.1f}s for {shots} shots; pre {int(((conv == 1) & (it <= 16)).sum())} "
          f"post {int(((conv == 1) & (it > 16)).sum())} osd {int((conv == 0).sum())}")
    arrs.update(rec.arrays("osd10_"))
    # a second, cheaper parameter set on the same syndromes: OSD order 0, post phase capped at 100 iterations (more OSD exits)
    kw2 = dict(kw, post_max_iter=100, osd_order=0)
    arrs["params_b"] = json.dumps(kw2)
    rec2 = Recorder(ref.osd_window(chk, channel_probs=dem.priors, **kw2), chk.shape[1], hist_every=0, has_hist=False)
    for j in range(shots):
        rec2.decode(det[j])
    print(f"  global144/b: osd exits {int((np.array(rec2.conv) == 0).sum())}")
    arrs.update(rec2.arrays("osd0_"))
    save("bb144_global_p004.npz", **arrs)


def gen_kat288(ref):
    """`Syndrome code.ipynb` cell 6: weight-2 syndromes of the [[288,12,18]] hx.  The notebook's
    stored output ((0,72) and (1,73) converge "with 14 VNs") is what the MULTI-thread reference
    prints; the deterministic single-thread path (the parity oracle for GDG) converges on more of
    them.  Both are recorded here from the reference itself."""
    from slidingwindowdecoder_amd.codes import bb_code
    from slidingwindowdecoder_amd import gf2
    code, _, _ = bb_code(288)
    span = gf2.Span()
    for v in gf2.rows_to_ints(code.hx.T):
        span.add(v)
    pairs = [(i, j) for i in range(144) for j in range(i + 1, 144) if span.reduce((1 << i) | (1 << j)) == 0]
    kw = dict(gdg_factor=0.625, max_step=40, max_tree_depth=4, max_side_depth=20, max_side_branch_step=30,
              max_tree_branch_step=30, low_error_mode=True, max_iter=8, ms_scaling_factor=0.625)
    res = {}
    for tag, mt in (("single", False), ("multi", True)):
        dec = ref.bpgdg_decoder(code.hx.astype(int), channel_probs=np.ones(288) * 0.01, multi_thread=mt, **kw)
        ok, outs = [], []
        for i, j in pairs:
            s = np.zeros(144)
            s[i] = s[j] = 1
            e = dec.decode(s)
            outs.append(np.asarray(e, np.uint8))
            if dec.converge:
                ok.append((i, j, int(e.sum())))
        res[tag] = np.array(ok, dtype=np.int32)
        if not mt:
            res["single_out"] = pack(np.array(outs))
        print(f"  kat288/{tag}: {len(ok)} converge: {ok[:4]}...")
    save("bb288_hx_wt2_kat.npz", pairs=np.array(pairs, np.int32), params=json.dumps(kw), **res)


def gen_bp4(ref):
    """bp4_osd (src/bp4_osd.pyx) on BB codes under depolarizing noise, the setting of Misc.ipynb cell 2
    (max_iter=100, ms_scaling_factor=0.625, osd_cs order 10) plus osd_e / osd_0 variants."""
    from slidingwindowdecoder_amd.codes import bb_code
    arrs = {}
    sets = [("bb72_cs10", 72, 0.08, dict(max_iter=32, ms_scaling_factor=0.625, osd_method="osd_cs", osd_order=10), 300),
            ("bb144_cs10", 144, 0.10, dict(max_iter=100, ms_scaling_factor=0.625, osd_method="osd_cs", osd_order=10), 300),
            ("bb72_e5", 72, 0.12, dict(max_iter=16, ms_scaling_factor=1.0, osd_method="osd_e", osd_order=5), 200),
            ("bb144_osd0", 144, 0.12, dict(max_iter=20, ms_scaling_factor=0.8, osd_method="osd_0", osd_order=0), 200)]
    for tag, N, p, kw, shots in sets:
        code, _, _ = bb_code(N)
        n = code.N
        px = py = pz = p / 3 * np.ones(n)
        dec = ref.bp4_osd(code.hx.astype(int), code.hz.astype(int), channel_probs_x=px, channel_probs_y=py,
                          channel_probs_z=pz, **kw)
        rng = np.random.default_rng(N + shots)
        sxs, szs, outs, conv, its, lprs, o0 = [], [], [], [], [], [], []
        for _ in range(shots):
            noise = rng.uniform(0, 1, n)
            err_z = np.logical_and(noise > px, noise < px + py + pz)
            err_x = noise < px + py
            sx = (err_z @ code.hx.T) % 2
            sz = (err_x @ code.hz.T) % 2
            out = dec.decode(sx, sz)
            sxs.append(sx); szs.append(sz); outs.append(np.asarray(out, np.uint8))
            conv.append(int(dec.converge)); its.append(int(dec.bp_iteration))
            lprs.append(np.asarray(dec.log_prob_ratios))
            o0.append(np.stack([dec.osd0_decoding_x, dec.osd0_decoding_z]).astype(np.uint8))
        arrs.update({tag + "_N": np.int32(N), tag + "_p": np.float64(p), tag + "_params": json.dumps(kw),
                     tag + "_sx": pack(np.array(sxs)), tag + "_sz": pack(np.array(szs)),
                     tag + "_out": pack(np.array(outs)), tag + "_osd0": pack(np.array(o0)),
                     tag + "_converge": np.array(conv, np.uint8), tag + "_bp_iteration": np.array(its, np.int32),
                     tag + "_lpr": np.array(lprs[:64])})
        print(f"  bp4/{tag}: converge {sum(conv)}/{shots}")
    save("bp4_depolarizing.npz", **arrs)


def gen_bp4_unequal_ranks(ref):
    """bp4_osd with rank(Hx) > rank(Hz) and a higher-order sweep: the reference sizes BOTH sweeps with kx = n - rank_x
    (bp4_osd.pyx:103-104, :284) -- well defined in this direction (the z-basis sweep simply walks fewer candidate columns); with
    rank(Hx) < rank(Hz) it reads past its column array.  Random ragged matrices, unequal X / Y / Z priors."""
    rng = np.random.default_rng(2718)
    arrs = {}
    for tag, mx, mz, n, kw in [("cs3", 14, 12, 40, dict(max_iter=6, ms_scaling_factor=0.8, osd_method="osd_cs", osd_order=3)),
                               ("e4", 16, 13, 48, dict(max_iter=5, ms_scaling_factor=1.0, osd_method="osd_e", osd_order=4))]:
        def rand_h(m):
            H = np.zeros((m, n), np.uint8)
            for c in range(n):
                H[rng.choice(m, size=3, replace=False), c] = 1
            return H
        Hx, Hz = rand_h(mx), rand_h(mz)
        Hz[mz - 1] = Hz[0] ^ Hz[1]  # a redundant check: rank(Hz) < mz <= rank(Hx)
        px, py, pz = rng.uniform(0.01, 0.04, n), rng.uniform(0.01, 0.04, n), rng.uniform(0.01, 0.04, n)
        dec = ref.bp4_osd(Hx.astype(int), Hz.astype(int), channel_probs_x=px, channel_probs_y=py, channel_probs_z=pz, **kw)
        sxs, szs, outs, conv, its = [], [], [], [], []
        for _ in range(250):
            pa = rng.choice(4, size=n, p=[0.85, 0.05, 0.05, 0.05])
            ex, ez = ((pa == 1) | (pa == 2)).astype(np.uint8), ((pa == 3) | (pa == 2)).astype(np.uint8)
            sx, sz = (Hx @ ez) % 2, (Hz @ ex) % 2
            out = dec.decode(sx, sz)
            sxs.append(sx); szs.append(sz); outs.append(np.asarray(out, np.uint8)); conv.append(int(dec.converge)); its.append(int(dec.bp_iteration))
        arrs.update({tag + "_hx": Hx, tag + "_hz": Hz, tag + "_px": px, tag + "_py": py, tag + "_pz": pz, tag + "_params": json.dumps(kw),
                     tag + "_sx": pack(np.array(sxs)), tag + "_sz": pack(np.array(szs)), tag + "_out": pack(np.array(outs)),
                     tag + "_converge": np.array(conv, np.uint8), tag + "_bp_iteration": np.array(its, np.int32)})
        print(f"  bp4_unequal/{tag}: converge {sum(conv)}/250")
    save("bp4_unequal_ranks.npz", **arrs)


def gen_bp4_camel(ref):
    """bp4_osd.camel_decode (src/bp4_osd.pyx:223-247, used by Misc.ipynb): one reference object per case, calls in
    sequence (the returned vectors persist in the object when no run converges)."""
    from slidingwindowdecoder_amd.codes import bb_code
    arrs = {}
    sets = [("bb72", 72, 0.06, dict(max_iter=32, ms_scaling_factor=0.625, osd_method="osd_0", osd_order=0), 200),
            ("bb144", 144, 0.08, dict(max_iter=50, ms_scaling_factor=0.8, osd_method="osd_0", osd_order=0), 150)]
    for tag, N, p, kw, shots in sets:
        code, _, _ = bb_code(N)
        n = code.N
        rng = np.random.default_rng(7 * N + 1)
        px, py, pz = (p / 3 * rng.uniform(0.5, 1.5, n) for _ in range(3))  # unequal priors: distinct path metrics
        dec = ref.bp4_osd(code.hx.astype(int), code.hz.astype(int), channel_probs_x=px, channel_probs_y=py,
                          channel_probs_z=pz, **kw)
        sxs, szs, outs, conv, its, pms = [], [], [], [], [], []
        for _ in range(shots):
            noise = rng.uniform(0, 1, n)
            err_z = np.logical_and(noise > px, noise < px + py + pz)
            err_x = noise < px + py
            sx = (err_z @ code.hx.T) % 2
            sz = (err_x @ code.hz.T) % 2
            out = dec.camel_decode(sx, sz)
            sxs.append(sx); szs.append(sz); outs.append(np.asarray(out, np.uint8))
            conv.append(int(dec.converge)); its.append(int(dec.bp_iteration)); pms.append(float(dec.min_pm))
        arrs.update({tag + "_N": np.int32(N), tag + "_params": json.dumps(kw), tag + "_px": px, tag + "_py": py, tag + "_pz": pz,
                     tag + "_sx": pack(np.array(sxs)), tag + "_sz": pack(np.array(szs)), tag + "_out": pack(np.array(outs)),
                     tag + "_converge": np.array(conv, np.uint8), tag + "_bp_iteration": np.array(its, np.int32),
                     tag + "_min_pm": np.array(pms, np.float64)})
        print(f"  bp4_camel/{tag}: converge {sum(conv)}/{shots}")
    save("bp4_camel.npz", **arrs)


def gen_bp4_shyps(ref):
    """bp4_osd on the SHYPS r=3 stabiliser matrices S_X = H^T (x) G, S_Z = G (x) H^T (src/build_SHYPS_circuit.py:37-45;
    21 x 49 each, column weight up to 9) under depolarizing code-capacity noise -- BASELINE config 5's decoder on
    config 5's code, the setting in which the reference itself can run BP4 (it has no circuit-level BP4)."""
    from slidingwindowdecoder_amd import shyps
    SX, SZ = shyps.shyps_stabilizers(3)
    n = SX.shape[1]
    arrs = {}
    arrs.update(graph_arrays(sp.csr_matrix(SX), np.zeros(n), "sx_"))
    arrs.update(graph_arrays(sp.csr_matrix(SZ), np.zeros(n), "sz_"))
    sets = [("osd0_p02", 0.02, dict(max_iter=32, ms_scaling_factor=0.625, osd_method="osd_0", osd_order=0), 400),
            ("osd0_p05", 0.05, dict(max_iter=32, ms_scaling_factor=1.0, osd_method="osd_0", osd_order=0), 400),
            ("cs10_p02", 0.02, dict(max_iter=32, ms_scaling_factor=0.625, osd_method="osd_cs", osd_order=10), 400),
            ("cs10_p05", 0.05, dict(max_iter=20, ms_scaling_factor=0.75, osd_method="osd_cs", osd_order=10), 400)]
    for tag, p, kw, shots in sets:
        px = py = pz = p / 3 * np.ones(n)
        dec = ref.bp4_osd(SX.astype(int), SZ.astype(int), channel_probs_x=px, channel_probs_y=py, channel_probs_z=pz, **kw)
        rng = np.random.default_rng(int(p * 1000) + shots + len(tag))
        sxs, szs, outs, conv, its, lprs, o0 = [], [], [], [], [], [], []
        for k in range(shots):
            noise = rng.uniform(0, 1, n)
            sc = 1.0 if k % 4 else 2.5                       # every fourth shot is heavier: more OSD exits
            err_z = np.logical_and(noise > sc * px, noise < sc * (px + py + pz))
            err_x = noise < sc * (px + py)
            sx = (err_z @ SX.T) % 2
            sz = (err_x @ SZ.T) % 2
            out = dec.decode(sx, sz)
            sxs.append(sx); szs.append(sz); outs.append(np.asarray(out, np.uint8))
            conv.append(int(dec.converge)); its.append(int(dec.bp_iteration))
            lprs.append(np.asarray(dec.log_prob_ratios))
            o0.append(np.stack([dec.osd0_decoding_x, dec.osd0_decoding_z]).astype(np.uint8))
        arrs.update({tag + "_p": np.float64(p), tag + "_params": json.dumps(kw),
                     tag + "_sx": pack(np.array(sxs)), tag + "_sz": pack(np.array(szs)),
                     tag + "_out": pack(np.array(outs)), tag + "_osd0": pack(np.array(o0)),
                     tag + "_converge": np.array(conv, np.uint8), tag + "_bp_iteration": np.array(its, np.int32),
                     tag + "_lpr": np.array(lprs)})
        print(f"  bp4_shyps/{tag}: converge {sum(conv)}/{shots}")
    save("bp4_shyps.npz", **arrs)


def main():
    ensure_reference()
    import src as ref
    which = sys.argv[1:] or ["bb72", "bb144", "bb288", "bb288_gdg", "kat288", "bp4", "camel", "bp4_shyps", "global144", "bp4_unequal"]
    if "bb72" in which:
        gen_bb72(ref)
    if "bb144" in which:
        gen_bb144(ref)
    if "bb288" in which:
        gen_bb288(ref)
    if "bb288_gdg" in which:
        gen_bb288_gdg(ref)
    if "kat288" in which:
        gen_kat288(ref)
    if "global144" in which:
        gen_global144(ref)
    if "bp4" in which:
        gen_bp4(ref)
    if "camel" in which:
        gen_bp4_camel(ref)
    if "bp4_shyps" in which:
        gen_bp4_shyps(ref)
    if "bp4_unequal" in which:
        gen_bp4_unequal_ranks(ref)


if __name__ == "__main__":
    main()
