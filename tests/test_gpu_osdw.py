"""GPU parity tests of the HIP osd_window path (through the C ABI) against
  (1) the golden vectors recorded from the reference extension, and
  (2) the CPU oracle on the same seeded inputs.
Everything integer is compared bit-for-bit; min_pm and the LLR history are compared with ==
(the north star allows 1e-5 relative for LLRs; the design goal is exact equality)."""
import numpy as np
import pytest

from tests import fixtures as fx

pytestmark = pytest.mark.gpu


def _dev_cls():
    from slidingwindowdecoder_amd import osd_window
    return osd_window


def _oracle():
    from oracle import oracle as O
    return O


def check_batch_vs_trace(dec, tr, kw_pre):
    out = dec.decode_batch(tr.synd)
    bad = np.flatnonzero((out != tr.out).any(axis=1))
    assert bad.size == 0, f"{bad.size} vectors differ, first {bad[:5]}"
    assert np.array_equal((dec.last_status & 0x100) != 0, tr.converge != 0)
    assert np.array_equal(dec.last_iterations, tr.bp_iteration)
    assert np.array_equal(dec.last_min_pm, tr.min_pm)


@pytest.mark.parametrize("tag", ["c1", "osd0", "cs10", "e6", "short"])
def test_bb72_sequential_decode_matches_reference(tag):
    """decode() one at a time on one object: also reproduces the reference object's stateful LLR
    history (hash after every decode)."""
    f = fx.load("bb72_capacity.npz")
    mat, priors = fx.graph(f, tag + "_")
    kw = fx.params(f, tag + "_params")
    dec = _dev_cls()(mat, channel_probs=priors, **kw)
    tr = fx.Trace(f, tag + "_", *mat.shape)
    for k in range(min(len(tr), 200)):
        out = dec.decode(tr.synd[k])
        assert (out == tr.out[k]).all(), f"decode {k}"
        assert dec.converge == tr.converge[k] and dec.bp_iteration == tr.bp_iteration[k]
        assert dec.min_pm == tr.min_pm[k]
        assert fx.h64(dec.log_prob_ratios) == tr.hist_hash[k], f"decode {k}: LLR history"
        if dec.exit_class == 2:
            assert (dec.osd0_decoding == tr.osd0[k]).all()


@pytest.mark.parametrize("tag", ["c1", "osd0", "cs10", "e6"])
def test_bb72_batch_matches_reference(tag):
    f = fx.load("bb72_capacity.npz")
    mat, priors = fx.graph(f, tag + "_")
    kw = fx.params(f, tag + "_params")
    dec = _dev_cls()(mat, channel_probs=priors, **kw)
    check_batch_vs_trace(dec, fx.Trace(f, tag + "_", *mat.shape), kw)


@pytest.mark.parametrize("tag", ["osd0", "osd10"])
def test_bb144_sliding_trace_batch(tag):
    """Config 2 shape: every window of the recorded reference sliding-window run (OSD-CS order 0 and
    the notebooks' default order 10), 192 shots per window in one launch."""
    f = fx.load("bb144_circuit_p003_w3f1.npz")
    kw = fx.params(f, tag + "_params")
    classes = np.zeros(6, int)
    for wi in range(11):
        mat, priors = fx.graph(f, f"win{wi}_")
        dec = _dev_cls()(mat, channel_probs=priors, **kw)
        tr = fx.Trace(f, f"{tag}_win{wi}_", *mat.shape)
        check_batch_vs_trace(dec, tr, kw)
        classes += np.bincount(dec.last_status & 0xFF, minlength=6)
    assert classes[0] > 500 and classes[1] > 300 and classes[2] > 50, classes


def test_bb144_history_values():
    f = fx.load("bb144_circuit_p003_w3f1.npz")
    wi = int(f["fresh_win"])
    mat, priors = fx.graph(f, f"win{wi}_")
    m, n = mat.shape
    dec = _dev_cls()(mat, channel_probs=priors, **fx.params(f, "osd0_params"))
    synd = fx.unpack(f["fresh_synd"], m)
    out = dec.decode_batch(synd, return_history=True)
    assert (out == fx.unpack(f["fresh_out"], n)).all()
    assert np.array_equal(dec.last_iterations, f["fresh_bp_iteration"])
    got = np.transpose(dec.last_history, (0, 2, 1))  # [B, n, 4]
    want = f["fresh_hist"]
    assert np.array_equal(got, want), f"max |diff| {np.abs(got - want).max()}"


@pytest.mark.parametrize("order", [0, 10])
def test_bb144_rank_deficient_inconsistent(order):
    f = fx.load("bb144_circuit_p003_w3f1.npz")
    mat, priors = fx.graph(f, "win10_")
    kw = dict(fx.params(f, "incons_params"), osd_order=order)
    dec = _dev_cls()(mat, channel_probs=priors, **kw)
    assert dec.rank == 210
    check_batch_vs_trace(dec, fx.Trace(f, f"incons{order}_", *mat.shape), kw)


def test_bb288_windows_match_reference():
    """Config 4 shape ([[288,12,18]] (4,1), 576 x 4896 windows, 16992 edges), OSD-CS order 10: the
    recorded reference run."""
    f = fx.load("bb288_circuit_p005_w4f1.npz")
    kw = fx.params(f, "osd10_params")
    for wi in range(4):
        mat, priors = fx.graph(f, f"win{wi}_")
        dec = _dev_cls()(mat, channel_probs=priors, **kw)
        check_batch_vs_trace(dec, fx.Trace(f, f"osd10_win{wi}_", *mat.shape), kw)


def test_bb288_windows_vs_oracle():
    """Same windows, OSD order 0, against the oracle on the recorded syndromes."""
    f = fx.load("bb288_circuit_p005_w4f1.npz")
    kw = dict(fx.params(f, "osd10_params"), osd_order=0)
    O = _oracle()
    for wi in range(4):
        mat, priors = fx.graph(f, f"win{wi}_")
        tr = fx.Trace(f, f"osd10_win{wi}_", *mat.shape)
        dec = _dev_cls()(mat, channel_probs=priors, **kw)
        ora = O.osd_window(mat, channel_probs=priors, **kw)
        out = dec.decode_batch(tr.synd)
        want, res = ora.decode_batch(tr.synd)
        assert (out == want).all()
        assert np.array_equal(dec.last_iterations, res["bp_iteration"])
        assert np.array_equal(dec.last_min_pm, res["min_pm"])
        assert np.array_equal(dec.last_status & 0xFF, res["exit_class"])


def test_random_small_codes_vs_oracle():
    """Ragged random matrices (uneven row/column weights, new_n < n, scaling factor != 1)."""
    rng = np.random.default_rng(11)
    O = _oracle()
    for trial in range(9):
        m, n = int(rng.integers(8, 21)), int(rng.integers(40, 200))
        H = (rng.random((m, n)) < 3.0 / m).astype(np.uint8)
        for c in range(n):
            if H[:, c].sum() == 0:
                H[rng.integers(m), c] = 1
        for r in range(m):
            if H[r].sum() == 0:
                H[r, rng.integers(n)] = 1
        if H.sum(axis=0).max() > 8:
            continue
        p = rng.uniform(0.01, 0.08, size=n)
        kw = dict(channel_probs=p, pre_max_iter=int(rng.integers(1, 9)), post_max_iter=int(rng.integers(1, 40)),
                  ms_scaling_factor=float(rng.choice([1.0, 0.9, 0.625])),
                  osd_method=["osd_0", "osd_cs", "osd_e"][trial % 3], osd_order=[0, 3, 4][trial % 3],
                  new_n=int(rng.integers(2 * m, n + 1)))
        dec, ora = _dev_cls()(H, **kw), O.osd_window(H, **kw)
        e = (rng.random((300, n)) < p).astype(np.uint8)
        synd = (e @ H.T) % 2
        out = dec.decode_batch(synd, return_history=True)
        want, res = ora.decode_batch(synd)
        assert (out == want).all(), f"trial {trial}"
        assert np.array_equal(dec.last_iterations, res["bp_iteration"])
        assert np.array_equal(dec.last_min_pm, res["min_pm"])
        assert np.array_equal(dec.last_status & 0xFF, res["exit_class"])


def test_failed_decimation_and_peel_exits_vs_oracle():
    """Random (mostly inconsistent) syndromes on small ragged codes with a short new_n: the "setting vn
    failed" and "peeling failed" exits (osd_window.pyx:179-186) leave an order-dependent partial result
    behind; the device replays the reference's order for them."""
    rng = np.random.default_rng(23)
    O = _oracle()
    seen = set()
    for trial in range(12):
        m, n = int(rng.integers(10, 25)), int(rng.integers(60, 220))
        H = (rng.random((m, n)) < 2.5 / m).astype(np.uint8)
        for c in range(n):
            if H[:, c].sum() == 0:
                H[rng.integers(m), c] = 1
        for r in range(m):
            if H[r].sum() == 0:
                H[r, rng.integers(n)] = 1
        if H.sum(axis=0).max() > 8 or H.sum(axis=1).max() > 60:
            continue
        p = rng.uniform(0.01, 0.08, size=n)
        kw = dict(channel_probs=p, pre_max_iter=int(rng.integers(1, 5)), post_max_iter=int(rng.integers(1, 20)),
                  ms_scaling_factor=1.0, osd_method="osd_0", osd_order=0, new_n=int(rng.integers(m, 2 * m)))
        dec, ora = _dev_cls()(H, **kw), O.osd_window(H, **kw)
        synd = (rng.random((400, m)) < 0.35).astype(np.uint8)
        out = dec.decode_batch(synd)
        want, res = ora.decode_batch(synd)
        assert (out == want).all(), f"trial {trial}"
        assert np.array_equal(dec.last_iterations, res["bp_iteration"])
        assert np.array_equal(dec.last_status & 0xFF, res["exit_class"])
        seen |= set(np.unique(res["exit_class"]).tolist())
    assert {3, 4} <= seen, seen


def test_constructor_and_decode_errors():
    f = fx.load("bb72_capacity.npz")
    mat, priors = fx.graph(f, "c1_")
    cls = _dev_cls()
    with pytest.raises(TypeError):
        cls([[1, 0], [0, 1]], channel_probs=[0.1, 0.1])
    with pytest.raises(ValueError):
        cls(mat, channel_probs=priors[:-1])
    with pytest.raises(ValueError):
        cls(mat, channel_probs=priors, osd_method="nope")
    with pytest.raises(ValueError):
        cls(mat, channel_probs=priors, osd_method="osd_cs", osd_order=43)
    d = cls(mat, channel_probs=priors)
    with pytest.raises(ValueError):
        d.decode(np.zeros(35))
    # float 0.0/1.0 syndromes are accepted like the reference's (char) cast (osd.py:165,178)
    out = d.decode(np.zeros(36, dtype=np.float64))
    assert out.dtype == np.int64 and not out.any()


def test_fuzz_medium_codes_vs_oracle():
    """Randomised matrices with 65..120 checks (the 256-thread kernel variants: a check is served by a thread of
    another wave than the one that initialised it) -- tests/fuzz_vs_oracle.py with a fixed seed."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_vs_oracle.py"), "40", "7"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_fuzz_large_m_column_form_elimination_vs_oracle():
    """Randomised matrices with 260..575 checks and 2100..3000 columns: the 1024-thread variants, whose OSD elimination
    runs in column form (`osd0_cols`: transform matrix in registers of nine waves, pivots resolved by wave 0, row
    operations through an LDS ring) -- rank-deficient matrices, inconsistent syndromes, all OSD methods."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_vs_oracle.py"), "24", "5", "260", "576", "osdw", "2100", "3000"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("seed", [7000, 7001])
def test_v7816_osd_exits_vs_oracle(seed):
    """The matrices of the fuzz campaign that exposed the round-3 miscompile (seeds 7000 / 7001, 121..300 checks: the
    <256, 7, 8, 16> kernel returned a wrong vector on every OSD exit when built with -structurizecfg-skip-uniform-regions and the
    explicitly scalar loop values; docs/history/DESIGN_rounds_1-5.md section 8) -- the same trials, deterministic, in the suite.  Fails if a compiler or
    flag change brings the wrong OSD ordering back."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_vs_oracle.py"), "30", str(seed), "121", "300", "osdw"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert int(last.split("osd_window comparisons")[1].split("shots through OSD")[0].split(",")[-1]) > 500, last  # OSD exits were compared


@pytest.mark.parametrize("mode", ["gd", "gdg", "bp"])
def test_fuzz_guessing_decoders_vs_oracle(mode):
    """Randomised matrices and parameters for bpgd_decoder / bpgdg_decoder / bp_history_decoder, including
    max_iter_per_step < 4 where BPGD's 4-slot history (bpgd.cpp:357-358) is only partly written per block."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_vs_oracle.py"), "30", "5", "6", "200", mode],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
