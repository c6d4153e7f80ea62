"""GPU parity of the guessing decoders (bpgdg single-thread gdg(), bpgd, plain BP) against the
reference's recorded runs and the oracle."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests import fixtures as fx

pytestmark = pytest.mark.gpu


def _cls(tag):
    import slidingwindowdecoder_amd as S
    return {"gdg": S.bpgdg_decoder, "gdg_low": S.bpgdg_decoder, "gd": S.bpgd_decoder}[tag]


@pytest.mark.parametrize("tag", ["gdg", "gdg_low", "gd"])
def test_bb72_guessing_batch(tag):
    f = fx.load("bb72_capacity.npz")
    mat, priors = fx.graph(f, tag + "_")
    kw = fx.params(f, tag + "_params")
    kw.pop("multi_thread", None)
    dec = _cls(tag)(mat, channel_probs=priors, **kw)
    tr = fx.Trace(f, tag + "_", *mat.shape)
    out = dec.decode_batch(tr.synd)
    bad = np.flatnonzero((out != tr.out).any(axis=1))
    assert bad.size == 0, f"{bad.size}/{len(tr)} vectors differ, first {bad[:8]}"
    assert np.array_equal((dec.last_status & 0x100) != 0, tr.converge != 0)


def test_bb72_guessing_single_decode():
    f = fx.load("bb72_capacity.npz")
    mat, priors = fx.graph(f, "gdg_")
    kw = fx.params(f, "gdg_params")
    kw.pop("multi_thread", None)
    dec = _cls("gdg")(mat, channel_probs=priors, **kw)
    tr = fx.Trace(f, "gdg_", *mat.shape)
    for k in range(60):
        out = dec.decode(tr.synd[k])
        assert out.dtype == np.int64 and (out == tr.out[k]).all()
        assert dec.converge == bool(tr.converge[k])


def test_bb144_gdg_windows():
    """Config 3 shape: every window of the recorded single-thread GDG sliding run."""
    import slidingwindowdecoder_amd as S
    f = fx.load("bb144_circuit_p003_w3f1.npz")
    kw = fx.params(f, "gdg_params")
    kw.pop("multi_thread")
    nconv = 0
    for wi in range(11):
        mat, priors = fx.graph(f, f"win{wi}_")
        dec = S.bpgdg_decoder(mat, channel_probs=priors, **kw)
        tr = fx.Trace(f, f"gdg_win{wi}_", *mat.shape)
        out = dec.decode_batch(tr.synd)
        bad = np.flatnonzero((out != tr.out).any(axis=1))
        assert bad.size == 0, f"window {wi}: {bad.size} vectors differ, first {bad[:8]}"
        assert np.array_equal((dec.last_status & 0x100) != 0, tr.converge != 0)
        nconv += int((dec.last_status & 0xFF == 1).sum())
    assert nconv > 300  # the decimation search really ran


def test_bb144_gdg_pipeline():
    import slidingwindowdecoder_amd as S
    from tests.test_gpu_pipeline import load_plan
    f = fx.load("bb144_circuit_p003_w3f1.npz")
    plan = load_plan(f, 11)
    kw = fx.params(f, "gdg_params")
    kw.pop("multi_thread")
    det = fx.unpack(f["det"], plan.chk.shape[0])
    dec = S.SlidingWindowDecoder(plan, decoder="bpgdg_decoder", **kw)
    total = dec.decode(det)
    want = fx.unpack(f["gdg_total"], plan.chk.shape[1])
    bad = np.flatnonzero((total != want).any(axis=1))
    assert bad.size == 0, f"{bad.size} shots differ: {bad[:8]}"


def test_gdg_pipeline_more_shots_than_workgroups():
    """Launches with more shots than the persistent grid holds start their shots heaviest syndrome first (shot_order_kernel) --
    parallel form (work items) and serial form (tickets): every shot still gets the reference's recorded result."""
    import os
    import slidingwindowdecoder_amd as S
    from tests.test_gpu_pipeline import load_plan
    f = fx.load("bb144_circuit_p003_w3f1.npz")
    plan = load_plan(f, 11)
    kw = fx.params(f, "gdg_params")
    kw.pop("multi_thread")
    reps = 9  # 1728 shots: more than two (parallel form) or three (serial form) workgroups per CU x 256 CUs
    rng = np.random.default_rng(3)
    perm = rng.permutation(192 * reps)
    det = np.tile(fx.unpack(f["det"], plan.chk.shape[0]), (reps, 1))[perm]
    want = np.tile(fx.unpack(f["gdg_total"], plan.chk.shape[1]), (reps, 1))[perm]
    for serial in (False, True):
        if serial:
            os.environ["SWD_GDG_SERIAL"] = "1"
        try:
            dec = S.SlidingWindowDecoder(plan, decoder="bpgdg_decoder", **kw)
        finally:
            os.environ.pop("SWD_GDG_SERIAL", None)
        total = dec.decode(det)
        bad = np.flatnonzero((total != want).any(axis=1))
        assert bad.size == 0, f"serial={serial}: {bad.size} shots differ: {bad[:8]}"


@pytest.mark.parametrize("serial_min", [1, None])
def test_gdg_stream_batches_take_the_serial_walk_and_equal_the_recorded_run(serial_min, monkeypatch):
    """Batches of 3072 shots or more pushed through a stream object run the serial tree walk (the next batch's grid fills the tail;
    swd_osdw.hip launch()) -- here with the threshold at 1 shot and at its default: host-buffer stream with ragged batches,
    device-buffer stream with alternating lanes, and a decode() call large enough to be cut in two halves on the lanes -- every shot
    gets the reference's recorded result, and the per-window statistics equal the one-launch (work-item) form's."""
    if serial_min is not None:
        monkeypatch.setenv("SWD_GDG_STREAM_SERIAL_MIN", str(serial_min))
    import torch
    import slidingwindowdecoder_amd as S
    from tests.test_gpu_pipeline import load_plan
    f = fx.load("bb144_circuit_p003_w3f1.npz")
    plan = load_plan(f, 11)
    kw = fx.params(f, "gdg_params")
    kw.pop("multi_thread")
    det = fx.unpack(f["det"], plan.chk.shape[0])
    want = fx.unpack(f["gdg_total"], plan.chk.shape[1])
    dec = S.SlidingWindowDecoder(plan, decoder="bpgdg_decoder", **kw)
    assert np.array_equal(dec.decode(det), want)
    st0, pm0, fl0, fg0 = dec.last_stats[..., :7].copy(), dec.last_min_pm.copy(), dec.last_obs_flips.copy(), dec.last_flagged.copy()
    cuts = [0, 50, 51, 120, 192]  # (statistics word 7 is a scheduling diagnostic of the work-item form: include/swd.h)
    got = list(dec.decode_stream([det[a:b] for a, b in zip(cuts[:-1], cuts[1:])]))
    for (a, b), (tot, st, pm, flips, flagged) in zip(zip(cuts[:-1], cuts[1:]), got):
        assert np.array_equal(tot, want[a:b]) and np.array_equal(st[..., :7], st0[a:b]) and np.array_equal(pm, pm0[a:b])
        assert np.array_equal(flips, fl0[a:b]) and np.array_equal(flagged, fg0[a:b])  # (a guessing decoder may leave a window unsolved)
    reps = 11  # 2112 shots: decode() cuts the call in two halves on the lanes
    assert np.array_equal(dec.decode(np.tile(det, (reps, 1))), np.tile(want, (reps, 1)))
    assert np.array_equal(dec.last_stats[..., :7], np.tile(st0, (reps, 1, 1)))
    dev = torch.device("cuda", 0)
    nrep = 3 if serial_min else 17  # (default threshold: 3264 shots per batch, the serial walk by the product's own rule)
    d_t = torch.from_numpy(np.tile(det, (nrep, 1))).to(dev)
    ncol = plan.chk.shape[1]
    outs = [dict(total=torch.empty((len(d_t), ncol), dtype=torch.uint8, device=dev), stats=torch.empty((len(d_t), 11, 8), dtype=torch.int32, device=dev))
            for _ in range(2)]
    s = dec.stream(len(d_t))
    for i in range(4):
        s.push_device(d_t, **outs[i % 2])
    s.wait()
    for o in outs:
        assert np.array_equal(o["total"].cpu().numpy(), np.tile(want, (nrep, 1))) and np.array_equal(o["stats"].cpu().numpy()[..., :7], np.tile(st0, (nrep, 1, 1)))
    dec.check_status()
    s.close()
    # a caller's own two streams in turn (no stream object): a launch that finds the previous one still running on the other stream is
    # scheduled like a stream batch -- same records
    lanes = [torch.cuda.Stream(dev), torch.cuda.Stream(dev, priority=-1)]
    for ln in lanes:
        ln.wait_stream(torch.cuda.current_stream(dev))
    for i in range(6):
        dec.decode_device(d_t, stream=lanes[i % 2], **outs[i % 2])
    torch.cuda.synchronize()
    for o in outs:
        assert np.array_equal(o["total"].cpu().numpy(), np.tile(want, (nrep, 1))) and np.array_equal(o["stats"].cpu().numpy()[..., :7], np.tile(st0, (nrep, 1, 1)))
    dec.check_status()


@pytest.mark.parametrize("tag", ["d4s20", "d3s10"])
def test_bb288_gdg_windows_and_pipeline(tag):
    """The reference's [[288,12,18]] guessing-decoder run (`Sliding Window GDG.ipynb` cell 8, guessing.py:160-197 with N = 288:
    (W,F) = (4,1) windows of 576 x 4752 / 4896, max_iter 16, max_step 60, D4 / S20, branch steps 40; and the default D3 / S10 on
    the same windows), recorded from the reference's single-thread gdg(): every window decode (vector, converge flag) and the whole
    window loop.  These windows take the large-graph kernels (1024 threads), which always run the SERIAL tree walk (Plan::finalize
    switches the work-item form off for them): both loop iterations below exercise that walk, once with and once without the
    environment switch that forces it -- the parallel form is covered on the [[144]] windows (test_bb144_gdg_windows)."""
    import os
    import slidingwindowdecoder_amd as S
    from tests.test_gpu_pipeline import load_plan
    f = fx.load("bb288_gdg_p005_w4f1.npz")
    kw = fx.params(f, tag + "_params")
    kw.pop("multi_thread")
    post = 0
    for wi in range(4):
        mat, priors = fx.graph(f, f"win{wi}_")
        dec = S.bpgdg_decoder(mat, channel_probs=priors, **kw)
        tr = fx.Trace(f, f"{tag}_win{wi}_", *mat.shape)
        out = dec.decode_batch(tr.synd)
        bad = np.flatnonzero((out != tr.out).any(axis=1))
        assert bad.size == 0, f"window {wi}: {bad.size} vectors differ, first {bad[:8]}"
        assert np.array_equal((dec.last_status & 0x100) != 0, tr.converge != 0)
        post += int(((dec.last_status & 0xFF) == 1).sum())
    assert post > 60  # the decimation search really ran
    plan = load_plan(f, 4)
    det = fx.unpack(f["det"], plan.chk.shape[0])
    want = fx.unpack(f[tag + "_total"], plan.chk.shape[1])
    for serial in (False, True):
        if serial:
            os.environ["SWD_GDG_SERIAL"] = "1"
        try:
            pl = S.SlidingWindowDecoder(plan, decoder="bpgdg_decoder", **kw)
        finally:
            os.environ.pop("SWD_GDG_SERIAL", None)
        assert pl.threads == 1024
        total = pl.decode(det)
        bad = np.flatnonzero((total != want).any(axis=1))
        assert bad.size == 0, f"serial={serial}: {bad.size} shots differ: {bad[:8]}"
        pl.check_status()


def test_threaded_ensemble_bb288_windows_vs_oracle():
    """bpgdg_decoder(multi_thread=True) with the reference's [[288,12,18]] shape D4 / S20 (32 threads, the notebook's own call) on
    the (4,1) windows: device against the oracle's ensemble (pinned to the reference's real threads on these windows in
    tests/test_oracle_vs_ref.py::test_threaded_ensemble_bb288_window_d4s20) on every shot, single windows and the window loop."""
    import slidingwindowdecoder_amd as S
    from oracle import oracle as O
    from slidingwindowdecoder_amd.windows import sliding_window_decode_host
    from tests.test_gpu_pipeline import load_plan
    f = fx.load("bb288_gdg_p005_w4f1.npz")
    kw = fx.params(f, "d4s20_params")
    kw.pop("multi_thread")
    for wi in (1, 3):
        mat, priors = fx.graph(f, f"win{wi}_")
        tr = fx.Trace(f, f"d4s20_win{wi}_", *mat.shape)
        post, ties = _ensemble_vs_oracle(mat, priors, kw, tr.synd[:32], 10)
        print(f"bb288 window {wi}: {post} ensembles, {ties} with a tied different vector")
    plan = load_plan(f, 4)
    det = fx.unpack(f["det"], plan.chk.shape[0])[:24]

    class Fresh:
        def __init__(self, w):
            self.w = w

        def decode(self, s):
            self.d = O.bpgdg_decoder(self.w.mat, channel_probs=self.w.prior, multi_thread=True, **kw)
            return self.d.decode(s)

    want, _ = sliding_window_decode_host(plan, det, Fresh)
    pl = S.SlidingWindowDecoder(plan, decoder="bpgdg_decoder", multi_thread=True, **kw)
    total = pl.decode(det)
    bad = np.flatnonzero((total != want).any(axis=1))
    assert bad.size == 0, f"{bad.size} shots differ from the oracle host loop: {bad[:8]}"
    pl.check_status()


def test_bb288_weight_two_kat():
    """Syndrome code.ipynb cell 6 (see tests/test_kat_syndrome_code.py): the device follows the
    deterministic single-thread search and reproduces its 22 converging syndromes vector for vector."""
    import slidingwindowdecoder_amd as S
    from slidingwindowdecoder_amd.codes import bb_code
    f = fx.load("bb288_hx_wt2_kat.npz")
    code, _, _ = bb_code(288)
    pairs = f["pairs"]
    synd = np.zeros((len(pairs), 144), np.uint8)
    for k, (i, j) in enumerate(pairs):
        synd[k, i] = synd[k, j] = 1
    dec = S.bpgdg_decoder(code.hx, channel_probs=np.ones(288) * 0.01, **fx.params(f, "params"))
    out = dec.decode_batch(synd)
    assert np.array_equal(out, fx.unpack(f["single_out"], 288))
    conv = (dec.last_status & 0x100) != 0
    got = [[int(i), int(j), int(out[k].sum())] for k, (i, j) in enumerate(pairs) if conv[k]]
    assert got == f["single"].tolist()
    assert [0, 72, 14] in got and [1, 73, 14] in got


def test_random_codes_vs_oracle():
    from oracle import oracle as O
    import slidingwindowdecoder_amd as S
    rng = np.random.default_rng(5)
    for trial in range(6):
        m, n = int(rng.integers(8, 21)), int(rng.integers(40, 160))
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
        kw = dict(channel_probs=p, max_iter=int(rng.integers(4, 9)), ms_scaling_factor=float(rng.choice([1.0, 0.625])),
                  max_iter_per_step=int(rng.choice([4, 6, 8])), max_step=int(rng.integers(5, 25)),
                  max_tree_depth=int(rng.integers(1, 4)), max_side_depth=int(rng.integers(4, 12)),
                  max_side_branch_step=int(rng.integers(3, 12)), gdg_factor=float(rng.choice([1.0, 0.625])),
                  low_error_mode=bool(trial % 2), new_n=int(rng.integers(2 * m, n + 1)))
        e = (rng.random((200, n)) < p).astype(np.uint8)
        synd = (e @ H.T) % 2
        for cls_d, cls_o in ((S.bpgdg_decoder, O.bpgdg_decoder), (S.bpgd_decoder, O.bpgd_decoder)):
            dec, ora = cls_d(H, **kw), cls_o(H, **kw)
            out = dec.decode_batch(synd)
            for k in range(synd.shape[0]):
                want = ora.decode(synd[k])
                assert (out[k] == want).all(), f"trial {trial} shot {k} {cls_d.__name__}"
                assert bool(dec.last_status[k] & 0x100) == bool(ora.converge)


def _both_forms(make, synd):
    """the same decoder built in the serial form (SWD_GDG_SERIAL=1 at construction) and in the parallel form"""
    import os
    os.environ["SWD_GDG_SERIAL"] = "1"
    try:
        ser = make()
    finally:
        del os.environ["SWD_GDG_SERIAL"]
    par = make()
    out_s = ser.decode_batch(synd)
    out_p = par.decode_batch(synd)
    return ser, par, out_s, out_p


def test_parallel_tree_search_equals_serial_in_every_record():
    """Side branches as tasks on other workgroups + replay of the serial bookkeeping (swd_gdg_kernel.h) against the
    one-workgroup serial walk: vectors, converge flags, path metrics and ALL statistics words (iterations, snapshots
    pushed, BP blocks run, min_converge_depth) have to be identical -- [[144]] mid and last windows of the recorded run,
    and the [[288]] weight-two setting (D = 4, S = 20, 30 steps per side branch)."""
    import slidingwindowdecoder_amd as S
    from slidingwindowdecoder_amd.codes import bb_code
    f = fx.load("bb144_circuit_p003_w3f1.npz")
    kw = fx.params(f, "gdg_params")
    kw.pop("multi_thread")
    hard = 0
    for wi in (0, 5, 10):
        mat, priors = fx.graph(f, f"win{wi}_")
        tr = fx.Trace(f, f"gdg_win{wi}_", *mat.shape)
        synd = np.tile(tr.synd, (3, 1))  # more units than one pass of the grid draws at once
        ser, par, out_s, out_p = _both_forms(lambda: S.bpgdg_decoder(mat, channel_probs=priors, **kw), synd)
        assert np.array_equal(out_s, out_p) and np.array_equal(out_p[:len(tr)], tr.out)
        # word 7 = snapshots the main branch queued as tasks: 0 in the serial form by definition
        assert np.array_equal(ser.last_stats[:, :7], par.last_stats[:, :7]), np.flatnonzero((ser.last_stats[:, :7] != par.last_stats[:, :7]).any(axis=1))[:8]
        assert np.array_equal(ser.last_min_pm, par.last_min_pm)
        assert not ser.last_stats[:, 7].any()
        hard += int((par.last_stats[:, 7] > 0).sum())
    assert hard > 50  # trees whose side branches ran as tasks on other workgroups
    k = fx.load("bb288_hx_wt2_kat.npz")
    code, _, _ = bb_code(288)
    pairs = k["pairs"]
    synd = np.zeros((len(pairs), 144), np.uint8)
    for q, (i, j) in enumerate(pairs):
        synd[q, i] = synd[q, j] = 1
    ser, par, out_s, out_p = _both_forms(lambda: S.bpgdg_decoder(code.hx, channel_probs=np.ones(288) * 0.01, **fx.params(k, "params")), synd)
    assert np.array_equal(out_s, out_p) and np.array_equal(ser.last_stats[:, :7], par.last_stats[:, :7]) and np.array_equal(ser.last_min_pm, par.last_min_pm)
    assert (par.last_stats[:, 4] > 20).any() and (par.last_stats[:, 7] > 0).any()  # deep trees, run as tasks


def test_hypothesis_ensemble_mode():
    """multi_thread=2 / hypotheses=64: this package's own ensemble over every leaf of gdg()'s tree (no reference counterpart).
    Properties: deterministic, never worse than the single-thread search in path metric, every converged answer reproduces
    the syndrome."""
    import slidingwindowdecoder_amd as S
    f = fx.load("bb144_circuit_p003_w3f1.npz")
    kw = fx.params(f, "gdg_params")
    kw.pop("multi_thread")
    mat, priors = fx.graph(f, "win5_")
    tr = fx.Trace(f, "gdg_win5_", *mat.shape)
    single = S.bpgdg_decoder(mat, channel_probs=priors, **kw)
    out1 = single.decode_batch(tr.synd)
    from slidingwindowdecoder_amd.decoders import hypotheses_shape
    assert hypotheses_shape(64) == (5, 6) and hypotheses_shape(32) == (4, 5) and hypotheses_shape(1) == (0, 0) and hypotheses_shape(100) == (5, 42)
    for h in (1, 2, 3, 7, 16, 64, 100, 161):  # leaves = 1 + (S - D) + 2 (2^D - 1)
        D, Sd = hypotheses_shape(h)
        assert 1 + (Sd - D) + 2 * (2 ** D - 1) == h and 0 <= D <= 6
    with pytest.raises(ValueError):
        hypotheses_shape(162)
    with pytest.raises(ValueError):
        hypotheses_shape(0)
    for extra in (dict(multi_thread=2), dict(hypotheses=64), dict(hypotheses=16), dict(hypotheses=7), dict(hypotheses=100), dict(hypotheses=2), dict(hypotheses=1)):
        ens = S.bpgdg_decoder(mat, channel_probs=priors, **dict(kw, **extra))
        out = ens.decode_batch(tr.synd)
        assert np.array_equal(out, ens.decode_batch(tr.synd))  # deterministic
        conv = (ens.last_status & 0x100) != 0
        H = mat.toarray().astype(np.int64)
        assert not ((out[conv].astype(np.int64) @ H.T + tr.synd[conv]) % 2).any()
        c1 = (single.last_status & 0x100) != 0
        if extra.get("hypotheses") == 16:  # D = 3, S = 4: a deeper ensemble of the same family can only find a smaller (or the same) metric
            ens16_pm, ens16_conv = ens.last_min_pm.copy(), conv.copy()
        if "multi_thread" in extra:  # same tree shape: a superset of the hypotheses the pruned search scores
            assert (conv | ~c1).all()
            both = conv & c1
            assert (ens.last_min_pm[both] <= single.last_min_pm[both]).all()


def _ensemble_vs_oracle(mat, priors, kw, synd, min_post=20):
    """device bpgdg_decoder(multi_thread=True) against the oracle's restatement of the threaded ensemble (which
    tests/test_oracle_vs_ref.py pins to the reference's real threads): vectors, converge flags and path metrics of EVERY shot --
    also the tied ones, the device breaks ties in the oracle's thread order -- plus the winner and the tie count per shot."""
    import slidingwindowdecoder_amd as S
    from oracle import oracle as O
    dev = S.bpgdg_decoder(mat, channel_probs=priors, multi_thread=True, **kw)
    ora = O.bpgdg_decoder(mat, channel_probs=priors, multi_thread=True, **kw)
    out = dev.decode_batch(synd)
    post = ties = 0
    for k in range(len(synd)):
        ora.clear_history()
        want = ora.decode(synd[k])
        assert np.array_equal(out[k], want), f"shot {k}: vectors differ"
        assert bool(dev.last_status[k] & 0x100) == bool(ora.converge), f"shot {k}: converge"
        if ora._res.exit_class == 0:
            continue
        pms, winner, nt = ora.ensemble_info()
        if (dev.last_status[k] & 0xFF) == 4:  # BPGD::reset failed (no ensemble ran)
            assert not want.any()
            continue
        post += 1
        ties += int(nt > 0)
        assert dev.last_min_pm[k] == ora.min_pm, f"shot {k}: min_pm {dev.last_min_pm[k]} vs {ora.min_pm}"
        assert dev.last_stats[k, 6] == winner and dev.last_stats[k, 7] == nt, f"shot {k}: winner / ties {dev.last_stats[k, 6:8]} vs {(winner, nt)}"
        # BP blocks: the device walks the prefix tree (a shared block runs once) but counts a block once per thread that would
        # have run it -- the number the oracle's thread-by-thread restatement counts
        assert dev.last_stats[k, 5] == ora.ensemble_blocks()[0], f"shot {k}: BP blocks {dev.last_stats[k, 5]} vs {ora.ensemble_blocks()[0]}"
    assert post >= min_post, post
    return post, ties


def test_threaded_ensemble_bb72_vs_oracle():
    f = fx.load("bb72_capacity.npz")
    mat, _ = fx.graph(f, "gdg_")
    rng = np.random.default_rng(23)
    priors = rng.uniform(0.03, 0.08, size=72)
    kw = dict(max_iter=8, ms_scaling_factor=1.0, max_iter_per_step=6, max_step=25, max_tree_depth=3, max_side_depth=10,
              max_tree_branch_step=10, max_side_branch_step=10, gdg_factor=1.0)
    H = mat.toarray().astype(np.int64)
    synd = np.array([(H @ (rng.random(72) < priors * 1.3).astype(np.int64) % 2) for _ in range(300)], dtype=np.uint8)
    post, ties = _ensemble_vs_oracle(mat, priors, kw, synd, 60)
    print(f"bb72: {post} ensembles, {ties} with a tied different vector")
    # other tree shapes and fewer iterations per block than history slots
    for D, S_, T in ((2, 5, 3), (1, 4, 6), (0, 3, 5)):
        kw2 = dict(kw, max_tree_depth=D, max_side_depth=S_, max_iter_per_step=T, max_step=12, max_tree_branch_step=4, max_side_branch_step=7,
                   gdg_factor=0.9, low_error_mode=bool(D == 1))
        _ensemble_vs_oracle(mat, priors, kw2, synd[:120], 20)


def test_threaded_ensemble_bb144_window_vs_oracle():
    """configs[2]'s window matrices and parameters (Sliding Window GDG.ipynb cell 3) with multi_thread=True"""
    f = fx.load("bb144_circuit_p003_w3f1.npz")
    kw = fx.params(f, "gdg_params")
    kw.pop("multi_thread")
    for wi in (0, 5, 10):
        mat, priors = fx.graph(f, f"win{wi}_")
        tr = fx.Trace(f, f"gdg_win{wi}_", *mat.shape)
        post, ties = _ensemble_vs_oracle(mat, priors, kw, tr.synd[:96], 15)
        print(f"bb144 window {wi}: {post} ensembles, {ties} with a tied different vector")


def test_threaded_ensemble_bb144_64_hypotheses_vs_oracle():
    """BASELINE configs[2] as written: "64 decimation hypotheses per shot" = max_tree_depth 5, max_side_depth 6 (main + 31 tree
    threads with two leaves each + one side thread) on the [[144,12,12]] (3,1) windows 0 / 5 / 10 of the recorded run.  Vector,
    converge flag, min_pm, winner and tie count of every shot against the oracle, whose restatement of this shape is pinned to
    the reference's real threads in tests/test_oracle_vs_ref.py::test_gdg_multi_64_hypotheses_bb144_window."""
    f = fx.load("bb144_circuit_p003_w3f1.npz")
    kw = fx.params(f, "gdg_params")
    kw.pop("multi_thread")
    kw.update(max_tree_depth=5, max_side_depth=6)
    for wi in (0, 5, 10):
        mat, priors = fx.graph(f, f"win{wi}_")
        tr = fx.Trace(f, f"gdg_win{wi}_", *mat.shape)
        post, ties = _ensemble_vs_oracle(mat, priors, kw, tr.synd[:96], 15)
        print(f"bb144 window {wi}, D = 5 / S = 6: {post} ensembles, {ties} with a tied different vector")


def test_ensemble_prefix_tree_shapes_vs_oracle():
    """The device walks the ensemble's prefix tree (shared BP blocks and scans run once, gdg_ensemble_tree); the oracle runs the
    thread bodies one after the other.  Shapes that stress the walk's bookkeeping: a tree deeper than max_step (the main thread's
    loop ends inside the shared part), D = 0 (no tree threads), D = 1, fewer iterations per block than history slots, tree threads
    without steps of their own (max_tree_branch_step = 0), low-error mode -- every record incl. the BP block count per shot."""
    f = fx.load("bb144_circuit_p003_w3f1.npz")
    base = fx.params(f, "gdg_params")
    base.pop("multi_thread")
    mat, priors = fx.graph(f, "win5_")
    tr = fx.Trace(f, "gdg_win5_", *mat.shape)
    shapes = [dict(max_tree_depth=4, max_side_depth=4, max_step=3, max_tree_branch_step=2),
              dict(max_tree_depth=0, max_side_depth=3),
              dict(max_tree_depth=1, max_side_depth=1, max_tree_branch_step=0),
              dict(max_tree_depth=2, max_side_depth=7, max_iter_per_step=3, low_error_mode=True),
              dict(max_tree_depth=6, max_side_depth=8, max_tree_branch_step=1, max_side_branch_step=2, max_step=8)]
    for sh in shapes:
        post, ties = _ensemble_vs_oracle(mat, priors, dict(base, **sh), tr.synd[:64], 8)
        print(f"{sh}: {post} ensembles, {ties} with a tied different vector")


def test_ensemble_main_thread_scan_after_an_early_converged_block():
    """tests/golden/ens_main_early_convergence.npz (found by tests/fuzz_pipeline.py, seed 13000): the main thread's block converges
    in its second iteration of six and the thread still runs its scan on it (bpgd.cpp:630-633) -- the history slots the block did not
    reach keep the previous block's values and no check is unmet.  The device has to record every iteration of a block the main
    thread takes part in, and must not count the parity words the exiting iteration re-armed as unmet checks."""
    import json
    import scipy.sparse as sp
    import slidingwindowdecoder_amd as S
    f = np.load(os.path.join(os.path.dirname(__file__), "golden", "ens_main_early_convergence.npz"))
    mat = sp.csr_matrix((np.ones(len(f["indices"]), np.uint8), f["indices"], f["indptr"]), shape=tuple(f["shape"]))
    kw = json.loads(str(f["kw"]))
    for over in (dict(), dict(max_side_depth=0), dict(max_tree_depth=2, max_side_depth=4), dict(max_iter_per_step=9)):
        k2 = dict(kw, **over)
        dev = S.bpgdg_decoder(mat, channel_probs=f["prior"], **k2)
        ora = O.bpgdg_decoder(mat, channel_probs=f["prior"], **k2)
        out = dev.decode_batch(np.repeat(f["synd"][None, :], 3, axis=0))
        want = ora.decode(f["synd"])
        assert np.array_equal(out[0], want) and np.array_equal(out[2], want), over
        assert dev.last_min_pm[0] == ora.min_pm, (over, dev.last_min_pm[0], ora.min_pm)
        if not over:
            assert np.array_equal(want, f["expect"]) and ora.min_pm == float(f["expect_pm"])


@pytest.mark.parametrize("D,S_", [(3, 10), (5, 6)])
def test_threaded_ensemble_pipeline_work_items_vs_oracle(D, S_, monkeypatch):
    """The whole (3,1) window loop with bpgdg_decoder(multi_thread=True) in every window.  In a pipeline launch the unit's owner walks
    the prefix tree and the main thread, the tree and side threads are TASKS on other workgroups and a FINAL item replays the offers
    (swd_gdg_kernel.h, gdg_ensemble_tree roles 1 / 2): total_e_hat, converge flags, path metrics, winners, tie counts and BP block
    counts of every (shot, window) against the oracle driven through the host-side window loop -- and the other two schedules
    (units from the ring without tasks; window-major tickets) must give the same records word for word."""
    import slidingwindowdecoder_amd as S
    from oracle import oracle as O
    from slidingwindowdecoder_amd.windows import sliding_window_decode_host
    from tests.test_gpu_pipeline import load_plan
    f = fx.load("bb144_circuit_p003_w3f1.npz")
    plan = load_plan(f, 11)
    kw = fx.params(f, "gdg_params")
    kw.pop("multi_thread")
    kw.update(max_tree_depth=D, max_side_depth=S_)
    B = 64
    det = fx.unpack(f["det"], plan.chk.shape[0])[:B]
    W = len(plan.windows)
    rec = {k: np.zeros((B, W)) for k in ("pm", "winner", "ties", "blocks", "conv", "post")}

    class Fresh:  # the device gives every decode the state of a newly built object
        def __init__(self, w):
            self.w = w

        def decode(self, s):
            self.d = O.bpgdg_decoder(self.w.mat, channel_probs=self.w.prior, multi_thread=True, **kw)
            return self.d.decode(s)

    def tap(wi, j, dec, s, e_hat):
        d = dec.d
        rec["conv"][j, wi] = bool(d.converge)
        rec["post"][j, wi] = d._res.exit_class != 0
        if d._res.exit_class != 0 and e_hat is not None:
            pms, winner, nt = d.ensemble_info()
            rec["pm"][j, wi], rec["winner"][j, wi], rec["ties"][j, wi] = d.min_pm, winner, nt
            rec["blocks"][j, wi] = d.ensemble_blocks()[0]

    want, _ = sliding_window_decode_host(plan, det, Fresh, on_decode=tap)
    results = []
    for env in ({}, {"SWD_ENS_NO_TASKS": "1"}, {"SWD_ENS_TICKETS": "1"}):
        for k in ("SWD_ENS_NO_TASKS", "SWD_ENS_TICKETS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        dec = S.SlidingWindowDecoder(plan, decoder="bpgdg_decoder", multi_thread=True, **kw)
        total = dec.decode(det)
        results.append((total.copy(), dec.last_stats.copy(), dec.last_min_pm.copy()))
    for k in ("SWD_ENS_NO_TASKS", "SWD_ENS_TICKETS"):
        monkeypatch.delenv(k, raising=False)
    total, st, pm = results[0]
    bad = np.flatnonzero((total != want).any(axis=1))
    assert bad.size == 0, f"{bad.size} shots differ from the oracle host loop: {bad[:8]}"
    assert np.array_equal((st[..., 0] & 0x100) != 0, rec["conv"] != 0)
    post = rec["post"] != 0
    ens = post & ((st[..., 0] & 0xFF) == 1)  # (a window whose BPGD::reset fails runs no ensemble: exit class 4)
    assert ens.sum() > 150, ens.sum()
    assert np.array_equal(pm[ens], rec["pm"][ens])
    assert np.array_equal(st[..., 6][ens], rec["winner"][ens]) and np.array_equal(st[..., 7][ens], rec["ties"][ens])
    assert np.array_equal(st[..., 5][ens], rec["blocks"][ens]), "BP blocks (counted once per thread that would run them)"
    for other, name in zip(results[1:], ("ring without tasks", "tickets")):
        assert np.array_equal(other[0], total), name
        assert np.array_equal(other[1], st), name
        assert np.array_equal(other[2], pm), name
    # batches of a stream object take the ticket schedule from 3072 shots (here: from one shot): the same records again
    monkeypatch.setenv("SWD_GDG_STREAM_SERIAL_MIN", "1")
    dec = S.SlidingWindowDecoder(plan, decoder="bpgdg_decoder", multi_thread=True, **kw)
    cuts = [0, 40, 41, B]
    for (a0, b0), (tot_s, st_s, pm_s, _, _) in zip(zip(cuts[:-1], cuts[1:]), dec.decode_stream([det[a:b] for a, b in zip(cuts[:-1], cuts[1:])])):
        assert np.array_equal(tot_s, total[a0:b0]) and np.array_equal(st_s, st[a0:b0]) and np.array_equal(pm_s, pm[a0:b0])
    dec.check_status()
    monkeypatch.delenv("SWD_GDG_STREAM_SERIAL_MIN")
    # more shots than workgroups and contexts to spare: the same shots tiled and shuffled, every copy the same record
    reps = 28
    perm = np.random.default_rng(D).permutation(B * reps)
    big = np.tile(det, (reps, 1))[perm]
    dec = S.SlidingWindowDecoder(plan, decoder="bpgdg_decoder", multi_thread=True, **kw)
    tot2 = dec.decode(big)
    assert np.array_equal(tot2, np.tile(total, (reps, 1))[perm])
    assert np.array_equal(dec.last_stats, np.tile(st, (reps, 1, 1))[perm]) and np.array_equal(dec.last_min_pm, np.tile(pm, (reps, 1))[perm])
    dec.check_status()


def test_threaded_ensemble_weight2_known_answer():
    """`Syndrome code.ipynb` cell 6 (:233-234): only (0,72) and (1,73) converge, both with 14 flipped variable nodes -- the stored
    output of the reference's multi_thread=True run (recorded again in the fixture)"""
    import slidingwindowdecoder_amd as S
    from slidingwindowdecoder_amd.codes import bb_code
    k = fx.load("bb288_hx_wt2_kat.npz")
    code, _, _ = bb_code(288)
    pairs = k["pairs"]
    synd = np.zeros((len(pairs), 144), np.uint8)
    for q, (i, j) in enumerate(pairs):
        synd[q, i] = synd[q, j] = 1
    dec = S.bpgdg_decoder(code.hx, channel_probs=np.ones(288) * 0.01, multi_thread=True, **fx.params(k, "params"))
    out = dec.decode_batch(synd)
    conv = np.flatnonzero(dec.last_status & 0x100)
    got = [(int(pairs[q][0]), int(pairs[q][1]), int(out[q].sum())) for q in conv]
    assert got == [tuple(int(x) for x in r) for r in k["multi"]] == [(0, 72, 14), (1, 73, 14)]


def test_deep_trees_beyond_64_snapshots_vs_oracle():
    """max_tree_depth 5, max_side_depth 12: max_guess = 2 (2^5 - 1) + 12 - 5 = 69 snapshots -- more than the parallel tree search
    holds per tree (64), so the plan takes the serial walk (160 snapshots); single-thread gdg() and the threaded ensemble vs the oracle"""
    import slidingwindowdecoder_amd as S
    from oracle import oracle as O
    f = fx.load("bb72_capacity.npz")
    mat, _ = fx.graph(f, "gdg_")
    rng = np.random.default_rng(5)
    priors = rng.uniform(0.04, 0.09, size=72)
    kw = dict(max_iter=6, ms_scaling_factor=1.0, max_iter_per_step=4, max_step=20, max_tree_depth=5, max_side_depth=12,
              max_tree_branch_step=6, max_side_branch_step=8, gdg_factor=1.0)
    H = mat.toarray().astype(np.int64)
    synd = np.array([(H @ (rng.random(72) < priors * 1.4).astype(np.int64) % 2) for _ in range(160)], dtype=np.uint8)
    dev, ora = S.bpgdg_decoder(mat, channel_probs=priors, **kw), O.bpgdg_decoder(mat, channel_probs=priors, **kw)
    out = dev.decode_batch(synd)
    deep = 0
    for k in range(len(synd)):
        ora.clear_history()
        assert np.array_equal(out[k], ora.decode(synd[k])) and bool(dev.last_status[k] & 0x100) == bool(ora.converge), k
        deep = max(deep, int(dev.last_stats[k, 4]))
    print("largest number of snapshots pushed by one tree:", deep)
    _ensemble_vs_oracle(mat, priors, kw, synd[:100], 20)


def test_pipeline_with_more_than_256_windows_takes_the_serial_form():
    """The parallel tree search packs the window number of a work item into 8 bits (swd_gdg_kernel.h, item_unit): a plan with more
    windows must run the serial walk (Plan::finalize) -- a 300-round block-banded detector error model, (2,1) windows ->
    299 windows, against the oracle driven through the host window loop."""
    import scipy.sparse as sp
    from oracle import oracle as O
    from slidingwindowdecoder_amd import SlidingWindowDecoder
    from slidingwindowdecoder_amd.windows import Window, WindowPlan, sliding_window_decode_host
    rng = np.random.default_rng(77)
    h, R, W = 6, 300, 2
    rows, cols, starts, nloc = [], [], [], []
    col = 0
    for r in range(R):
        starts.append(col)
        loc = [[r * h + i] for i in range(h)] + [sorted(rng.choice(h, 2, replace=False) + r * h) for _ in range(4)]
        span = [] if r == R - 1 else [[r * h + int(rng.integers(h)), (r + 1) * h + int(rng.integers(h))] for _ in range(5)]
        nloc.append(len(loc))
        for rr in loc + span:
            rows += list(rr); cols += [col] * len(rr); col += 1
    starts.append(col)
    chk = sp.csr_matrix((np.ones(len(rows), np.uint8), (rows, cols)), shape=(R * h, col))
    priors = rng.uniform(0.01, 0.05, size=col)
    obs = sp.csr_matrix((rng.random((3, col)) < 0.1).astype(np.uint8))
    wins = []
    for top in range(R - W + 1):
        last = top + W >= R
        r0, r1, c0 = top * h, (top + W) * h, starts[top]
        if not last:
            c1 = starts[top + W - 1] + nloc[top + W - 1]
            ident = sp.csr_matrix((np.ones(h, np.uint8), (np.arange(W * h - h, W * h), np.arange(h))), shape=(W * h, h))
            mat = sp.hstack((chk[r0:r1, c0:c1], ident), format="csr")
            prior = np.concatenate((priors[c0:c1], np.full(h, 0.03)))
            ncg, commit = c1 - c0, starts[top + 1] - c0
        else:
            mat, prior = sp.csr_matrix(chk[r0:r1, c0:col]), priors[c0:col].copy()
            ncg = commit = col - c0
        mat.sort_indices()
        wins.append(Window(r0, r1, c0, ncg, commit, mat, prior, last))
    assert len(wins) == 299
    plan = WindowPlan(chk, obs, priors, np.arange(col), [(r * h, starts[r]) for r in range(R)] + [(R * h, col)], wins, 0.03, h)
    kw = dict(max_iter=6, ms_scaling_factor=1.0, max_iter_per_step=4, max_step=8, max_tree_depth=2, max_side_depth=5,
              max_tree_branch_step=10, max_side_branch_step=6)
    e = (rng.random((6, col)) < priors * 1.5).astype(np.uint8)
    det = ((sp.csr_matrix(e) @ chk.T.astype(np.int32)).toarray() % 2).astype(np.uint8)
    dev = SlidingWindowDecoder(plan, decoder="bpgdg_decoder", **kw)
    total = dev.decode(det)

    class Fresh:
        def __init__(self, w): self.w = w
        def decode(self, s): return O.bpgdg_decoder(self.w.mat, channel_probs=self.w.prior, **kw).decode(s)
    want, _ = sliding_window_decode_host(plan, det, Fresh)
    assert np.array_equal(total, want)


def test_reused_ensemble_object_semantics_vs_the_reference_object():
    """bpgdg_decoder(multi_thread=True).decode() one syndrome at a time against ONE re-used BPGD_main_thread of the reference
    (oracle/_ref, the reference's own bpgd.cpp compiled where it lies; the library travels to the GPU box): when BPGD::reset fails
    the reference hands back its PREVIOUS decode's position vector over this decode's sorted columns (bpgd.cpp:597-599, 619-625,
    bp_guessing_decoder.pyx:247-251) -- round 5 returned zeros there.  Every decode's vector must equal the reference object's,
    stale ones included (decodes whose winning path metric is tied between different vectors are left out: thread timing)."""
    import ctypes as C
    import os
    import scipy.sparse as sp
    import slidingwindowdecoder_amd as S
    from oracle import oracle as O
    REF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libswd_ref.so")
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/libswd_ref.so did not travel to this box (built from /root/reference by make -C oracle ref)")
    from tests.test_oracle_vs_ref import Pcm
    R = C.CDLL(REF)
    vp, i32 = C.c_void_p, C.c_int32
    R.ref_pcm_new.restype = vp
    R.ref_pcm_new.argtypes = [i32, i32, vp, vp]
    R.ref_pcm_free.argtypes = [vp]
    R.ref_gdg_multi_new.restype = vp
    R.ref_gdg_multi_new.argtypes = [i32] * 9 + [C.c_double]
    R.ref_gdg_multi_free.argtypes = [vp]
    R.ref_gdg_multi_decode.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp]
    f = fx.load("bb72_capacity.npz")
    mat, _ = fx.graph(f, "gdg_")
    m, n = mat.shape
    rng = np.random.default_rng(5)
    priors = rng.uniform(0.03, 0.08, size=n)
    new_n = 24
    kw = dict(max_iter=8, ms_scaling_factor=1.0, max_iter_per_step=6, max_step=25, max_tree_depth=3, max_side_depth=10,
              max_tree_branch_step=10, max_side_branch_step=10, gdg_factor=1.0, new_n=new_n)
    H = sp.csr_matrix(mat).astype(np.int64)
    synds = [(H @ (rng.random(n) < priors * 1.3).astype(np.int64) % 2).astype(np.uint8) for _ in range(120)]
    synds += [(rng.random(m) < 0.3).astype(np.uint8) for _ in range(120)]
    order = rng.permutation(len(synds))
    dev = S.bpgdg_decoder(mat, channel_probs=priors, multi_thread=True, **kw)
    fresh = S.bpgdg_decoder(mat, channel_probs=priors, multi_thread=True, reuse_object=False, **kw)
    ora = O.bpgdg_decoder(mat, channel_probs=priors, multi_thread=True, **kw)
    p = Pcm(R, mat)
    llr = np.ascontiguousarray(np.log((1 - priors) / priors))
    obj = R.ref_gdg_multi_new(m, new_n, kw["max_iter_per_step"], kw["max_step"], kw["max_tree_depth"], kw["max_side_depth"],
                              kw["max_tree_branch_step"], kw["max_side_branch_step"], 0, 1.0)
    ran = fails = stale = 0
    for k in order:
        s = synds[k]
        ora.clear_history()
        o_out = ora.decode(s)
        got = dev.decode(s)
        if ora._res.exit_class == 0:  # the pre-processing BP converged: the ensemble object is not touched
            assert np.array_equal(got, o_out)
            continue
        cols = np.ascontiguousarray(ora.cols)
        err, rpm, rpms = np.zeros(new_n, np.uint8), C.c_double(), np.zeros(256)
        su = np.ascontiguousarray(s, np.uint8)
        R.ref_gdg_multi_decode(obj, p.h, m, new_n, cols.ctypes.data, llr.ctypes.data, su.ctypes.data, err.ctypes.data, C.byref(rpm), rpms.ctypes.data)
        ref_out = np.zeros(n, np.int64)
        ref_out[cols[:new_n]] = err
        ran += 1
        if ora.ensemble_blocks()[0] == 0:  # BPGD::reset failed
            fails += 1
            stale += int(err.any())
            assert np.array_equal(got, ref_out), "a failed reset must hand back the previous decode's position vector"
            assert not fresh.decode(s).any() and not dev.converge
            continue
        if ora.ensemble_info()[2] == 0:
            assert np.array_equal(got, ref_out)
            assert bool(dev.converge) == (rpm.value < 9999.0)
        else:  # tied winners: the device and the reference may pick different vectors; keep the two objects' states aligned
            dev._prev_pos = err.copy()
    R.ref_gdg_multi_free(obj)
    assert ran >= 80 and fails >= 5 and stale >= 3, (ran, fails, stale)
