"""GPU parity of the whole sliding-window pipeline (one launch for all shots and windows) against
the reference's own sliding run recorded in tests/golden (osd.py:130-179 semantics)."""
import numpy as np
import pytest
import scipy.sparse as sp

from tests import fixtures as fx

pytestmark = pytest.mark.gpu


def load_plan(f, nwin):
    from slidingwindowdecoder_amd.windows import Window, WindowPlan
    chk, priors = fx.graph(f, "chk_")
    wins = []
    for wi in range(nwin):
        mat, pr = fx.graph(f, f"win{wi}_")
        r0, r1, c0, ncg, commit, last = (int(x) for x in f[f"win{wi}_meta"])
        wins.append(Window(r0, r1, c0, ncg, commit, mat, pr, bool(last)))
    obs = fx.graph(f, "obs_")[0] if "obs_indptr" in f else sp.csr_matrix((0, chk.shape[1]), dtype=np.uint8)
    return WindowPlan(chk, obs, priors, np.arange(chk.shape[1]), [tuple(a) for a in f["anchors"]], wins,
                      float(f["noisy_prior"]), 0)


@pytest.mark.parametrize("tag", ["osd0", "osd10"])
def test_bb144_pipeline_matches_reference_run(tag):
    from slidingwindowdecoder_amd import SlidingWindowDecoder
    f = fx.load("bb144_circuit_p003_w3f1.npz")
    plan = load_plan(f, 11)
    kw = fx.params(f, tag + "_params")
    shots = int(f["num_shots"])
    det = fx.unpack(f["det"], plan.chk.shape[0])
    dec = SlidingWindowDecoder(plan, **kw)
    total = dec.decode(det)
    want = fx.unpack(f[tag + "_total"], plan.chk.shape[1])
    bad = np.flatnonzero((total != want).any(axis=1))
    assert bad.size == 0, f"{bad.size}/{shots} shots differ: {bad[:8]}"
    # every window decode agrees with the recorded one (iterations, converge, min_pm)
    for wi in range(11):
        tr = fx.Trace(f, f"{tag}_win{wi}_", *plan.windows[wi].mat.shape)
        assert np.array_equal(dec.last_stats[:, wi, 1], tr.bp_iteration), f"window {wi}: bp_iteration"
        assert np.array_equal((dec.last_stats[:, wi, 0] & 0x100) != 0, tr.converge != 0), f"window {wi}: converge"
        assert np.array_equal(dec.last_min_pm[:, wi], tr.min_pm), f"window {wi}: min_pm"
    # logical-error accounting of osd.py:184-191 on the same shots
    from slidingwindowdecoder_amd.windows import logical_error_stats
    obs = fx.unpack(f["obs_data"], 12)
    flagged, logical = logical_error_stats(plan, det, obs, total)
    assert not flagged.any()
    assert np.array_equal(logical.astype(np.uint8), f[tag + "_logical"])
    # the same accounting done on the device: predicted observable flips + flagged bit per shot
    pred = (sp.csr_matrix(total) @ plan.obs.T.astype(np.int32)).toarray() % 2
    pred_mask = (pred.astype(np.uint32) << np.arange(12, dtype=np.uint32)).sum(axis=1).astype(np.uint32)
    assert np.array_equal(dec.last_obs_flips, pred_mask)
    assert not dec.last_flagged.any()
    obs_mask = (obs.astype(np.uint32) << np.arange(12, dtype=np.uint32)).sum(axis=1).astype(np.uint32)
    assert np.array_equal((dec.last_obs_flips != obs_mask) | dec.last_flagged, logical)


def test_bb144_pipeline_device_tensors_and_stats():
    import torch
    from slidingwindowdecoder_amd import SlidingWindowDecoder
    f = fx.load("bb144_circuit_p003_w3f1.npz")
    plan = load_plan(f, 11)
    det = fx.unpack(f["det"], plan.chk.shape[0])
    dec = SlidingWindowDecoder(plan, **fx.params(f, "osd0_params"))
    d = torch.from_numpy(np.ascontiguousarray(det)).cuda()
    total, stats, pm = dec.decode_device(d)
    torch.cuda.synchronize()
    assert np.array_equal(total.cpu().numpy(), fx.unpack(f["osd0_total"], plan.chk.shape[1]))
    st = stats.cpu().numpy()
    cls = st[..., 0] & 0xFF
    assert (st[..., 1] == st[..., 2] + st[..., 3]).all()
    post = cls >= 1
    # shortened graphs keep at most new_n = 432 live variable nodes
    assert (st[..., 4][post] <= 432).all() and (st[..., 6][post] < 5976).all()
    assert (st[..., 7][cls == 2] > 0).all() and (st[..., 7][cls != 2] == 0).all()


def test_bb288_pipeline_vs_oracle_host_loop():
    """[[288,12,18]] (4,1): device pipeline against the oracle driven through the host-side window
    loop (same commit rule), OSD order 0."""
    from oracle import oracle as O
    from slidingwindowdecoder_amd import SlidingWindowDecoder
    from slidingwindowdecoder_amd.windows import sliding_window_decode_host
    f = fx.load("bb288_circuit_p005_w4f1.npz")
    plan = load_plan(f, 4)
    kw = dict(fx.params(f, "osd10_params"), osd_order=0)
    det = fx.unpack(f["det"], plan.chk.shape[0])
    dec = SlidingWindowDecoder(plan, **kw)
    total = dec.decode(det)
    want, _ = sliding_window_decode_host(plan, det, lambda w: O.osd_window(w.mat, channel_probs=w.prior, **kw))
    bad = np.flatnonzero((total != want).any(axis=1))
    assert bad.size == 0, f"shots {bad.tolist()} differ; exit classes {(dec.last_stats[bad, :, 0] & 0xFF).tolist()}, iterations {dec.last_stats[bad, :, 1].tolist()}"


@pytest.mark.parametrize("decoder", ["osd_window", "bpgdg_decoder", "bpgd_decoder", "ens"])
def test_fuzz_random_window_plans_vs_oracle_host_loop(decoder):
    """Random block-banded detector error models, (W, F), priors and decoder parameters: one device launch
    against the host window loop driven with the oracle (tests/fuzz_pipeline.py, fixed seed).  "ens" =
    bpgdg_decoder(multi_thread=True) with random tree shapes: the threaded ensemble with its threads as work items."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_pipeline.py"), "12", "3", decoder],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_window_placement_is_validated():
    """A commit count larger than the window (osd.py:170-173 would fail to broadcast) is refused at create time."""
    from slidingwindowdecoder_amd import SlidingWindowDecoder
    f = fx.load("bb144_circuit_p003_w3f1.npz")
    plan = load_plan(f, 11)
    w = plan.windows[0]
    plan.windows[0] = type(w)(w.row0, w.row1, w.col0, w.ncols_global, w.mat.shape[1] + 1, w.mat, w.prior, w.is_last)
    with pytest.raises(ValueError, match="invalid window placement"):
        SlidingWindowDecoder(plan, **fx.params(f, "osd0_params"))


@pytest.mark.parametrize("method,W,F,noisy", [(0, 3, 1, None), (2, 3, 1, None), (2, 4, 2, None), (1, 3, 1, 0.05), (2, 3, 2, 0.02)])
def test_plan_windows_methods_vs_oracle_host_loop(method, W, F, noisy):
    """Window extraction variants of osd.py:79-121: method 0 (plain sub-matrices), method 2 (identity block behind ALL faults
    of the last round block) and a caller-given noisy-syndrome prior -- [[72,12,6]] circuit-level DEM, device pipeline against
    the oracle driven through the host-side window loop."""
    from oracle import oracle as O
    from slidingwindowdecoder_amd import SlidingWindowDecoder
    from slidingwindowdecoder_amd.circuit import bb_dem
    from slidingwindowdecoder_amd.codes import bb_code
    from slidingwindowdecoder_amd.windows import plan_windows, sample_dem, sliding_window_decode_host
    code, A, B = bb_code(72)
    dem = bb_dem(code, A, B, 0.004, 6)
    plan = plan_windows(dem.chk, dem.obs, dem.priors, 36, W, F, method=method, noisy_prior=noisy)
    if noisy is not None:
        assert plan.noisy_prior == noisy and all(np.all(w.prior[-36:] == noisy) for w in plan.windows[:-1])
    det, obs, _ = sample_dem(plan.chk, plan.obs, plan.priors, 96, seed=100 + method)
    kw = dict(pre_max_iter=8, post_max_iter=40, ms_scaling_factor=1.0, osd_method="osd_cs", osd_order=4)
    dec = SlidingWindowDecoder(plan, **kw)
    total = dec.decode(det)
    want, _ = sliding_window_decode_host(plan, det, lambda w: O.osd_window(w.mat, channel_probs=w.prior, **kw))
    bad = np.flatnonzero((total != want).any(axis=1))
    assert bad.size == 0, f"shots {bad.tolist()} differ"
    assert len(np.unique(dec.last_stats[..., 0] & 0xFF)) >= 2


def test_x_basis_sliding_window_experiment():
    """The x-basis memory experiment of `Sliding Window OSD.ipynb` (N = 144, p = 0.004, 12 rounds, (W,F) = (5,2), method 1,
    z_basis=False; notebook: 0 flagged, 140 / 10 000 logical errors -- with the third-party ldpc BP+OSD-CS 10 in the windows,
    shorten=False, so that count is context, not a parity target): the 360 x 3096 windows on the device against the oracle host
    loop (bit-exact), then 10 000 device-sampled shots with osd_window(osd_cs 10): no flagged shot, a logical error rate of the
    notebook's order of magnitude."""
    import torch
    from oracle import oracle as O
    from slidingwindowdecoder_amd import DemSampler, SlidingWindowDecoder
    from slidingwindowdecoder_amd.circuit import bb_dem
    from slidingwindowdecoder_amd.codes import bb_code
    from slidingwindowdecoder_amd.windows import plan_windows, sample_dem, sliding_window_decode_host
    code, A, B = bb_code(144)
    dem = bb_dem(code, A, B, 0.004, 12, z_basis=False)
    plan = plan_windows(dem.chk, dem.obs, dem.priors, 72, 5, 2, method=1, z_basis=False)
    assert plan.noisy_prior == 0.05900506726184526
    kw = dict(pre_max_iter=8, post_max_iter=200, ms_scaling_factor=1.0, osd_method="osd_cs", osd_order=0)
    dec = SlidingWindowDecoder(plan, **kw)
    det, obs, _ = sample_dem(plan.chk, plan.obs, plan.priors, 24, seed=9)
    total = dec.decode(det)
    want, _ = sliding_window_decode_host(plan, det, lambda w: O.osd_window(w.mat, channel_probs=w.prior, **kw))
    assert np.array_equal(total, want)
    shots, ref = 10000, 140
    dec = SlidingWindowDecoder(plan, **dict(kw, osd_order=10))
    d, flips = DemSampler(plan.chk, plan.obs, plan.priors).sample_device(shots, seed=20240318)
    shot = torch.empty((shots, 2), dtype=torch.int32, device="cuda")
    dec.decode_device(d, shot_result=shot)
    dec.check_status()
    sr = shot.cpu().numpy()
    flagged = sr[:, 1] != 0
    nerr = int((((sr[:, 0].astype(np.int64) & 0xFFFFFFFF) != (flips.cpu().numpy().astype(np.int64) & 0xFFFFFFFF)) | flagged).sum())
    print(f"x-basis (5,2) p=0.004: {nerr}/{shots} logical errors (notebook 140/10000), flagged {int(flagged.sum())}")
    assert not flagged.any()
    assert ref // 3 < nerr < 2 * ref, (nerr, ref)


def test_packed_output_and_streaming_equal_the_one_shot_decode():
    """The host path of the notebooks: total_e_hat travels bit-packed (decode(packed=True) returns it as it travels), a call of
    2048 shots or more is cut in two halves on the two lanes, and the streaming form (two batches in flight, ragged batch sizes)
    returns batch by batch what separate decode() calls return -- all against the reference's recorded run."""
    import torch
    from slidingwindowdecoder_amd import SlidingWindowDecoder
    f = fx.load("bb144_circuit_p003_w3f1.npz")
    plan = load_plan(f, 11)
    kw = fx.params(f, "osd10_params")
    det = fx.unpack(f["det"], plan.chk.shape[0])
    want = fx.unpack(f["osd10_total"], plan.chk.shape[1])
    dec = SlidingWindowDecoder(plan, **kw)
    ncol = plan.chk.shape[1]
    bits = dec.decode(det, packed=True)
    assert bits.shape == (len(det), (ncol + 7) // 8)
    assert np.array_equal(np.unpackbits(bits, axis=1, count=ncol, bitorder="little"), want)
    st0, pm0, fl0 = dec.last_stats.copy(), dec.last_min_pm.copy(), dec.last_obs_flips.copy()
    # a call large enough to be cut in two (the 192 recorded shots tiled): both halves, in order
    reps = 11
    big = np.tile(det, (reps, 1))
    total = dec.decode(big)
    assert np.array_equal(total, np.tile(want, (reps, 1)))
    assert np.array_equal(dec.last_stats, np.tile(st0, (reps, 1, 1))) and np.array_equal(dec.last_min_pm, np.tile(pm0, (reps, 1)))
    assert np.array_equal(dec.last_obs_flips, np.tile(fl0, reps)) and not dec.last_flagged.any()
    # streaming: ragged batches, two in flight
    cuts = [0, 64, 65, 130, 192]
    batches = [det[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    got = list(dec.decode_stream(batches))
    assert len(got) == len(batches)
    for (a, b), (tot, st, pm, flips, flagged) in zip(zip(cuts[:-1], cuts[1:]), got):
        assert np.array_equal(tot, want[a:b]) and np.array_equal(st, st0[a:b]) and np.array_equal(pm, pm0[a:b])
        assert np.array_equal(flips, fl0[a:b]) and not flagged.any()
    # packed stream without statistics; a third push without a pop is refused
    s = dec.stream(192, packed=True, want_stats=False)
    s.push(det); s.push(det[:100])
    with pytest.raises(RuntimeError):
        s.push(det)
    t1, st, pm, _, _ = s.pop()
    assert st is None and pm is None and np.array_equal(np.unpackbits(t1, axis=1, count=ncol, bitorder="little"), want)
    t2 = s.pop()[0]
    assert np.array_equal(np.unpackbits(t2, axis=1, count=ncol, bitorder="little"), want[:100])
    with pytest.raises(RuntimeError):
        s.pop()
    s.close()
    # device form: launches alternate between the two lanes, one set of output tensors per lane
    dev = torch.device("cuda", 0)
    d_t = torch.from_numpy(det).to(dev)
    outs = [dict(total=torch.empty((len(det), ncol), dtype=torch.uint8, device=dev), stats=torch.empty((len(det), 11, 8), dtype=torch.int32, device=dev),
                 shot_result=torch.empty((len(det), 2), dtype=torch.int32, device=dev)) for _ in range(2)]
    s = dec.stream(len(det))
    for i in range(6):
        s.push_device(d_t, **outs[i % 2])
    s.wait()
    for o in outs:
        assert np.array_equal(o["total"].cpu().numpy(), want) and np.array_equal(o["stats"].cpu().numpy(), st0)
        assert np.array_equal(o["shot_result"].cpu().numpy()[:, 0].astype(np.uint32), fl0)
    s.wait(torch.cuda.current_stream(dev))
    dec.check_status()
    s.close()


def test_push_device_orders_behind_the_default_stream_and_survives_destroy_order():
    """Round-4 advisor findings on the streaming form: (1) the lanes are non-blocking streams, and PyTorch's default stream has
    handle 0 -- `after` = NULL must still order the launch behind the producer of det on that stream (a long sleep kernel followed
    by the copy that fills det) and behind the last reader of the lane's outputs; (2) a pipeline destroyed before its stream object
    leaves a stream whose calls fail cleanly instead of touching freed memory."""
    import torch
    from slidingwindowdecoder_amd import SlidingWindowDecoder
    f = fx.load("bb144_circuit_p003_w3f1.npz")
    plan = load_plan(f, 11)
    kw = fx.params(f, "osd10_params")
    det = fx.unpack(f["det"], plan.chk.shape[0])
    want = fx.unpack(f["osd10_total"], plan.chk.shape[1])
    ncol = plan.chk.shape[1]
    dec = SlidingWindowDecoder(plan, **kw)
    dev = torch.device("cuda", 0)
    assert torch.cuda.current_stream(dev).cuda_stream == 0  # the case the finding is about
    src = torch.from_numpy(det).to(dev)
    d_t = torch.zeros_like(src)
    outs = [dict(total=torch.empty((len(det), ncol), dtype=torch.uint8, device=dev),
                 shot_result=torch.empty((len(det), 2), dtype=torch.int32, device=dev)) for _ in range(2)]
    s = dec.stream(len(det))
    torch.cuda.synchronize()
    for i in range(4):
        d_t.zero_()
        torch.cuda._sleep(200_000_000)       # ~0.1 s of default-stream work in front of the producer
        d_t.copy_(src)                       # det is only valid once this has run
        s.push_device(d_t, **outs[i % 2])    # after=None -> the current (= default) stream
        s.wait(torch.cuda.current_stream(dev))
        assert np.array_equal(outs[i % 2]["total"].cpu().numpy(), want), f"push {i}: the launch did not wait for its input"
    # explicit opt-out still works when the caller has synchronised
    torch.cuda.synchronize()
    s.push_device(d_t, after=False, **outs[0])
    s.wait()
    assert np.array_equal(outs[0]["total"].cpu().numpy(), want)
    dec.check_status()
    # destroy order: the pipeline first, then calls on the orphaned stream fail with a message, then the stream is freed
    h = dec._h
    dec._h = None
    from slidingwindowdecoder_amd import _lib
    _lib.lib().swd_pipeline_destroy(h)
    with pytest.raises(RuntimeError, match="destroyed"):
        s.push_device(d_t, **outs[0])
    with pytest.raises(RuntimeError, match="destroyed"):
        s.push(det)
    s.close()
