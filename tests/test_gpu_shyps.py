"""SHYPS r=3 memory experiment (BASELINE config 5's circuit): (3,1) sliding windows of the stim-free DEM decoded on
the device vs the oracle driven through the host-side window loop (osd_window semantics; the reference's notebook
uses the third-party ldpc decoder there, for which no pinned oracle exists)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("order", [0, 10])
def test_shyps_windows_vs_oracle(order):
    from oracle import oracle as O
    from slidingwindowdecoder_amd import SlidingWindowDecoder, shyps
    from slidingwindowdecoder_amd.windows import logical_error_stats, plan_windows, sample_dem, sliding_window_decode_host
    dem = shyps.shyps_dem(3, 0.004, 6)
    plan = plan_windows(dem.chk, dem.obs, dem.priors, 21, 3, 1, method=1)
    assert [w.mat.shape for w in plan.windows][:2] == [(63, 476), (63, 476)]
    det, obs, _ = sample_dem(plan.chk, plan.obs, plan.priors, 400, seed=11)
    kw = dict(pre_max_iter=8, post_max_iter=100, ms_scaling_factor=1.0, osd_method="osd_cs", osd_order=order)
    dec = SlidingWindowDecoder(plan, **kw)
    total = dec.decode(det)
    want, _ = sliding_window_decode_host(plan, det, lambda w: O.osd_window(w.mat, channel_probs=w.prior, **kw))
    assert np.array_equal(total, want)
    cls = np.bincount((dec.last_stats[..., 0] & 0xFF).ravel(), minlength=6)
    assert cls[2] > 0 and cls[1] > 0  # OSD and post-BP exits both exercised
    # (p = 0.004 is a parity point far above the regime of the notebook -- some shots end flagged there; the logical error rate
    # is checked below at the notebook's p = 0.001)


def test_shyps_twelve_round_window_vs_oracle():
    """BASELINE config 5 names a "12-round window": W = 12 rounds of 21 detectors -> 252 x 2240 window matrices
    (column weight 9, row weight 44), (W,F) = (12,1) over a 14-round experiment, decoded on the device vs the oracle."""
    from oracle import oracle as O
    from slidingwindowdecoder_amd import SlidingWindowDecoder, shyps
    from slidingwindowdecoder_amd.windows import logical_error_stats, plan_windows, sample_dem, sliding_window_decode_host
    dem = shyps.shyps_dem(3, 0.004, 14)
    plan = plan_windows(dem.chk, dem.obs, dem.priors, 21, 12, 1, method=1)
    assert [w.mat.shape for w in plan.windows] == [(252, 2240)] * 3 + [(252, 2205)]
    det, obs, _ = sample_dem(plan.chk, plan.obs, plan.priors, 160, seed=12)
    kw = dict(pre_max_iter=8, post_max_iter=100, ms_scaling_factor=1.0, osd_method="osd_cs", osd_order=10)
    dec = SlidingWindowDecoder(plan, **kw)
    assert dec.threads == 1024
    total = dec.decode(det)
    want, _ = sliding_window_decode_host(plan, det, lambda w: O.osd_window(w.mat, channel_probs=w.prior, **kw))
    assert np.array_equal(total, want)
    cls = np.bincount((dec.last_stats[..., 0] & 0xFF).ravel(), minlength=6)
    assert cls[2] > 0 and cls[1] > 0
    flagged, _ = logical_error_stats(plan, det, obs, total)
    assert not flagged.any()  # every shot's committed faults reproduce its detector data


def _device_run(plan, shots, **kw):
    import torch
    from slidingwindowdecoder_amd import DemSampler, SlidingWindowDecoder
    dec = SlidingWindowDecoder(plan, **kw)
    det, flips = DemSampler(plan.chk, plan.obs, plan.priors).sample_device(shots, seed=20240318)
    shot = torch.empty((shots, 2), dtype=torch.int32, device="cuda")
    _, stats, _ = dec.decode_device(det, shot_result=shot)
    dec.check_status()
    sr = shot.cpu().numpy()
    true = flips.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    return sr[:, 1] != 0, (sr[:, 0].astype(np.int64) & 0xFFFFFFFF) != true, stats.cpu().numpy()


def test_shyps_p001_logical_error_rate_matches_the_notebook():
    """/root/reference/SHYPS.ipynb cell 2 (outputs at :212-234): r = 3, p = 0.001, 4 rounds, (W,F) = (3,1), OSD order 0, 20 000 shots ->
    0 flagged, 170 logical errors, 2.13e-3 per round (there with the third-party ldpc decoder inside the windows).  Same
    experiment, same number of shots, osd_window in the windows: no flagged shot and a logical count within three standard
    deviations of the difference of two such samples."""
    from slidingwindowdecoder_amd import shyps
    from slidingwindowdecoder_amd.windows import plan_windows
    dem = shyps.shyps_dem(3, 0.001, 4)
    assert dem.chk.shape == (105, 833)
    plan = plan_windows(dem.chk, dem.obs, dem.priors, 21, 3, 1, method=1)
    shots, ref = 20000, 170
    flagged, logical, _ = _device_run(plan, shots, pre_max_iter=8, post_max_iter=100, ms_scaling_factor=1.0, osd_method="osd_cs", osd_order=0)
    nerr = int((logical | flagged).sum())
    per_round = 1.0 - (1.0 - nerr / shots) ** (1.0 / 4)
    print(f"SHYPS r=3 p=0.001 (3,1): {nerr}/{shots} logical errors, {per_round:.3e} per round (notebook: 170/20000, 2.13e-3)")
    assert not flagged.any()
    assert abs(nerr - ref) < 3.0 * (nerr + ref) ** 0.5, (nerr, ref)


def test_shyps_twelve_round_windows_p001_rate():
    """the twelve-round windows at the notebook's noise level: a wider window must not decode worse than the (3,1) run
    (2.13e-3 per round, SHYPS.ipynb:234) -- 14 rounds, 8192 device-sampled shots"""
    from slidingwindowdecoder_amd import shyps
    from slidingwindowdecoder_amd.windows import plan_windows
    dem = shyps.shyps_dem(3, 0.001, 14)
    plan = plan_windows(dem.chk, dem.obs, dem.priors, 21, 12, 1, method=1)
    shots = 8192
    flagged, logical, _ = _device_run(plan, shots, pre_max_iter=8, post_max_iter=100, ms_scaling_factor=1.0, osd_method="osd_cs", osd_order=10)
    nerr = int((logical | flagged).sum())
    per_round = 1.0 - (1.0 - nerr / shots) ** (1.0 / 14)
    print(f"SHYPS r=3 p=0.001 (12,1), 14 rounds: {nerr}/{shots} logical errors, {per_round:.3e} per round")
    assert not flagged.any()
    assert per_round < 2.6e-3  # 2.13e-3 + three standard deviations of this sample size
