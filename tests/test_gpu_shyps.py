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
    flagged, logical = logical_error_stats(plan, det, obs, total)
    assert logical.mean() < 0.5


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
