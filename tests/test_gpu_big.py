"""Row J of the scope table: osd_window on graphs whose fp64 messages do not fit one CU's LDS.  /root/reference/IBM.ipynb:119
builds it on the un-windowed 936 x 8784 detector error model (30 672 edges = 245 KB of messages); nothing in
/root/reference/src/osd_window.pyx:8-126 bounds m, n or the weights.  The large-graph kernels (pipeline_kernel<..., BIG>) keep
the scratch region of the layout in HBM and are selected automatically when no LDS-resident variant fits."""
import numpy as np
import pytest
import scipy.sparse as sp

from tests import fixtures as fx

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag,pkey", [("osd10_", "params"), ("osd0_", "params_b")])
def test_global_dem_matches_the_reference_run(tag, pkey):
    """all 288 recorded decodes of the reference on the global [[144,12,12]] DEM at p = 0.004, as IBM.ipynb configures it"""
    from slidingwindowdecoder_amd import osd_window
    f = fx.load("bb144_global_p004.npz")
    mat, priors = fx.graph(f, "chk_")
    tr = fx.Trace(f, tag, *mat.shape)
    dec = osd_window(mat, channel_probs=priors, **fx.params(f, pkey))
    assert (dec.m, dec.n, dec.new_n) == (936, 8784, 1872)
    out = dec.decode_batch(tr.synd, return_osd0=True)
    bad = np.flatnonzero((out != tr.out).any(axis=1))
    assert bad.size == 0, f"{bad.size}/{len(tr)} shots differ: {bad[:8]}, exit classes {(dec.last_status[bad[:8]] & 0xFF).tolist()}"
    assert np.array_equal(dec.last_iterations, tr.bp_iteration)
    assert np.array_equal((dec.last_status & 0x100) != 0, tr.converge != 0)
    assert np.array_equal(dec.last_min_pm, tr.min_pm)  # float ==
    osd = (dec.last_status & 0xFF) == 2
    assert osd.sum() >= 30 and np.array_equal(dec.last_osd0[osd], tr.osd0[osd])
    cls = np.bincount(dec.last_status & 0xFF, minlength=3)
    assert cls[0] > 0 and cls[1] > 0 and cls[2] > 0  # pre, post and OSD exits


def test_global_dem_single_decode_surface():
    """decode() one syndrome at a time (stateful history like the reference object) through the same kernels"""
    from slidingwindowdecoder_amd import osd_window
    f = fx.load("bb144_global_p004.npz")
    mat, priors = fx.graph(f, "chk_")
    tr = fx.Trace(f, "osd0_", *mat.shape)
    dec = osd_window(mat, channel_probs=priors, **fx.params(f, "params_b"))
    for k in range(12):
        out = dec.decode(tr.synd[k])
        assert np.array_equal(out, tr.out[k]) and dec.bp_iteration == tr.bp_iteration[k] and dec.min_pm == tr.min_pm[k]
        assert bool(dec.converge) == bool(tr.converge[k])


@pytest.mark.parametrize("m,n,colw,pre,post", [(512, 9000, 3, 8, 16), (1024, 8192, 3, 4, 8), (1000, 9216, 5, 5, 7), (1024, 4500, 9, 3, 12)])
def test_random_large_matrices_vs_oracle(m, n, colw, pre, post):
    """ragged random matrices beyond the LDS-resident variants (more than 8192 columns, or more than 160 KB of messages, or both),
    unequal priors, shortening lengths, both kernel forms (history ring in HBM when an iteration cap is no multiple of four)"""
    from oracle import oracle as O
    from slidingwindowdecoder_amd import osd_window
    rng = np.random.default_rng(m + n)
    deg = rng.integers(max(1, colw - 2), colw + 1, size=n)
    rows = np.concatenate([rng.choice(m, size=d, replace=False) for d in deg])
    H = sp.csr_matrix((np.ones(len(rows), np.uint8), (rows, np.repeat(np.arange(n), deg))), shape=(m, n))
    if np.diff(H.indptr).max() > 64:
        pytest.skip("row weight above 64")
    p = rng.uniform(0.0005, 0.004, size=n)
    kw = dict(channel_probs=p, pre_max_iter=pre, post_max_iter=post, ms_scaling_factor=float(rng.choice([1.0, 0.9])),
              osd_method="osd_cs", osd_order=3, new_n=int(rng.integers(m, min(n, 3 * m))))
    dev, ora = osd_window(H, **kw), O.osd_window(H, **kw)
    B = 12
    e = (rng.random((B, n)) < p * 1.5).astype(np.uint8)
    synd = ((sp.csr_matrix(e) @ H.T.astype(np.int32)).toarray() % 2).astype(np.uint8)
    synd[-2:] = (rng.random((2, m)) < 0.1).astype(np.uint8)  # inconsistent tail
    want, res = ora.decode_batch(synd)
    out = dev.decode_batch(synd)
    bad = np.flatnonzero((out != want).any(axis=1))
    assert bad.size == 0, f"shots {bad.tolist()} differ; exit classes {(dev.last_status[bad] & 0xFF).tolist()} vs {res['exit_class'][bad].tolist()}"
    assert np.array_equal(dev.last_iterations, res["bp_iteration"]) and np.array_equal(dev.last_min_pm, res["min_pm"])
    assert np.array_equal(dev.last_status & 0xFF, res["exit_class"])


@pytest.mark.parametrize("ring", ["8", "32"])
def test_column_form_elimination_with_a_short_ring(ring, monkeypatch):
    """osd0_colsw (large graphs, 256 < m <= 960) when a batch of 64 sorted columns holds more pivots than its ring has entries: the
    batch closes early and the next one starts behind its last pivot column (SWD_OWIDE_RING is read when the layout is made)"""
    monkeypatch.setenv("SWD_OWIDE_RING", ring)
    test_random_large_matrices_vs_oracle(900, 8600, 4, 3, 5)
    test_random_large_matrices_vs_oracle(512, 9000, 3, 8, 16)


def test_bb288_wide_windows_pipeline_vs_oracle():
    """[[288,12,18]] with (W,F) = (5,1): 720 x 6336 window matrices, 173 KB of messages -- the sliding-window pipeline on the
    large-graph kernels against the oracle driven through the host window loop"""
    from oracle import oracle as O
    from slidingwindowdecoder_amd import SlidingWindowDecoder
    from slidingwindowdecoder_amd.circuit import bb_dem
    from slidingwindowdecoder_amd.codes import bb_code
    from slidingwindowdecoder_amd.windows import plan_windows, sample_dem, sliding_window_decode_host
    code, A, B = bb_code(288)
    dem = bb_dem(code, A, B, 0.004, 7)
    plan = plan_windows(dem.chk, dem.obs, dem.priors, 144, 5, 1, method=1)
    assert plan.windows[1].mat.shape[0] == 720 and plan.windows[1].mat.nnz * 8 > 160 * 1024
    det, obs, _ = sample_dem(plan.chk, plan.obs, plan.priors, 10, seed=5)
    kw = dict(pre_max_iter=8, post_max_iter=60, ms_scaling_factor=1.0, osd_method="osd_cs", osd_order=0)
    dec = SlidingWindowDecoder(plan, **kw)
    assert dec.threads == 1024 and dec.lds_bytes < plan.windows[1].mat.nnz * 8  # the full graph's messages are not in LDS
    total = dec.decode(det)
    want, _ = sliding_window_decode_host(plan, det, lambda w: O.osd_window(w.mat, channel_probs=w.prior, **kw))
    assert np.array_equal(total, want)
    assert not dec.last_flagged.any()
