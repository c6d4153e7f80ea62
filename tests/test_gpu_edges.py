"""Edge cases of the C-ABI entry points: empty and single-shot batches, all-zero syndromes, the largest
matrices a kernel variant takes (matrices no variant takes: the general form, tests/test_gpu_huge.py)."""
import numpy as np
import pytest
import scipy.sparse as sp

from tests import fixtures as fx

pytestmark = pytest.mark.gpu


def _rand_h(rng, m, n, colw=3):
    rows = np.concatenate([rng.choice(m, size=colw, replace=False) for _ in range(n)])
    cols = np.repeat(np.arange(n), colw)
    H = sp.csr_matrix((np.ones(len(rows), np.uint8), (rows, cols)), shape=(m, n))
    H.data[:] = 1
    return H


def test_empty_and_single_shot_batches():
    from oracle import oracle as O
    from slidingwindowdecoder_amd import osd_window, bpgdg_decoder, bp4_osd, DemSampler
    rng = np.random.default_rng(3)
    H = _rand_h(rng, 24, 60).toarray()
    p = np.full(60, 0.02)
    dec = osd_window(H, channel_probs=p, pre_max_iter=4, post_max_iter=10, osd_method="osd_0")
    out = dec.decode_batch(np.zeros((0, 24), np.uint8))
    assert out.shape == (0, 60)
    one = (rng.random((1, 24)) < 0.2).astype(np.uint8)
    want, _ = O.osd_window(H, channel_probs=p, pre_max_iter=4, post_max_iter=10, osd_method="osd_0").decode_batch(one)
    assert np.array_equal(dec.decode_batch(one), want)
    # all-zero syndrome: the zero vector after one iteration (osd_window.pyx:473-485)
    z = dec.decode_batch(np.zeros((5, 24), np.uint8))
    assert not z.any() and (dec.last_iterations == 1).all()
    g = bpgdg_decoder(H, channel_probs=p, max_iter=6, max_iter_per_step=4, max_step=5, max_tree_depth=2, max_side_depth=4,
                      max_tree_branch_step=4, max_side_branch_step=4)
    assert g.decode_batch(np.zeros((0, 24), np.uint8)).shape == (0, 60)
    q = bp4_osd(H, H, channel_probs_x=p, channel_probs_y=p, channel_probs_z=p, max_iter=5)
    assert q.decode_batch(np.zeros((0, 24), np.uint8), np.zeros((0, 24), np.uint8)).shape == (0, 2, 60)
    s = DemSampler(sp.csr_matrix(H), sp.csr_matrix(np.ones((1, 60), np.uint8)), p)
    det, obs = s.sample(0)[:2]
    assert det.shape == (0, 24) and obs.shape[0] == 0


def test_empty_batch_pipeline():
    from slidingwindowdecoder_amd import SlidingWindowDecoder
    from tests.test_gpu_pipeline import load_plan
    f = fx.load("bb144_circuit_p003_w3f1.npz")
    plan = load_plan(f, 11)
    dec = SlidingWindowDecoder(plan, **fx.params(f, "osd0_params"))
    total = dec.decode(np.zeros((0, plan.chk.shape[0]), np.uint8))
    assert total.shape == (0, plan.chk.shape[1])
    one = dec.decode(fx.unpack(f["det"], plan.chk.shape[0])[:1])
    assert np.array_equal(one, fx.unpack(f["osd0_total"], plan.chk.shape[1])[:1])


@pytest.mark.parametrize("m,n", [(64, 256), (256, 1792), (640, 1600), (600, 2000)])
def test_largest_matrices_of_a_variant_vs_oracle(m, n):
    """m and n at the upper edge of a kernel variant (threads x variable nodes per thread)."""
    from oracle import oracle as O
    from slidingwindowdecoder_amd import osd_window
    rng = np.random.default_rng(m)
    H = _rand_h(rng, m, n, 3)
    if np.diff(H.indptr).max() > 60:
        pytest.skip("row degree above the bound")
    p = rng.uniform(0.001, 0.01, size=n)
    kw = dict(channel_probs=p, pre_max_iter=4, post_max_iter=8, ms_scaling_factor=0.9, osd_method="osd_0", new_n=min(n, 2 * m))
    dev, ora = osd_window(H, **kw), O.osd_window(H, **kw)
    B = 16 if m < 1024 else 4
    e = (rng.random((B, n)) < p * 2).astype(np.uint8)
    synd = (e @ H.T.toarray()) % 2
    want, res = ora.decode_batch(synd)
    out = dev.decode_batch(synd)
    assert np.array_equal(out, want)
    assert np.array_equal(dev.last_iterations, res["bp_iteration"]) and np.array_equal(dev.last_min_pm, res["min_pm"])


# (matrices beyond every kernel variant are no longer refused: tests/test_gpu_huge.py)
