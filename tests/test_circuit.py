"""Host-side problem construction against the reference's notebook known-answers
 and against the committed fixtures."""
import numpy as np
import pytest
import scipy.sparse as sp

from slidingwindowdecoder_amd import gf2
from slidingwindowdecoder_amd.circuit import bb_dem
from slidingwindowdecoder_amd.codes import bb_code
from slidingwindowdecoder_amd.windows import plan_windows
from tests import fixtures as fx


@pytest.fixture(scope="module")
def dem144():
    code, A, B = bb_code(144)
    return code, bb_dem(code, A, B, 0.003, 12)


def test_bb_codes_match_reference_matrices():
    # hx / hz recorded from the reference's create_bivariate_bicycle_codes (codes_q.py:235-246)
    f72 = fx.load("bb72_capacity.npz")
    code, _, _ = bb_code(72)
    assert np.array_equal(code.hx, f72["hx"]) and np.array_equal(code.hz, f72["hz"])
    f288 = fx.load("bb288_circuit_p005_w4f1.npz")
    code, _, _ = bb_code(288)
    assert np.array_equal(code.hx, f288["hx"])
    assert (code.N, code.K) == (288, 12)


def test_logicals_are_valid():
    code, _, _ = bb_code(144)
    assert code.K == 12 and code.lz.shape == (12, 144)
    assert not ((code.hx.astype(int) @ code.lz.T) % 2).any()       # commute with X checks
    stacked = np.vstack((code.hz, code.lz))
    assert gf2.rank(stacked) == gf2.rank(code.hz) + 12             # independent of stabilisers
    assert gf2.rank((code.lx.astype(int) @ code.lz.T) % 2) == 12   # pair up with lx


def test_dem_structure_known_answers(dem144):
    _, dem = dem144
    # IBM.ipynb:165  -> (936, 8784); Round Analysis.ipynb:207-208 -> 864 = 144+216+504 faults/round
    assert dem.chk.shape == (936, 8784) and dem.chk.nnz == 30672
    cw = np.asarray(dem.chk.sum(axis=0)).ravel()
    assert {int(k): int((cw == k).sum()) for k in np.unique(cw)} == {2: 864, 3: 5328, 4: 864, 5: 864, 6: 864}
    rw = np.asarray(dem.chk.sum(axis=1)).ravel()
    assert rw.min() == 16 and rw.max() == 35
    assert dem.obs.shape == (12, 8784)
    assert len(np.unique(np.round(dem.priors, 13))) == 8


@pytest.mark.parametrize("p,expect", [(0.003, 0.027499817877069083), (0.004, 0.036622121785736664),
                                      (0.005, 0.04572241379526658)])
def test_noisy_syndrome_prior_known_answers(p, expect):
    # Sliding Window OSD.ipynb:665 / 293 / 471 "prior for noisy syndrome"
    code, A, B = bb_code(144)
    dem = bb_dem(code, A, B, p, 4)
    plan = plan_windows(dem.chk, dem.obs, dem.priors, 72, 3, 1, method=1)
    assert plan.noisy_prior == pytest.approx(expect, rel=0, abs=2e-16)


def test_window_geometry_known_answers(dem144):
    _, dem = dem144
    plan = plan_windows(dem.chk, dem.obs, dem.priors, 72, 3, 1, method=1)
    # Round Analysis.ipynb:322 anchors
    assert plan.anchors[:6] == [(0, 0), (72, 648), (144, 1368), (216, 2088), (288, 2808), (360, 3528)]
    assert plan.anchors[-1] == (936, 8784)
    assert len(plan.windows) == 11
    shapes = [w.mat.shape for w in plan.windows]
    assert shapes[0] == (216, 1656) and shapes[-1] == (216, 1656) and set(shapes[1:-1]) == {(216, 1728)}
    assert [w.mat.nnz for w in plan.windows] == [5544] + [5976] * 9 + [5904]
    assert [w.commit for w in plan.windows] == [648] + [720] * 9 + [1656]
    # mid windows are translates of one another
    a, b = plan.windows[3], plan.windows[7]
    assert (a.mat != b.mat).nnz == 0 and np.array_equal(a.prior, b.prior)


def test_dem_matches_committed_fixture(dem144):
    """The fixture's matrices are what the reference decoders were run on; the generator must
    keep producing exactly those (column order included)."""
    _, dem = dem144
    plan = plan_windows(dem.chk, dem.obs, dem.priors, 72, 3, 1, method=1)
    f = fx.load("bb144_circuit_p003_w3f1.npz")
    chk, priors = fx.graph(f, "chk_")
    assert (sp.csr_matrix(plan.chk) != chk).nnz == 0
    assert np.array_equal(plan.priors, priors)
    for wi, w in enumerate(plan.windows):
        mat, pr = fx.graph(f, f"win{wi}_")
        assert (w.mat != mat).nnz == 0 and np.array_equal(w.prior, pr)
        assert list(f[f"win{wi}_meta"]) == [w.row0, w.row1, w.col0, w.ncols_global, w.commit, int(w.is_last)]


def test_bb288_window_shapes():
    code, A, B = bb_code(288)
    dem = bb_dem(code, A, B, 0.005, 6)
    plan = plan_windows(dem.chk, dem.obs, dem.priors, 144, 4, 1, method=1)
    assert [w.mat.shape for w in plan.windows] == [(576, 4752), (576, 4896), (576, 4896), (576, 4752)]
    assert plan.windows[1].mat.nnz == 16992


def test_merged_prior_is_per_row_like_the_reference():
    """osd.py:79-89 sums the replaced faults' priors per row of the identity block: a vector.  Uniform for the BB
    circuits, not in general; a caller-given scalar is broadcast."""
    import scipy.sparse as sp
    from slidingwindowdecoder_amd.windows import plan_windows
    rng = np.random.default_rng(2)
    h, R = 4, 4
    cols = []
    for r in range(R):          # per round: 3h local faults (one per detector, three times), then h faults into the next round
        for rep in range(3):
            for i in range(h):
                cols.append([r * h + i])
        if r < R - 1:
            for i in range(h):
                cols.append([r * h + i, (r + 1) * h + (i + 1) % h])
    rows = np.concatenate(cols)
    cidx = np.concatenate([[j] * len(c) for j, c in enumerate(cols)])
    chk = sp.csr_matrix((np.ones(len(rows), np.uint8), (rows, cidx)), shape=(R * h, len(cols)))
    priors = rng.uniform(0.001, 0.01, size=len(cols))
    obs = sp.csr_matrix((1, len(cols)), dtype=np.uint8)
    plan = plan_windows(chk, obs, priors, h, 2, 1, method=1)
    w0 = plan.windows[0]
    tail = w0.prior[-h:]
    a1, b1 = plan.anchors[1], plan.anchors[2]
    block = sp.csr_matrix(plan.chk)[a1[0]:b1[0], a1[1] + 3 * h:b1[1]]
    want = np.asarray(block.multiply(plan.priors[a1[1] + 3 * h:b1[1]]).sum(axis=1)).ravel()
    assert np.allclose(tail, want) and len(np.unique(np.round(tail, 12))) > 1
    plan2 = plan_windows(chk, obs, priors, h, 2, 1, method=1, noisy_prior=0.02)
    assert np.allclose(plan2.windows[0].prior[-h:], 0.02)


def test_x_basis_experiment_known_answer():
    """`Sliding Window OSD.ipynb` (x-basis run: N = 144, p = 0.004, 12 rounds, (W,F) = (5,2), method 1, z_basis=False) prints
    "prior for noisy syndrome 0.05900506726184526".  The x-basis windows are cut `c[1] + n` columns into a region
    (/root/reference/osd.py:83, 105), so the value depends on Stim's column order inside a region; the general two-sensitivity
    sweep + the reference's column order (bb_dem's default, column_order="stim") reproduce it to the last digit."""
    code, A, B = bb_code(144)
    dem = bb_dem(code, A, B, 0.004, 12, z_basis=False)
    assert dem.chk.shape == (936, 8784) and dem.obs.shape == (12, 8784)
    plan = plan_windows(dem.chk, dem.obs, dem.priors, 72, 5, 2, method=1, z_basis=False)
    assert plan.noisy_prior == 0.05900506726184526
    assert [w.mat.shape for w in plan.windows] == [(360, 3096)] * 4 + [(360, 3024)]
    # the x-basis observables are the X logicals: every mechanism's observable flips follow from lx
    assert not ((code.hz.astype(int) @ code.lx.T) % 2).any()


def test_default_is_the_reference_column_order_and_prior_merge():
    """bb_dem's default (both bases) is the reference's dem_to_check_matrices order and its SUMMED priors across circuit units
    (column_order="stim", build_circuit.py:251-299): the three notebook priors come out of the DEFAULT to the last digit
    (test_noisy_syndrome_prior_known_answers) and the committed fixtures hold exactly these priors (test_dem_matches_committed_fixture).
    "circuit" (this module's own order of rounds 1-4) stays available: the same mechanisms in another order."""
    code, A, B = bb_code(144)
    p, expect = 0.003, 0.027499817877069083
    d = bb_dem(code, A, B, p, 12)
    s = bb_dem(code, A, B, p, 12, column_order="stim")
    assert (sp.csc_matrix(d.chk) != sp.csc_matrix(s.chk)).nnz == 0 and np.array_equal(d.priors, s.priors)
    plan = plan_windows(d.chk, d.obs, d.priors, 72, 3, 1, method=1)
    assert plan.noisy_prior == expect
    c = bb_dem(code, A, B, p, 12, column_order="circuit")
    assert c.chk.shape == (936, 8784)
    key = lambda m: sorted(tuple(m.indices[m.indptr[j]:m.indptr[j + 1]]) for j in range(m.shape[1]))
    assert key(sp.csc_matrix(d.chk)) == key(sp.csc_matrix(c.chk))  # the same mechanisms, another order
    assert 0.0 <= np.sort(d.priors)[-1] - np.sort(c.priors)[-1] < 1e-4
    with pytest.raises(ValueError):
        bb_dem(code, A, B, p, 12, column_order="other")


def test_z_basis_prior_merge_modes():
    """"circuit" XOR-merges a mechanism's probabilities over the whole circuit, the reference's dem_to_check_matrices ADDS them
    across the units of the circuit (the default "stim" mode, build_circuit.py:262-270).  Mechanism by mechanism (matched by
    symptom): the summed prior is never smaller, the gap is O(p^2) -- what rounds 1-4's fixtures and bench inputs deviated by."""
    code, A, B = bb_code(144)
    p = 0.003
    d, c = bb_dem(code, A, B, p, 12), bb_dem(code, A, B, p, 12, column_order="circuit")

    def by_symptom(dem):
        chk, obs = sp.csc_matrix(dem.chk), sp.csc_matrix(dem.obs)
        return {(tuple(chk.indices[chk.indptr[j]:chk.indptr[j + 1]]), tuple(obs.indices[obs.indptr[j]:obs.indptr[j + 1]])): dem.priors[j]
                for j in range(chk.shape[1])}
    ps, pc = by_symptom(d), by_symptom(c)
    assert ps.keys() == pc.keys() and len(ps) == 8784
    gap = np.array([ps[k] - pc[k] for k in ps])
    assert (gap >= -1e-18).all()
    assert 0.0 < gap.max() < 4.2e-5, gap.max()          # 2 p q for the largest pair of unit probabilities
    assert (gap > 1e-12).sum() < 0.45 * len(gap)        # mechanisms emitted by one unit only have identical priors
    assert gap.max() / min(pc.values()) < 0.11          # ... and the relative deviation stays below 11 % of the smallest prior
