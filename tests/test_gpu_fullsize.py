"""Full-size properties that need no oracle (BASELINE sizes: 4096+ shots x 11 windows): the committed faults
explain the detector data of every unflagged shot, no shot is flagged, and the logical error rate per round lands
where the reference's notebook reports it for the same decoder (`Sliding Window OSD.ipynb:369-372,402`:
[[144,12,12]], p = 0.004, (3,1), osd_window with OSD-CS 10 -> 1.54e-3 per round over 10 000 shots)."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def test_bb144_p004_full_batch_properties():
    import bench
    from slidingwindowdecoder_amd import DemSampler, SlidingWindowDecoder
    plan = bench.build_problem(p=0.004)
    shots = 16384
    det, obs = DemSampler(plan.chk, plan.obs, plan.priors).sample(shots, seed=11)
    dec = SlidingWindowDecoder(plan, pre_max_iter=8, post_max_iter=200, ms_scaling_factor=1.0, osd_method="osd_cs", osd_order=10)
    total = dec.decode(det)
    # residual syndrome of the whole run (osd.py:184-185): zero for every shot, like in the reference's runs
    resid = (sp.csr_matrix(total) @ sp.csr_matrix(plan.chk).T.astype(np.int32)).toarray() % 2 != det
    assert not resid.any()
    assert not dec.last_flagged.any()
    pred = (sp.csr_matrix(total) @ sp.csr_matrix(plan.obs).T.astype(np.int32)).toarray() % 2
    logical = (pred != obs).any(axis=1)
    mask = (obs.astype(np.uint32) << np.arange(obs.shape[1], dtype=np.uint32)).sum(axis=1).astype(np.uint32)
    assert np.array_equal(logical, dec.last_obs_flips != mask)  # the device's own accounting
    per_round = 1.0 - (1.0 - logical.mean()) ** (1.0 / 12)
    assert 1.1e-3 < per_round < 2.0e-3, per_round
