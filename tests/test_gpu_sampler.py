"""Device DEM sampler vs its numpy restatement (tests/philox_ref.py) and vs the model's statistics."""
import numpy as np
import pytest
import scipy.sparse as sp

from tests import philox_ref

pytestmark = pytest.mark.gpu


def _plan():
    import bench
    return bench.build_problem()


def test_sampler_matches_philox_restatement():
    from slidingwindowdecoder_amd import DemSampler
    plan = _plan()
    s = DemSampler(plan.chk, plan.obs, plan.priors)
    det, obs, faults = s.sample(300, seed=20240318, first_shot=5, return_faults=True)
    want = philox_ref.sample_faults(plan.priors, 300, 20240318, first_shot=5)
    assert np.array_equal(faults, want)
    chk, ob = sp.csr_matrix(plan.chk).astype(np.int32), sp.csr_matrix(plan.obs).astype(np.int32)
    assert np.array_equal(det, (sp.csr_matrix(want) @ chk.T).toarray() % 2)
    assert np.array_equal(obs, (sp.csr_matrix(want) @ ob.T).toarray() % 2)
    # cutting the batch differently gives the same shots
    det2, obs2 = s.sample(100, seed=20240318, first_shot=105)
    assert np.array_equal(det2, det[100:200]) and np.array_equal(obs2, obs[100:200])


def test_sampler_statistics_and_device_path():
    import torch
    from slidingwindowdecoder_amd import DemSampler
    plan = _plan()
    s = DemSampler(plan.chk, plan.obs, plan.priors)
    det, flips = s.sample_device(8192, seed=3)
    torch.cuda.synchronize()
    d = det.cpu().numpy()
    _, _, faults = s.sample(8192, seed=3, return_faults=True)
    assert np.array_equal(d, s.sample(8192, seed=3)[0])
    # mean 32.5 faults and 87.7 fired detectors per shot at p = 0.003
    assert abs(faults.sum(axis=1).mean() - plan.priors.sum()) < 0.3
    assert abs(d.sum(axis=1).mean() - 87.7) < 1.5
    # per-column rates within 5 sigma of the priors
    rate, sig = faults.mean(axis=0), np.sqrt(plan.priors * (1 - plan.priors) / 8192)
    assert (np.abs(rate - plan.priors) < 5 * sig + 1e-9).all()


def test_sampler_errors():
    from slidingwindowdecoder_amd import DemSampler
    plan = _plan()
    with pytest.raises(ValueError):
        DemSampler(plan.chk, plan.obs[:, :-1], plan.priors)
