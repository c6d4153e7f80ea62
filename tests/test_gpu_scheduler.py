"""The work-unit scheduler (one workgroup per (window, shot), window-major tickets on a persistent grid) must
give every shot the result it gets alone: odd batch sizes, batches smaller and larger than the grid, repeated
launches on one plan."""
import numpy as np
import pytest

from tests import fixtures as fx
from tests.test_gpu_pipeline import load_plan

pytestmark = pytest.mark.gpu


def test_results_do_not_depend_on_batching():
    from slidingwindowdecoder_amd import SlidingWindowDecoder
    f = fx.load("bb144_circuit_p003_w3f1.npz")
    plan = load_plan(f, 11)
    kw = fx.params(f, "osd0_params")
    det = fx.unpack(f["det"], plan.chk.shape[0])
    want = fx.unpack(f["osd0_total"], plan.chk.shape[1])
    dec = SlidingWindowDecoder(plan, **kw)
    for size in (1, 3, 97, len(det)):
        got = dec.decode(det[:size])
        assert np.array_equal(got, want[:size]), f"batch of {size}"
    # a batch much larger than the persistent grid (2 workgroups per CU): every shot repeated
    big = np.tile(det, (12, 1))
    got = dec.decode(big)
    assert np.array_equal(got, np.tile(want, (12, 1)))
    flips = dec.last_obs_flips.reshape(12, -1)
    assert (flips == flips[0]).all()
