"""The numpy Philox4x32-10 used to check the device sampler, against the Random123 known answers."""
import numpy as np

from tests import philox_ref


def test_philox4x32_10_known_answers():
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = philox_ref.philox4x32_10(*ctr, *key).ravel()
        assert [int(x) for x in got] == list(want)


def test_sample_faults_rates():
    p = np.array([0.0, 1.0, 0.25, 0.003] * 50)
    f = philox_ref.sample_faults(p, 4000, seed=9)
    assert f[:, 0::4].sum() == 0 and (f[:, 1::4] == 1).all()
    assert abs(f[:, 2::4].mean() - 0.25) < 0.01 and abs(f[:, 3::4].mean() - 0.003) < 0.001
