"""Host-side logic that needs no GPU: the tree shape behind ``hypotheses=H`` and the layout of the bench's algorithmic-bytes helpers."""
import numpy as np
import pytest


def test_hypotheses_shape_covers_every_count_up_to_the_device_limit():
    """leaves of the decimation tree = 1 (main) + (S - D) side branches + 2 (2^D - 1) tree leaves (bpgd.cpp:576-577,
    bp_guessing_decoder.pyx:181); the deepest full tree that fits, at most depth 6 and 160 snapshots"""
    from slidingwindowdecoder_amd.decoders import hypotheses_shape
    assert hypotheses_shape(64) == (5, 6)  # BASELINE configs[2]: 64 hypotheses per shot
    for h in range(1, 162):
        D, S = hypotheses_shape(h)
        assert 0 <= D <= 6 and S >= D
        assert 1 + (S - D) + 2 * (2 ** D - 1) == h
        assert 2 * (2 ** D - 1) + (S - D) <= 160
        if D < 6:
            assert 2 * (2 ** (D + 1) - 1) + 1 > h  # no deeper full tree would fit
    for bad in (0, -3, 162, 1000):
        with pytest.raises(ValueError):
            hypotheses_shape(bad)


def test_bench_algorithmic_bytes_helpers():
    """32 B per live edge and iteration; the large-graph workload leaves the post-phase iterations out of its HBM figure"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import scipy.sparse as sp

    class W:  # a window with 3 checks, 5 nodes, 8 edges
        mat = sp.csr_matrix(np.array([[1, 1, 0, 1, 0], [0, 1, 1, 0, 0], [1, 0, 0, 1, 1]], np.uint8))
    class P:
        windows = [W]
    st = np.zeros((2, 1, 8), np.int32)
    st[0, 0] = [1, 12, 4, 8, 3, 2, 5, 0]   # exit post: 4 pre + 8 post iterations on 5 live edges
    st[1, 0] = [0, 2, 2, 0, 0, 0, 0, 0]    # exit pre after 2 iterations
    assert bench.lds_algorithmic_bytes(P, st) == 32.0 * 8 * (4 + 2) + 32.0 * 5 * 8
    full = 40.0 * 8 + 17.0 * 5 + 2.0 * 3
    short = 40.0 * 5 + 17.0 * 3 + 2.0 * 2
    a_all = bench.algorithmic_bytes(P, st, 8)
    a_pre = bench.algorithmic_bytes(P, st, 8, post_in_lds=True)
    assert a_all - a_pre == 8 * short
    assert a_pre == (4 + 2) * full + 16.0 * 5 + 2 * (3 + 5)
    lo, up = bench.gdg_lds_algorithmic_bytes(P, st)
    assert lo == 32.0 * 8 * 6 and up >= lo
