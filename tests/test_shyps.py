"""Known answers of the reference's SHYPS notebook (/root/reference/SHYPS.ipynb cell 2, r=3, 4 rounds, p=1e-3):
printed H and G, 49 gates per CNOT layer, detector matrix (105, 833), row weight 28..44, column weight 2..9,
9 observables."""
import numpy as np

from slidingwindowdecoder_amd import gf2, shyps


def test_simplex_matrices_match_notebook_printout():
    H, G = shyps.simplex_matrices(3)
    assert H.tolist() == [[1, 0, 1, 1, 0, 0, 0], [0, 1, 0, 1, 1, 0, 0], [0, 0, 1, 0, 1, 1, 0], [0, 0, 0, 1, 0, 1, 1],
                          [1, 0, 0, 0, 1, 0, 1], [1, 1, 0, 0, 0, 1, 0], [0, 1, 1, 0, 0, 0, 1]]
    assert G.tolist() == [[1, 0, 1, 1, 1, 0, 0], [0, 1, 0, 1, 1, 1, 0], [0, 0, 1, 0, 1, 1, 1]]
    P = gf2.left_inverse(G.T)
    assert np.array_equal(P @ G.T % 2, np.identity(3, dtype=np.int64))


def test_cnot_layers():
    H, _ = shyps.simplex_matrices(3)
    eye = np.identity(7, dtype=np.int64)
    for gauge in (np.kron(eye, H.T), np.kron(H.T, eye)):
        layers = shyps.cnot_layers(gauge)
        assert [len(x) for x in layers] == [49, 49, 49]
        seen = set()
        for layer in layers:  # a layer touches every gauge ancilla and every data qubit once
            assert len({u for u, _ in layer}) == 49 and len({v for _, v in layer}) == 49
            seen |= set(layer)
        assert len(seen) == 147 and all(gauge[u, v] for u, v in seen)


def test_shyps_dem_structure():
    dem = shyps.shyps_dem(3, 0.001, 4)
    chk = dem.chk.toarray()
    assert chk.shape == (105, 833)
    rw, cw = chk.sum(axis=1), chk.sum(axis=0)
    assert (rw.max(), cw.max(), rw.min(), cw.min()) == (44, 9, 28, 2)
    assert dem.obs.shape == (9, 833)
    assert (dem.priors > 0).all() and (dem.priors < 0.02).all()
    # every mechanism is detectable (no undetectable logical fault of weight one at this distance)
    assert (cw > 0).all()
