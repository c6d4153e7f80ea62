"""Stim-free detector error model of the SHYPS memory experiment (BASELINE config 5).

The reference builds a Stim circuit for the subsystem hypergraph-product simplex code
(/root/reference/src/build_SHYPS_circuit.py:9-191) and lets Stim derive the detector error model
(/root/reference/SHYPS.ipynb cell 1).  Stim is not available on the MI355X boxes; this module restates
the gate / noise / measurement schedule of the z-basis experiment as the flat op list of
``circuit.dem_from_ops`` (backward X-sensitivity sweep, Stim's per-component probabilities, merge of
equal symptoms), like ``circuit.bb_memory_ops`` does for the BB codes.

Known answers of the notebook (r = 3, 4 rounds): 7 x 7 circulant H of h(x) = 1 + x^2 + x^3, three CNOT
layers of 49 gates per basis, detector matrix 105 x 833, row weight 28..44, column weight 2..9, nine
observables -- asserted in tests/test_shyps.py.
"""
from __future__ import annotations

from collections import deque

import numpy as np

from . import gf2
from .circuit import CX, DEP1, DEP2, DETECTOR, M, MX, OBSERVABLE, R, RX, XERR, DEM, dem_from_ops

# h(x) = 1 + x^a + x^b with gcd(h, x^(2^r-1) - 1) primitive of degree r (build_SHYPS_circuit.py:13-18)
_PRIMITIVE = {3: (0, 2, 3), 4: (0, 3, 4), 5: (0, 2, 5)}


def _poly_div_gf2(num: list[int], den: list[int]) -> list[int]:
    """Quotient of num / den over GF(2); coefficient lists in increasing degree, exact division."""
    num = num[:]
    dn = len(den) - 1
    q = [0] * (len(num) - dn)
    for s in range(len(num) - 1 - dn, -1, -1):
        if num[s + dn]:
            q[s] = 1
            for i, c in enumerate(den):
                num[s + i] ^= c
    assert not any(num), "h(x) does not divide x^n - 1"
    return q


def simplex_matrices(r: int):
    """(H, G): the overcomplete n_r x n_r parity-check matrix (circulant of h) and the r x n_r generator
    matrix (shifts of g = (x^n_r - 1) / h) of the classical simplex code, n_r = 2^r - 1."""
    if r not in _PRIMITIVE:
        raise ValueError(f"unsupported r={r}")
    nr = 2 ** r - 1
    h = [0] * (max(_PRIMITIVE[r]) + 1)
    for e in _PRIMITIVE[r]:
        h[e] = 1
    first = np.zeros(nr, dtype=np.int64)
    first[:len(h)] = h
    H = np.array([np.roll(first, i) for i in range(nr)])
    xn = [1] + [0] * (nr - 1) + [1]  # x^nr - 1 == x^nr + 1
    g = _poly_div_gf2(xn, h)
    gfirst = np.zeros(nr, dtype=np.int64)
    gfirst[:len(g)] = g
    G = np.array([np.roll(gfirst, i) for i in range(r)])
    assert not (G @ H % 2).any()  # build_SHYPS_circuit.py:33
    return H, G


def shyps_stabilizers(r: int):
    """(S_X, S_Z) of the subsystem hypergraph-product simplex code: S_X = H^T (x) G, S_Z = G (x) H^T
    (build_SHYPS_circuit.py:37-45), both (n_r r) x n_r^2 -- the check matrices `bp4_osd(S_X, S_Z, ...)` takes for
    code-capacity decoding (r = 3: 21 x 49, row weight 12, column weight <= 9, rank 12 each)."""
    H, G = simplex_matrices(r)
    return (np.kron(H.T, G) % 2).astype(np.uint8), (np.kron(G, H.T) % 2).astype(np.uint8)


def maximum_matching(adj: dict[int, list[int]], left: list[int]) -> dict[int, int]:
    """Hopcroft-Karp maximum matching of a bipartite graph given as left -> ordered neighbour lists.
    Free left vertices are taken in `left` order and neighbours in list order, which fixes the matching
    (and through it the CNOT layers, utils.py:517-574)."""
    INF = 1 << 30
    mate_l: dict[int, int | None] = {u: None for u in left}
    mate_r: dict[int, int | None] = {}
    dist: dict[int | None, int] = {}

    def layers() -> bool:
        q = deque()
        for u in left:
            if mate_l[u] is None:
                dist[u] = 0
                q.append(u)
            else:
                dist[u] = INF
        dist[None] = INF
        while q:
            u = q.popleft()
            if dist[u] < dist[None]:
                for v in adj.get(u, ()):
                    w = mate_r.get(v)
                    if w is None:
                        dist[None] = dist[u] + 1
                    elif dist[w] == INF:
                        dist[w] = dist[u] + 1
                        q.append(w)
        return dist[None] != INF

    def augment(u) -> bool:
        if u is None:
            return True
        for v in adj.get(u, ()):
            w = mate_r.get(v)
            if w is None or (dist[w] == dist[u] + 1 and augment(w)):
                mate_l[u] = v
                mate_r[v] = u
                return True
        dist[u] = INF
        return False

    while layers():
        for u in left:
            if mate_l[u] is None:
                augment(u)
    return {u: v for u, v in mate_l.items() if v is not None}


def cnot_layers(adj_mat: np.ndarray) -> list[list[tuple[int, int]]]:
    """Edge colouring of the bipartite gauge/data graph by repeatedly removing a maximum matching
    (utils.py:577-623); layer c lists the (gauge, data) pairs of colour c."""
    nrow = adj_mat.shape[0]
    left = list(range(nrow))
    rest = {u: [int(v) for v in np.nonzero(adj_mat[u])[0]] for u in left}
    layers = []
    while any(rest[u] for u in left):
        mt = maximum_matching(rest, left)
        layers.append([(u, v) for u, v in mt.items()])
        for u, v in mt.items():
            rest[u].remove(v)
    return layers


def shyps_memory_ops(r: int, p: float, num_repeat: int):
    """Op list of the z-basis SHYPS memory circuit (build_SHYPS_circuit.py:96-189) and (S_Z, L_Z)."""
    H, G = simplex_matrices(r)
    nr = 2 ** r - 1
    N = nr * nr
    eye = np.identity(nr, dtype=np.int64)
    gauge_x, gauge_z = np.kron(H.T, eye), np.kron(eye, H.T)
    agg_z = np.kron(G, eye)                 # S_Z = agg_z @ gauge_z
    S_Z = np.kron(G, H.T)
    P = gf2.left_inverse(G.T)               # P G^T = I
    L_Z = np.kron(G, P) % 2
    lz, lx = cnot_layers(gauge_z), cnot_layers(gauge_x)
    assert len(lz) == 3 and len(lx) == 3
    XG, DT, ZG = 0, N, 2 * N                # qubit offsets: X gauge ancillas, data, Z gauge ancillas

    ops = []
    nmeas = 0
    zprev = None

    def block(repeat: bool):
        nonlocal nmeas, zprev
        if repeat:
            for i in range(N):
                ops.append((XERR, ZG + i, p))
                ops.append((DEP1, DT + i, p))
        for layer in lz:
            for zg, d in layer:
                ops.append((CX, DT + d, ZG + zg))
                ops.append((DEP2, DT + d, ZG + zg, p))
        zcur = []
        for i in range(N):
            ops.append((XERR, ZG + i, p))
            ops.append((M, ZG + i, nmeas))
            zcur.append(nmeas)
            nmeas += 1
        for row in agg_z:
            idx = np.nonzero(row)[0]
            ops.append((DETECTOR, [zcur[i] for i in idx] + ([zprev[i] for i in idx] if repeat else [])))
        zprev = zcur
        for i in range(N):
            ops.append((RX, XG + i))
        for layer in lx:
            for xg, d in layer:
                ops.append((CX, XG + xg, DT + d))
                ops.append((DEP2, XG + xg, DT + d, p))
        for i in range(N):
            ops.append((MX, XG + i, nmeas))
            nmeas += 1
        for i in range(N):
            ops.append((R, ZG + i))
            ops.append((XERR, ZG + i, p))

    for i in range(N):
        ops.append((RX, XG + i))
        ops.append((R, ZG + i))
        ops.append((XERR, ZG + i, p))
    for i in range(N):
        ops.append((R, DT + i))
        ops.append((XERR, DT + i, p))
    block(False)
    for _ in range(num_repeat - 1):
        block(True)
    dmeas = []
    for i in range(N):
        ops.append((XERR, DT + i, p))
        ops.append((M, DT + i, nmeas))
        dmeas.append(nmeas)
        nmeas += 1
    for k, row in enumerate(S_Z):
        ops.append((DETECTOR, [dmeas[i] for i in np.nonzero(row)[0]] + [zprev[i] for i in np.nonzero(agg_z[k])[0]]))
    for row in L_Z:
        ops.append((OBSERVABLE, [dmeas[i] for i in np.nonzero(row)[0]]))
    return ops


def shyps_dem(r: int, p: float, num_repeat: int) -> DEM:
    """(chk, obs, priors) of the z-basis SHYPS memory experiment -- counterpart of
    ``dem_to_check_matrices(build_SHYPS_circuit(r, p, num_repeat).detector_error_model())``."""
    return dem_from_ops(shyps_memory_ops(r, p, num_repeat))
