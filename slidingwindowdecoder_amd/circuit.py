"""Stim-free detector-error-model (DEM) derivation for the BB memory circuit.

The reference builds a Stim circuit (/root/reference/src/build_circuit.py:6-234), asks
Stim for its detector error model and turns that into (chk, obs, priors)
(/root/reference/src/build_circuit.py:251-299).  Stim is not available on the MI355X
boxes, so this module derives the same model directly:

* ``bb_memory_ops`` restates the gate / noise / measurement schedule of the reference
  circuit as a flat op list (``use_both=False``, ``HZH=False``; ``z_basis=True`` -- the variant
  every BASELINE config uses -- or ``z_basis=False``, build_circuit.py:147-152, 167-172, 204-221).
* ``dem_from_ops`` sweeps that list BACKWARDS keeping, per qubit, the set of detectors
  and observables an X flip and a Z flip at that point of time would toggle (two
  "sensitivities"): CX c,t -> sx[c] ^= sx[t], sz[t] ^= sz[c]; H swaps them; MR -> sx = this
  measurement's detectors, sz = 0; M -> sx ^= ...; MRX / MX the same with the roles swapped;
  R / RX -> both 0.  Every noise channel then contributes independent fault mechanisms with
  Stim's per-component probabilities (DEPOLARIZE1: (1-sqrt(1-4p/3))/2 for each of X,Y,Z;
  DEPOLARIZE2: (1-(1-16p/15)^(1/8))/2 for each of the 15 Paulis; a Y flips what X and Z flip)
  and mechanisms with identical symptoms are merged with p <- p(1-q) + q(1-p).

Host-side, runs once per experiment.  The structural known answers of the reference's
notebooks (shape 936 x 8784 for 12 rounds of the [[144,12,12]] code, column-weight
histogram, window anchors, merged "noisy syndrome" prior 0.027499817877069083 at p=0.003)
are asserted in tests/test_circuit.py.  Column order and the merging of a symptom's
probabilities follow the reference's ``dem_to_check_matrices`` by default (``bb_dem``,
``column_order="stim"``); "circuit" keeps this module's own order of rounds 1-4.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp

# op codes
R, RX, H, CX, M, MR, MRX, XERR, DEP1, DEP2, DETECTOR, OBSERVABLE, MX, ZERR = range(14)


def bb_memory_ops(code, A_list, B_list, p: float, num_repeat: int, z_basis: bool = True, return_segments: bool = False):
    """Op list of the BB memory circuit (z- or x-basis experiment).  Each op is a tuple
    (kind, qubits..., prob) ; DETECTOR/OBSERVABLE carry absolute measurement indices.
    ``return_segments``: also the segment of every op -- 0 for the initialisation and the encoding round, k for the k-th
    repetition of the round block (the reference appends it as ``(num_repeat - 1) * rep_circuit``, a REPEAT block), the last
    value for the final data measurement: the units in which Stim's error analysis emits its mechanisms (dem_from_ops)."""
    n = code.N
    h = n // 2
    a1, a2, a3 = A_list
    b1, b2, b3 = B_list
    A1, A2, A3 = (t.col_of_row() for t in (a1, a2, a3))
    B1, B2, B3 = (t.col_of_row() for t in (b1, b2, b3))
    A1T, A2T, A3T = (t.row_of_col() for t in (a1, a2, a3))
    B1T, B2T, B3T = (t.row_of_col() for t in (b1, b2, b3))
    XC, LD, RD, ZC = 0, h, n, 3 * h  # qubit offsets: X ancillas, L data, R data, Z ancillas

    ops = []
    seg_start = {}  # segment -> index of its first op
    nmeas = 0
    zmeas_prev = xmeas_prev = None

    def cx(c, t):
        ops.append((CX, c, t))
        ops.append((DEP2, c, t, p))

    for i in range(h):
        ops.append((R, XC + i))
        ops.append((R, ZC + i))
        ops.append((XERR, XC + i, p))
        ops.append((XERR, ZC + i, p))
    for i in range(n):
        ops.append((R if z_basis else RX, LD + i))
        ops.append((XERR if z_basis else ZERR, LD + i, p))

    for rnd in range(num_repeat):
        repeat = rnd > 0
        if repeat:
            seg_start[rnd] = len(ops)
        if repeat:
            for i in range(h):
                ops.append((XERR, ZC + i, p))
                ops.append((ZERR, XC + i, p))  # (invisible to Z-basis detectors)
                ops.append((DEP1, RD + i, p))
        else:
            for i in range(h):
                ops.append((H, XC + i))
        for i in range(h):  # layer 1
            cx(RD + A1T[i], ZC + i)
            ops.append((DEP1, LD + i, p))
        for i in range(h):  # layer 2
            cx(XC + i, LD + A2[i])
            cx(RD + A3T[i], ZC + i)
        for i in range(h):  # layer 3
            cx(XC + i, RD + B2[i])
            cx(LD + B1T[i], ZC + i)
        for i in range(h):  # layer 4
            cx(XC + i, RD + B1[i])
            cx(LD + B2T[i], ZC + i)
        for i in range(h):  # layer 5
            cx(XC + i, RD + B3[i])
            cx(LD + B3T[i], ZC + i)
        for i in range(h):  # layer 6
            cx(XC + i, LD + A1[i])
            cx(RD + A2T[i], ZC + i)
        zmeas = []
        for i in range(h):  # layer 7
            cx(XC + i, LD + A3[i])
            ops.append((XERR, ZC + i, p))
            ops.append((MR, ZC + i, nmeas))
            zmeas.append(nmeas)
            nmeas += 1
        if z_basis:
            for i in range(h):
                ops.append((DETECTOR, (zmeas[i], zmeas_prev[i]) if repeat else (zmeas[i],)))
        zmeas_prev = zmeas
        xmeas = []
        for i in range(h):  # layer 8
            ops.append((ZERR, XC + i, p))
            ops.append((MRX, XC + i, nmeas))
            xmeas.append(nmeas)
            nmeas += 1
        if not z_basis:
            for i in range(h):
                ops.append((DETECTOR, (xmeas[i], xmeas_prev[i]) if repeat else (xmeas[i],)))
        xmeas_prev = xmeas

    seg_start[num_repeat] = len(ops)
    dmeas = []
    for i in range(n):
        ops.append((M if z_basis else MX, LD + i, nmeas))
        dmeas.append(nmeas)
        nmeas += 1
    last = zmeas_prev if z_basis else xmeas_prev
    for i, row in enumerate(code.hz if z_basis else code.hx):
        ops.append((DETECTOR, tuple(dmeas[j] for j in np.flatnonzero(row)) + (last[i],)))
    for i, row in enumerate(code.lz if z_basis else code.lx):
        ops.append((OBSERVABLE, tuple(dmeas[j] for j in np.flatnonzero(row))))
    if not return_segments:
        return ops
    seg = np.zeros(len(ops), dtype=np.int32)
    for k, i0 in seg_start.items():
        seg[i0:] = k
    return ops, seg


@dataclass
class DEM:
    chk: sp.csc_matrix  # detectors x mechanisms
    obs: sp.csc_matrix  # observables x mechanisms
    priors: np.ndarray


def dem_from_ops(ops, segments=None) -> DEM:
    """``segments is None``: columns in circuit order of first appearance, equal symptoms merged with p <- p(1-q) + q(1-p)
    (bb_dem's "circuit" mode: the order the z-basis fixtures of rounds 1-4 were made with).
    ``segments`` (bb_memory_ops(..., return_segments=True)): the column order and priors of
    ``dem_to_check_matrices(circuit.detector_error_model())`` (/root/reference/src/build_circuit.py:251-299) -- Stim analyses
    the circuit backwards and flushes its mechanisms per unit (what precedes the REPEAT block, every iteration of it, what
    follows), merged and sorted by symptom (detector ids ascending, observables last) inside a unit; the reference then walks
    the flattened model and ADDS the probabilities of a symptom that several units emit (:262-270).  The x-basis windows of
    osd.py:83, 105 cut INSIDE a region of columns (``c[1] + n``), so there the order matters: with it the notebook's
    "prior for noisy syndrome 0.05900506726184526" (`Sliding Window OSD.ipynb`, x-basis (5,2) run) comes out to the last digit."""
    ndet = sum(1 for o in ops if o[0] == DETECTOR)
    nobs = sum(1 for o in ops if o[0] == OBSERVABLE)
    nq = 0
    for o in ops:
        if o[0] in (CX, DEP2):
            nq = max(nq, o[1] + 1, o[2] + 1)
        elif o[0] not in (DETECTOR, OBSERVABLE):
            nq = max(nq, o[1] + 1)
    # measurement -> bitmask over [detectors | observables]
    meas_mask: dict[int, int] = {}
    d = 0
    k = 0
    for o in ops:
        if o[0] == DETECTOR:
            for mi in o[1]:
                meas_mask[mi] = meas_mask.get(mi, 0) ^ (1 << d)
            d += 1
        elif o[0] == OBSERVABLE:
            for mi in o[1]:
                meas_mask[mi] = meas_mask.get(mi, 0) ^ (1 << (ndet + k))
            k += 1

    sx = [0] * nq  # detectors / observables an X flip on the qubit would toggle from here on
    sz = [0] * nq  # ... a Z flip
    mech: dict[int, float] = {}  # symptom bitmask -> probability
    order: list[int] = []

    cur_seg = 0
    seg_mech: dict[tuple, float] = {}  # (segment, symptom) -> probability (Stim order)

    def emit(sym: int, q: float) -> None:
        if sym == 0:
            return
        if segments is not None:
            key = (cur_seg, sym)
            pp = seg_mech.get(key)
            seg_mech[key] = q if pp is None else pp * (1.0 - q) + q * (1.0 - pp)
            return
        if sym in mech:
            pp = mech[sym]
            mech[sym] = pp * (1.0 - q) + q * (1.0 - pp)
        else:
            mech[sym] = q
            order.append(sym)

    # the 15 two-qubit Paulis, (x-part, z-part) per qubit; grouped by their X pattern (first, second, both, none) so that a
    # circuit whose detectors only see X flips yields its mechanisms in the order of the z-basis derivation of round 1
    I_, X_, Y_, Z_ = (0, 0), (1, 0), (1, 1), (0, 1)
    PAIRS = [(X_, I_), (Y_, I_), (X_, Z_), (Y_, Z_), (I_, X_), (I_, Y_), (Z_, X_), (Z_, Y_),
             (X_, X_), (X_, Y_), (Y_, X_), (Y_, Y_), (Z_, I_), (I_, Z_), (Z_, Z_)]

    for idx in range(len(ops) - 1, -1, -1):
        o = ops[idx]
        kind = o[0]
        if segments is not None:
            cur_seg = int(segments[idx])
        if kind == CX:
            sx[o[1]] ^= sx[o[2]]
            sz[o[2]] ^= sz[o[1]]
        elif kind == DEP2:
            q = 0.5 * (1.0 - (1.0 - 16.0 * o[3] / 15.0) ** 0.125)
            a, b = o[1], o[2]
            for (xa, za), (xb, zb) in PAIRS:
                emit((sx[a] if xa else 0) ^ (sz[a] if za else 0) ^ (sx[b] if xb else 0) ^ (sz[b] if zb else 0), q)
        elif kind == DEP1:
            q = 0.5 * (1.0 - (1.0 - 4.0 * o[2] / 3.0) ** 0.5)
            emit(sx[o[1]], q)             # X
            emit(sx[o[1]] ^ sz[o[1]], q)  # Y
            emit(sz[o[1]], q)             # Z
        elif kind == XERR:
            emit(sx[o[1]], o[2])
        elif kind == ZERR:
            emit(sz[o[1]], o[2])
        elif kind == H:
            sx[o[1]], sz[o[1]] = sz[o[1]], sx[o[1]]
        elif kind == MR:
            sx[o[1]], sz[o[1]] = meas_mask.get(o[2], 0), 0
        elif kind == M:
            sx[o[1]] ^= meas_mask.get(o[2], 0)
        elif kind == MRX:
            sx[o[1]], sz[o[1]] = 0, meas_mask.get(o[2], 0)
        elif kind == MX:
            sz[o[1]] ^= meas_mask.get(o[2], 0)
        elif kind in (R, RX):
            sx[o[1]] = sz[o[1]] = 0
    dmask = (1 << ndet) - 1
    if segments is None:
        order.reverse()  # forward-circuit order of first appearance
    else:
        def sort_key(sym: int):
            ds, v = [], sym & dmask
            while v:
                low = v & -v
                ds.append(low.bit_length() - 1)
                v ^= low
            v = sym >> ndet
            while v:
                low = v & -v
                ds.append((1 << 40) + low.bit_length() - 1)
                v ^= low
            return tuple(ds)
        first: dict[int, int] = {}
        for (sg, sym), q in seg_mech.items():
            if sym not in first or sg < first[sym]:
                first[sym] = sg
            mech[sym] = mech.get(sym, 0.0) + q
        order = sorted(first, key=lambda sym: (first[sym], sort_key(sym)))

    rows, cols, orow, ocol = [], [], [], []
    for j, sym in enumerate(order):
        v = sym & dmask
        while v:
            low = v & -v
            rows.append(low.bit_length() - 1)
            cols.append(j)
            v ^= low
        v = sym >> ndet
        while v:
            low = v & -v
            orow.append(low.bit_length() - 1)
            ocol.append(j)
            v ^= low
    nm = len(order)
    chk = sp.csc_matrix((np.ones(len(rows), np.uint8), (rows, cols)), shape=(ndet, nm))
    obs = sp.csc_matrix((np.ones(len(orow), np.uint8), (orow, ocol)), shape=(nobs, nm))
    priors = np.array([mech[s] for s in order], dtype=np.float64)
    return DEM(chk, obs, priors)


def bb_dem(code, A_list, B_list, p: float, num_repeat: int, z_basis: bool = True, column_order: str = "stim") -> DEM:
    """(chk, obs, priors) of the BB memory experiment -- counterpart of
    ``dem_to_check_matrices(build_circuit(..., z_basis=z_basis).detector_error_model())``
    (/root/reference/osd.py:35-37).  ``column_order``:
    "stim" (default, both bases): the reference's own column order and priors -- mechanisms flushed per unit of the circuit (what
    precedes the REPEAT block, each iteration, what follows), sorted by symptom inside a unit, faults of one unit merged by
    p(1-q) + q(1-p) and the merged probabilities of a symptom that several units emit ADDED, which is what
    ``dem_to_check_matrices`` does (/root/reference/src/build_circuit.py:262-270).  The x-basis windows are cut inside a column
    region (osd.py:83, 105), so there the order is part of the result; for z-basis it decides stable-sort ties only.
    "circuit": columns in circuit order of first appearance and every fault of the whole circuit merged by p(1-q) + q(1-p) --
    rounds 1-4 used this for z-basis (a symptom emitted by two units gets p + q - 2pq instead of the reference's p + q: 1 584 of
    the 8 784 priors of the [[144,12,12]] experiment differ, by at most 4.1e-5; tests/test_circuit.py::test_z_basis_prior_merge_modes).
    Since round 5 the default, every recorded fixture, the headline bench and the full-size tests use the reference's inputs."""
    if column_order == "circuit":
        return dem_from_ops(bb_memory_ops(code, A_list, B_list, p, num_repeat, z_basis))
    if column_order != "stim":
        raise ValueError("column_order: 'circuit' or 'stim'")
    ops, seg = bb_memory_ops(code, A_list, B_list, p, num_repeat, z_basis, return_segments=True)
    return dem_from_ops(ops, seg)
