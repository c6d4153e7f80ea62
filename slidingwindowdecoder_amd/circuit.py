"""Stim-free detector-error-model (DEM) derivation for the BB memory circuit.

The reference builds a Stim circuit (/root/reference/src/build_circuit.py:6-234), asks
Stim for its detector error model and turns that into (chk, obs, priors)
(/root/reference/src/build_circuit.py:251-299).  Stim is not available on the MI355X
boxes, so this module derives the same model directly:

* ``bb_memory_ops`` restates the gate / noise / measurement schedule of the reference
  circuit as a flat op list (z-basis memory experiment, ``use_both=False``, ``HZH=False``
  -- the variant every BASELINE config uses).
* ``dem_from_ops`` sweeps that list BACKWARDS keeping, per qubit, the set of detectors
  and observables an X flip at that point of time would toggle ("sensitivity"), which is
  all that matters for Z-basis detectors: CX c,t -> sens[c] ^= sens[t]; MR -> sens = this
  measurement's detectors; M -> sens ^= ...; R / MRX / H -> sens = 0.  Every noise channel
  then contributes independent fault mechanisms with Stim's per-component probabilities
  (DEPOLARIZE1: (1-sqrt(1-4p/3))/2 for each of X,Y,Z; DEPOLARIZE2: (1-(1-16p/15)^(1/8))/2
  for each of the 15 Paulis) and mechanisms with identical symptoms are merged with
  p <- p(1-q) + q(1-p).

Host-side, runs once per experiment.  The structural known answers of the reference's
notebooks (shape 936 x 8784 for 12 rounds of the [[144,12,12]] code, column-weight
histogram, window anchors, merged "noisy syndrome" prior 0.027499817877069083 at p=0.003)
are asserted in tests/test_circuit.py.  Column order is this module's own (first
appearance in circuit order); the decoder results depend on it only through stable-sort
ties and parity is always checked on identical matrices.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp

# op codes
R, RX, H, CX, M, MR, MRX, XERR, DEP1, DEP2, DETECTOR, OBSERVABLE, MX = range(13)


def bb_memory_ops(code, A_list, B_list, p: float, num_repeat: int):
    """Op list of the z-basis BB memory circuit.  Each op is a tuple
    (kind, qubits..., prob) ; DETECTOR/OBSERVABLE carry absolute measurement indices."""
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
    nmeas = 0
    zmeas_prev = None

    def cx(c, t):
        ops.append((CX, c, t))
        ops.append((DEP2, c, t, p))

    for i in range(h):
        ops.append((R, XC + i))
        ops.append((R, ZC + i))
        ops.append((XERR, XC + i, p))
        ops.append((XERR, ZC + i, p))
    for i in range(n):
        ops.append((R, LD + i))
        ops.append((XERR, LD + i, p))

    for rnd in range(num_repeat):
        repeat = rnd > 0
        if repeat:
            for i in range(h):
                ops.append((XERR, ZC + i, p))
                # Z_ERROR on the X ancilla: invisible to Z-basis detectors
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
        for i in range(h):
            if repeat:
                ops.append((DETECTOR, (zmeas[i], zmeas_prev[i])))
            else:
                ops.append((DETECTOR, (zmeas[i],)))
        zmeas_prev = zmeas
        for i in range(h):  # layer 8
            ops.append((MRX, XC + i, nmeas))
            nmeas += 1

    dmeas = []
    for i in range(n):
        ops.append((M, LD + i, nmeas))
        dmeas.append(nmeas)
        nmeas += 1
    for i, row in enumerate(code.hz):
        ops.append((DETECTOR, tuple(dmeas[j] for j in np.flatnonzero(row)) + (zmeas_prev[i],)))
    for i, row in enumerate(code.lz):
        ops.append((OBSERVABLE, tuple(dmeas[j] for j in np.flatnonzero(row))))
    return ops


@dataclass
class DEM:
    chk: sp.csc_matrix  # detectors x mechanisms
    obs: sp.csc_matrix  # observables x mechanisms
    priors: np.ndarray


def dem_from_ops(ops) -> DEM:
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

    sens = [0] * nq
    mech: dict[int, float] = {}  # symptom bitmask -> probability
    order: list[int] = []

    def emit(sym: int, q: float) -> None:
        if sym == 0:
            return
        if sym in mech:
            pp = mech[sym]
            mech[sym] = pp * (1.0 - q) + q * (1.0 - pp)
        else:
            mech[sym] = q
            order.append(sym)

    for o in reversed(ops):
        kind = o[0]
        if kind == CX:
            sens[o[1]] ^= sens[o[2]]
        elif kind == DEP2:
            q = 0.5 * (1.0 - (1.0 - 16.0 * o[3] / 15.0) ** 0.125)
            sa, sb = sens[o[1]], sens[o[2]]
            for sym in (sa, sb, sa ^ sb):  # X-part on first / second / both; 4 Paulis each
                for _ in range(4):
                    emit(sym, q)
        elif kind == DEP1:
            q = 0.5 * (1.0 - (1.0 - 4.0 * o[2] / 3.0) ** 0.5)
            emit(sens[o[1]], q)  # X
            emit(sens[o[1]], q)  # Y
        elif kind == XERR:
            emit(sens[o[1]], o[2])
        elif kind == MR:
            sens[o[1]] = meas_mask.get(o[2], 0)
        elif kind == M:
            sens[o[1]] ^= meas_mask.get(o[2], 0)
        elif kind in (R, RX, MRX, H):
            sens[o[1]] = 0
        # MX: an X flip commutes with the X-basis measurement -- neither its outcome nor the state after it changes
    # forward-circuit order of first appearance
    order.reverse()

    rows, cols, orow, ocol = [], [], [], []
    dmask = (1 << ndet) - 1
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


def bb_dem(code, A_list, B_list, p: float, num_repeat: int) -> DEM:
    """(chk, obs, priors) of the z-basis BB memory experiment -- counterpart of
    ``dem_to_check_matrices(build_circuit(...).detector_error_model())``
    (/root/reference/osd.py:35-37)."""
    return dem_from_ops(bb_memory_ops(code, A_list, B_list, p, num_repeat))
