"""Small GF(2) linear-algebra helpers for host-side problem construction.

Rows are held as Python integers (bit j of the integer = column j), which keeps
elimination on the few-hundred-column matrices of this package short and exact.
Host-side setup only (runs once per experiment); nothing here is on the decode
hot path.  Plays the role that /root/reference/src/utils.py:309-430
(row_echelon / rank / kernel) plays for the reference's code constructions, with
a different algorithmic layout (integer bitsets instead of boolean ndarrays).
"""
from __future__ import annotations

import numpy as np


def rows_to_ints(mat: np.ndarray) -> list[int]:
    """Pack each row of a 0/1 matrix into an int (bit j <-> column j)."""
    mat = np.asarray(mat)
    out = []
    for r in mat:
        nz = np.flatnonzero(r)
        v = 0
        for j in nz:
            v |= 1 << int(j)
        out.append(v)
    return out


def ints_to_rows(rows: list[int], ncols: int) -> np.ndarray:
    out = np.zeros((len(rows), ncols), dtype=np.uint8)
    for i, v in enumerate(rows):
        while v:
            low = v & -v
            out[i, low.bit_length() - 1] = 1
            v ^= low
    return out


class Span:
    """Incrementally built row space over GF(2) with reduced basis vectors."""

    def __init__(self) -> None:
        self.by_lead: dict[int, int] = {}  # leading bit -> basis vector

    def reduce(self, v: int) -> int:
        while v:
            lead = v.bit_length() - 1
            b = self.by_lead.get(lead)
            if b is None:
                return v
            v ^= b
        return 0

    def add(self, v: int) -> bool:
        """Insert v; True if it enlarged the span."""
        v = self.reduce(v)
        if v == 0:
            return False
        self.by_lead[v.bit_length() - 1] = v
        return True

    @property
    def dim(self) -> int:
        return len(self.by_lead)


def rank(mat: np.ndarray) -> int:
    s = Span()
    for v in rows_to_ints(mat):
        s.add(v)
    return s.dim


def nullspace(mat: np.ndarray) -> np.ndarray:
    """Basis (rows) of {x : mat @ x = 0 mod 2}."""
    mat = np.asarray(mat) % 2
    m, n = mat.shape
    # eliminate on columns of mat^T augmented with identity: track combinations
    rows = rows_to_ints(mat.T)  # n vectors of length m
    comb = [1 << i for i in range(n)]  # which original coordinates were combined
    lead_owner: dict[int, int] = {}
    kernel = []
    for i in range(n):
        v, c = rows[i], comb[i]
        while v:
            lead = v.bit_length() - 1
            k = lead_owner.get(lead)
            if k is None:
                lead_owner[lead] = i
                rows[i], comb[i] = v, c
                break
            v ^= rows[k]
            c ^= comb[k]
        if v == 0:
            kernel.append(c)
    return ints_to_rows(kernel, n)


def left_inverse(mat: np.ndarray) -> np.ndarray:
    """P with P @ mat = I over GF(2) for a full-column-rank n x k matrix (any such P; the reference's
    utils.inverse serves the same purpose, /root/reference/src/utils.py:476-514)."""
    a = (np.asarray(mat) % 2).astype(np.uint8)
    n, k = a.shape
    aug = np.concatenate([a, np.identity(n, dtype=np.uint8)], axis=1)
    row = 0
    for col in range(k):
        piv = next((i for i in range(row, n) if aug[i, col]), None)
        if piv is None:
            raise ValueError("matrix does not have full column rank")
        if piv != row:
            aug[[row, piv]] = aug[[piv, row]]
        for i in range(n):
            if i != row and aug[i, col]:
                aug[i] ^= aug[row]
        row += 1
    return aug[:k, k:].astype(np.int64)
