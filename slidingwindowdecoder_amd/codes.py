"""Bivariate-bicycle (BB) code construction for the sliding-window harness.

Host-side, runs once per experiment.  Gives the same check matrices as the
reference's ``create_bivariate_bicycle_codes`` (/root/reference/src/codes_q.py:235-246)
-- hx = [A | B], hz = [B^T | A^T] with A = sum of the three A monomials and
B = sum of the three B monomials in the cyclic shifts x = S_l (x) I_m,
y = I_l (x) S_m -- but is built from index arithmetic on the (i, j) torus rather
than from Kronecker products.  Logical operators are any basis of
ker(hx) / rowspace(hz); logical-error accounting ("any observable flipped",
/root/reference/osd.py:186-187) is basis independent.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import gf2


@dataclass
class Monomial:
    """x^px * y^py acting on the l x m torus; a permutation of l*m points."""
    l: int
    m: int
    px: int
    py: int

    def col_of_row(self) -> np.ndarray:
        """perm[r] = column holding the 1 of row r (row r=(i,j) -> ((i+px)%l, (j+py)%m))."""
        i, j = np.divmod(np.arange(self.l * self.m), self.m)
        return ((i + self.px) % self.l) * self.m + ((j + self.py) % self.m)

    def row_of_col(self) -> np.ndarray:
        """Inverse permutation = col_of_row of the transpose."""
        p = self.col_of_row()
        inv = np.empty_like(p)
        inv[p] = np.arange(p.size)
        return inv

    def dense(self) -> np.ndarray:
        p = self.col_of_row()
        a = np.zeros((p.size, p.size), dtype=np.uint8)
        a[np.arange(p.size), p] = 1
        return a


@dataclass
class CSSCode:
    hx: np.ndarray
    hz: np.ndarray
    lx: np.ndarray = field(default=None)
    lz: np.ndarray = field(default=None)
    name: str = ""

    def __post_init__(self) -> None:
        self.hx = np.asarray(self.hx, dtype=np.uint8) % 2
        self.hz = np.asarray(self.hz, dtype=np.uint8) % 2
        assert self.hx.shape[1] == self.hz.shape[1]
        assert not ((self.hx.astype(int) @ self.hz.T.astype(int)) % 2).any(), "not a CSS code"
        self.N = int(self.hx.shape[1])
        rx, rz = gf2.rank(self.hx), gf2.rank(self.hz)
        self.K = self.N - rx - rz
        if self.lz is None:
            self.lz = _logicals(self.hx, self.hz)
        if self.lx is None:
            self.lx = _logicals(self.hz, self.hx)
        assert self.lz.shape[0] == self.K and self.lx.shape[0] == self.K


def _logicals(h_commute: np.ndarray, h_stab: np.ndarray) -> np.ndarray:
    """Basis of ker(h_commute) modulo rowspace(h_stab)."""
    ker = gf2.nullspace(h_commute)
    span = gf2.Span()
    for v in gf2.rows_to_ints(h_stab):
        span.add(v)
    out = []
    for v in gf2.rows_to_ints(ker):
        if span.add(v):
            out.append(v)
    return gf2.ints_to_rows(out, h_commute.shape[1])


# (l, m, A_x_pows, A_y_pows, B_x_pows, B_y_pows) keyed by block length N, the table of
# /root/reference/osd.py:17-30.
BB_PARAMS = {
    72: (6, 6, [3], [1, 2], [1, 2], [3]),
    90: (15, 3, [9], [1, 2], [2, 7], [0]),
    108: (9, 6, [3], [1, 2], [1, 2], [3]),
    144: (12, 6, [3], [1, 2], [1, 2], [3]),
    288: (12, 12, [3], [2, 7], [1, 2], [3]),
    360: (30, 6, [9], [1, 2], [25, 26], [3]),
    756: (21, 18, [3], [10, 17], [3, 19], [5]),
}


def create_bivariate_bicycle_codes(l, m, A_x_pows, A_y_pows, B_x_pows, B_y_pows, name=None):
    """Same signature and return convention as the reference
    (/root/reference/src/codes_q.py:235): ``(code, A_list, B_list)`` where the lists
    hold the three monomials of A (x powers first, then y powers) and of B (y powers
    first, then x powers)."""
    A_list = [Monomial(l, m, p, 0) for p in A_x_pows] + [Monomial(l, m, 0, p) for p in A_y_pows]
    B_list = [Monomial(l, m, 0, p) for p in B_y_pows] + [Monomial(l, m, p, 0) for p in B_x_pows]
    A = sum(t.dense().astype(int) for t in A_list) % 2
    B = sum(t.dense().astype(int) for t in B_list) % 2
    hx = np.hstack((A, B))
    hz = np.hstack((B.T, A.T))
    code = CSSCode(hx, hz, name=name or f"BB_n{2 * l * m}")
    return code, A_list, B_list


def bb_code(N: int):
    return create_bivariate_bicycle_codes(*BB_PARAMS[N])
