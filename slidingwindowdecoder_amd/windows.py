"""(W, F) sliding-window geometry over a BB detector error model.

Counterpart of the window-extraction half of the reference's harness
(/root/reference/osd.py:42-121, identical code in guessing.py:49-126 and the notebooks):

1. columns are permuted into "regions" keyed by the first and last detector round they
   touch (osd.py:42-68);
2. ``anchors`` = (first detector row, first column) of every round block (osd.py:70-77);
3. window t spans rows of W consecutive blocks; for ``method=1`` every non-final window
   keeps the faults local to its last block and replaces the faults that reach into the
   next block by an h x h identity ("noisy syndrome") with the merged prior
   ``sum(chk[c0:b0, c1:b1] * priors[c1:b1])`` (osd.py:79-89, 103-113);
4. ``num_win = ceil((len(anchors) - W + F - 1) / F)`` (osd.py:91);
5. after decoding window t the first ``anchors[t+F].col - anchors[t].col`` entries of the
   estimate are committed (whole estimate for the last window) (osd.py:140, 170-173).

Everything here is host-side setup that runs once per experiment; matrices are kept
sparse (CSR) because the decoder consumes CSR edge lists.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp


@dataclass
class Window:
    row0: int          # a[0]
    row1: int          # b[0]
    col0: int          # a[1]   first global column of the window
    ncols_global: int  # how many leading window columns are global DEM columns
    commit: int        # number of leading columns committed after decoding
    mat: sp.csr_matrix  # (row1-row0) x n_window check matrix (incl. identity block)
    prior: np.ndarray   # n_window fault probabilities
    is_last: bool


@dataclass
class WindowPlan:
    chk: sp.csr_matrix      # region-permuted global check matrix
    obs: sp.csr_matrix
    priors: np.ndarray
    perm: np.ndarray        # new column j = old column perm[j]
    anchors: list
    windows: list
    noisy_prior: float | None
    n_half: int


def region_permutation(chk: sp.spmatrix, n_half: int) -> np.ndarray:
    """Column order of osd.py:42-68: regions enumerated as (i, i+h), (i, i+2h) for
    i = 0, h, 2h, ...; inside a region the original order is kept."""
    chk = sp.csc_matrix(chk)
    num_row, num_col = chk.shape
    n = 2 * n_half
    bounds = []
    i = 0
    while i < num_row:
        bounds.append((i, i + n_half))
        if i + n > num_row:
            break
        bounds.append((i, i + n))
        i += n_half
    region_of = {lu: k for k, lu in enumerate(bounds)}
    keys = np.empty(num_col, dtype=np.int64)
    for j in range(num_col):
        r = chk.indices[chk.indptr[j]:chk.indptr[j + 1]]
        lo = int(r.min()) // n_half * n_half
        hi = (int(r.max()) // n_half + 1) * n_half
        keys[j] = region_of[(lo, hi)]
    return np.argsort(keys, kind="stable")


def find_anchors(chk: sp.spmatrix, n_half: int) -> list:
    """osd.py:70-77."""
    chk = sp.csc_matrix(chk)
    num_row, num_col = chk.shape
    anchors = []
    j = 0
    for i in range(num_col):
        r = chk.indices[chk.indptr[i]:chk.indptr[i + 1]]
        if r.min() >= j:
            anchors.append((j, i))
            j += n_half
    anchors.append((num_row, num_col))
    return anchors


def plan_windows(chk, obs, priors, n_half: int, W: int, F: int, method: int = 1,
                 z_basis: bool = True, noisy_prior=None) -> WindowPlan:
    perm = region_permutation(chk, n_half)
    chk = sp.csc_matrix(chk)[:, perm]
    obs = sp.csc_matrix(obs)[:, perm]
    priors = np.asarray(priors, dtype=np.float64)[perm]
    anchors = find_anchors(chk, n_half)
    n = 2 * n_half
    chk_r = sp.csr_matrix(chk)

    def shifted(c):
        if method == 1:
            return (c[0], c[1] + (n_half * 3 if z_basis else n))
        return c

    # osd.py:79-89: one merged prior PER ROW of the identity block (np.sum(..., axis=1)); the rows agree for
    # the BB circuits (the reference prints element 0) but not e.g. for SHYPS.  A caller-given scalar is broadcast.
    noisy_vec = None
    if noisy_prior is None and method != 0:
        b = anchors[W]
        c = shifted(anchors[W - 1])
        block = chk_r[c[0]:b[0], c[1]:b[1]]
        noisy_vec = np.asarray(block.multiply(priors[c[1]:b[1]]).sum(axis=1)).ravel()
        noisy_prior = float(noisy_vec[0])
    elif method != 0:
        noisy_vec = np.ones(n_half) * np.asarray(noisy_prior, dtype=np.float64)
        noisy_prior = float(noisy_vec[0])

    num_win = math.ceil((len(anchors) - W + F - 1) / F)
    windows = []
    top_left = 0
    for i in range(num_win):
        a = anchors[top_left]
        b = anchors[min(top_left + W, len(anchors) - 1)]
        last = i == num_win - 1
        if not last and method != 0:
            c = shifted(anchors[top_left + W - 1])
            sub = chk_r[a[0]:b[0], a[1]:c[1]]
            nrow = b[0] - a[0]
            ident = sp.csr_matrix((np.ones(n_half, np.uint8),
                                   (np.arange(nrow - n_half, nrow), np.arange(n_half))),
                                  shape=(nrow, n_half))
            mat = sp.hstack((sub, ident), format="csr")
            prior = np.concatenate((priors[a[1]:c[1]], noisy_vec))
            ncg = c[1] - a[1]
        else:
            mat = sp.csr_matrix(chk_r[a[0]:b[0], a[1]:b[1]])
            prior = priors[a[1]:b[1]].copy()
            ncg = b[1] - a[1]
        commit = ncg if last else anchors[top_left + F][1] - a[1]
        mat.sort_indices()
        windows.append(Window(a[0], b[0], a[1], ncg, commit, mat, prior, last))
        top_left += F
    return WindowPlan(sp.csr_matrix(chk), sp.csr_matrix(obs), priors, perm, anchors, windows,
                      noisy_prior, n_half)


def sample_dem(chk, obs, priors, num_shots: int, seed: int = 20240318):
    """Bernoulli(priors) fault sampling -> (det_data, obs_data, faults); what
    ``dem.compile_sampler().sample`` provides to the reference harness (osd.py:124-125)."""
    rng = np.random.default_rng(seed)
    chk = sp.csr_matrix(chk)
    obs = sp.csr_matrix(obs)
    e = (rng.random((num_shots, chk.shape[1])) < priors).astype(np.uint8)
    det = (sp.csr_matrix(e) @ chk.T.astype(np.int32)).toarray() % 2
    ob = (sp.csr_matrix(e) @ obs.T.astype(np.int32)).toarray() % 2
    return det.astype(np.uint8), ob.astype(np.uint8), e


def sliding_window_decode_host(plan: WindowPlan, det_data: np.ndarray, decoder_factory,
                               on_decode=None):
    """Host-side window loop with the commit rule of osd.py:130-179.

    ``decoder_factory(window) -> object with .decode(syndrome)``; any decoder with the
    reference's class surface fits (the product's classes, the oracle, or the reference
    itself).  Returns (total_e_hat[shots, num_col] uint8, flagged_per_window).
    ``on_decode(win_idx, shot, decoder, syndrome, e_hat)`` is an optional tap used by the
    fixture generator and the parity tests.
    """
    num_shots = det_data.shape[0]
    num_col = plan.chk.shape[1]
    chk_t = sp.csr_matrix(plan.chk.T.astype(np.int32))
    total = np.zeros((num_shots, num_col), dtype=np.uint8)
    cur = det_data.copy()
    flagged = []
    for wi, w in enumerate(plan.windows):
        dec = decoder_factory(w)
        nflag = 0
        mat_i = w.mat.astype(np.int32)
        for j in range(num_shots):
            s = cur[j, w.row0:w.row1]
            e_hat = np.asarray(dec.decode(s))
            if on_decode is not None:
                on_decode(wi, j, dec, s, e_hat)
            nflag += int(((mat_i @ e_hat + s) % 2).any())
            total[j, w.col0:w.col0 + w.commit] = e_hat[:w.commit]
        flagged.append(nflag)
        cur = ((det_data + (sp.csr_matrix(total) @ chk_t).toarray()) % 2).astype(np.uint8)
    return total, flagged


def logical_error_stats(plan: WindowPlan, det_data, obs_data, total_e_hat):
    """osd.py:184-191: flagged = residual syndrome non-zero; logical = any observable wrong."""
    t = sp.csr_matrix(total_e_hat)
    resid = (det_data + (t @ plan.chk.T.astype(np.int32)).toarray()) % 2
    flagged = resid.any(axis=1)
    logical = ((obs_data + (t @ plan.obs.T.astype(np.int32)).toarray()) % 2).any(axis=1)
    return flagged, np.logical_or(flagged, logical)
