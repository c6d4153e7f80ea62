"""Loaders for the golden vectors in tests/golden/ (written by tests/golden/make_golden.py
from the reference's own compiled extension)."""
from __future__ import annotations

import functools
import hashlib
import json
import os

import numpy as np
import scipy.sparse as sp

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@functools.lru_cache(maxsize=None)
def load(name: str):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def graph(f, prefix: str):
    shape = tuple(int(x) for x in f[prefix + "shape"])
    nnz = int(f[prefix + "indptr"][-1])
    mat = sp.csr_matrix((np.ones(nnz, np.uint8), f[prefix + "indices"], f[prefix + "indptr"]), shape=shape)
    return mat, f[prefix + "priors"]


def params(f, key: str) -> dict:
    return json.loads(str(f[key]))


def unpack(a, nbits: int) -> np.ndarray:
    return np.unpackbits(a, axis=-1)[..., :nbits]


def h64(a: np.ndarray) -> np.uint64:
    return np.frombuffer(hashlib.blake2b(np.ascontiguousarray(a).tobytes(), digest_size=8).digest(),
                         dtype=np.uint64)[0]


class Trace:
    """One recorded sequence of reference decodes."""

    def __init__(self, f, prefix: str, m: int, n: int):
        self.synd = unpack(f[prefix + "synd"], m)
        self.out = unpack(f[prefix + "out"], n)
        self.converge = f[prefix + "converge"]
        self.bp_iteration = f[prefix + "bp_iteration"] if prefix + "bp_iteration" in f else None
        self.min_pm = f[prefix + "min_pm"] if prefix + "min_pm" in f else None
        self.hist_hash = f[prefix + "hist_hash"] if prefix + "hist_hash" in f else None
        self.hist_idx = f[prefix + "hist_idx"] if prefix + "hist_idx" in f else None
        self.hist = f[prefix + "hist"] if prefix + "hist" in f else None
        self.osd0 = unpack(f[prefix + "osd0"], n) if prefix + "osd0" in f else None

    def __len__(self):
        return self.synd.shape[0]
