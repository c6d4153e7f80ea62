"""ctypes front-end of the CPU parity oracle (oracle/swd_oracle.c).

TEST INFRASTRUCTURE ONLY.  Allowed importers: tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.

The classes mirror the reference's Cython classes (constructor kwargs, ``decode``,
properties) so that parity tests read like calls into the reference:
  osd_window          /root/reference/src/osd_window.pyx
  bpgdg_decoder etc.  /root/reference/src/bp_guessing_decoder.pyx
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libswd_oracle.so")
    src = os.path.join(_HERE, "swd_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libswd_oracle.so"], stdout=subprocess.DEVNULL)
    return so


class _Result(C.Structure):
    _fields_ = [("converge", C.c_int32), ("bp_iteration", C.c_int32), ("exit_class", C.c_int32),
                ("reserved", C.c_int32), ("min_pm", C.c_double)]


RESULT_DTYPE = np.dtype([("converge", "<i4"), ("bp_iteration", "<i4"), ("exit_class", "<i4"),
                         ("reserved", "<i4"), ("min_pm", "<f8")])


class _OsdwParams(C.Structure):
    _fields_ = [("pre_max_iter", C.c_int32), ("post_max_iter", C.c_int32),
                ("ms_scaling_factor", C.c_double), ("new_n", C.c_int32),
                ("osd_method", C.c_int32), ("osd_order", C.c_int32)]


class _GdgParams(C.Structure):
    _fields_ = [("max_iter", C.c_int32), ("ms_scaling_factor", C.c_double),
                ("max_iter_per_step", C.c_int32), ("max_step", C.c_int32),
                ("max_tree_depth", C.c_int32), ("max_side_depth", C.c_int32),
                ("max_tree_branch_step", C.c_int32), ("max_side_branch_step", C.c_int32),
                ("gdg_factor", C.c_double), ("new_n", C.c_int32), ("low_error_mode", C.c_int32)]


class _Bp4Params(C.Structure):
    _fields_ = [("max_iter", C.c_int32), ("ms_scaling_factor", C.c_double), ("osd_method", C.c_int32),
                ("osd_order", C.c_int32)]


def lib():
    global _LIB
    if _LIB is None:
        # SWD_ORACLE_SO: an alternative build of the same source, e.g. libswd_oracle_asan.so (tests/test_oracle_asan.py)
        L = C.CDLL(os.environ.get("SWD_ORACLE_SO") or build())
        L.swo_bp4_create.restype = C.c_void_p
        L.swo_bp4_create.argtypes = [C.c_int] * 3 + [C.c_void_p] * 7 + [C.POINTER(_Bp4Params)]
        L.swo_bp4_free.argtypes = [C.c_void_p]
        L.swo_bp4_decode.argtypes = [C.c_void_p] * 5 + [C.POINTER(_Result)] + [C.c_void_p] * 3
        L.swo_bp4_camel_decode.argtypes = [C.c_void_p] * 5 + [C.POINTER(_Result)]
        L.swo_bp4_ranks.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.swo_graph_create.restype = C.c_void_p
        L.swo_graph_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.swo_graph_free.argtypes = [C.c_void_p]
        L.swo_graph_rank.argtypes = [C.c_void_p]
        L.swo_osdw_create.restype = C.c_void_p
        L.swo_osdw_create.argtypes = [C.c_void_p, C.POINTER(_OsdwParams)]
        L.swo_osdw_free.argtypes = [C.c_void_p]
        L.swo_osdw_clear_history.argtypes = [C.c_void_p]
        L.swo_osdw_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(_Result)]
        L.swo_osdw_decode_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        for f in ("swo_osdw_history", "swo_osdw_osd0", "swo_osdw_bp", "swo_gdg_history"):
            getattr(L, f).restype = C.c_void_p
            getattr(L, f).argtypes = [C.c_void_p]
        L.swo_gdg_create.restype = C.c_void_p
        L.swo_gdg_create.argtypes = [C.c_void_p, C.POINTER(_GdgParams)]
        L.swo_gdg_free.argtypes = [C.c_void_p]
        L.swo_gdg_clear_history.argtypes = [C.c_void_p]
        L.swo_gdg_decode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(_Result)]
        L.swo_gdg_ensemble_info.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.swo_gdg_cols.restype = C.POINTER(C.c_int)
        L.swo_gdg_cols.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


_OSD_METHODS = {  # osd_window.pyx:69-79
    0: ["osd_0", "0", "osd0"],
    1: ["osd_e", "1", "osde", "exhaustive", "e"],
    2: ["osd_cs", "2", "osdcs", "combination_sweep", "cs"],
}


def parse_osd_method(osd_method, osd_order):
    s = str(osd_method).lower()
    for k, names in _OSD_METHODS.items():
        if s in names:
            return k, (0 if k == 0 else int(osd_order))
    raise ValueError(f"ERROR: OSD method '{osd_method}' invalid. Please choose from the following "
                     "methods: 'OSD_0', 'OSD_E' or 'OSD_CS'.")


class _Graph:
    def __init__(self, pcm, channel_probs):
        if not (isinstance(pcm, np.ndarray) or sp.issparse(pcm)):
            raise TypeError("The input matrix is of an invalid type. Please input a np.ndarray or "
                            f"scipy.sparse.spmatrix object, not {type(pcm)}")
        a = sp.csr_matrix(pcm)
        a.eliminate_zeros()
        a.sort_indices()
        self.m, self.n = a.shape
        probs = np.ascontiguousarray(channel_probs, dtype=np.float64)
        if len(probs) != self.n:
            raise ValueError("The length of the channel probability vector must be eqaul to the "
                             f"block length n={self.n}.")
        self.row_ptr = np.ascontiguousarray(a.indptr, dtype=np.int32)
        self.col_idx = np.ascontiguousarray(a.indices, dtype=np.int32)
        self.probs = probs
        self.h = lib().swo_graph_create(self.m, self.n, self.row_ptr.ctypes.data,
                                        self.col_idx.ctypes.data, probs.ctypes.data)
        self.rank = lib().swo_graph_rank(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().swo_graph_free(self.h)
            self.h = None


def _synd_u8(s, m):
    s = np.asarray(s)
    if s.shape[0] != m:
        raise ValueError(f"The input to the ldpc.bp_decoder.decode must be a syndrome (of length={m}). "
                         f"The inputted vector has length={s.shape[0]}.")
    return np.ascontiguousarray(s.astype(np.int64).astype(np.uint8))


class osd_window:
    def __init__(self, parity_check_matrix, **kw):
        self._g = _Graph(parity_check_matrix, kw.get("channel_probs"))
        self.m, self.n = self._g.m, self._g.n
        method, order = parse_osd_method(kw.get("osd_method", "osd_0"), kw.get("osd_order", 0))
        new_n = kw.get("new_n", None)
        p = _OsdwParams(int(kw.get("pre_max_iter", 8)), int(kw.get("post_max_iter", 100)),
                        float(kw.get("ms_scaling_factor", 1.0)), int(new_n) if new_n else 0,
                        method, order)
        self.new_n = min(self.n, 2 * self.m) if not new_n else min(int(new_n), self.n)
        self.rank = self._g.rank
        self._h = lib().swo_osdw_create(self._g.h, C.byref(p))
        if not self._h:
            raise ValueError("For this code, the OSD order should be set in the range "
                             f"0<=osd_oder<={self.new_n - self.rank}.")
        self._res = _Result()
        self._out = np.zeros(self.n, dtype=np.uint8)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().swo_osdw_free(self._h)
            self._h = None

    def clear_history(self):
        lib().swo_osdw_clear_history(self._h)

    def decode(self, syndrome):
        s = _synd_u8(syndrome, self.m)
        lib().swo_osdw_decode(self._h, s.ctypes.data, self._out.ctypes.data, C.byref(self._res))
        return self._out.astype(np.int64)

    def decode_batch(self, syndromes):
        s = np.ascontiguousarray(np.asarray(syndromes).astype(np.uint8))
        B = s.shape[0]
        out = np.zeros((B, self.n), dtype=np.uint8)
        res = np.zeros(B, dtype=RESULT_DTYPE)
        lib().swo_osdw_decode_batch(self._h, B, s.ctypes.data, out.ctypes.data, res.ctypes.data)
        return out, res

    bp_iteration = property(lambda self: self._res.bp_iteration)
    converge = property(lambda self: self._res.converge)
    min_pm = property(lambda self: self._res.min_pm)
    exit_class = property(lambda self: self._res.exit_class)

    def _vec(self, fn):
        p = getattr(lib(), fn)(self._h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (self.n,)).astype(np.int64)

    bp_decoding = property(lambda self: self._vec("swo_osdw_bp"))
    osd0_decoding = property(lambda self: self._vec("swo_osdw_osd0"))

    @property
    def osdw_decoding(self):
        return self._out.astype(np.int64)

    @property
    def log_prob_ratios(self):
        p = lib().swo_osdw_history(self._h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), (self.n, 4)).copy()


class _GdgBase:
    _mode = 2

    def __init__(self, parity_check_matrix, **kw):
        self._g = _Graph(parity_check_matrix, kw.get("channel_probs"))
        self.m, self.n = self._g.m, self._g.n
        new_n = kw.get("new_n", None)
        p = _GdgParams(int(kw.get("max_iter", 50)), float(kw.get("ms_scaling_factor", 1.0)),
                       int(kw.get("max_iter_per_step", 6)), int(kw.get("max_step", 25)),
                       int(kw.get("max_tree_depth", 3)), int(kw.get("max_side_depth", 10)),
                       int(kw.get("max_tree_branch_step", 10)), int(kw.get("max_side_branch_step", 10)),
                       float(kw.get("gdg_factor", kw.get("gd_factor", 1.0))),
                       int(new_n) if new_n else 0, int(bool(kw.get("low_error_mode", False))))
        self._h = lib().swo_gdg_create(self._g.h, C.byref(p))
        self._res = _Result()
        self._out = np.zeros(self.n, dtype=np.uint8)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().swo_gdg_free(self._h)
            self._h = None

    def clear_history(self):
        lib().swo_gdg_clear_history(self._h)

    def decode(self, syndrome):
        s = _synd_u8(syndrome, self.m)
        lib().swo_gdg_decode(self._h, self._mode, s.ctypes.data, self._out.ctypes.data, C.byref(self._res))
        return self._out.astype(np.int64)

    converge = property(lambda self: self._res.converge)
    min_pm = property(lambda self: self._res.min_pm)

    @property
    def log_prob_ratios(self):
        p = lib().swo_gdg_history(self._h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), (self.n, 4)).copy()


class bp_history_decoder(_GdgBase):
    _mode = 2


class bpgdg_decoder(_GdgBase):
    """multi_thread=False: the deterministic gdg() (bp_guessing_decoder.pyx:254-338); multi_thread=True: the threaded ensemble
    (bpgd.cpp:419-688) with its thread bodies run in a fixed order -- main, tree threads by id, side threads by index, ties of the
    path metric to the earliest (swd_oracle.c: gdg_multi_run)."""
    _mode = 0

    def __init__(self, parity_check_matrix, **kw):
        super().__init__(parity_check_matrix, **kw)
        if kw.get("multi_thread", False):
            self._mode = 3

    def ensemble_info(self):
        """after a multi_thread decode that ran the post-processing: (path metric per hypothesis [main, tree 1.., side 1..],
        10000.0 = not converged; index of the winner or -1; number of converged hypotheses sharing the winning metric with a
        different vector -- non-zero: the reference's answer for this syndrome depends on thread timing)"""
        pm = np.zeros(256)
        w, t = C.c_int32(), C.c_int32()
        k = lib().swo_gdg_ensemble_info(self._h, pm.ctypes.data, 256, C.byref(w), C.byref(t))
        return pm[:k].copy(), w.value, t.value

    def ensemble_blocks(self):
        """(BP blocks of the last ensemble decode, those at depth < max_tree_depth of the main / tree threads, the latter counted
        once per distinct direction prefix)"""
        t, p, u = C.c_int32(), C.c_int32(), C.c_double()
        lib().swo_gdg_ensemble_blocks.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double)]
        lib().swo_gdg_ensemble_blocks(self._h, C.byref(t), C.byref(p), C.byref(u))
        return t.value, p.value, u.value

    @property
    def cols(self):
        return np.ctypeslib.as_array(lib().swo_gdg_cols(self._h), (self.n,)).astype(np.int32)


class bpgd_decoder(_GdgBase):
    _mode = 1


class bp4_osd:
    """src/bp4_osd.pyx restated (quaternary BP + one OSD per basis)."""

    def __init__(self, Hx, Hz, **kw):
        if not (isinstance(Hx, np.ndarray) or sp.issparse(Hx)):
            raise TypeError("The input matrix is of an invalid type. Please input a np.ndarray or scipy.sparse.spmatrix object.")
        if Hx.shape[1] != Hz.shape[1]:
            raise ValueError("Hx, Hz blocklength does not match!")
        ax, az = sp.csr_matrix(Hx), sp.csr_matrix(Hz)
        for a in (ax, az):
            a.eliminate_zeros(); a.sort_indices()
        self.mx, self.n = ax.shape
        self.mz = az.shape[0]
        px, py, pz = (np.ascontiguousarray(kw.get(k), dtype=np.float64) for k in ("channel_probs_x", "channel_probs_y", "channel_probs_z"))
        if len(px) != self.n:
            raise ValueError(f"The length of the channel probability vector must be eqaul to the block length n={self.n}.")
        method, order = parse_osd_method(kw.get("osd_method", "osd_0"), kw.get("osd_order", 0))
        p = _Bp4Params(int(kw.get("max_iter", 32)), float(kw.get("ms_scaling_factor", 1.0)), method, order)
        self._keep = [np.ascontiguousarray(x, dtype=np.int32) for x in (ax.indptr, ax.indices, az.indptr, az.indices)] + [px, py, pz]
        self._h = lib().swo_bp4_create(self.mx, self.mz, self.n, *[k.ctypes.data for k in self._keep], C.byref(p))
        if not self._h:
            raise ValueError("For this code, the OSD order should be set in the range 0<=osd_oder<=n-rank.")
        self._res = _Result()
        self._lpr = np.zeros((self.n, 3))
        self._o0x = np.zeros(self.n, np.uint8); self._o0z = np.zeros(self.n, np.uint8)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().swo_bp4_free(self._h)
            self._h = None

    def decode(self, sx, sz):
        sx, sz = _synd_u8(sx, self.mx), _synd_u8(sz, self.mz)
        ox, oz = np.zeros(self.n, np.uint8), np.zeros(self.n, np.uint8)
        lib().swo_bp4_decode(self._h, sx.ctypes.data, sz.ctypes.data, ox.ctypes.data, oz.ctypes.data, C.byref(self._res),
                             self._lpr.ctypes.data, self._o0x.ctypes.data, self._o0z.ctypes.data)
        return np.stack([ox, oz]).astype(np.int64)

    def camel_decode(self, sx, sz):
        sx, sz = _synd_u8(sx, self.mx), _synd_u8(sz, self.mz)
        ox, oz = np.zeros(self.n, np.uint8), np.zeros(self.n, np.uint8)
        lib().swo_bp4_camel_decode(self._h, sx.ctypes.data, sz.ctypes.data, ox.ctypes.data, oz.ctypes.data, C.byref(self._res))
        self._o0x[:], self._o0z[:] = ox, oz
        return np.stack([ox, oz]).astype(np.int64)

    converge = property(lambda self: self._res.converge)
    bp_iteration = property(lambda self: self._res.bp_iteration)
    min_pm = property(lambda self: self._res.min_pm)
    log_prob_ratios = property(lambda self: self._lpr.copy())
    def _bpdec(self, z):
        L = lib()
        L.swo_bp4_bp_decoding.restype = C.c_void_p
        L.swo_bp4_bp_decoding.argtypes = [C.c_void_p, C.c_int]
        p = L.swo_bp4_bp_decoding(self._h, z)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int8)), (self.n,)).astype(np.int64)

    bp_decoding_x = property(lambda self: self._bpdec(0))
    bp_decoding_z = property(lambda self: self._bpdec(1))
    osd0_decoding_x = property(lambda self: self._o0x.astype(np.int64))
    osd0_decoding_z = property(lambda self: self._o0z.astype(np.int64))
