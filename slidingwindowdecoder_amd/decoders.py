"""Python class surface of the reference, backed by the HIP library.

``osd_window`` keeps the constructor kwargs, ``decode`` and the properties of the reference's
Cython class (/root/reference/src/osd_window.pyx:8-126, 158-199, 487-517) so that the
notebook / script loops (/root/reference/osd.py:152-167) run unchanged, and adds
``decode_batch`` (shape modelled on the batched decoders the reference's notebooks compare
against) which is the path that actually uses the GPU well: one launch, one workgroup per
syndrome.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import _lib

EXIT_PRE, EXIT_POST, EXIT_OSD, EXIT_FAIL_SET, EXIT_FAIL_PEEL, EXIT_NO_OSD, EXIT_SCHED_FAULT = range(7)
STATUS_CONVERGE = 0x100

_OSD_METHODS = {  # aliases of osd_window.pyx:69-79
    0: ["osd_0", "0", "osd0"],
    1: ["osd_e", "1", "osde", "exhaustive", "e"],
    2: ["osd_cs", "2", "osdcs", "combination_sweep", "cs"],
}


def _parse_osd_method(osd_method, osd_order):
    s = str(osd_method).lower()
    for k, names in _OSD_METHODS.items():
        if s in names:
            return k, (0 if k == 0 else int(osd_order))
    raise ValueError(f"ERROR: OSD method '{osd_method}' invalid. Please choose from the following "
                     "methods: 'OSD_0', 'OSD_E' or 'OSD_CS'.")


class _Csr:
    """Validated CSR + priors kept alive for the C call."""

    def __init__(self, pcm, channel_probs):
        if not (isinstance(pcm, np.ndarray) or sp.issparse(pcm)):
            raise TypeError("The input matrix is of an invalid type. Please input a np.ndarray or "
                            f"scipy.sparse.spmatrix object, not {type(pcm)}")
        a = sp.csr_matrix(pcm)
        a.data = (np.asarray(a.data) != 0).astype(np.uint8)
        a.eliminate_zeros()
        a.sort_indices()
        self.m, self.n = a.shape
        if channel_probs is None:
            raise ValueError("channel_probs is required")
        probs = np.ascontiguousarray(channel_probs, dtype=np.float64)
        if len(probs) != self.n:
            raise ValueError("The length of the channel probability vector must be eqaul to the "
                             f"block length n={self.n}.")
        self.row_ptr = np.ascontiguousarray(a.indptr, dtype=np.int32)
        self.col_idx = np.ascontiguousarray(a.indices, dtype=np.int32)
        self.probs = probs
        self.desc = _lib.GraphDesc(self.m, self.n, int(self.row_ptr[-1]), self.row_ptr.ctypes.data,
                                   self.col_idx.ctypes.data, probs.ctypes.data)


def _as_synd(x, m):
    x = np.asarray(x)
    if x.ndim != 1 or x.shape[0] != m:
        n_in = x.shape[0] if x.ndim >= 1 else 0
        raise ValueError(f"The input to the ldpc.bp_decoder.decode must be a syndrome (of length={m}). "
                         f"The inputted vector has length={n_in}. Valid formats are `np.ndarray` or "
                         "`scipy.sparse.spmatrix`.")
    return np.ascontiguousarray((x.astype(np.int64) & 0xFF).astype(np.uint8))  # C (char) cast, c_util.pyx:6-9


class osd_window:
    """BP + OSD on a shortened window matrix (reference: src/osd_window.pyx)."""

    def __init__(self, parity_check_matrix, **kwargs):
        L = _lib.lib()
        self._csr = _Csr(parity_check_matrix, kwargs.get("channel_probs"))
        self.m, self.n = self._csr.m, self._csr.n
        method, order = _parse_osd_method(kwargs.get("osd_method", "osd_0"), kwargs.get("osd_order", 0))
        new_n = kwargs.get("new_n", None)
        self.pre_max_iter = int(kwargs.get("pre_max_iter", 8))
        self.post_max_iter = int(kwargs.get("post_max_iter", 100))
        self.ms_scaling_factor = float(kwargs.get("ms_scaling_factor", 1.0))
        self.osd_method, self.osd_order = method, order
        self.device = int(kwargs.get("device", 0))
        p = _lib.OsdwParams(self.pre_max_iter, self.post_max_iter, self.ms_scaling_factor,
                            int(new_n) if new_n else 0, method, order)
        self._h = L.swd_osdw_create(C.byref(self._csr.desc), C.byref(p), self.device)
        if not self._h:
            msg = _lib.last_error()
            if "OSD order" in msg or "invalid" in msg:
                raise ValueError(msg)
            raise RuntimeError(f"swd_osdw_create failed: {msg}")
        i = [C.c_int32() for _ in range(4)]
        L.swd_osdw_info(self._h, *[C.byref(x) for x in i])
        self.new_n, self.rank = i[2].value, i[3].value
        # per-object state the reference keeps between decodes
        self._hist = np.zeros((4, self.n), dtype=np.float64)
        self._last = dict(status=EXIT_PRE, iters=0, min_pm=0.0)
        self._bp_decoding = np.zeros(self.n, dtype=np.int64)
        self._osdw_decoding = np.zeros(self.n, dtype=np.int64)
        self._osd0_decoding = np.zeros(self.n, dtype=np.int64)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().swd_osdw_destroy(h)
            except Exception:
                pass
            self._h = None

    # ---- reference surface -------------------------------------------------------------
    def decode(self, input_vector):
        """One syndrome -> int64[n] (osd_window.pyx:158-199).  The LLR history persists between
        calls exactly like the reference object's."""
        s = _as_synd(input_vector, self.m)
        out = np.zeros(self.n, dtype=np.uint8)
        st = np.zeros(_lib.STAT_WORDS, np.int32)
        pm = np.zeros(1, np.float64)
        osd0 = np.zeros(self.n, dtype=np.uint8)
        bpd = np.zeros(self.n, dtype=np.uint8)
        rc = _lib.lib().swd_osdw_decode_batch(self._h, 1, s.ctypes.data, out.ctypes.data, st.ctypes.data,
                                              pm.ctypes.data, self._hist.ctypes.data, 1, osd0.ctypes.data, bpd.ctypes.data)
        if rc:
            raise RuntimeError(f"swd_osdw_decode_batch failed: {_lib.last_error()}")
        self._last = dict(status=int(st[0]), iters=int(st[1]), min_pm=float(pm[0]))
        res = out.astype(np.int64)
        if (int(st[0]) & 0xFF) == EXIT_OSD:
            self._osdw_decoding = res
            self._osd0_decoding = osd0.astype(np.int64)
            self._bp_decoding = bpd.astype(np.int64)  # the BP decisions the OSD started from (osd_window.pyx:499-501)
        else:
            self._bp_decoding = res
        return res

    def decode_batch(self, syndromes, return_history=False, return_osd0=False):
        """B syndromes [B, m] -> uint8 [B, n]; every shot starts from a zero LLR history (the state
        of a freshly constructed reference object).  Per-shot results are kept in
        ``last_status`` / ``last_iterations`` / ``last_min_pm`` (and ``last_history`` [B,4,n])."""
        s = np.asarray(syndromes)
        if s.ndim != 2 or s.shape[1] != self.m:
            raise ValueError(f"syndromes must have shape [B, {self.m}]")
        s = np.ascontiguousarray((s.astype(np.int64) & 0xFF).astype(np.uint8))
        B = s.shape[0]
        out = np.zeros((B, self.n), dtype=np.uint8)
        st = np.zeros((B, _lib.STAT_WORDS), np.int32)
        pm = np.zeros(B, np.float64)
        hist = np.zeros((B, 4, self.n), np.float64) if return_history else None
        osd0 = np.zeros((B, self.n), np.uint8) if return_osd0 else None
        rc = _lib.lib().swd_osdw_decode_batch(self._h, B, s.ctypes.data, out.ctypes.data, st.ctypes.data,
                                              pm.ctypes.data, hist.ctypes.data if hist is not None else None, 0,
                                              osd0.ctypes.data if osd0 is not None else None, None)
        if rc:
            raise RuntimeError(f"swd_osdw_decode_batch failed: {_lib.last_error()}")
        self.last_stats = st
        self.last_status, self.last_iterations, self.last_min_pm = st[:, 0].copy(), st[:, 1].copy(), pm
        self.last_history, self.last_osd0 = hist, osd0
        return out

    def decode_batch_device(self, synd, out=None, stats=None, min_pm=None, stream=None):
        """Device-resident batch: ``synd`` is a torch uint8 CUDA tensor [B, m] (row stride free).
        Asynchronous on the current (or given) torch stream.  Returns (out, stats[B, 8], min_pm)
        tensors; stats[:, 0] = exit class | 0x100 * converge, stats[:, 1] = bp_iteration."""
        import torch
        B = synd.shape[0]
        dev = synd.device
        out = torch.empty((B, self.n), dtype=torch.uint8, device=dev) if out is None else out
        stats = torch.empty((B, _lib.STAT_WORDS), dtype=torch.int32, device=dev) if stats is None else stats
        min_pm = torch.empty(B, dtype=torch.float64, device=dev) if min_pm is None else min_pm
        st = torch.cuda.current_stream(dev) if stream is None else stream
        rc = _lib.lib().swd_osdw_decode_batch_dev(self._h, B, synd.data_ptr(), synd.stride(0), out.data_ptr(),
                                                  out.stride(0), stats.data_ptr(), min_pm.data_ptr(), None, 0, None, None,
                                                  st.cuda_stream)
        if rc:
            raise RuntimeError(f"swd_osdw_decode_batch_dev failed: {_lib.last_error()}")
        return out, stats, min_pm

    def set_timing(self, on=True):
        _lib.lib().swd_osdw_set_timing(self._h, 1 if on else 0)

    def get_timing(self):
        ms, k = C.c_double(), C.c_int64()
        _lib.lib().swd_osdw_get_timing(self._h, C.byref(ms), C.byref(k))
        return ms.value, k.value

    @property
    def bp_iteration(self):
        return self._last["iters"]

    @property
    def converge(self):
        return 1 if (self._last["status"] & STATUS_CONVERGE) else 0

    @property
    def min_pm(self):
        return self._last["min_pm"]

    @property
    def exit_class(self):
        return self._last["status"] & 0xFF

    @property
    def bp_decoding(self):
        return self._bp_decoding.copy()

    @property
    def osdw_decoding(self):
        return self._osdw_decoding.copy()

    @property
    def osd0_decoding(self):
        return self._osd0_decoding.copy()

    @property
    def log_prob_ratios(self):
        return np.ascontiguousarray(self._hist.T)


def hypotheses_shape(hyp):
    """(max_tree_depth D, max_side_depth S) of the decimation tree with exactly ``hyp`` root-to-leaf hypotheses per shot:
    leaves = 1 (main) + (S - D) (side branches below the tree) + 2 (2^D - 1) (two per tree node; bpgd.cpp:576-577,
    bp_guessing_decoder.pyx:181) with the deepest full tree that fits -- 64 -> D = 5, S = 6; 32 -> (4, 5); 16 -> (3, 4); 100 -> (5, 42).
    The device holds trees of depth <= 6 and at most 160 snapshots (2 (2^D - 1) + S - D)."""
    hyp = int(hyp)
    if hyp < 1:
        raise ValueError("hypotheses must be a positive integer")
    D = 0
    while D < 6 and 2 * (2 ** (D + 1) - 1) + 1 <= hyp:
        D += 1
    side = hyp - (2 * (2 ** D - 1) + 1)
    if 2 * (2 ** D - 1) + side > 160:
        raise ValueError(f"hypotheses={hyp}: the device holds at most 161 hypotheses per shot (tree depth 6 would need "
                         f"{2 * (2 ** D - 1) + side} snapshots, limit 160)")
    return D, D + side


def _gdg_params(kwargs, mode):
    """kwargs of bp_guessing_decoder.pyx:7-9, 162-171, 475-478.  ``hypotheses=H`` (not a reference kwarg) picks the tree
    shape with H hypotheses per shot (``hypotheses_shape``: 64 -> max_tree_depth 5, max_side_depth 6) and scores every leaf
    of gdg()'s tree (``multi_thread=2``, this package's own ensemble)."""
    kwargs = dict(kwargs)
    hyp = kwargs.pop("hypotheses", None)
    if hyp is not None:
        D, S_ = hypotheses_shape(hyp)
        kwargs.update(max_tree_depth=D, max_side_depth=S_, multi_thread=2)  # 2: every leaf of gdg()'s tree (not a reference mode)
    new_n = kwargs.get("new_n", None)
    return _lib.GdgParams(int(kwargs.get("max_iter", 50)), float(kwargs.get("ms_scaling_factor", 1.0)),
                          int(kwargs.get("max_iter_per_step", 6)), int(kwargs.get("max_step", 25)),
                          int(kwargs.get("max_tree_depth", 3)), int(kwargs.get("max_side_depth", 10)),
                          int(kwargs.get("max_tree_branch_step", 10)), int(kwargs.get("max_side_branch_step", 10)),
                          float(kwargs.get("gdg_factor", kwargs.get("gd_factor", 1.0))),
                          int(new_n) if new_n else 0, int(bool(kwargs.get("low_error_mode", False))), mode,
                          (2 if kwargs.get("multi_thread", False) == 2 else int(bool(kwargs.get("multi_thread", False)))) if mode == 0 else 0)


class bp_history_decoder:
    """Plain min-sum BP with a 4-deep posterior history (reference: src/bp_guessing_decoder.pyx:5-158)
    and base class of the guessing decoders.  ``bpgdg_decoder``: the side branches of a shot's decimation tree run
    concurrently on different workgroups; with ``multi_thread=False`` (default) the result is that of the reference's
    deterministic single-thread ``gdg()`` bit for bit; ``multi_thread=True`` runs the reference's threaded ensemble
    (bpgd.cpp:419-688: main thread, 2^D - 1 tree threads, S - D side threads) with the thread bodies in a fixed order, equal to
    the reference on every syndrome whose winning path metric is unique (``last_stats[:, 7]`` counts the tied, different
    vectors) -- the hypotheses are the leaves of a prefix tree whose shared BP blocks and scans run once (gdg_ensemble_tree).
    Every decode of the ensemble has the state of a NEWLY BUILT reference object: when ``BPGD::reset`` fails the zero vector is
    returned (a re-used reference object returns its previous decode's ``min_pm_error``, bpgd.cpp:583, 619-625), and each thread's
    posterior history starts from zeros (a re-used reference thread keeps its own stale slots when ``max_iter_per_step < 4``);
    ``hypotheses=H`` is this package's own ensemble over every leaf of gdg()'s tree with H leaves (no reference counterpart)."""
    _mode = 2

    def __init__(self, parity_check_matrix, **kwargs):
        L = _lib.lib()
        self._csr = _Csr(parity_check_matrix, kwargs.get("channel_probs"))
        self.m, self.n = self._csr.m, self._csr.n
        self.device = int(kwargs.get("device", 0))
        p = _gdg_params(kwargs, self._mode)
        self._h = L.swd_gdg_create(C.byref(self._csr.desc), C.byref(p), self.device)
        if not self._h:
            raise RuntimeError(f"swd_gdg_create failed: {_lib.last_error()}")
        self._hist = np.zeros((4, self.n), dtype=np.float64)
        self._last = dict(status=0, iters=0, min_pm=0.0)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().swd_gdg_destroy(h)
            except Exception:
                pass
            self._h = None

    def decode(self, input_vector):
        s = _as_synd(input_vector, self.m)
        out = np.zeros(self.n, dtype=np.uint8)
        st = np.zeros(_lib.STAT_WORDS, np.int32)
        pm = np.zeros(1, np.float64)
        rc = _lib.lib().swd_gdg_decode_batch(self._h, 1, s.ctypes.data, out.ctypes.data, st.ctypes.data,
                                             pm.ctypes.data, None, 0)
        if rc:
            raise RuntimeError(f"swd_gdg_decode_batch failed: {_lib.last_error()}")
        self._last = dict(status=int(st[0]), iters=int(st[1]), min_pm=float(pm[0]))
        return out.astype(np.int64)

    def decode_batch(self, syndromes):
        s = np.asarray(syndromes)
        if s.ndim != 2 or s.shape[1] != self.m:
            raise ValueError(f"syndromes must have shape [B, {self.m}]")
        s = np.ascontiguousarray((s.astype(np.int64) & 0xFF).astype(np.uint8))
        B = s.shape[0]
        out = np.zeros((B, self.n), dtype=np.uint8)
        st = np.zeros((B, _lib.STAT_WORDS), np.int32)
        pm = np.zeros(B, np.float64)
        rc = _lib.lib().swd_gdg_decode_batch(self._h, B, s.ctypes.data, out.ctypes.data, st.ctypes.data,
                                             pm.ctypes.data, None, 0)
        if rc:
            raise RuntimeError(f"swd_gdg_decode_batch failed: {_lib.last_error()}")
        self.last_stats, self.last_min_pm = st, pm
        self.last_status = st[:, 0].copy()
        return out

    @property
    def converge(self):
        return bool(self._last["status"] & STATUS_CONVERGE)


class bpgdg_decoder(bp_history_decoder):
    """BP + guided decimation guessing (reference: src/bp_guessing_decoder.pyx:160-442).

    ``multi_thread=True`` and ``decode()`` one syndrome at a time: the reference keeps ONE ``BPGD_main_thread`` for the object's
    lifetime (bp_guessing_decoder.pyx:238-251) whose ``min_pm_error`` -- a vector over the POSITIONS of the sorted order, not over
    columns -- is never cleared (bpgd.cpp:597-599); when ``BPGD::reset`` fails, ``do_work`` returns before anything is written to it
    (:619-625) and the decoder copies the PREVIOUS decode's position vector over this decode's first new_n sorted columns.  The
    object reproduces that (``reuse_object=True``, the default; round 6): the device decodes with the state of a new object, the
    sorted order of the decode is recomputed from the pre-processing history it returns (stable argsort of the slot-order sum,
    bp_guessing_decoder.pyx:240-244), and the position vector is kept between calls.  ``decode_batch`` gives every shot a new
    object's state (zero vector on a failed reset)."""
    _mode = 0

    def __init__(self, parity_check_matrix, **kwargs):
        super().__init__(parity_check_matrix, **kwargs)
        nn = kwargs.get("new_n", None)
        self._new_n = min(int(nn), self.n) if nn else min(self.n, 2 * self.m)  # bp_guessing_decoder.pyx:186-189
        self._ens = kwargs.get("multi_thread", False) is True or kwargs.get("multi_thread", False) == 1
        self._ens = bool(self._ens) and "hypotheses" not in kwargs
        self._reuse = bool(kwargs.get("reuse_object", True))
        self._prev_pos = np.zeros(self._new_n, np.uint8)  # min_pm_error of a new object: zeros

    def decode(self, input_vector):
        if not (self._ens and self._reuse):
            return super().decode(input_vector)
        s = _as_synd(input_vector, self.m)
        out = np.zeros(self.n, dtype=np.uint8)
        st = np.zeros(_lib.STAT_WORDS, np.int32)
        pm = np.zeros(1, np.float64)
        hist = np.zeros((4, self.n), np.float64)
        rc = _lib.lib().swd_gdg_decode_batch(self._h, 1, s.ctypes.data, out.ctypes.data, st.ctypes.data, pm.ctypes.data, hist.ctypes.data, 0)
        if rc:
            raise RuntimeError(f"swd_gdg_decode_batch failed: {_lib.last_error()}")
        self._last = dict(status=int(st[0]), iters=int(st[1]), min_pm=float(pm[0]))
        exit_class = int(st[0]) & 0xFF
        if exit_class == EXIT_PRE:  # the pre-processing BP converged: the ensemble object is not touched
            return out.astype(np.int64)
        llr_sum = ((hist[0] + hist[1]) + hist[2]) + hist[3]
        cols = np.argsort(llr_sum, kind="stable")[:self._new_n]
        if exit_class == EXIT_FAIL_PEEL:  # BPGD::reset failed: the previous decode's position vector over THIS decode's sorted columns
            out[:] = 0
            out[cols] = self._prev_pos
        else:
            self._prev_pos = out[cols].copy()
        return out.astype(np.int64)


class bpgd_decoder(bp_history_decoder):
    """BP + guided decimation (reference: src/bp_guessing_decoder.pyx:473-571)."""
    _mode = 1


class bp4_osd:
    """Quaternary BP + OSD (reference: src/bp4_osd.pyx).  ``decode(sx, sz)`` returns the int64 array of
    shape (2, n) the reference returns (row 0: X string, row 1: Z string)."""

    def __init__(self, Hx, Hz, **kwargs):
        L = _lib.lib()
        if not (isinstance(Hx, np.ndarray) or sp.issparse(Hx)):
            raise TypeError("The input matrix is of an invalid type. Please input a np.ndarray or "
                            "scipy.sparse.spmatrix object.")
        if Hx.shape[1] != Hz.shape[1]:
            raise ValueError("Hx, Hz blocklength does not match!")
        n = Hx.shape[1]
        probs = []
        for key in ("channel_probs_x", "channel_probs_y", "channel_probs_z"):
            v = kwargs.get(key)
            if v is None:
                raise ValueError(f"{key} is required")
            v = np.ascontiguousarray(v, dtype=np.float64)
            if len(v) != n:
                raise ValueError("The length of the channel probability vector must be eqaul to the "
                                 f"block length n={n}.")
            probs.append(v)
        self._px, self._py, self._pz = probs
        self._cx, self._cz = _Csr(Hx, self._px), _Csr(Hz, self._pz)
        self.mx, self.mz, self.n = self._cx.m, self._cz.m, n
        method, order = _parse_osd_method(kwargs.get("osd_method", "osd_0"), kwargs.get("osd_order", 0))
        self.device = int(kwargs.get("device", 0))
        p = _lib.Bp4Params(int(kwargs.get("max_iter", 32)), float(kwargs.get("ms_scaling_factor", 1.0)), method, order)
        self._h = L.swd_bp4_create(C.byref(self._cx.desc), C.byref(self._cz.desc), self._px.ctypes.data,
                                   self._py.ctypes.data, self._pz.ctypes.data, C.byref(p), self.device)
        if not self._h:
            msg = _lib.last_error()
            if "OSD order" in msg or "invalid" in msg or "blocklength" in msg:
                raise ValueError(msg)
            raise RuntimeError(f"swd_bp4_create failed: {msg}")
        i = [C.c_int32() for _ in range(5)]
        L.swd_bp4_info(self._h, *[C.byref(x) for x in i])
        self.rank_x, self.rank_z = i[3].value, i[4].value
        self._last = dict(status=0, iters=0)
        self._lpr = np.zeros((3, n))
        self._osd0 = np.zeros((2, n), np.int64)
        self._out = np.zeros((2, n), np.int64)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().swd_bp4_destroy(h)
            except Exception:
                pass
            self._h = None

    def decode_batch(self, synd_x, synd_z, return_llr=False, details=True):
        """[B, mx], [B, mz] -> uint8 [B, 2, n] (X string, Z string per decode).  ``details=False`` leaves the posteriors, the OSD-0
        vectors and the BP decisions on the device (``last_llr`` / ``last_osd0`` / ``last_bp_decoding`` become None): 2 n + 32 bytes
        come back per decode instead of 30 n."""
        sx, sz = np.asarray(synd_x), np.asarray(synd_z)
        if sx.ndim != 2 or sx.shape[1] != self.mx or sz.ndim != 2 or sz.shape[1] != self.mz or sx.shape[0] != sz.shape[0]:
            raise ValueError(f"syndromes must have shapes [B, {self.mx}] and [B, {self.mz}]")
        sx = np.ascontiguousarray((sx.astype(np.int64) & 0xFF).astype(np.uint8))
        sz = np.ascontiguousarray((sz.astype(np.int64) & 0xFF).astype(np.uint8))
        B = sx.shape[0]
        out = np.zeros((B, 2, self.n), np.uint8)
        st = np.zeros((B, _lib.STAT_WORDS), np.int32)
        lpr = np.zeros((B, 3, self.n)) if details else None
        osd0 = np.zeros((B, 2, self.n), np.uint8) if details else None
        bpd = np.zeros((B, 2, self.n), np.uint8) if details else None
        rc = _lib.lib().swd_bp4_decode_batch(self._h, B, sx.ctypes.data, sz.ctypes.data, out.ctypes.data, st.ctypes.data,
                                             lpr.ctypes.data if details else None, osd0.ctypes.data if details else None,
                                             bpd.ctypes.data if details else None)
        if rc:
            raise RuntimeError(f"swd_bp4_decode_batch failed: {_lib.last_error()}")
        self.last_stats, self.last_status, self.last_iterations = st, st[:, 0].copy(), st[:, 1].copy()
        self.last_llr, self.last_osd0, self.last_bp_decoding = lpr, osd0, bpd
        return out

    def decode_batch_device(self, synd_x, synd_z, out=None, stats=None, llr=None, stream=None):
        """Device-resident batch: ``synd_x`` [B, mx], ``synd_z`` [B, mz] contiguous torch uint8 CUDA tensors ->
        (out uint8 [B, 2, n], stats int32 [B, 8]); ``llr`` (float64 [B, 3, n], optional) receives the posterior LLRs.
        Asynchronous on the current (or given) torch stream."""
        import torch
        B, dev = synd_x.shape[0], synd_x.device
        for t, m in ((synd_x, self.mx), (synd_z, self.mz)):
            if t.dtype != torch.uint8 or t.dim() != 2 or t.shape != (B, m) or not t.is_contiguous() or not t.is_cuda:
                raise ValueError(f"syndromes must be contiguous uint8 CUDA tensors [B, {self.mx}] and [B, {self.mz}]")
        out = torch.empty((B, 2, self.n), dtype=torch.uint8, device=dev) if out is None else out
        stats = torch.empty((B, _lib.STAT_WORDS), dtype=torch.int32, device=dev) if stats is None else stats
        st = torch.cuda.current_stream(dev) if stream is None else stream
        rc = _lib.lib().swd_bp4_decode_batch_dev(self._h, B, synd_x.data_ptr(), synd_z.data_ptr(), out.data_ptr(), stats.data_ptr(),
                                                 llr.data_ptr() if llr is not None else None, None, None, st.cuda_stream)
        if rc:
            raise RuntimeError(f"swd_bp4_decode_batch_dev failed: {_lib.last_error()}")
        return out, stats

    def decode(self, input_vector_x, input_vector_z):
        sx, sz = np.asarray(input_vector_x), np.asarray(input_vector_z)
        if sx.shape[0] != self.mx or sz.shape[0] != self.mz:
            raise ValueError(f"The input to the bp4_osd.decode must be a syndrome (of length={self.mx}).")
        out = self.decode_batch(sx[None, :], sz[None, :])
        self._last = dict(status=int(self.last_status[0]), iters=int(self.last_iterations[0]))
        self._lpr = self.last_llr[0]
        self._out = out[0].astype(np.int64)
        self._bp = self.last_bp_decoding[0].astype(np.int64)
        if (self._last["status"] & 0xFF) in (EXIT_PRE, EXIT_OSD):
            self._osd0 = self.last_osd0[0].astype(np.int64)
        if (self._last["status"] & 0xFF) == EXIT_OSD:  # osdw_decoding_* only change when the OSD ran (bp4_osd.pyx:217-219)
            self._osdw = self._out.copy()
        return self._out.copy()

    def camel_decode_batch(self, synd_x, synd_z):
        """camel_decode (bp4_osd.pyx:223-247) for a batch: uint8 [B, 2, n]; ``last_status`` / ``last_iterations`` /
        ``last_min_pm`` per shot.  Every shot sees a newly built object: zero vectors when no run converges."""
        sx, sz = np.asarray(synd_x), np.asarray(synd_z)
        if sx.ndim != 2 or sx.shape[1] != self.mx or sz.ndim != 2 or sz.shape[1] != self.mz or sx.shape[0] != sz.shape[0]:
            raise ValueError(f"syndromes must have shapes [B, {self.mx}] and [B, {self.mz}]")
        sx = np.ascontiguousarray((sx.astype(np.int64) & 0xFF).astype(np.uint8))
        sz = np.ascontiguousarray((sz.astype(np.int64) & 0xFF).astype(np.uint8))
        B = sx.shape[0]
        out = np.zeros((B, 2, self.n), np.uint8)
        st = np.zeros((B, _lib.STAT_WORDS), np.int32)
        pm = np.full(B, 10000.0)
        rc = _lib.lib().swd_bp4_camel_decode_batch(self._h, B, sx.ctypes.data, sz.ctypes.data, out.ctypes.data, st.ctypes.data,
                                                   pm.ctypes.data)
        if rc:
            raise RuntimeError(f"swd_bp4_camel_decode_batch failed: {_lib.last_error()}")
        self.last_stats, self.last_status, self.last_iterations, self.last_min_pm = st, st[:, 0].copy(), st[:, 1].copy(), pm
        return out

    def camel_decode(self, input_vector_x, input_vector_z):
        """Same call as the reference's ``camel_decode``; the returned (2, n) array is also what
        ``osd0_decoding_x`` / ``osd0_decoding_z`` hold afterwards."""
        sx, sz = np.asarray(input_vector_x), np.asarray(input_vector_z)
        if sx.shape[0] != self.mx or sz.shape[0] != self.mz:
            raise ValueError(f"The input to the bp4_osd.decode must be a syndrome (of length={self.mx}).")
        out = self.camel_decode_batch(sx[None, :], sz[None, :])
        self._last = dict(status=int(self.last_status[0]), iters=int(self.last_iterations[0]))
        self._min_pm = float(self.last_min_pm[0])
        self._osd0 = out[0].astype(np.int64)
        return self._osd0.copy()

    converge = property(lambda self: 1 if (self._last["status"] & STATUS_CONVERGE) else 0)
    bp_iteration = property(lambda self: self._last["iters"])
    min_pm = property(lambda self: getattr(self, "_min_pm", 0.0))
    bp_decoding_x = property(lambda self: getattr(self, "_bp", np.zeros((2, self.n), np.int64))[0].copy())
    bp_decoding_z = property(lambda self: getattr(self, "_bp", np.zeros((2, self.n), np.int64))[1].copy())
    osdw_decoding_x = property(lambda self: getattr(self, "_osdw", np.zeros((2, self.n), np.int64))[0].copy())
    osdw_decoding_z = property(lambda self: getattr(self, "_osdw", np.zeros((2, self.n), np.int64))[1].copy())
    osd0_decoding_x = property(lambda self: self._osd0[0].copy())
    osd0_decoding_z = property(lambda self: self._osd0[1].copy())

    @property
    def log_prob_ratios(self):
        return np.ascontiguousarray(self._lpr.T)


class _HostPool:
    """Result arrays of the host-buffer calls, recycled.  A fresh ``np.empty`` of 36 MB (total_e_hat of a 4096-shot batch of the
    [[144,12,12]] experiment) is 9 000 untouched pages that the unpacking threads fault in one by one -- a quarter of the call.  The
    arrays handed out are views of pooled base arrays; a base whose views have all been dropped by the caller (reference count back
    at the pool's own) is handed out again, warm.  A caller that keeps every result simply gets fresh arrays as before."""

    def __init__(self, per_shape=3, shapes=6):
        self._bufs, self._per_shape, self._shapes = {}, per_shape, shapes
        # What sys.getrefcount reports for a pooled base nobody else holds depends on the interpreter (borrowed operand-stack
        # references, free-threaded builds): measure it with the loop shape take() uses instead of assuming CPython 3.10's 3, and
        # check the probe the other way -- one live view must read exactly one more.  A probe that disagrees switches recycling off.
        self._idle = self._probe(hold_view=False)
        if self._idle is None or self._probe(hold_view=True) != self._idle + 1:
            self._idle = None

    @staticmethod
    def _probe(hold_view):
        import sys
        lst = [np.empty(4, np.uint8)]
        view = lst[0][...] if hold_view else None
        n = None
        for base in lst:
            n = sys.getrefcount(base)
        del view
        return n

    def take(self, shape, dtype):
        import sys
        key = (tuple(int(x) for x in shape), np.dtype(dtype).str)
        if self._idle is None:  # reference counts are not readable the way the probe expects: plain fresh arrays
            return np.empty(key[0], dtype)
        lst = self._bufs.get(key)
        if lst is None:
            if len(self._bufs) >= self._shapes:
                self._bufs.pop(next(iter(self._bufs)))
            lst = self._bufs[key] = []
        for base in lst:
            if sys.getrefcount(base) == self._idle:  # the calibrated count of a base with no view alive
                return base[...]
        base = np.empty(key[0], dtype)
        if len(lst) < self._per_shape:
            lst.append(base)
        return base[...]


class SlidingWindowDecoder:
    """The (W,F) sliding-window loop of the reference harness (/root/reference/osd.py:130-179) for
    a whole batch of shots in ONE launch: a workgroup carries a shot through all its windows,
    committing ``commit`` columns per window and folding them back into the residual syndrome.

    ``plan`` is a ``windows.WindowPlan``; decoder kwargs are those of ``osd_window`` and apply to
    every window like in osd.py:152-161."""

    def __init__(self, plan, device=0, decoder="osd_window", **kwargs):
        L = _lib.lib()
        self.plan = plan
        self.W = len(plan.windows)
        self.num_det, self.num_col = plan.chk.shape
        self.decoder = decoder
        if decoder == "osd_window":
            method, order = _parse_osd_method(kwargs.get("osd_method", "osd_0"), kwargs.get("osd_order", 0))
            new_n = kwargs.get("new_n", None)
            p = _lib.OsdwParams(int(kwargs.get("pre_max_iter", 8)), int(kwargs.get("post_max_iter", 100)),
                                float(kwargs.get("ms_scaling_factor", 1.0)), int(new_n) if new_n else 0, method, order)
        elif decoder in ("bpgdg_decoder", "bpgd_decoder", "bp_history_decoder"):
            p = _gdg_params(kwargs, {"bpgdg_decoder": 0, "bpgd_decoder": 1, "bp_history_decoder": 2}[decoder])
        else:
            raise ValueError(f"unknown window decoder {decoder!r}")
        self._keep = []
        descs = (_lib.WindowDesc * self.W)()
        for i, w in enumerate(plan.windows):
            c = _Csr(w.mat, w.prior)
            self._keep.append(c)
            descs[i].graph = c.desc
            descs[i].row0, descs[i].col0, descs[i].commit = int(w.row0), int(w.col0), int(w.commit)
        chk = _Csr(plan.chk, plan.priors)
        self._keep.append(chk)
        self.device = int(device)
        self._pool = _HostPool()
        create = L.swd_pipeline_create if decoder == "osd_window" else L.swd_pipeline_create_gdg
        self._h = create(self.W, C.cast(descs, C.c_void_p), C.byref(chk.desc), C.byref(p), self.device)
        self._loop = None
        if not self._h:
            msg = _lib.last_error()
            if "OSD order" in msg or "invalid" in msg:
                raise ValueError(msg)
            if decoder == "osd_window" and ("no kernel variant" in msg or "exceeds" in msg or "needs" in msg):
                # windows beyond every kernel variant (e.g. the un-windowed [[288,12,18]] model, 2736 x 26 208): the window loop of
                # /root/reference/osd.py:130-179 as the reference runs it -- one osd_window per window (device: the general form,
                # csrc/swd_huge.hip), the residual syndrome det ^ chk @ total_e_hat between windows (osd.py:165, 178) on the host
                self._loop = [osd_window(w.mat, channel_probs=w.prior, device=self.device,
                                         **{k: v for k, v in kwargs.items() if k != "device"}) for w in plan.windows]
                self._chk_t = sp.csr_matrix(plan.chk.T.astype(np.int32))
                self.lds_bytes, self.threads = 0, 1024
                self.num_obs = int(plan.obs.shape[0]) if plan.obs is not None else 0
                self._obs_t = sp.csr_matrix(plan.obs.T.astype(np.int32)) if self.num_obs else None
                return
            raise RuntimeError(f"swd_pipeline_create failed: {msg}")
        i = [C.c_int32() for _ in range(5)]
        L.swd_pipeline_info(self._h, *[C.byref(x) for x in i])
        self.lds_bytes, self.threads = i[3].value, i[4].value
        self.num_obs = int(plan.obs.shape[0]) if plan.obs is not None else 0
        if 0 < self.num_obs <= 32:
            a = sp.csr_matrix(plan.obs)
            a.sort_indices()
            rp, ci = np.ascontiguousarray(a.indptr, np.int32), np.ascontiguousarray(a.indices, np.int32)
            od = _lib.GraphDesc(a.shape[0], a.shape[1], int(rp[-1]), rp.ctypes.data, ci.ctypes.data, None)
            if L.swd_pipeline_set_observables(self._h, C.byref(od)):
                raise RuntimeError(f"swd_pipeline_set_observables failed: {_lib.last_error()}")

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().swd_pipeline_destroy(h)
            except Exception:
                pass
            self._h = None

    def _check_det(self, det_data):
        d = np.asarray(det_data)
        if d.ndim != 2 or d.shape[1] != self.num_det:
            raise ValueError(f"det_data must have shape [B, {self.num_det}]")
        if d.dtype != np.uint8 or not d.flags.c_contiguous:
            d = np.ascontiguousarray((d.astype(np.int64) & 0xFF).astype(np.uint8))
        return d

    def decode(self, det_data, packed=False):
        """det_data [B, num_det] (host) -> total_e_hat uint8 [B, num_col]; per-window records in
        ``last_stats`` [B, W, 8] and ``last_min_pm`` [B, W].  ``packed=True`` returns total_e_hat bit-packed instead,
        uint8 [B, ceil(num_col / 8)] (``np.unpackbits(bits, axis=1, count=num_col, bitorder="little")`` gives the bytes): the
        faults always travel device -> host in that form (1098 B per shot instead of 8784 for the [[144,12,12]] experiment)."""
        d = self._check_det(det_data)
        B = d.shape[0]
        if self._loop is not None:
            return self._decode_window_loop(d, packed)
        pool = self._pool
        total = pool.take((B, (self.num_col + 7) // 8 if packed else self.num_col), np.uint8)
        st = pool.take((B, self.W, _lib.STAT_WORDS), np.int32)
        pm = pool.take((B, self.W), np.float64)
        shot = np.empty((B, 2), np.int32)
        fn = _lib.lib().swd_pipeline_decode_packed if packed else _lib.lib().swd_pipeline_decode
        rc = fn(self._h, B, d.ctypes.data, total.ctypes.data, st.ctypes.data, pm.ctypes.data, shot.ctypes.data)
        if rc:
            raise RuntimeError(f"swd_pipeline_decode failed: {_lib.last_error()}")
        self.last_stats, self.last_min_pm = st, pm
        self.last_obs_flips, self.last_flagged = shot[:, 0].astype(np.uint32), shot[:, 1].astype(bool)
        return total

    def _decode_window_loop(self, d, packed):
        """the window loop for plans no pipeline kernel takes: per window one batched osd_window decode on the device, commit, residual"""
        B = d.shape[0]
        total = np.zeros((B, self.num_col), np.uint8)
        st = np.zeros((B, self.W, _lib.STAT_WORDS), np.int32)
        pm = np.zeros((B, self.W), np.float64)
        cur = d
        for wi, (w, dec) in enumerate(zip(self.plan.windows, self._loop)):
            out = dec.decode_batch(np.ascontiguousarray(cur[:, w.row0:w.row1])) if B else np.zeros((0, w.mat.shape[1]), np.uint8)
            total[:, w.col0:w.col0 + w.commit] = out[:, :w.commit]
            if B:
                st[:, wi, 0], st[:, wi, 1], pm[:, wi] = dec.last_status, dec.last_iterations, dec.last_min_pm
            cur = ((d.astype(np.int32) + (sp.csr_matrix(total) @ self._chk_t).toarray()) % 2).astype(np.uint8)  # osd.py:178
        self.last_stats, self.last_min_pm = st, pm
        self.last_flagged = cur.any(axis=1)
        flips = np.zeros(B, np.uint32)
        if self._obs_t is not None and self.num_obs <= 32 and B:
            bits = ((sp.csr_matrix(total) @ self._obs_t).toarray() % 2).astype(np.uint32)
            flips = (bits << np.arange(self.num_obs, dtype=np.uint32)).sum(axis=1).astype(np.uint32)
        self.last_obs_flips = flips
        return np.packbits(total, axis=1, bitorder="little") if packed else total

    def _no_loop(self, what):
        if self._loop is not None:
            raise RuntimeError(f"{what} needs the one-launch pipeline; this plan's windows are beyond every kernel variant and run as a "
                               "window loop of osd_window decodes (use decode())")

    def stream(self, max_shots, packed=False, want_stats=True):
        """Streaming form for consecutive batches (``SlidingWindowStream``): two batches in flight on two lanes of this
        pipeline, copies and unpacking of batch k overlapped with the launch of batch k + 1."""
        self._no_loop("stream()")
        return SlidingWindowStream(self, max_shots, packed=packed, want_stats=want_stats)

    def decode_stream(self, batches, packed=False, want_stats=True):
        """Generator over an iterable of host batches [B_k, num_det]: yields (total_e_hat, stats, min_pm, obs_flips, flagged) per
        batch, in order, keeping two batches in flight (the deployment form of the shots loop of /root/reference/osd.py:130-191)."""
        st, it = None, iter(batches)
        try:
            for d in it:
                d = self._check_det(d)
                if st is not None and d.shape[0] > st.max_shots:  # a larger batch than the lanes were sized for: drain, then re-create
                    while st.pending:
                        yield st.pop()
                    st.close()
                    st = None
                if st is None:
                    st = self.stream(d.shape[0], packed=packed, want_stats=want_stats)
                if st.pending == 2:
                    yield st.pop()
                st.push(d)
            while st is not None and st.pending:
                yield st.pop()
        finally:
            if st is not None:
                st.close()

    def decode_device(self, det, total=None, stats=None, min_pm=None, shot_result=None, stream=None,
                      want_stats=True):
        """torch uint8 CUDA tensor [B, num_det] -> (total [B, num_col], stats [B, W, 8], min_pm [B, W]);
        asynchronous on the current torch stream.  ``shot_result`` (int32 [B, 2] CUDA tensor, optional)
        receives the predicted observable-flip mask and the flagged bit of every shot."""
        import torch
        self._no_loop("decode_device()")
        if det.dtype != torch.uint8 or det.dim() != 2 or det.shape[1] != self.num_det or det.stride(1) != 1:
            raise ValueError(f"det must be a uint8 tensor [B, {self.num_det}] with unit column stride")
        if not det.is_cuda or det.device.index != self.device:
            raise ValueError(f"det lives on {det.device}, the pipeline on cuda:{self.device}")
        if shot_result is not None and self.num_obs > 32:
            raise ValueError("shot_result carries at most 32 observables; decode the observables on the host for more")
        B, dev = det.shape[0], det.device
        total = torch.empty((B, self.num_col), dtype=torch.uint8, device=dev) if total is None else total
        if want_stats:
            stats = torch.empty((B, self.W, _lib.STAT_WORDS), dtype=torch.int32, device=dev) if stats is None else stats
            min_pm = torch.empty((B, self.W), dtype=torch.float64, device=dev) if min_pm is None else min_pm
        st = torch.cuda.current_stream(dev) if stream is None else stream
        rc = _lib.lib().swd_pipeline_decode_dev(self._h, B, det.data_ptr(), det.stride(0), total.data_ptr(),
                                                total.stride(0), stats.data_ptr() if stats is not None else None,
                                                min_pm.data_ptr() if min_pm is not None else None,
                                                shot_result.data_ptr() if shot_result is not None else None,
                                                st.cuda_stream)
        if rc:
            raise RuntimeError(f"swd_pipeline_decode_dev failed: {_lib.last_error()}")
        return total, stats, min_pm

    def check_status(self):
        """Synchronises the device and raises if any launch of this pipeline since the last check recorded a
        scheduling fault (a window whose predecessor never finished: exit class 6 in ``stats``).  The host-buffer
        ``decode`` checks by itself; callers of the asynchronous ``decode_device`` call this after their launches."""
        if self._loop is not None:
            return
        flags = C.c_uint32(0)
        if _lib.lib().swd_pipeline_status(self._h, C.byref(flags)):
            raise RuntimeError(f"swd_pipeline_status failed: {_lib.last_error()}")
        if flags.value:
            raise RuntimeError(f"sliding-window pipeline recorded a scheduling fault (flags 0x{flags.value:x})")

    def set_profiling(self, on=True):
        _lib.lib().swd_pipeline_set_profiling(self._h, 1 if on else 0)

    def get_profile(self, B):
        """[B, W, 8] int64 ticks (100 MHz) per phase of the last launch (diagnostics)."""
        out = np.zeros((B, self.W, 8), np.int64)
        if _lib.lib().swd_pipeline_get_profile(self._h, B, out.ctypes.data):
            raise RuntimeError(_lib.last_error())
        return out

    def set_timing(self, on=True):
        _lib.lib().swd_pipeline_set_timing(self._h, 1 if on else 0)

    def get_timing(self):
        ms, k = C.c_double(), C.c_int64()
        _lib.lib().swd_pipeline_get_timing(self._h, C.byref(ms), C.byref(k))
        return ms.value, k.value


class SlidingWindowStream:
    """Two-lane stream of a ``SlidingWindowDecoder`` (C ABI: swd_pipeline_stream_*).  Host form: ``push(det)`` returns at once,
    ``pop()`` waits for the oldest batch in flight -> (total_e_hat, stats, min_pm, obs_flips, flagged).  Device form:
    ``push_device`` launches on the next lane with the caller's CUDA tensors (keep one set of outputs per lane), ``wait()`` joins."""

    def __init__(self, dec, max_shots, packed=False, want_stats=True):
        self.dec, self.packed, self.want_stats = dec, bool(packed), bool(want_stats)
        self.max_shots = int(max_shots)
        flags = (_lib.STREAM_PACKED if packed else 0) | (0 if want_stats else _lib.STREAM_NO_STATS)
        self._h = _lib.lib().swd_pipeline_stream_create(dec._h, self.max_shots, flags)
        if not self._h:
            raise RuntimeError(f"swd_pipeline_stream_create failed: {_lib.last_error()}")
        self._sizes = []

    def close(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().swd_pipeline_stream_destroy(h)
            except Exception:
                pass
            self._h = None

    __del__ = close

    @property
    def pending(self):
        return len(self._sizes)

    def push(self, det_data):
        d = self.dec._check_det(det_data)
        if _lib.lib().swd_pipeline_stream_push(self._h, d.shape[0], d.ctypes.data):
            raise RuntimeError(f"swd_pipeline_stream_push failed: {_lib.last_error()}")
        self._sizes.append(d.shape[0])  # (the library has copied the detector bytes into its page-locked block)

    def pop(self):
        if not self._sizes:
            raise RuntimeError("stream pop: no batch in flight")
        B, dec = self._sizes.pop(0), self.dec
        pool = dec._pool
        total = pool.take((B, (dec.num_col + 7) // 8 if self.packed else dec.num_col), np.uint8)
        st = pool.take((B, dec.W, _lib.STAT_WORDS), np.int32) if self.want_stats else None
        pm = pool.take((B, dec.W), np.float64) if self.want_stats else None
        shot = np.empty((B, 2), np.int32)
        rc = _lib.lib().swd_pipeline_stream_pop(self._h, total.ctypes.data, st.ctypes.data if st is not None else None,
                                                pm.ctypes.data if pm is not None else None, shot.ctypes.data)
        if rc != B:
            raise RuntimeError(f"swd_pipeline_stream_pop failed: {_lib.last_error()}")
        return total, st, pm, shot[:, 0].astype(np.uint32), shot[:, 1].astype(bool)

    def push_device(self, det, total, stats=None, min_pm=None, shot_result=None, after=None):
        """CUDA tensors as in ``SlidingWindowDecoder.decode_device``; ``after``: a torch stream whose work so far must precede
        the launch -- the producer of ``det`` and the last reader of this lane's output tensors (default: the current stream,
        also when that is PyTorch's default stream, whose handle is 0); ``after=False``: no dependency, the caller has
        synchronised inputs and outputs itself."""
        import torch
        dec = self.dec
        if det.dtype != torch.uint8 or det.dim() != 2 or det.shape[1] != dec.num_det or det.stride(1) != 1 or not det.is_cuda:
            raise ValueError(f"det must be a uint8 CUDA tensor [B, {dec.num_det}] with unit column stride")
        if after is False:
            aft = _lib.STREAM_NO_DEPENDENCY
        else:
            aft = (torch.cuda.current_stream(det.device) if after is None else after).cuda_stream or None  # 0 -> NULL = the legacy default stream
        rc = _lib.lib().swd_pipeline_stream_push_dev(self._h, det.shape[0], det.data_ptr(), det.stride(0), total.data_ptr(), total.stride(0),
                                                     stats.data_ptr() if stats is not None else None,
                                                     min_pm.data_ptr() if min_pm is not None else None,
                                                     shot_result.data_ptr() if shot_result is not None else None, aft)
        if rc:
            raise RuntimeError(f"swd_pipeline_stream_push_dev failed: {_lib.last_error()}")

    def wait(self, stream=None):
        """stream=None: the host waits for both lanes; a torch stream: that stream waits (device-side)."""
        if _lib.lib().swd_pipeline_stream_wait(self._h, stream.cuda_stream if stream is not None else None):
            raise RuntimeError(f"swd_pipeline_stream_wait failed: {_lib.last_error()}")


class DemSampler:
    """Samples shots of a detector error model on the device: ``det, obs = sampler.sample(shots)`` is what
    ``dem.compile_sampler().sample(shots)`` gives the reference harness (/root/reference/osd.py:124-125).
    Faults are Bernoulli(priors) per column from a Philox4x32-10 stream that is a pure function of
    (seed, shot number, column): batches and ranks can be cut anywhere (``first_shot``)."""

    def __init__(self, chk, obs, priors, device=0):
        L = _lib.lib()
        self._chk = _Csr(chk, priors)
        self.num_det, self.num_col = self._chk.m, self._chk.n
        self.num_obs = 0
        od = None
        if obs is not None:
            a = sp.csr_matrix(obs)
            a.sort_indices()
            self._orp, self._oci = np.ascontiguousarray(a.indptr, np.int32), np.ascontiguousarray(a.indices, np.int32)
            if a.shape[1] != self.num_col:
                raise ValueError(f"obs has {a.shape[1]} columns, chk has {self.num_col}")
            if a.shape[0] > 32:
                raise ValueError("at most 32 observables")
            self.num_obs = int(a.shape[0])
            od = _lib.GraphDesc(a.shape[0], a.shape[1], int(self._orp[-1]), self._orp.ctypes.data, self._oci.ctypes.data, None)
        self.device = int(device)
        self._h = L.swd_sampler_create(C.byref(self._chk.desc), C.byref(od) if od is not None else None, self.device)
        if not self._h:
            raise RuntimeError(f"swd_sampler_create failed: {_lib.last_error()}")

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                _lib.lib().swd_sampler_destroy(self._h)
            except Exception:  # interpreter shutdown: the module globals may be gone already
                pass
            self._h = None

    def sample(self, shots, seed=20240318, first_shot=0, return_faults=False):
        """-> det uint8 [shots, num_det], obs uint8 [shots, num_obs] (, faults uint8 [shots, num_col])."""
        det = np.zeros((shots, self.num_det), np.uint8)
        flips = np.zeros(shots, np.uint32)
        faults = np.zeros((shots, self.num_col), np.uint8) if return_faults else None
        if _lib.lib().swd_sampler_sample(self._h, shots, int(seed), int(first_shot), det.ctypes.data, flips.ctypes.data,
                                         faults.ctypes.data if return_faults else None):
            raise RuntimeError(f"swd_sampler_sample failed: {_lib.last_error()}")
        obs = ((flips[:, None] >> np.arange(self.num_obs, dtype=np.uint32)) & 1).astype(np.uint8)
        return (det, obs, faults) if return_faults else (det, obs)

    def sample_device(self, shots, seed=20240318, first_shot=0):
        """-> torch tensors on the device: det uint8 [shots, num_det], observable-flip bit masks int32 [shots]."""
        import torch
        dev = torch.device("cuda", self.device)
        det = torch.empty((shots, self.num_det), dtype=torch.uint8, device=dev)
        flips = torch.empty((shots,), dtype=torch.int32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        if _lib.lib().swd_sampler_sample_dev(self._h, shots, int(seed), int(first_shot), det.data_ptr(), 0, flips.data_ptr(),
                                             None, 0, st):
            raise RuntimeError(f"swd_sampler_sample_dev failed: {_lib.last_error()}")
        return det, flips
