"""ctypes binding of libswd_hip.so (C ABI: include/swd.h).

There is no CPU fallback: if the HIP library is missing or no MI355X is visible, construction of
any decoder raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported
from this package.)
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("SWD_LIB", "libswd_hip.so"))  # SWD_LIB: a development build next to it
_lib = None


class GraphDesc(C.Structure):
    _fields_ = [("m", C.c_int32), ("n", C.c_int32), ("nnz", C.c_int32),
                ("row_ptr", C.c_void_p), ("col_idx", C.c_void_p), ("channel_probs", C.c_void_p)]


class WindowDesc(C.Structure):
    _fields_ = [("graph", GraphDesc), ("row0", C.c_int32), ("col0", C.c_int32), ("commit", C.c_int32),
                ("reserved", C.c_int32)]


class OsdwParams(C.Structure):
    _fields_ = [("pre_max_iter", C.c_int32), ("post_max_iter", C.c_int32),
                ("ms_scaling_factor", C.c_double), ("new_n", C.c_int32),
                ("osd_method", C.c_int32), ("osd_order", C.c_int32)]


class GdgParams(C.Structure):
    _fields_ = [("max_iter", C.c_int32), ("ms_scaling_factor", C.c_double), ("max_iter_per_step", C.c_int32),
                ("max_step", C.c_int32), ("max_tree_depth", C.c_int32), ("max_side_depth", C.c_int32),
                ("max_tree_branch_step", C.c_int32), ("max_side_branch_step", C.c_int32),
                ("gdg_factor", C.c_double), ("new_n", C.c_int32), ("low_error_mode", C.c_int32),
                ("mode", C.c_int32), ("multi_thread", C.c_int32)]


class Bp4Params(C.Structure):
    _fields_ = [("max_iter", C.c_int32), ("ms_scaling_factor", C.c_double), ("osd_method", C.c_int32),
                ("osd_order", C.c_int32)]


# every symbol include/swd.h declares: (name, restype, argtypes)
_vp, _i32, _i64, _dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
SYMBOLS = [
    ("swd_last_error", C.c_char_p, []),
    ("swd_abi_version", C.c_int, []),
    ("swd_device_count", C.c_int, []),
    ("swd_osdw_create", _vp, [C.POINTER(GraphDesc), C.POINTER(OsdwParams), C.c_int]),
    ("swd_osdw_destroy", None, [_vp]),
    ("swd_osdw_info", C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    ("swd_osdw_decode_batch", C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    ("swd_osdw_decode_batch_dev", C.c_int, [_vp, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    ("swd_osdw_set_timing", C.c_int, [_vp, _i32]),
    ("swd_osdw_get_timing", C.c_int, [_vp, C.POINTER(_dbl), C.POINTER(_i64)]),
    ("swd_gdg_create", _vp, [C.POINTER(GraphDesc), C.POINTER(GdgParams), C.c_int]),
    ("swd_gdg_destroy", None, [_vp]),
    ("swd_gdg_decode_batch", C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32]),
    ("swd_gdg_decode_batch_dev", C.c_int, [_vp, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    ("swd_bp4_create", _vp, [C.POINTER(GraphDesc), C.POINTER(GraphDesc), _vp, _vp, _vp, C.POINTER(Bp4Params), C.c_int]),
    ("swd_bp4_destroy", None, [_vp]),
    ("swd_bp4_info", C.c_int, [_vp] + [C.POINTER(_i32)] * 5),
    ("swd_bp4_decode_batch", C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("swd_bp4_decode_batch_dev", C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("swd_bp4_camel_decode_batch", C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    ("swd_bp4_camel_decode_batch_dev", C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("swd_pipeline_create_gdg", _vp, [_i32, _vp, C.POINTER(GraphDesc), C.POINTER(GdgParams), C.c_int]),
    ("swd_pipeline_create", _vp, [_i32, _vp, C.POINTER(GraphDesc), C.POINTER(OsdwParams), C.c_int]),
    ("swd_pipeline_destroy", None, [_vp]),
    ("swd_pipeline_info", C.c_int, [_vp] + [C.POINTER(_i32)] * 5),
    ("swd_pipeline_set_observables", C.c_int, [_vp, C.POINTER(GraphDesc)]),
    ("swd_pipeline_decode", C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    ("swd_pipeline_decode_dev", C.c_int, [_vp, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp]),
    ("swd_pipeline_decode_packed", C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    ("swd_pipeline_stream_create", _vp, [_vp, _i32, _i32]),
    ("swd_pipeline_stream_destroy", None, [_vp]),
    ("swd_pipeline_stream_push", C.c_int, [_vp, _i32, _vp]),
    ("swd_pipeline_stream_pop", C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    ("swd_pipeline_stream_pending", C.c_int, [_vp]),
    ("swd_pipeline_stream_push_dev", C.c_int, [_vp, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp]),
    ("swd_pipeline_stream_wait", C.c_int, [_vp, _vp]),
    ("swd_pipeline_status", C.c_int, [_vp, C.POINTER(C.c_uint32)]),
    ("swd_pipeline_set_profiling", C.c_int, [_vp, _i32]),
    ("swd_pipeline_get_profile", C.c_int, [_vp, _i32, _vp]),
    ("swd_pipeline_set_timing", C.c_int, [_vp, _i32]),
    ("swd_pipeline_get_timing", C.c_int, [_vp, C.POINTER(_dbl), C.POINTER(_i64)]),
    ("swd_sampler_create", _vp, [C.POINTER(GraphDesc), C.POINTER(GraphDesc), C.c_int]),
    ("swd_sampler_destroy", None, [_vp]),
    ("swd_sampler_info", C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    ("swd_sampler_sample", C.c_int, [_vp, _i32, C.c_uint64, C.c_uint64, _vp, _vp, _vp]),
    ("swd_sampler_sample_dev", C.c_int, [_vp, _i32, C.c_uint64, C.c_uint64, _vp, _i64, _vp, _vp, _i64, _vp]),
    ("swd_diag_occupy", C.c_int, [C.c_int, _i32, _i32, _i32, _i32, _vp]),
]

STAT_WORDS = 8
STREAM_PACKED, STREAM_NO_STATS = 1, 2
STREAM_NO_DEPENDENCY = C.c_void_p(-1).value  # include/swd.h: SWD_STREAM_NO_DEPENDENCY


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  Two HIP runtimes in
    one process do not share a device context (the second one to initialise sees 0 devices), so
    when torch is installed its copy is loaded first with RTLD_GLOBAL and libswd_hip.so binds to
    it by SONAME; torch tensors' data_ptr()s and streams are then valid in our launches."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def lib():
    """Load libswd_hip.so; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C slidingwindowdecoder_amd/csrc). "
                "This package has no CPU fallback.")
        _preload_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            f = getattr(L, name, None)
            if f is None:
                if os.environ.get("SWD_LIB"):  # a development / bisect build may predate a symbol
                    continue
                raise RuntimeError(f"{LIB_PATH} does not export {name}: rebuild it")
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def last_error() -> str:
    return (lib().swd_last_error() or b"").decode()
