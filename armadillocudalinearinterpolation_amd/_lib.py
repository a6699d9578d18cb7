"""ctypes binding of libmi355interp.so (the C ABI in include/mi355_interp.h).

There is NO CPU fallback: if the HIP library is missing or fails to load, every
entry point raises.  PyTorch is only used by callers for device memory and
streams; nothing torch-typed crosses this boundary.
"""
import ctypes as C
import os

from . import _build

_lib = None


class MiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libmi355interp error %d: %s" % (code, msg))
        self.code = code


class EdmParams(C.Structure):
    """mi_edm_params (include/mi355_interp.h)."""
    _fields_ = [
        ("vth", C.c_float), ("a1", C.c_float), ("a2", C.c_float), ("b1", C.c_float), ("b2", C.c_float),
        ("I", C.c_float), ("L", C.c_float),
        ("newton_tol", C.c_double),
        ("newton_max_iter", C.c_uint32),
        ("n_spikes", C.c_uint32),
        ("time_horizon", C.c_float),
        ("n_grid", C.c_uint32),
        ("n_real", C.c_uint32),
        ("beta_mean", C.c_float),
        ("beta_stddev", C.c_float),
        ("seed", C.c_uint64),
        ("math_mode", C.c_int),
        ("mean_quirk", C.c_int),
        ("max_events", C.c_uint32),
        ("real_offset", C.c_uint32),
        ("dedup_identical", C.c_uint32),
    ]


_vp, _sz, _dbl, _i32, _u32, _f32 = C.c_void_p, C.c_size_t, C.c_double, C.c_int, C.c_uint32, C.c_float
_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes).  Kept in one table so tests can check it against
# the declarations in include/mi355_interp.h.
SIGNATURES = {
    "mi_abi_version": (_i32, []),
    "mi_debug_pinned_ranges": (_sz, []),
    "mi_last_error": (C.c_char_p, [_vp]),
    "mi_ctx_create": (_i32, [_i32, _pp]),
    "mi_ctx_destroy": (_i32, [_vp]),
    "mi_ctx_set_stream": (_i32, [_vp, _vp]),
    "mi_ctx_own_stream": (_i32, [_vp]),
    "mi_ctx_synchronize": (_i32, [_vp]),
    "mi_ctx_set_query_order": (_i32, [_vp, _i32]),
    "mi_ctx_device_info": (_i32, [_vp, C.c_char_p, _sz, C.POINTER(_i32), C.POINTER(_sz)]),
    "mi_timer_create": (_i32, [_vp, _pp]),
    "mi_timer_destroy": (_i32, [_vp]),
    "mi_timer_start": (_i32, [_vp]),
    "mi_timer_stop": (_i32, [_vp]),
    "mi_timer_elapsed_ms": (_i32, [_vp, C.POINTER(_f32)]),
    "mi_grid1_create": (_i32, [_vp, _vp, _vp, _sz, C.c_uint, _pp]),
    "mi_grid1_create_uniform": (_i32, [_vp, _dbl, _dbl, _vp, _sz, C.c_uint, _pp]),
    "mi_grid1_destroy": (_i32, [_vp]),
    "mi_grid1_info": (_i32, [_vp, C.POINTER(_sz), C.POINTER(_i32), C.POINTER(_sz)]),
    "mi_interp1_f64_dev": (_i32, [_vp, _vp, _vp, _vp, _sz, _dbl]),
    "mi_interp1_f64_host": (_i32, [_vp, _vp, _vp, _vp, _sz, _dbl]),
    "mi_interp1_f64": (_i32, [_vp, _vp, _vp, _sz, _vp, _vp, _sz, _dbl]),
    "mi_grid2_create": (_i32, [_vp, _vp, _sz, _vp, _sz, _vp, C.c_uint, _pp]),
    "mi_grid2_create_uniform": (_i32, [_vp, _dbl, _dbl, _sz, _dbl, _dbl, _sz, _vp, C.c_uint, _pp]),
    "mi_grid2_destroy": (_i32, [_vp]),
    "mi_grid2_info": (_i32, [_vp, C.POINTER(_sz)]),
    "mi_interp2_f64_dev": (_i32, [_vp, _vp, _vp, _vp, _vp, _sz, _dbl]),
    "mi_interp2_f64_host": (_i32, [_vp, _vp, _vp, _vp, _vp, _sz, _dbl]),
    "mi_restrict_f32_dev": (_i32, [_vp, _vp, _vp, _vp, _vp, _f32, _f32, _u32, _vp, _sz]),
    "mi_restrict_f32_host": (_i32, [_vp, _vp, _vp, _vp, _vp, _f32, _f32, _u32, _vp, _sz]),
    "mi_masked_mean_f32_dev": (_i32, [_vp, _vp, _vp, _sz, _sz, _i32, _vp, _vp, _vp]),
    "mi_restrict_mean_f32_dev": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _u32, _sz, _sz, _i32,
                                        _vp, _vp, _vp, _vp]),
    "mi_group_create": (_i32, [_i32, C.POINTER(_i32), _pp]),
    "mi_group_destroy": (_i32, [_vp]),
    "mi_group_size": (_i32, [_vp]),
    "mi_group_ctx": (_vp, [_vp, _i32]),
    "mi_group_synchronize": (_i32, [_vp]),
    "mi_shard_bounds": (None, [_sz, _i32, _i32, C.POINTER(_sz), C.POINTER(_sz)]),
    "mi_group_set_reduce": (_i32, [_vp, _i32]),
    "mi_group_rccl_ranks": (_i32, [_vp]),
    "mi_group_grid1_create": (_i32, [_vp, _vp, _vp, _sz, C.c_uint, _pp]),
    "mi_group_grid1_destroy": (_i32, [_vp]),
    "mi_group_interp1_f64_host": (_i32, [_vp, _vp, _vp, _vp, _sz, _dbl]),
    "mi_group_interp1_f64_dev": (_i32, [_vp, _vp, _pp, _pp, _sz, _dbl, _pp]),
    "mi_group_grid2_create": (_i32, [_vp, _vp, _sz, _vp, _sz, _vp, C.c_uint, _pp]),
    "mi_group_grid2_destroy": (_i32, [_vp]),
    "mi_group_interp2_f64_host": (_i32, [_vp, _vp, _vp, _vp, _vp, _sz, _dbl]),
    "mi_group_interp2_f64_dev": (_i32, [_vp, _vp, _pp, _pp, _pp, _sz, _dbl, _pp]),
    "mi_group_set_gather_chunks": (_i32, [_vp, _i32]),
    "mi_group_edm_create": (_i32, [_vp, C.POINTER(EdmParams), _pp]),
    "mi_group_edm_destroy": (_i32, [_vp]),
    "mi_group_edm_set_params": (_i32, [_vp, C.POINTER(EdmParams)]),
    "mi_group_edm_compute_f": (_i32, [_vp, _vp, _vp, _vp]),
    "mi_group_edm_shard": (_vp, [_vp, _i32]),
    "mi_group_edm_shard_bounds": (_i32, [_vp, _i32, C.POINTER(_sz), C.POINTER(_sz)]),
    "mi_edm_default_params": (None, [C.POINTER(EdmParams)]),
    "mi_edm_create": (_i32, [_vp, C.POINTER(EdmParams), _pp]),
    "mi_edm_destroy": (_i32, [_vp]),
    "mi_edm_set_params": (_i32, [_vp, C.POINTER(EdmParams)]),
    "mi_edm_set_kernel_choice": (_i32, [_vp, _i32, _i32]),
    "mi_edm_compute_f": (_i32, [_vp, _vp, _vp, _vp]),
    "mi_edm_compute_f_begin": (_i32, [_vp, _vp]),
    "mi_edm_compute_f_end": (_i32, [_vp, _vp, _vp]),
    "mi_edm_residual_from_sums": (_i32, [C.POINTER(EdmParams), _vp, _vp, _vp]),
    "mi_edm_debug_read": (_i32, [_vp] + [_vp] * 10),
    "mi_edm_debug_counters": (_i32, [_vp, C.POINTER(C.c_uint64 * 8)]),
    "mi_edm_last_timings": (_i32, [_vp, C.POINTER(_f32 * 4)]),
    "mi_edm_math_probe": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _sz]),
}


def lib_path():
    return _build.LIB_PATH


def load(build_if_missing=True, strict=True):
    """Load libmi355interp.so; raises (never falls back) when unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB_PATH
    if not os.path.exists(path):
        if not build_if_missing:
            raise RuntimeError("%s is missing: run __graft_entry__.build() (no CPU fallback exists)" % path)
        _build.build_lib()
    # PyTorch wheels bundle their own libamdhip64; two HIP runtimes in one process cannot both own the
    # device ("No HIP GPUs are available").  Import torch first so that libmi355interp.so binds to the
    # runtime torch already loaded (same SONAME) and streams/pointers are shared.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    missing = []
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing and strict:
        raise RuntimeError("libmi355interp.so lacks symbols declared in mi355_interp.h: %s" % ", ".join(missing))
    if L.mi_abi_version() != 4:
        raise RuntimeError("libmi355interp.so ABI version %d != 4" % L.mi_abi_version())
    _lib = L
    return L


def check(status, ctx=None):
    if status != 0:
        msg = load().mi_last_error(ctx)
        raise MiError(status, (msg or b"?").decode("utf-8", "replace"))
