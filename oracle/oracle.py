"""ctypes/numpy front end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None

_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u16p = np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile oracle/*.c -> liboracle.so with the committed Makefile."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "clean", "all"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    sz, dbl, i32 = C.c_size_t, C.c_double, C.c_int
    L.orc_interp1_scan_sorted.argtypes = [_f64p, _f64p, sz, _f64p, sz, dbl, _f64p]
    L.orc_interp1_scan_sorted.restype = None
    L.orc_interp1_bracket.argtypes = [_f64p, _f64p, sz, _f64p, sz, dbl, _f64p, i32]
    L.orc_interp1_bracket.restype = None
    L.orc_interp1_arma.argtypes = [_f64p, _f64p, sz, _f64p, sz, dbl, _f64p]
    L.orc_interp1_arma.restype = i32
    L.orc_interp1_uniform.argtypes = [dbl, dbl, _f64p, sz, _f64p, sz, dbl, _f64p, i32]
    L.orc_interp1_uniform.restype = None
    L.orc_interp2_bilinear.argtypes = [_f64p, sz, _f64p, sz, _f64p, _f64p, _f64p, sz, dbl, _f64p, i32]
    L.orc_interp2_bilinear.restype = None
    L.orc_interp2_bilinear_uniform.argtypes = [dbl, dbl, sz, dbl, dbl, sz, _f64p, _f64p, _f64p, sz,
                                               dbl, _f64p, i32]
    L.orc_interp2_bilinear_uniform.restype = None
    L.orc_restrict_f32.argtypes = [_f32p, _u16p, _f32p, _u16p, C.c_float, C.c_float, C.c_uint32, _f32p, sz]
    L.orc_restrict_f32.restype = None
    L.orc_masked_mean_f32.argtypes = [_f32p, _u32p, sz, sz, i32, _f32p, C.POINTER(C.c_uint32)]
    L.orc_masked_mean_f32.restype = None
    L.orc_splitmix_uniform.argtypes = [C.c_uint64, _f64p, sz]
    L.orc_splitmix_uniform.restype = None
    L.orc_max_threads.argtypes = []
    L.orc_max_threads.restype = i32
    _lib = L
    return L


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def max_threads():
    return int(lib().orc_max_threads())


def splitmix_uniform(seed, n):
    out = np.empty(n, dtype=np.float64)
    lib().orc_splitmix_uniform(C.c_uint64(seed), out, n)
    return out


def interp1_scan_sorted(xg, yg, xi, extrap=np.nan):
    xg, yg, xi = _c64(xg), _c64(yg), _c64(xi)
    out = np.empty_like(xi)
    lib().orc_interp1_scan_sorted(xg, yg, xg.size, xi, xi.size, extrap, out)
    return out


def interp1_bracket(xg, yg, xi, extrap=np.nan, nthreads=1):
    xg, yg, xi = _c64(xg), _c64(yg), _c64(xi)
    out = np.empty_like(xi)
    lib().orc_interp1_bracket(xg, yg, xg.size, xi, xi.size, extrap, out, nthreads)
    return out


def interp1_arma(x, y, xi, extrap=np.nan):
    x, y, xi = _c64(x), _c64(y), _c64(xi)
    out = np.empty_like(xi)
    rc = lib().orc_interp1_arma(x, y, x.size, xi, xi.size, extrap, out)
    if rc != 0:
        raise ValueError("orc_interp1_arma failed with code %d" % rc)
    return out


def interp1_uniform(x0, dx, yg, xi, extrap=np.nan, nthreads=1):
    yg, xi = _c64(yg), _c64(xi)
    out = np.empty_like(xi)
    lib().orc_interp1_uniform(x0, dx, yg, yg.size, xi, xi.size, extrap, out, nthreads)
    return out


def interp2_bilinear(xg, yg, z, xq, yq, extrap=np.nan, nthreads=1):
    """z: (ny, nx) array; passed to C in column-major (arma::mat) layout."""
    xg, yg, xq, yq = _c64(xg), _c64(yg), _c64(xq), _c64(yq)
    z = np.asarray(z, dtype=np.float64)
    assert z.shape == (yg.size, xg.size)
    zcm = np.ascontiguousarray(z.T).reshape(-1)      # column-major flat
    out = np.empty_like(xq)
    lib().orc_interp2_bilinear(xg, xg.size, yg, yg.size, zcm, xq, yq, xq.size, extrap, out, nthreads)
    return out


def interp2_bilinear_uniform(x0, dx, nx, y0, dy, ny, z, xq, yq, extrap=np.nan, nthreads=1):
    xq, yq = _c64(xq), _c64(yq)
    z = np.asarray(z, dtype=np.float64)
    assert z.shape == (ny, nx)
    zcm = np.ascontiguousarray(z.T).reshape(-1)
    out = np.empty_like(xq)
    lib().orc_interp2_bilinear_uniform(x0, dx, nx, y0, dy, ny, zcm, xq, yq, xq.size, extrap, out, nthreads)
    return out


def restrict_f32(t0, i0, t1, i1, T, L, ngrid):
    t0 = np.ascontiguousarray(t0, dtype=np.float32)
    t1 = np.ascontiguousarray(t1, dtype=np.float32)
    i0 = np.ascontiguousarray(i0, dtype=np.uint16)
    i1 = np.ascontiguousarray(i1, dtype=np.uint16)
    out = np.empty_like(t0)
    lib().orc_restrict_f32(t0, i0, t1, i1, T, L, ngrid, out, t0.size)
    return out


def masked_mean_f32(x, accept, nspikes, quirk=False):
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
    accept = np.ascontiguousarray(accept, dtype=np.uint32)
    nreal = accept.size
    assert x.size == nspikes * nreal
    mean = np.empty(nspikes, dtype=np.float32)
    cnt = C.c_uint32(0)
    lib().orc_masked_mean_f32(x, accept, nreal, nspikes, int(bool(quirk)), mean, C.byref(cnt))
    return mean, int(cnt.value)


# ---- EventDrivenMap pipeline (oracle/edm_oracle.c) --------------------------------------------

class EdmParams(C.Structure):
    """orc_edm_params: same field order as mi_edm_params."""
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
    ]


class EdmCounters(C.Structure):
    """orc_edm_counters: decision-coverage counts of orc_edm_compute_f_counted (see oracle/edm_oracle.c)."""
    _fields_ = [(n, C.c_uint64) for n in (
        "realisations", "accepted", "events", "argmin_ties", "no_firing_events", "nan_times", "argmin_tree_mismatch", "unwritten_last_slots",
        "unwritten_last_slots_all", "seed_scans_empty", "newton_cap_hits", "event_cap_hits", "time_cap_exits",
        "newton_solves", "newton_iters", "wave64_rounds")] + [("max_newton_iter", C.c_uint32), ("max_events_one", C.c_uint32)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


_variants = {}


def _variant_lib(name):
    """Test-only SENSITIVITY builds of the oracle sources (oracle/Makefile): 'contract' (-ffp-contract=fast -mfma) and
    'libm' (libm expf/logf/powf).  Never the parity oracle."""
    if name not in _variants:
        path = os.path.join(_HERE, "liboracle_%s.so" % name)
        srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")] + [os.path.join(_HERE, "Makefile")]
        if not os.path.exists(path) or any(os.path.getmtime(path) < os.path.getmtime(s_) for s_ in srcs):
            subprocess.check_call(["make", "-s", "-B", "-C", _HERE, os.path.basename(path)])
        _variants[name] = C.CDLL(path)
    return _variants[name]


def _edm_lib(variant=None):
    L = lib() if variant is None else _variant_lib(variant)
    if not getattr(L, "_edm_ready", False):
        pp = C.POINTER(EdmParams)
        vp = C.c_void_p
        L.orc_edm_default_params.argtypes = [pp]
        L.orc_edm_default_params.restype = None
        L.orc_edm_compute_f.argtypes = [pp, _f64p, _f64p, _u16p] + [vp] * 10 + [C.c_int]
        L.orc_edm_compute_f.restype = C.c_int
        L.orc_edm_compute_f_counted.argtypes = [pp, _f64p, _f64p, _u16p] + [vp] * 10 + [C.c_int, C.POINTER(EdmCounters)]
        L.orc_edm_compute_f_counted.restype = C.c_int
        L.orc_edm_math_probe.argtypes = [C.c_int, _f32p, _f32p, _f32p, C.c_size_t]
        L.orc_edm_math_probe.restype = None
        L.orc_edm_coupling.argtypes = [pp, _f32p]
        L.orc_edm_coupling.restype = None
        L.orc_edm_lift.argtypes = [pp, _f32p, _f32p, _f32p]
        L.orc_edm_lift.restype = None
        L.orc_edm_seed_indices.argtypes = [pp, _f64p, _u16p]
        L.orc_edm_seed_indices.restype = None
        L.orc_edm_event_time.argtypes = [pp, C.c_float, C.c_float, C.c_float]
        L.orc_edm_event_time.restype = C.c_float
        L.orc_edm_beta.argtypes = [pp, C.c_uint32, C.c_uint32]
        L.orc_edm_beta.restype = C.c_float
        L.orc_edm_residual_from_sums.argtypes = [pp, _f64p, _f64p, _f64p]
        L.orc_edm_residual_from_sums.restype = None
        L.orc_edm_argmin_reference_tree.argtypes = [_f32p, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
        L.orc_edm_argmin_reference_tree.restype = None
        L.orc_edm_argmin.argtypes = [_f32p, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
        L.orc_edm_argmin.restype = None
        L._edm_ready = True
    return L


def edm_default_params(**overrides):
    p = EdmParams()
    _edm_lib().orc_edm_default_params(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def edm_math_probe(op, a, b=None):
    """op: 0 exp, 1 log, 2 pow(a,b), 3 erfinv."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(a if b is None else b, dtype=np.float32)
    out = np.empty_like(a)
    _edm_lib().orc_edm_math_probe(int(op), a, b, out, a.size)
    return out


def edm_coupling(p):
    w = np.empty(p.n_grid, dtype=np.float32)
    _edm_lib().orc_edm_coupling(C.byref(p), w)
    return w


def edm_lift(p, U):
    U = np.ascontiguousarray(U, dtype=np.float32)
    v = np.empty(p.n_grid, dtype=np.float32)
    s = np.empty(p.n_grid, dtype=np.float32)
    _edm_lib().orc_edm_lift(C.byref(p), U, v, s)
    return v, s


def edm_seed_indices(p, Z, prev=None):
    ind = np.zeros(p.n_spikes, dtype=np.uint16) if prev is None else np.array(prev, dtype=np.uint16)
    _edm_lib().orc_edm_seed_indices(C.byref(p), _c64(Z), ind)
    return ind


def edm_beta(p, r, i):
    return float(_edm_lib().orc_edm_beta(C.byref(p), int(r), int(i)))


def edm_argmin_reference_tree(times):
    """The reference's blockReduceMin (EventDrivenMap.cu:843-881) emulated literally for 32-wide warps: (time, index)."""
    times = np.ascontiguousarray(times, dtype=np.float32)
    t, i = C.c_float(0), C.c_uint32(0)
    _edm_lib().orc_edm_argmin_reference_tree(times, times.size, C.byref(t), C.byref(i))
    return float(t.value), int(i.value)


def edm_argmin(times):
    """The arg-min rule the oracle's event loop uses (orc_edm_argmin: closed form of the reference's reduction): (time, index)."""
    times = np.ascontiguousarray(times, dtype=np.float32)
    t, i = C.c_float(0), C.c_uint32(0)
    _edm_lib().orc_edm_argmin(times, times.size, C.byref(t), C.byref(i))
    return float(t.value), int(i.value)


def edm_residual_from_sums(p, Z, sums):
    """f from the element-wise sum of the shards' partial blocks (2S+1 doubles)."""
    f = np.empty(p.n_spikes, dtype=np.float64)
    _edm_lib().orc_edm_residual_from_sums(C.byref(p), _c64(Z), _c64(sums), f)
    return f


def edm_compute_f(p, Z, seed_ind=None, nthreads=1, debug=True, counters=None, variant=None):
    """Whole residual.  Returns (f, dbg) with dbg holding every stage output.
    counters: an EdmCounters that this evaluation's decision-coverage counts are ADDED to.
    variant: None (the oracle) or 'contract' / 'libm' (sensitivity builds, tests only)."""
    Z = _c64(Z)
    S, R, N = p.n_spikes, p.n_real, p.n_grid
    f = np.empty(S, dtype=np.float64)
    ind = np.zeros(S, dtype=np.uint16) if seed_ind is None else np.array(seed_ind, dtype=np.uint16)
    dbg = {
        "v": np.empty(N, np.float32), "s": np.empty(N, np.float32), "w": np.empty(N, np.float32),
        "t0": np.empty(S * R, np.float32), "i0": np.empty(S * R, np.uint16),
        "t1": np.empty(S * R, np.float32), "i1": np.empty(S * R, np.uint16),
        "accept": np.empty(R, np.uint32), "restricted": np.empty(S * R, np.float32),
        "sums": np.empty(2 * S + 1, np.float64),        # partial block [sums | count | x0]
    }
    order = ["v", "s", "w", "t0", "i0", "t1", "i1", "accept", "restricted", "sums"]
    ptrs = [C.c_void_p(dbg[k].ctypes.data) if debug else None for k in order]
    L = _edm_lib(variant)
    if counters is not None:
        rc = L.orc_edm_compute_f_counted(C.byref(p), Z, f, ind, *ptrs, int(nthreads), C.byref(counters))
    else:
        rc = L.orc_edm_compute_f(C.byref(p), Z, f, ind, *ptrs, int(nthreads))
    if rc != 0:
        raise ValueError("orc_edm_compute_f failed (%d)" % rc)
    dbg["seed_ind"] = ind
    return f, dbg
