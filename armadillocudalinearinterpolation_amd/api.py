"""Python host side over the C ABI (include/mi355_interp.h).

PyTorch is plumbing only: device memory (torch.Tensor on cuda:N), streams and,
in bench.py, torch.distributed.  Every computation runs in libmi355interp.so;
nothing here has a CPU or eager-torch fallback.

Names follow the reference's domain: grids / nodes / queries for the tables,
realisations / spikes for EventDrivenMap (EventDrivenMap.hpp:11-121).
"""
import ctypes as C
import os
import math
import weakref

import numpy as np

from . import _lib
from ._lib import EdmParams, MiError, check  # noqa: F401

MI_GRID_SANITISE = 0x1
MI_GRID_DEVICE_PTRS = 0x2
MI_GRID2_COMPACT = 0x4
MATH_EXACT = 0
MATH_FAST = 1


def _torch():
    import torch
    return torch


def _ptr(t):
    """Device/host address of a torch tensor or numpy array (must be contiguous)."""
    if isinstance(t, np.ndarray):
        assert t.flags["C_CONTIGUOUS"]
        return C.c_void_p(t.ctypes.data)
    assert t.is_contiguous()
    return C.c_void_p(t.data_ptr())


def _np64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))


class Context:
    """mi_ctx: one device + one stream.  By default it follows torch's current stream."""

    def __init__(self, device=0, stream="torch"):
        self._L = _lib.load()
        h = C.c_void_p()
        check(self._L.mi_ctx_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)
        self._children = weakref.WeakSet()     # grids / problems created on this context: closed before it is
        if stream == "torch":
            self.use_torch_stream()
        elif stream is not None:
            self.set_stream(stream)

    def use_torch_stream(self):
        torch = _torch()
        self.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def set_stream(self, raw_stream):
        check(self._L.mi_ctx_set_stream(self._h, C.c_void_p(raw_stream)), self._h)

    def own_stream(self):
        """Give this context a non-blocking stream of its own (work of several contexts can then overlap)."""
        check(self._L.mi_ctx_own_stream(self._h), self._h)

    def synchronize(self):
        check(self._L.mi_ctx_synchronize(self._h), self._h)

    def set_query_order(self, order):
        """0 auto (device-side probe), 1 queries are unordered, 2 queries are ordered/clustered."""
        check(self._L.mi_ctx_set_query_order(self._h, int(order)), self._h)

    def device_info(self):
        name = C.create_string_buffer(128)
        cus, hbm = C.c_int(0), C.c_size_t(0)
        check(self._L.mi_ctx_device_info(self._h, name, 128, C.byref(cus), C.byref(hbm)), self._h)
        return {"name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": hbm.value}

    def timer(self):
        return Timer(self)

    def close(self):
        """Closes the handles created on this context first (their device state belongs to it), then the context."""
        for child in list(getattr(self, "_children", ()) or ()):
            try:
                child.close()
            except Exception:
                pass
        if getattr(self, "_h", None):
            self._L.mi_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Timer:
    """HIP events recorded on the context's stream."""

    def __init__(self, ctx):
        self._ctx = ctx
        self._L = ctx._L
        h = C.c_void_p()
        check(self._L.mi_timer_create(ctx._h, C.byref(h)), ctx._h)
        self._h = h

    def start(self):
        check(self._L.mi_timer_start(self._h), self._ctx._h)

    def stop(self):
        check(self._L.mi_timer_stop(self._h), self._ctx._h)

    def elapsed_ms(self):
        ms = C.c_float(0)
        check(self._L.mi_timer_elapsed_ms(self._h, C.byref(ms)), self._ctx._h)
        return float(ms.value)

    def close(self):
        if getattr(self, "_h", None):
            self._L.mi_timer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Grid1:
    """HBM-resident 1-D table (mi_grid1)."""

    def __init__(self, ctx, handle):
        self._ctx, self._h, self._L = ctx, handle, ctx._L
        if hasattr(ctx, "_children"):
            ctx._children.add(self)

    @classmethod
    def from_nodes(cls, ctx, x, y, sanitise=True):
        """Explicit abscissae -- the arma::interp1(X, Y, ...) table."""
        x, y = _np64(x), _np64(y)
        if x.size != y.size:
            raise ValueError("X and Y must have the same number of elements")
        h = C.c_void_p()
        check(ctx._L.mi_grid1_create(ctx._h, _ptr(x), _ptr(y), x.size,
                                     MI_GRID_SANITISE if sanitise else 0, C.byref(h)), ctx._h)
        return cls(ctx, h)

    @classmethod
    def from_device_nodes(cls, ctx, x, y):
        """Explicit, strictly increasing abscissae already resident on the device (float64 CUDA tensors)."""
        if x.numel() != y.numel():
            raise ValueError("X and Y must have the same number of elements")
        h = C.c_void_p()
        check(ctx._L.mi_grid1_create(ctx._h, _ptr(x), _ptr(y), x.numel(), MI_GRID_DEVICE_PTRS, C.byref(h)), ctx._h)
        return cls(ctx, h)

    @classmethod
    def uniform(cls, ctx, x0, dx, y):
        """Implicit grid X_i = fma(i, dx, x0)."""
        y = _np64(y)
        h = C.c_void_p()
        check(ctx._L.mi_grid1_create_uniform(ctx._h, float(x0), float(dx), _ptr(y), y.size, 0, C.byref(h)), ctx._h)
        return cls(ctx, h)

    def info(self):
        n, mode, tb = C.c_size_t(0), C.c_int(0), C.c_size_t(0)
        check(self._L.mi_grid1_info(self._h, C.byref(n), C.byref(mode), C.byref(tb)), self._ctx._h)
        return {"n_nodes": n.value, "mode": mode.value, "table_bytes": tb.value}

    def interp(self, xq, out=None, extrap=math.nan):
        """xq: float64 cuda tensor -> float64 cuda tensor (asynchronous on the ctx stream)."""
        torch = _torch()
        if not (xq.is_cuda and xq.dtype == torch.float64 and xq.is_contiguous()):
            raise ValueError("xq must be a contiguous float64 CUDA tensor")
        if out is None:
            out = torch.empty_like(xq)
        elif not (out.is_cuda and out.dtype == torch.float64 and out.is_contiguous() and out.numel() == xq.numel()):
            raise ValueError("out must be a contiguous float64 CUDA tensor of the same size")
        check(self._L.mi_interp1_f64_dev(self._ctx._h, self._h, _ptr(xq), _ptr(out), xq.numel(), float(extrap)),
              self._ctx._h)
        return out

    def interp_host(self, xq, extrap=math.nan, out=None):
        xq = _np64(xq)
        if out is None:
            out = np.empty_like(xq)
        elif out.dtype != np.float64 or out.size != xq.size or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be a contiguous float64 array of the query size")
        check(self._L.mi_interp1_f64_host(self._ctx._h, self._h, _ptr(xq), _ptr(out), xq.size, float(extrap)),
              self._ctx._h)
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._L.mi_grid1_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def interp1(ctx, X, Y, XI, extrap=math.nan):
    """arma::interp1(X, Y, XI, YI, "linear", extrap) on host arrays, computed on the GPU."""
    X, Y, XI = _np64(X), _np64(Y), _np64(XI)
    if X.size != Y.size:
        raise ValueError("X and Y must have the same number of elements")
    YI = np.empty_like(XI)
    check(ctx._L.mi_interp1_f64(ctx._h, _ptr(X), _ptr(Y), X.size, _ptr(XI), _ptr(YI), XI.size, float(extrap)), ctx._h)
    return YI


class Grid2:
    """HBM-resident 2-D table (mi_grid2); z is (ny, nx), stored column-major like arma::mat."""

    def __init__(self, ctx, handle):
        self._ctx, self._h, self._L = ctx, handle, ctx._L
        if hasattr(ctx, "_children"):
            ctx._children.add(self)

    @staticmethod
    def _colmajor(z, ny, nx):
        z = np.asarray(z, dtype=np.float64)
        if z.shape != (ny, nx):
            raise ValueError("Z must have shape (len(y), len(x))")
        return np.ascontiguousarray(z.T).reshape(-1)

    @classmethod
    def from_axes(cls, ctx, x, y, z, compact=False):
        x, y = _np64(x), _np64(y)
        zc = cls._colmajor(z, y.size, x.size)
        h = C.c_void_p()
        check(ctx._L.mi_grid2_create(ctx._h, _ptr(x), x.size, _ptr(y), y.size, _ptr(zc),
                                     MI_GRID2_COMPACT if compact else 0, C.byref(h)), ctx._h)
        return cls(ctx, h)

    @classmethod
    def uniform(cls, ctx, x0, dx, nx, y0, dy, ny, z, compact=False):
        """z: (ny, nx) numpy array, or a column-major float64 CUDA tensor of ny*nx elements."""
        h = C.c_void_p()
        if isinstance(z, np.ndarray) or not hasattr(z, "is_cuda"):
            zc = cls._colmajor(z, ny, nx)
            flags = 0
        else:
            if z.numel() != nx * ny:
                raise ValueError("Z must have nx*ny elements")
            zc, flags = z, MI_GRID_DEVICE_PTRS
        if compact:
            flags |= MI_GRID2_COMPACT
        check(ctx._L.mi_grid2_create_uniform(ctx._h, float(x0), float(dx), nx, float(y0), float(dy), ny,
                                             _ptr(zc), flags, C.byref(h)), ctx._h)
        return cls(ctx, h)

    def info(self):
        tb = C.c_size_t(0)
        check(self._L.mi_grid2_info(self._h, C.byref(tb)))
        return {"table_bytes": tb.value}

    def interp(self, xq, yq, out=None, extrap=math.nan):
        torch = _torch()
        for t in (xq, yq):
            if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
                raise ValueError("queries must be contiguous float64 CUDA tensors")
        if xq.numel() != yq.numel():
            raise ValueError("xq and yq must have the same number of elements")
        if out is None:
            out = torch.empty_like(xq)
        check(self._L.mi_interp2_f64_dev(self._ctx._h, self._h, _ptr(xq), _ptr(yq), _ptr(out), xq.numel(),
                                         float(extrap)), self._ctx._h)
        return out

    def interp_host(self, xq, yq, extrap=math.nan):
        xq, yq = _np64(xq), _np64(yq)
        out = np.empty_like(xq)
        check(self._L.mi_interp2_f64_host(self._ctx._h, self._h, _ptr(xq), _ptr(yq), _ptr(out), xq.size,
                                          float(extrap)), self._ctx._h)
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._L.mi_grid2_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- the reference's own step: Restrict + masked mean (EventDrivenMap.cu:769-824) ----

def restrict(ctx, t0, i0, t1, i1, final_time, half_length, ngrid, out=None):
    """RestrictKernel: CUDA tensors t0,t1 float32 and i0,i1 uint16 (or int16 views), [spike][realisation]."""
    torch = _torch()
    if out is None:
        out = torch.empty_like(t0)
    check(ctx._L.mi_restrict_f32_dev(ctx._h, _ptr(t0), _ptr(i0), _ptr(t1), _ptr(i1), float(final_time),
                                     float(half_length), int(ngrid), _ptr(out), t0.numel()), ctx._h)
    return out


def masked_mean(ctx, x, accept, nspikes, quirk=False, want_sums=False):
    torch = _torch()
    nreal = accept.numel()
    assert x.numel() == nspikes * nreal
    mean = torch.empty(nspikes, dtype=torch.float32, device=x.device)
    count = torch.empty(1, dtype=torch.int32, device=x.device)
    sums = torch.empty(2 * nspikes + 1, dtype=torch.float64, device=x.device) if want_sums else None   # [sums | count | x0]
    check(ctx._L.mi_masked_mean_f32_dev(ctx._h, _ptr(x), _ptr(accept), nreal, nspikes, int(bool(quirk)),
                                        _ptr(mean), _ptr(count), _ptr(sums) if want_sums else None), ctx._h)
    return (mean, count, sums) if want_sums else (mean, count)


def restrict_mean(ctx, t0, i0, t1, i1, accept, final_time, half_length, ngrid, nspikes, quirk=False,
                  want_restricted=False, want_sums=False):
    torch = _torch()
    nreal = accept.numel()
    assert t0.numel() == nspikes * nreal
    mean = torch.empty(nspikes, dtype=torch.float32, device=t0.device)
    count = torch.empty(1, dtype=torch.int32, device=t0.device)
    sums = torch.empty(2 * nspikes + 1, dtype=torch.float64, device=t0.device) if want_sums else None   # [sums | count | x0]
    restricted = torch.empty_like(t0) if want_restricted else None
    check(ctx._L.mi_restrict_mean_f32_dev(ctx._h, _ptr(t0), _ptr(i0), _ptr(t1), _ptr(i1), _ptr(accept),
                                          float(final_time), float(half_length), int(ngrid), nreal, nspikes,
                                          int(bool(quirk)), _ptr(restricted) if want_restricted else None,
                                          _ptr(mean), _ptr(count), _ptr(sums) if want_sums else None), ctx._h)
    return {"mean": mean, "count": count, "sums": sums, "restricted": restricted}


# ---- EventDrivenMap (EventDrivenMap.hpp:11-121) ---------------------------------------

def default_edm_params(**overrides):
    p = EdmParams()
    _lib.load().mi_edm_default_params(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError("mi_edm_params has no field %r" % k)
        setattr(p, k, v)
    return p


class EventDrivenMap:
    """Mirror of the reference class: ComputeF(Z) -> f through lift/evolve/restrict/average."""

    def __init__(self, ctx, parameters, noReal, **overrides):
        self._ctx, self._L = ctx, ctx._L
        par = np.atleast_1d(np.asarray(parameters, dtype=np.float64))
        self.params = default_edm_params(beta_mean=float(np.float32(par[0])), n_real=int(noReal), **overrides)
        h = C.c_void_p()
        check(self._L.mi_edm_create(ctx._h, C.byref(self.params), C.byref(h)), ctx._h)
        self._h = h
        # test / tuning hooks of the Python layer (the library itself reads no environment for this): force an evolve
        # kernel form, switch the exact quotient by launch-uniform divisors off.  Results are bit-identical either way.
        wpr = os.environ.get("MI_EDM_WAVES_PER_REALISATION", "")
        if wpr in ("1", "4") or "MI_EDM_NO_UNIFORM_DIV" in os.environ:
            self.set_kernel_choice(int(wpr) if wpr in ("1", "4") else 0, "MI_EDM_NO_UNIFORM_DIV" not in os.environ)
        if hasattr(ctx, "_children"):
            ctx._children.add(self)

    def _push(self):
        check(self._L.mi_edm_set_params(self._h, C.byref(self.params)), self._ctx._h)

    def set_kernel_choice(self, waves_per_realisation=0, uniform_division=True):
        """mi_edm_set_kernel_choice: 0 / 1 / 4 waves per realisation, exact quotient by uniform divisors on / off."""
        check(self._L.mi_edm_set_kernel_choice(self._h, int(waves_per_realisation), int(bool(uniform_division))), self._ctx._h)

    # setters of EventDrivenMap.hpp:27-51
    def SetTimeHorizon(self, T):
        assert T > 0
        self.params.time_horizon = float(T)
        self._push()

    def SetNoRealisations(self, noReal):
        assert noReal > 0
        self.params.n_real = int(noReal)
        self._push()

    def SetNoThreads(self, noThreads):
        assert 0 < noThreads <= 1024
        self.params.n_grid = int(noThreads)
        self._push()

    def SetParameterStdDev(self, sigma):
        assert sigma >= 0
        self.params.beta_stddev = float(sigma)
        self._push()

    def SetParameters(self, parId, parVal):
        assert parId == 0
        self.params.beta_mean = float(parVal)
        self._push()

    def SetSeed(self, seed):
        self.params.seed = int(seed)
        self._push()

    def PostProcess(self):
        """EventDrivenMap.cu:343-346 draws a new seed; here: advance it deterministically."""
        self.params.seed = (int(self.params.seed) * 6364136223846793005 + 1442695040888963407) & (2**64 - 1)
        self._push()

    def ComputeF(self, Z, want_partial=False):
        Z = _np64(Z)
        S = int(self.params.n_spikes)
        if Z.size != S:
            raise ValueError("Z must have n_spikes=%d elements" % S)
        f = np.empty(S, dtype=np.float64)
        partial = np.empty(2 * S + 1, dtype=np.float64) if want_partial else None   # MI_EDM_PARTIAL_LEN(S)
        check(self._L.mi_edm_compute_f(self._h, _ptr(Z), _ptr(f), _ptr(partial) if want_partial else None),
              self._ctx._h)
        return (f, partial) if want_partial else f

    def begin(self, Z):
        """Enqueue one evaluation and return at once (mi_edm_compute_f_begin); collect it with end()."""
        Z = _np64(Z)
        if Z.size != int(self.params.n_spikes):
            raise ValueError("Z must have n_spikes=%d elements" % int(self.params.n_spikes))
        check(self._L.mi_edm_compute_f_begin(self._h, _ptr(Z)), self._ctx._h)

    def end(self, want_partial=False):
        S = int(self.params.n_spikes)
        f = np.empty(S, dtype=np.float64)
        partial = np.empty(2 * S + 1, dtype=np.float64) if want_partial else None   # MI_EDM_PARTIAL_LEN(S)
        check(self._L.mi_edm_compute_f_end(self._h, _ptr(f), _ptr(partial) if want_partial else None), self._ctx._h)
        return (f, partial) if want_partial else f

    def ComputeFBatch(self, Zs, want_partial=False):
        """Several independent evaluations at once (the columns of a finite-difference Jacobian): every
        evaluation runs on a replica of this problem with a context and stream of its own, all are enqueued before
        any is waited for, so they overlap on the device.  Returns F[b] (and partial[b]) in the order of Zs; each
        equals what ComputeF(Zs[b]) returns."""
        Zs = [np.asarray(z, dtype=np.float64) for z in Zs]
        reps = getattr(self, "_replicas", None)
        if reps is None:
            reps = self._replicas = []
        while len(reps) < len(Zs):
            ctx = Context(self._ctx.device, stream=None)
            ctx.own_stream()
            h = C.c_void_p()
            par = EdmParams()
            C.memmove(C.byref(par), C.byref(self.params), C.sizeof(EdmParams))
            check(self._L.mi_edm_create(ctx._h, C.byref(par), C.byref(h)), ctx._h)
            rep = EventDrivenMap.__new__(EventDrivenMap)
            rep._ctx, rep._L, rep.params, rep._h = ctx, self._L, par, h
            reps.append(rep)
        mine = bytes(self.params)
        for rep in reps[:len(Zs)]:
            if bytes(rep.params) != mine:                      # a setter ran on the parent since the last batch
                C.memmove(C.byref(rep.params), C.byref(self.params), C.sizeof(EdmParams))
                rep._push()
        begun = []
        try:
            for rep, z in zip(reps, Zs):
                rep.begin(z)
                begun.append(rep)
            out = [rep.end(want_partial) for rep in reps[:len(Zs)]]
            begun = []
        finally:
            for rep in begun:                                  # a begin() or end() raised: nothing may stay pending
                try:
                    rep._push()                                # mi_edm_set_params drains the stream and abandons the evaluation
                except Exception:
                    pass
        if want_partial:
            return np.stack([o[0] for o in out]), np.stack([o[1] for o in out])
        return np.stack(out)

    def residual_from_sums(self, Z, sums_and_count):
        Z, sc = _np64(Z), _np64(sums_and_count)
        f = np.empty(int(self.params.n_spikes), dtype=np.float64)
        check(self._L.mi_edm_residual_from_sums(C.byref(self.params), _ptr(Z), _ptr(sc), _ptr(f)), self._ctx._h)
        return f

    def debug_read(self):
        """Stage outputs of the last ComputeF (the reference's Save* taps, EventDrivenMap.cu:406-503)."""
        S, R, N = int(self.params.n_spikes), int(self.params.n_real), int(self.params.n_grid)
        out = {
            "v": np.empty(N, np.float32), "s": np.empty(N, np.float32), "w": np.empty(N, np.float32),
            "t0": np.empty(S * R, np.float32), "i0": np.empty(S * R, np.uint16),
            "t1": np.empty(S * R, np.float32), "i1": np.empty(S * R, np.uint16),
            "accept": np.empty(R, np.uint32), "restricted": np.empty(S * R, np.float32),
            "seed_ind": np.empty(S, np.uint16),
        }
        order = ["v", "s", "w", "t0", "i0", "t1", "i1", "accept", "restricted", "seed_ind"]
        check(self._L.mi_edm_debug_read(self._h, *[_ptr(out[k]) for k in order]), self._ctx._h)
        return out

    def debug_counters(self):
        """Decision-coverage taps of the last ComputeF (mi_edm_debug_counters: one more, tapped, evolve)."""
        c = (C.c_uint64 * 8)()
        check(self._L.mi_edm_debug_counters(self._h, C.byref(c)), self._ctx._h)
        names = ("events", "max_events_one", "max_newton_iter", "newton_cap_hits", "event_cap_hits", "accepted",
                 "no_firing_events", "argmin_ties")
        return {n: int(c[i]) for i, n in enumerate(names)}

    def last_timings(self):
        ms = (C.c_float * 4)()
        check(self._L.mi_edm_last_timings(self._h, C.byref(ms)), self._ctx._h)
        return {"lift_ms": ms[0], "evolve_ms": ms[1], "restrict_mean_ms": ms[2], "total_ms": ms[3]}

    def close(self):
        for rep in getattr(self, "_replicas", None) or []:
            rep.close()
        self._replicas = []
        if getattr(self, "_h", None):
            self._L.mi_edm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- several GPUs of one node, one host process (mi_group_*) ---------------------------------------------------

def shard_bounds(n, rank, world):
    """mi_shard_bounds: contiguous balanced split (host arithmetic of the C ABI; equals sharding.shard_bounds)."""
    lo, hi = C.c_size_t(0), C.c_size_t(0)
    _lib.load().mi_shard_bounds(int(n), int(rank), int(world), C.byref(lo), C.byref(hi))
    return lo.value, hi.value


class _GroupCtx:
    """borrowed view of a group member's context (owned by the group)"""

    def __init__(self, L, h, device):
        self._L, self._h, self.device = L, h, device

    def timer(self):
        """HIP events on this member's stream (the stream its shard's kernels are launched on)"""
        return Timer(self)

    def device_info(self):
        return Context.device_info(self)


class Group:
    """One process driving several GPUs: devices = [0, 1, ..] (a repeated ordinal rehearses the sharding on one GPU)."""
    REDUCE_HOST, REDUCE_RCCL = 0, 1

    def __init__(self, devices):
        self._L = _lib.load()
        devices = list(range(devices)) if isinstance(devices, int) else [int(d) for d in devices]
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        check(self._L.mi_group_create(len(devices), arr, C.byref(h)))
        self._h, self.devices = h, devices

    def __len__(self):
        return int(self._L.mi_group_size(self._h))

    def ctx(self, rank):
        return _GroupCtx(self._L, C.c_void_p(self._L.mi_group_ctx(self._h, int(rank))), self.devices[rank])

    def set_reduce(self, mode):
        check(self._L.mi_group_set_reduce(self._h, int(mode)))

    def set_gather_chunks(self, chunks):
        """mi_group_set_gather_chunks: >= 2 hides the gather of interp_dev(gather=True) behind the kernels, chunk by chunk"""
        check(self._L.mi_group_set_gather_chunks(self._h, int(chunks)))

    def rccl_ranks(self):
        """ranks of the group's RCCL communicator as RCCL reports it (ncclCommCount); 0: no communicator possible"""
        return int(self._L.mi_group_rccl_ranks(self._h))

    def synchronize(self):
        check(self._L.mi_group_synchronize(self._h))

    def grid1(self, X, Y, sanitise=True):
        X, Y = _np64(X), _np64(Y)
        t = C.c_void_p()
        check(self._L.mi_group_grid1_create(self._h, _ptr(X), _ptr(Y), X.size, 1 if sanitise else 0, C.byref(t)))
        return GroupGrid1(self, t)

    def grid2(self, x, y, z, compact=False):
        """2-D table replicated on every device; z is (len(y), len(x))"""
        x, y = _np64(x), _np64(y)
        zc = Grid2._colmajor(z, y.size, x.size)
        t = C.c_void_p()
        check(self._L.mi_group_grid2_create(self._h, _ptr(x), x.size, _ptr(y), y.size, _ptr(zc),
                                            MI_GRID2_COMPACT if compact else 0, C.byref(t)))
        return GroupGrid2(self, t)

    def edm(self, parameters, noReal, **overrides):
        return GroupEventDrivenMap(self, parameters, noReal, **overrides)

    def close(self):
        if getattr(self, "_h", None):
            self._L.mi_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GroupGrid1:
    def __init__(self, group, h):
        self._g, self._L, self._h = group, group._L, h

    def interp_host(self, xq, extrap=math.nan):
        """host arrays in, host array out; the queries are sharded over the group's devices"""
        xq = _np64(xq)
        out = np.empty_like(xq)
        check(self._L.mi_group_interp1_f64_host(self._g._h, self._h, _ptr(xq), _ptr(out), xq.size, float(extrap)))
        return out

    def interp_dev(self, xq_shards, extrap=math.nan, gather=False, out=None, gathered=None, sync=True):
        """device-resident shards (one float64 tensor per group member, equal sizes); returns the per-shard results and,
        with gather=True, one buffer per member holding every shard (RCCL all-gather, or device copies in a rehearsal group).
        out / gathered: caller-owned result tensors (no allocation in the call); sync=False: return with the shards'
        kernels enqueued on the members' streams (Group.synchronize waits) -- the shape bench.py --backend group times"""
        torch = _torch()
        P, n = len(self._g), int(xq_shards[0].numel())
        assert len(xq_shards) == P and all(int(t.numel()) == n for t in xq_shards)
        outs = out if out is not None else [torch.empty_like(t) for t in xq_shards]
        assert len(outs) == P and all(int(t.numel()) == n and t.dtype == torch.float64 and t.is_contiguous() for t in outs)
        full = gathered
        if gather and full is None:
            full = [torch.empty(P * n, dtype=torch.float64, device=t.device) for t in xq_shards]
        if gather:
            assert len(full) == P and all(int(t.numel()) == P * n for t in full)
        arr = lambda ts: (C.c_void_p * P)(*[t.data_ptr() for t in ts])  # noqa: E731
        check(self._L.mi_group_interp1_f64_dev(self._g._h, self._h, arr(xq_shards), arr(outs), n, float(extrap),
                                               arr(full) if gather else None))
        if sync:
            self._g.synchronize()
        return (outs, full) if gather else outs

    def close(self):
        if getattr(self, "_h", None):
            self._L.mi_group_grid1_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GroupGrid2:
    def __init__(self, group, h):
        self._g, self._L, self._h = group, group._L, h

    def interp_host(self, xq, yq, extrap=math.nan):
        """host arrays in, host array out; the scattered queries are sharded over the group's devices"""
        xq, yq = _np64(xq), _np64(yq)
        if xq.size != yq.size:
            raise ValueError("xq and yq must have the same number of elements")
        out = np.empty_like(xq)
        check(self._L.mi_group_interp2_f64_host(self._g._h, self._h, _ptr(xq), _ptr(yq), _ptr(out), xq.size, float(extrap)))
        return out

    def interp_dev(self, xq_shards, yq_shards, extrap=math.nan, gather=False):
        """device-resident shards (one pair of float64 tensors per group member, equal sizes)"""
        torch = _torch()
        P, n = len(self._g), int(xq_shards[0].numel())
        assert len(xq_shards) == P and len(yq_shards) == P
        assert all(int(t.numel()) == n for t in xq_shards) and all(int(t.numel()) == n for t in yq_shards)
        outs = [torch.empty_like(t) for t in xq_shards]
        full = [torch.empty(P * n, dtype=torch.float64, device=t.device) for t in xq_shards] if gather else None
        arr = lambda ts: (C.c_void_p * P)(*[t.data_ptr() for t in ts])  # noqa: E731
        check(self._L.mi_group_interp2_f64_dev(self._g._h, self._h, arr(xq_shards), arr(yq_shards), arr(outs), n,
                                               float(extrap), arr(full) if gather else None))
        self._g.synchronize()
        return (outs, full) if gather else outs

    def close(self):
        if getattr(self, "_h", None):
            self._L.mi_group_grid2_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GroupEventDrivenMap:
    """EventDrivenMap with the realisations sharded over a Group; noReal is the total."""

    def __init__(self, group, parameters, noReal, **overrides):
        self._g, self._L = group, group._L
        par = np.atleast_1d(np.asarray(parameters, dtype=np.float64))
        self.params = default_edm_params(beta_mean=float(np.float32(par[0])), n_real=int(noReal), **overrides)
        h = C.c_void_p()
        check(self._L.mi_group_edm_create(group._h, C.byref(self.params), C.byref(h)))
        self._h = h

    def _push(self):
        check(self._L.mi_group_edm_set_params(self._h, C.byref(self.params)))

    def ComputeF(self, Z, want_partial=False):
        Z = _np64(Z)
        S = int(self.params.n_spikes)
        f = np.empty(S, dtype=np.float64)
        partial = np.empty(2 * S + 1, dtype=np.float64)
        check(self._L.mi_group_edm_compute_f(self._h, _ptr(Z), _ptr(f), _ptr(partial)))
        return (f, partial) if want_partial else f

    def shard_bounds(self, rank):
        lo, hi = C.c_size_t(0), C.c_size_t(0)
        check(self._L.mi_group_edm_shard_bounds(self._h, int(rank), C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def close(self):
        if getattr(self, "_h", None):
            self._L.mi_group_edm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
