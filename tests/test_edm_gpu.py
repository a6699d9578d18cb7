"""GPU parity suite for the EventDrivenMap residual (lift -> evolve -> restrict -> average) through the C ABI.
EXACT math mode must reproduce oracle/edm_oracle.c BIT FOR BIT at every stage tap (the reference's Save*
debug dumps, EventDrivenMap.cu:406-503); FAST mode (hardware exp/log) within stated tolerances."""
import ctypes as C

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

Z_DRIVER = [0.3310, 0.6914, 1.3557]


@pytest.fixture(autouse=True, params=["auto", "wave_per_realisation", "workgroup_per_realisation"])
def evolve_form(request, monkeypatch):
    """The evolve kernel has two forms (one wave, or a workgroup of four waves, per realisation) chosen by the
    realisation count; every test of this file runs with the automatic choice and with each form forced
    (MI_EDM_WAVES_PER_REALISATION, read when an EventDrivenMap handle is created), so both are held to the same bit-exact parity."""
    if request.param == "auto":
        monkeypatch.delenv("MI_EDM_WAVES_PER_REALISATION", raising=False)
    else:
        monkeypatch.setenv("MI_EDM_WAVES_PER_REALISATION", "1" if request.param == "wave_per_realisation" else "4")
    return request.param


def _probe(ctx, mode, op, a, b=None):
    import torch
    from armadillocudalinearinterpolation_amd import _lib
    L = _lib.load()
    ta = torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    tb = torch.from_numpy(np.ascontiguousarray(a if b is None else b, np.float32)).cuda()
    out = torch.empty_like(ta)
    _lib.check(L.mi_edm_math_probe(ctx._h, mode, op, C.c_void_p(ta.data_ptr()), C.c_void_p(tb.data_ptr()),
                                   C.c_void_p(out.data_ptr()), ta.numel()), ctx._h)
    return out.cpu().numpy()


def test_uniform_divisor_quotient_is_the_ieee_quotient(mi_ctx, evolve_form):
    """div_by (csrc/mi_edm_math.hpp): the five-operation quotient by a wave-uniform divisor against IEEE division, bit for
    bit, over random bit patterns (every exponent, both signs, zeros, subnormals, infinities, NaNs) and numerators spread
    over the exponents the shortcut accepts and a little beyond, for the divisors of the reference's model (1 - beta,
    beta - 1, vth - I) and awkward ones (significand all ones, powers of two, the edges of the accepted range and beyond)."""
    if evolve_form != "auto":
        pytest.skip("no evolve kernel involved")
    rng = np.random.default_rng(7)
    n = 1 << 24
    bits = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    mant = rng.integers(0, 1 << 23, n, dtype=np.uint32)
    expo = rng.integers(127 - 104, 127 + 104, n, dtype=np.uint32)
    spread = ((rng.integers(0, 2, n, dtype=np.uint32) << 31) | (expo << 23) | mant).view(np.float32)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1.1754944e-38, 3.4028235e38, 2.0 ** -100, 2.0 ** 100,
                        np.nextafter(np.float32(2.0 ** -100), np.float32(0)), np.nextafter(np.float32(2.0 ** 100), np.float32(np.inf))], np.float32)
    a = np.concatenate([bits, spread, special])
    beta = np.float32(13.0589)
    divisors = [np.float32(1.0) - beta, beta - np.float32(1.0), np.float32(1.0) - np.float32(0.9), np.float32(0.1), np.float32(3.0),
                np.float32(-7.0), np.float32(1.9999999), np.float32(1.0000001), np.float32(0.5), np.float32(2.0 ** -20), np.float32(2.0 ** 20),
                np.float32(2.0 ** -21), np.float32(2.0 ** 21), np.float32(1e-30), np.float32(np.inf), np.float32(0.0)]
    for c in divisors:
        cb = np.concatenate([np.full(4, c, np.float32), a[4:]])
        got = _probe(mi_ctx, 0, 4, a, cb)
        want = _probe(mi_ctx, 0, 5, a, cb)
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (c, a[~same][:5], got[~same][:5], want[~same][:5])
        with np.errstate(all="ignore"):
            host = (a / c).astype(np.float32)
        ok = (want.view(np.uint32) == host.view(np.uint32)) | (np.isnan(want) & np.isnan(host))
        assert ok.all(), ("device IEEE division differs from numpy", c)


def test_one_correction_step_of_the_uniform_divisor_quotient(mi_ctx, evolve_form):
    """With rc = RN(1 / c), q0 = RN(a rc) and ONE step q1 = RN(q0 + RN(a - c q0) rc), q1 is the IEEE quotient for every
    numerator: all 2^23 significands at four exponents (the quotient scales exactly with the numerator's exponent inside
    the guarded range [2^-100, 2^101)), both signs, for the reference's divisors, awkward ones (significand all ones, one
    ulp above a power of two) and 40 random ones.  The launch does not rely on this observation -- it proves it for its
    own divisor on the device before it picks a one-step kernel (divisor_check_kernel, csrc/mi_edm.hip) -- but the
    product only gets faster where the check passes, and this is the evidence that it does."""
    if evolve_form != "auto":
        pytest.skip("no evolve kernel involved")
    rng = np.random.default_rng(23)
    mant = np.arange(1 << 23, dtype=np.uint32)
    a = np.concatenate([((np.uint32(s) << 31) | (np.uint32(e) << 23) | mant).view(np.float32)
                        for s, e in ((0, 127), (1, 127), (0, 27), (1, 227), (0, 200))])
    beta = np.float32(13.0589)
    divisors = [np.float32(1.0) - beta, beta - np.float32(1.0), np.float32(1.0) - np.float32(0.9), np.float32(3.0), np.float32(-7.0),
                np.float32(1.9999999), np.float32(0.99999994), np.float32(1.0000001), np.float32(2.0 ** -20), np.float32(2.0 ** 20)]
    divisors += list((rng.uniform(1.0, 2.0, 40) * 2.0 ** rng.integers(-6, 7, 40) * rng.choice([-1.0, 1.0], 40)).astype(np.float32))
    for c in divisors:
        cb = np.concatenate([np.full(4, c, np.float32), a[4:]])
        got = _probe(mi_ctx, 0, 13, a, cb)
        want = _probe(mi_ctx, 0, 5, a, cb)
        same = got.view(np.uint32) == want.view(np.uint32)
        assert same.all(), (c, a[~same][:5], got[~same][:5], want[~same][:5])


def test_firing_test_pre_decision_never_changes_the_decision(mi_ctx, evolve_form):
    """will_fire (csrc/mi_edm_math.hpp) settles cases that are clear of the threshold with the hardware transcendentals and
    leaves the rest to the exact path: both forms must give the same decision everywhere -- checked on synaptic values over
    20 decades with the membrane value placed AT the exact threshold, within a few ulp of it, within the margin, just
    outside it and far away, for three values of beta."""
    if evolve_form != "auto":
        pytest.skip("no evolve kernel involved")
    rng = np.random.default_rng(11)
    n = 1 << 21
    gap = np.float32(1.0) - np.float32(0.9)
    s0 = (10.0 ** rng.uniform(-12, 8, n)).astype(np.float32)
    s0[: n // 8] = (gap * (10.0 ** rng.uniform(-2, 2, n // 8))).astype(np.float32)        # ratio around 1
    for op_exact, beta in ((6, 13.0589), (8, 1.5), (10, 0.7)):
        # locate the exact threshold in v0 for every s0 by bisection on the exact path's decision (monotone in v0)
        lo = np.full(n, -1e6, np.float32)
        hi = np.full(n, 1e6, np.float32)
        for _ in range(60):
            mid = ((lo.astype(np.float64) + hi.astype(np.float64)) * 0.5).astype(np.float32)
            fire = _probe(mi_ctx, 0, op_exact, mid, s0) > 0.5
            hi = np.where(fire, mid, hi)
            lo = np.where(fire, lo, mid)
        thr = hi                                                                            # smallest v0 found that fires
        scale = np.maximum(np.abs(thr), np.float32(1e-3))
        cands = [thr, lo, np.nextafter(thr, np.float32(np.inf)), np.nextafter(lo, np.float32(-np.inf))]
        for rel in (1e-7, 1e-6, 1e-5, 5e-5, 9e-5, 1.1e-4, 2e-4, 1e-3, 1e-2, 1.0):
            cands += [(thr + np.float32(rel) * scale).astype(np.float32), (thr - np.float32(rel) * scale).astype(np.float32)]
        cands += [rng.standard_normal(n).astype(np.float32), np.full(n, np.nan, np.float32), np.full(n, np.inf, np.float32)]
        for v0 in cands:
            a = _probe(mi_ctx, 0, op_exact, v0, s0)
            b = _probe(mi_ctx, 0, op_exact + 1, v0, s0)
            assert np.array_equal(a, b), (beta, v0[a != b][:5], s0[a != b][:5])
    # special synaptic values
    sv = np.array([0.0, -0.0, np.nan, np.inf, -np.inf, -1.0, 1e-45, 3e38, 1e-30, 1e30], np.float32)
    vv = np.array([0.95, 0.5, 0.2, 1.0, 1.5, 0.99, 0.9, 0.91, 1.01, -3.0], np.float32)
    for op_exact in (6, 8, 10):
        for shift in range(len(vv)):
            v0 = np.roll(vv, shift)
            assert np.array_equal(_probe(mi_ctx, 0, op_exact, v0, sv), _probe(mi_ctx, 0, op_exact + 1, v0, sv))


def test_lane_pair_exchange_of_the_paired_solves(mi_ctx, evolve_form):
    """edm::other_half (v_permlane32_swap_b32): every lane must receive the value of lane ^ 32 -- the paired firing-time
    solves (edm::newton_time_paired) exchange their exponentials and f / f' through it."""
    if evolve_form != "auto":
        pytest.skip("pure math probe: no evolve kernel involved")
    a = np.arange(1024, dtype=np.float32) * 0.5 - 3.0
    out = _probe(mi_ctx, 0, 12, a)
    assert np.array_equal(out, a[np.arange(1024) ^ 32])


def test_uniform_division_path_changes_nothing(mi_ctx, monkeypatch):
    """N = 512 (the reference's Driver.cu grid) takes the exact quotient by uniform divisors in the wave-per-realisation
    kernel; MI_EDM_NO_UNIFORM_DIV switches it off.  Every event array must be the same either way."""
    monkeypatch.setenv("MI_EDM_WAVES_PER_REALISATION", "1")
    taps = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("MI_EDM_NO_UNIFORM_DIV", "1")
        else:
            monkeypatch.delenv("MI_EDM_NO_UNIFORM_DIV", raising=False)
        edm, f, partial, dbg = _run(mi_ctx, n_grid=512, n_real=3500)       # enough realisations for the path to be chosen
        taps.append((f, partial, dbg))
    for k in ("t0", "i0", "t1", "i1", "accept", "restricted"):
        assert np.array_equal(taps[0][2][k], taps[1][2][k], equal_nan=True), k
    assert np.array_equal(taps[0][0], taps[1][0]) and np.array_equal(taps[0][1], taps[1][1])


def test_device_math_bit_identical_to_oracle(mi_ctx, evolve_form):
    if evolve_form != "auto":
        pytest.skip("no evolve kernel involved")
    rng = np.random.default_rng(0)
    x = np.concatenate([np.linspace(-110, 95, 300001), rng.standard_normal(200000) * 20,
                        [0.0, -0.0, np.inf, -np.inf, np.nan, 88.72284, -103.9, -87.4, 1e-30, -1e-30]]).astype(np.float32)
    assert np.array_equal(_probe(mi_ctx, 0, 0, x), oracle.edm_math_probe(0, x), equal_nan=True)      # exp, incl. subnormal results
    # EVERY float from 88.70 to 89.5 and from -104.5 to -103.9, and the whole subnormal-result range in steps of 64 ulp: the
    # device routine has no select for the oracle's overflow / underflow cases (ldexpf's own rounding must deliver them)
    def every_float(lo, hi, step=1):
        a, b = np.float32(lo).view(np.uint32), np.float32(hi).view(np.uint32)
        return np.arange(min(a, b), max(a, b) + 1, step, dtype=np.uint32).view(np.float32)
    xe = np.concatenate([every_float(88.70, 89.5), every_float(-103.9, -104.5), every_float(-87.0, -103.9, 64),
                         np.array([1e30, 3.4e38, -1e30, -3.4e38, 128.0, -150.0, -151.0], np.float32)])
    assert np.array_equal(_probe(mi_ctx, 0, 0, xe), oracle.edm_math_probe(0, xe), equal_nan=True)
    xl = np.concatenate([np.exp(rng.uniform(-100, 88, 400000)), [0.0, -1.0, np.inf, np.nan, 1e-42, 1.0]]).astype(np.float32)
    assert np.array_equal(_probe(mi_ctx, 0, 1, xl), oracle.edm_math_probe(1, xl), equal_nan=True)    # log, incl. subnormal inputs
    a = np.concatenate([rng.uniform(-5, 400, 200000), [0.0, -0.0]]).astype(np.float32)
    b = rng.uniform(0.02, 0.2, a.size).astype(np.float32)
    assert np.array_equal(_probe(mi_ctx, 0, 2, a, b), oracle.edm_math_probe(2, a, b), equal_nan=True)
    u = rng.uniform(-1, 1, 200000).astype(np.float32)
    assert np.array_equal(_probe(mi_ctx, 0, 3, u), oracle.edm_math_probe(3, u), equal_nan=True)
    # FAST mode: hardware transcendentals, a few ulp
    xs = np.linspace(-60, 60, 100001).astype(np.float32)
    fast = _probe(mi_ctx, 1, 0, xs).astype(np.float64)
    assert np.max(np.abs(fast - np.exp(xs.astype(np.float64))) / np.exp(xs.astype(np.float64))) < 2e-5


def _run(mi_ctx, **kw):
    import armadillocudalinearinterpolation_amd as mi
    R = kw.pop("n_real", 8)
    edm = mi.EventDrivenMap(mi_ctx, [13.0589], R, **kw)
    f, partial = edm.ComputeF(Z_DRIVER, want_partial=True)
    return edm, f, partial, edm.debug_read()


@pytest.mark.parametrize("mean_quirk", [1, 0])      # 1 (the default): the reference's averaging; 0: the true mean
@pytest.mark.parametrize("n_grid,n_real", [(1024, 8), (512, 5), (1000, 3), (64, 4), (1024, 1)])
def test_compute_f_exact_mode_bit_parity(mi_ctx, n_grid, n_real, mean_quirk):
    edm, f, partial, dbg = _run(mi_ctx, n_grid=n_grid, n_real=n_real, mean_quirk=mean_quirk)
    p = oracle.edm_default_params(n_grid=n_grid, n_real=n_real, mean_quirk=mean_quirk)
    assert edm.params.mean_quirk == mean_quirk and oracle.edm_default_params().mean_quirk == 1
    fo, d = oracle.edm_compute_f(p, Z_DRIVER, nthreads=8)
    assert np.array_equal(dbg["seed_ind"], d["seed_ind"])
    for k in ("w", "v", "s"):
        assert np.array_equal(dbg[k], d[k], equal_nan=True), k           # coupling table + lift profile
    for k in ("t0", "i0", "t1", "i1", "accept"):
        assert np.array_equal(dbg[k], d[k], equal_nan=True), k           # evolve: last / crossed events
    assert np.array_equal(dbg["restricted"], d["restricted"], equal_nan=True)
    S = 3                                                                # partial block [sums | count | x0]
    assert partial.shape == (2 * S + 1,) and partial[S] == d["sums"][S]
    assert np.allclose(partial[:S], d["sums"][:S], rtol=1e-12, atol=0, equal_nan=True)
    assert np.array_equal(partial[S + 1:], d["sums"][S + 1:], equal_nan=True)
    if mean_quirk:                                                       # realisation 0 travels separately
        assert np.array_equal(partial[S + 1:], dbg["restricted"].reshape(S, n_real)[:, 0].astype(np.float64), equal_nan=True)
    else:
        assert np.all(partial[S + 1:] == 0.0)
    assert np.allclose(f, fo, rtol=0, atol=2e-7, equal_nan=True)         # 1 ulp(fp32) of the mean (sum order)
    # residual recomputed from the partial sums (what a multi-GPU all-reduce feeds) equals the device path
    assert np.allclose(edm.residual_from_sums(Z_DRIVER, partial), f, rtol=0, atol=1e-15, equal_nan=True)


def test_compute_f_heterogeneous_beta_bit_parity(mi_ctx):
    kw = dict(n_grid=512, n_real=6, beta_stddev=0.4, seed=987654321)
    edm, f, partial, dbg = _run(mi_ctx, **kw)
    p = oracle.edm_default_params(**kw)
    fo, d = oracle.edm_compute_f(p, Z_DRIVER, nthreads=8)
    for k in ("t0", "i0", "t1", "i1", "accept", "restricted"):
        assert np.array_equal(dbg[k], d[k], equal_nan=True), k
    assert len(set(dbg["t0"][:6].tolist())) > 1                          # realisations really differ
    assert np.allclose(f, fo, rtol=0, atol=2e-7, equal_nan=True)


@pytest.mark.parametrize("Z", [[0.3310, 0.6914], [0.3310, 0.35, 0.6914, 1.0, 1.3557]])
def test_other_spike_counts(mi_ctx, Z):
    """noSpikes is a compile-time 3 in the reference (parameters.hpp:12); the path takes 1..8 (generic kernel)."""
    import armadillocudalinearinterpolation_amd as mi
    S = len(Z)
    edm = mi.EventDrivenMap(mi_ctx, [13.0589], 3, n_grid=512, n_spikes=S)
    f, partial = edm.ComputeF(Z, want_partial=True)
    dbg = edm.debug_read()
    p = oracle.edm_default_params(n_grid=512, n_real=3, n_spikes=S)
    fo, d = oracle.edm_compute_f(p, Z, nthreads=3)
    for k in ("seed_ind", "v", "s", "t0", "i0", "t1", "i1", "accept", "restricted"):
        assert np.array_equal(dbg[k], d[k], equal_nan=True), k
    assert np.allclose(f, fo, rtol=0, atol=2e-7, equal_nan=True)


@pytest.mark.parametrize("kw", [dict(vth=0.0), dict(vth=-1.0), dict(I=1.5), dict(beta_mean=1.0), dict(beta_mean=0.0),
                                dict(time_horizon=1e-3), dict(a1=0.0, a2=0.0), dict(newton_max_iter=0),
                                dict(newton_tol=0.0), dict(time_horizon=40.0), dict(I=0.999),
                                dict(a1=50.0, max_events=3000)])      # runaway excitation: only the event cap ends it
def test_pathological_parameters_terminate_and_match(mi_ctx, kw):
    """Degenerate models (division by zero in 1-beta, non-positive threshold, no coupling, zero Newton budget,
    long horizons ...): the kernel must terminate (bounded event loop, every wave reaches its exit) and still
    reproduce the oracle bit for bit, NaNs and rejected realisations included."""
    import armadillocudalinearinterpolation_amd as mi
    ov = dict(kw)
    bm = ov.pop("beta_mean", 13.0589)
    edm = mi.EventDrivenMap(mi_ctx, [bm], 2, n_grid=512, **ov)
    f, partial = edm.ComputeF(Z_DRIVER, want_partial=True)
    dbg = edm.debug_read()
    p = oracle.edm_default_params(n_grid=512, n_real=2, **kw)
    fo, d = oracle.edm_compute_f(p, Z_DRIVER, nthreads=2)
    for k in ("v", "s", "t0", "i0", "t1", "i1", "accept"):
        assert np.array_equal(dbg[k], d[k], equal_nan=True), (kw, k)
    assert np.array_equal(np.isnan(f), np.isnan(fo)) and np.allclose(f, fo, rtol=0, atol=2e-7, equal_nan=True)


@pytest.mark.parametrize("kw", [dict(L=12.0), dict(L=24.0, b1=4.0, b2=3.0), dict(L=12.0, time_horizon=60.0), dict(n_grid=992),
                                dict(a1=3e38, a2=1e38), dict(a1=1.1e-29, a2=0.7e-29), dict(a1=7.0, a2=7.0, b1=3.5, b2=3.5)])
def test_quotient_range_tracking_of_the_state_pass(mi_ctx, evolve_form, kw):
    """The throughput kernel (one wave per realisation, homogeneous beta, EXACT math) divides by 1 - beta with a five-operation
    exact quotient that is valid for numerators in [2^-100, 2^101); instead of testing that in every slice, the state
    pass tracks the range of |s| per lane and takes an unguarded pass when the range allows it (csrc/mi_edm.hip).
    Cases in which the two passes alternate: rings on which the activity dies after 50-100 events and the last event is a
    "no neuron fires" step of 100 time units (exp(-100) pushes every numerator below 2^-100: guarded pass, IEEE expansion),
    a grid with padding lanes, couplings that overflow to infinity, sit below 2^-100 altogether, or vanish exactly.
    Every tap must be the oracle's bits."""
    if evolve_form == "workgroup_per_realisation":
        pytest.skip("the latency form has no such pass")
    import armadillocudalinearinterpolation_amd as mi
    R = 3500
    kw = dict(dict(n_grid=1024, max_events=800), **kw)
    edm = mi.EventDrivenMap(mi_ctx, [13.0589], R, **kw)
    edm.ComputeF(Z_DRIVER)
    dbg = edm.debug_read()
    # beta_stddev = 0: the R realisations are R copies of one computation -- the oracle evolves that one
    _, d = oracle.edm_compute_f(oracle.edm_default_params(n_real=1, **kw), Z_DRIVER)
    for k in ("w", "v", "s"):
        assert np.array_equal(dbg[k], d[k], equal_nan=True), (kw, k)
    S = edm.params.n_spikes
    for k in ("t0", "i0", "t1", "i1"):
        assert np.array_equal(np.asarray(dbg[k]).reshape(S, R), np.repeat(np.asarray(d[k]).reshape(S, 1), R, axis=1), equal_nan=True), (kw, k)
    assert np.array_equal(np.asarray(dbg["accept"]), np.repeat(np.asarray(d["accept"]), R))
    edm.close()


def test_setters_and_second_call(mi_ctx):
    """SetNoThreads(512) then ComputeF (Driver.cu:69-71) and a perturbed Z (finite-difference column)."""
    import armadillocudalinearinterpolation_amd as mi
    edm = mi.EventDrivenMap(mi_ctx, [13.0589], 4)
    f1024 = edm.ComputeF(Z_DRIVER)
    edm.SetNoThreads(512)
    f512 = edm.ComputeF(Z_DRIVER)
    p = oracle.edm_default_params(n_grid=512, n_real=4)
    fo, _ = oracle.edm_compute_f(p, Z_DRIVER, nthreads=4)
    assert np.allclose(f512, fo, rtol=0, atol=2e-7)
    assert not np.array_equal(f1024, f512)
    Zp = [Z_DRIVER[0] + 1e-2, Z_DRIVER[1], Z_DRIVER[2]]                  # NewtonSolver.cpp:184-191, eps = 1e-2
    fp = edm.ComputeF(Zp)
    fpo, _ = oracle.edm_compute_f(p, Zp, nthreads=4)
    assert np.allclose(fp, fpo, rtol=0, atol=2e-7)
    edm.SetTimeHorizon(4.0)
    p.time_horizon = 4.0
    ft, _ = oracle.edm_compute_f(p, Z_DRIVER, nthreads=4)
    assert np.allclose(edm.ComputeF(Z_DRIVER), ft, rtol=0, atol=2e-7)
    with pytest.raises(mi.MiError):
        mi.EventDrivenMap(mi_ctx, [13.0589], 4, n_grid=2048)


def test_fast_mode_within_tolerance(mi_ctx):
    import armadillocudalinearinterpolation_amd as mi
    _, f, _, dbg = _run(mi_ctx, n_real=4, math_mode=mi.MATH_FAST)
    p = oracle.edm_default_params(n_real=4)
    fo, d = oracle.edm_compute_f(p, Z_DRIVER, nthreads=4)
    assert np.all(dbg["accept"] == 1)
    assert np.max(np.abs(dbg["i0"].astype(int) - d["i0"].astype(int))) <= 1
    assert np.max(np.abs(dbg["restricted"] - d["restricted"])) < 5e-3      # positions: < 1 grid cell (5.9e-3)
    assert np.max(np.abs(f - fo)) < 5e-3


def test_many_realisations_are_identical_when_sigma_is_zero(mi_ctx):
    """R = 20000 (grid-stride path, > resident waves): sigma = 0 => every realisation equals realisation 0."""
    edm, f, partial, dbg = _run(mi_ctx, n_grid=512, n_real=20000)
    for k in ("t0", "i0", "t1", "i1"):
        a = dbg[k].reshape(3, 20000)
        assert np.all(a == a[:, :1]), k
    assert np.all(dbg["accept"] == dbg["accept"][0])
    p = oracle.edm_default_params(n_grid=512, n_real=2)
    fo, d = oracle.edm_compute_f(p, Z_DRIVER)
    assert np.array_equal(dbg["t0"].reshape(3, 20000)[:, 0], d["t0"].reshape(3, 2)[:, 0])
    t = edm.last_timings()
    assert t["evolve_ms"] > 0 and t["total_ms"] >= t["evolve_ms"]


def test_dedup_identical_is_bit_identical_to_full_evolution(mi_ctx):
    """Opt-in shortcut for sigma = 0 (all realisations are one computation): evolve one, replicate its events.
    Every stage tap, the count, the sums and f must equal the full evolution; with sigma > 0 the flag is ignored."""
    for n_grid, n_real in ((1024, 4099), (512, 20000)):
        _, f_full, p_full, d_full = _run(mi_ctx, n_grid=n_grid, n_real=n_real)
        edm, f_dd, p_dd, d_dd = _run(mi_ctx, n_grid=n_grid, n_real=n_real, dedup_identical=1)
        assert np.array_equal(f_full, f_dd) and np.array_equal(p_full, p_dd)
        for k in ("t0", "i0", "t1", "i1", "accept", "restricted"):
            assert np.array_equal(d_full[k], d_dd[k]), k
        assert edm.last_timings()["evolve_ms"] > 0
    _, f_full, p_full, d_full = _run(mi_ctx, n_grid=512, n_real=3000, beta_stddev=0.3)
    _, f_dd, p_dd, d_dd = _run(mi_ctx, n_grid=512, n_real=3000, beta_stddev=0.3, dedup_identical=1)
    assert np.array_equal(f_full, f_dd) and np.array_equal(p_full, p_dd)
    assert np.array_equal(d_full["t0"], d_dd["t0"]) and not np.all(d_dd["t0"].reshape(3, 3000) == d_dd["t0"].reshape(3, 3000)[:, :1])


def test_concurrent_evaluations_equal_sequential_ones(mi_ctx):
    """mi_edm_compute_f_begin/_end on replicas with streams of their own (ComputeFBatch: the columns of the
    finite-difference Jacobian evaluated together) give, column by column, exactly what ComputeF gives; setters on the
    parent reach the replicas; the Newton iterates with and without concurrent columns are identical."""
    import armadillocudalinearinterpolation_amd as mi
    from armadillocudalinearinterpolation_amd import newton
    edm = mi.EventDrivenMap(mi_ctx, [13.0589], 700, n_grid=512)
    Zs = [np.array(Z_DRIVER) + d for d in ([0, 0, 0], [1e-2, 0, 0], [0, 1e-2, 0], [0, 0, 1e-2], [-5e-3, 2e-3, 0])]
    seq = np.stack([edm.ComputeF(z) for z in Zs])
    assert np.array_equal(edm.ComputeFBatch(Zs), seq)
    F, P = edm.ComputeFBatch(Zs[:2], want_partial=True)
    assert np.array_equal(F, seq[:2]) and P.shape == (2, 7) and np.all(P[:, 3] == P[0, 3])
    edm.SetParameterStdDev(0.3)                          # the replicas must pick this up
    seq2 = np.stack([edm.ComputeF(z) for z in Zs[:3]])
    assert not np.array_equal(seq2, seq[:3])
    assert np.array_equal(edm.ComputeFBatch(Zs[:3]), seq2)
    with pytest.raises(mi.MiError):                      # one evaluation per handle at a time
        edm.begin(Zs[0])
        edm.begin(Zs[1])
    edm.end()
    edm.close()
    runs = []
    for concurrent in (True, False):
        prob = mi.EventDrivenMap(mi_ctx, [13.0589], 300, n_grid=1024)
        pars = newton.ParameterList(tolerance=1e-4, maxIterations=10, printOutput=False, finiteDifferenceEpsilon=1e-2)
        solver = newton.NewtonSolver(prob, np.array(Z_DRIVER), pars)
        solver.concurrent_columns = concurrent
        runs.append(solver.Solve())
        prob.close()
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1], equal_nan=True)
    assert runs[0][2] == runs[1][2] and runs[0][3] == runs[1][3]


def test_python_newton_on_gpu_matches_oracle_newton(mi_ctx):
    """The replicated Newton loop used for multi-GPU runs (newton.py), here on one GPU with 1024 grid points."""
    import armadillocudalinearinterpolation_amd as mi
    from armadillocudalinearinterpolation_amd import newton
    Z0 = [float(np.float32(z)) for z in Z_DRIVER]
    edm = mi.EventDrivenMap(mi_ctx, [13.0589], 64, mean_quirk=0)          # the true mean does not depend on R when sigma = 0
    pars = newton.ParameterList(tolerance=1e-4, maxIterations=10, printOutput=False, finiteDifferenceEpsilon=1e-2)
    u, hist, conv, it = newton.NewtonSolver(edm, Z0, pars).Solve()

    class Orc:
        def ComputeF(self, Z):
            return oracle.edm_compute_f(oracle.edm_default_params(n_real=1, mean_quirk=0), Z)[0]
    uo, ho, co, io = newton.NewtonSolver(Orc(), Z0, pars).Solve()
    assert conv and co and it == io
    assert np.allclose(u, uo, rtol=0, atol=1e-4) and np.allclose(hist, ho, rtol=0, atol=5e-6)


def _one_realisation_reference(n_grid, Z):
    """One oracle realisation (sigma = 0: every realisation is this one) and the residual of R copies of it under both
    averaging rules: reference (R-1 copies summed, R in the divisor; EventDrivenMap.cu:800-802,:817,:822) and true mean.
    The device sums the fp32 position in fp64 -- exact -- so the expected mean is fl32(k x / R), bit for bit."""
    p = oracle.edm_default_params(n_grid=n_grid, n_real=1)
    _, d = oracle.edm_compute_f(p, Z)
    x = d["restricted"].astype(np.float64)
    U0 = np.array([Z[0], 0.0, Z[1], Z[2]])

    def f(R, quirk):
        k = R - 1 if (quirk and R > 1) else R
        mean = (k * x / R).astype(np.float32).astype(np.float64)
        return (-U0[0] * U0[1:] - mean) + U0[0] * 5.0
    return d, x, f


@pytest.mark.parametrize("n_grid", [1024, 512])
def test_reference_mean_at_the_driver_size(mi_ctx, n_grid):
    """Driver.cu:19 R = 1000, every realisation accepted, the reference's averaging (the default): f must be the
    oracle's bit for bit, and differ from the true-mean residual by x_m / 1000 (ten Newton tolerances)."""
    d, x, fref = _one_realisation_reference(n_grid, Z_DRIVER)
    R = 1000
    edm, f, partial, dbg = _run(mi_ctx, n_grid=n_grid, n_real=R)
    assert edm.params.mean_quirk == 1 and np.all(dbg["accept"] == 1) and partial[3] == R
    assert np.array_equal(dbg["restricted"].reshape(3, R), np.repeat(d["restricted"].reshape(3, 1), R, axis=1))
    assert np.array_equal(partial[:3], (R - 1) * x) and np.array_equal(partial[4:], x)
    assert np.array_equal(f, fref(R, True))
    _, ft, pt, _ = _run(mi_ctx, n_grid=n_grid, n_real=R, mean_quirk=0)
    assert np.array_equal(ft, fref(R, False)) and np.array_equal(pt[:3], R * x) and np.all(pt[4:] == 0)
    assert np.allclose(f - ft, x / R, rtol=0, atol=2e-7) and np.all(np.abs(f - ft) > 5e-4)


def test_driver_grid_many_realisations_match_the_oracle(mi_ctx):
    """The reference's Driver.cu grid (N = 512) with enough realisations for the wave-per-realisation kernel to take the
    exact quotient by uniform divisors (csrc/mi_edm_math.hpp div_by): every row must equal row 0, and row 0 the oracle's
    realisation -- the oracle divides with `/`."""
    R = 20_000
    d, x, fref = _one_realisation_reference(512, Z_DRIVER)
    edm, f, partial, dbg = _run(mi_ctx, n_grid=512, n_real=R)
    for k in ("t0", "i0", "t1", "i1"):
        a = dbg[k].reshape(3, R)
        assert np.all(a == a[:, :1]), k
        assert np.array_equal(a[:, 0], d[k].reshape(3, 1)[:, 0]), k
    assert np.all(dbg["accept"] == 1) and partial[3] == R
    assert np.array_equal(dbg["restricted"].reshape(3, R)[:, 0], d["restricted"])
    assert np.array_equal(f, fref(R, True))
    edm.close()


@pytest.mark.timeout(600)
def test_config4_per_gpu_share_125k_realisations(mi_ctx):
    """BASELINE configs[3] at one GPU's share: 125 000 realisations x 1024 grid points, EXACT math, sigma = 0.  Every
    row must equal row 0, row 0 must equal the oracle's realisation, count == R, and f must be the oracle's for both
    averaging rules (bit for bit: the sums are exact)."""
    R = 125_000
    d, x, fref = _one_realisation_reference(1024, Z_DRIVER)
    edm, f, partial, dbg = _run(mi_ctx, n_grid=1024, n_real=R)
    for k in ("t0", "i0", "t1", "i1"):
        a = dbg[k].reshape(3, R)
        assert np.all(a == a[:, :1]), k
        assert np.array_equal(a[:, 0], d[k].reshape(3, 1)[:, 0]), k
    assert np.all(dbg["accept"] == 1) and partial[3] == R
    assert np.array_equal(dbg["restricted"].reshape(3, R)[:, 0], d["restricted"])
    assert np.array_equal(partial[:3], (R - 1) * x) and np.array_equal(f, fref(R, True))
    # the decision-coverage taps over the whole share: 848 events in every realisation, at most 9 Newton iterations, no cap
    # reached, no event without a firing neuron, no exact tie -- none of D0 / D1 / D8 is exercised on config 4's inputs
    taps = edm.debug_counters()
    assert taps["accepted"] == R and taps["events"] == R * taps["max_events_one"] and taps["max_events_one"] == 848
    assert taps["max_newton_iter"] <= 9 and taps["newton_cap_hits"] == 0 and taps["event_cap_hits"] == 0
    assert taps["no_firing_events"] == 0 and taps["argmin_ties"] == 0
    edm.params.mean_quirk = 0
    edm._push()
    ft, pt = edm.ComputeF(Z_DRIVER, want_partial=True)
    assert np.array_equal(ft, fref(R, False)) and np.array_equal(pt[:3], R * x) and pt[3] == R
    edm.close()


@pytest.mark.parametrize("kw", [
    dict(n_grid=1024, n_real=6),                                  # config 4's grid, Driver.cu:24's inputs
    dict(n_grid=512, n_real=6),                                   # Driver.cu:69's grid
    dict(n_grid=1024, n_real=5, beta_stddev=0.3, seed=7),         # per-neuron beta
    dict(n_grid=1024, n_real=3, max_events=10),                   # event cap reached (D8)
    dict(n_grid=512, n_real=3, newton_max_iter=2),                # Newton cap reached (D0)
])
def test_device_decision_counters_equal_the_oracles(mi_ctx, kw):
    """mi_edm_debug_counters re-runs Evolve with the tapped instantiation of the kernel: its counts of events, Newton
    iterations, cap hits and no-firing events must equal oracle/edm_oracle.c's orc_edm_counters on the same inputs --
    and at the BASELINE inputs (first two cases) none of the documented decisions D0 / D1 / D8 is reached on the device
    either (tests/test_edm_oracle_cpu.py states the same over the whole config-5 solve on the CPU)."""
    kw = dict(kw)
    edm, f, partial, dbg = _run(mi_ctx, **dict(kw))
    dev = edm.debug_counters()
    again = edm.debug_read()                                       # the tapped run must leave the very same events behind
    for k in ("t0", "i0", "t1", "i1", "accept"):
        assert np.array_equal(again[k], dbg[k], equal_nan=True), k
    c = oracle.EdmCounters()
    oracle.edm_compute_f(oracle.edm_default_params(**kw), Z_DRIVER, nthreads=8, counters=c)
    o = c.as_dict()
    for name in ("events", "max_events_one", "max_newton_iter", "newton_cap_hits", "event_cap_hits", "accepted", "no_firing_events"):
        assert dev[name] == o[name], (name, dev, o)
    assert (dev["argmin_ties"] > 0) == (o["argmin_ties"] > 0)
    if set(kw) <= {"n_grid", "n_real"}:
        assert dev["newton_cap_hits"] == dev["event_cap_hits"] == dev["no_firing_events"] == dev["argmin_ties"] == 0
        assert dev["accepted"] == kw["n_real"] and dev["max_newton_iter"] <= 12


@pytest.mark.parametrize("kw", [dict(newton_max_iter=0, max_events=300), dict(I=0.5)])
@pytest.mark.parametrize("n_grid", [1024, 512, 992, 1000, 64])
def test_arg_min_ties_follow_the_reference_reduction(mi_ctx, n_grid, kw):
    """Exact ties of firing times are broken as the reference's blockReduceMin breaks them on 32-wide warps
    (EventDrivenMap.cu:843-881; oracle/edm_oracle.c orc_edm_argmin, checked there against a literal emulation of the
    shuffle trees): newton_max_iter = 0 gives some 80 events with many-way ties of real firing times, I = 0.5 an event at
    which nobody fires (1024 threads: neuron 1023 wins; fewer warps: the padding pair (100.0f, 0)); N = 1000 is not made of
    whole warps (lowest index).  Every event array equal to the oracle's, in both kernel forms."""
    kw = dict(kw, n_grid=n_grid, n_real=5)
    edm, f, partial, dbg = _run(mi_ctx, **dict(kw))
    c = oracle.EdmCounters()
    fo, d = oracle.edm_compute_f(oracle.edm_default_params(**kw), Z_DRIVER, nthreads=5, counters=c)
    for k in ("t0", "i0", "t1", "i1", "accept"):
        assert np.array_equal(dbg[k], d[k], equal_nan=True), k
    assert c.no_firing_events >= 5 and c.argmin_tree_mismatch == 0
    if "newton_max_iter" in kw and n_grid >= 512:
        assert c.argmin_ties >= 100
    dev = edm.debug_counters()
    assert dev["no_firing_events"] == c.no_firing_events and (dev["argmin_ties"] > 0) == (c.argmin_ties > 0)
