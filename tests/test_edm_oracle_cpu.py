"""CPU suite for the EventDrivenMap oracle (oracle/edm_oracle.c): math routines against libm/scipy, stage
known-answers, and the physics sanity of the residual at the reference's own inputs (Driver.cu:16,24).
The reference holds no expected outputs for any of this ("parity unpinned", see the oracle header)."""
import numpy as np
import pytest
from scipy.special import erfinv

import oracle

Z_DRIVER = [0.3310, 0.6914, 1.3557]          # Driver.cu:24
BETA = 13.0589                               # Driver.cu:16


def test_math_routines_accuracy_and_special_values():
    x = np.linspace(-87, 88, 400001).astype(np.float32)
    e = oracle.edm_math_probe(0, x).astype(np.float64)
    ref = np.exp(x.astype(np.float64))
    assert np.max(np.abs(e - ref) / ref) < 1.5e-7                       # ~1.2 ulp
    sp = oracle.edm_math_probe(0, np.array([0.0, -0.0, np.inf, -np.inf, 89.0, -104.0], np.float32))
    assert sp[0] == 1.0 and sp[1] == 1.0 and np.isinf(sp[2]) and sp[3] == 0.0 and np.isinf(sp[4]) and sp[5] == 0.0
    assert np.isnan(oracle.edm_math_probe(0, np.array([np.nan], np.float32))[0])
    xl = np.exp(np.linspace(-80, 80, 400001)).astype(np.float32)
    l = oracle.edm_math_probe(1, xl).astype(np.float64)
    refl = np.log(xl.astype(np.float64))
    assert np.max(np.abs(l - refl) / np.maximum(np.abs(refl), 1.0)) < 2e-7
    sl = oracle.edm_math_probe(1, np.array([0.0, -1.0, np.inf, 1.0], np.float32))
    assert np.isneginf(sl[0]) and np.isnan(sl[1]) and np.isposinf(sl[2]) and sl[3] == 0.0
    a = np.linspace(0.01, 300, 5000).astype(np.float32)
    b = np.full_like(a, 1.0 / BETA)
    pw = oracle.edm_math_probe(2, a, b).astype(np.float64)
    assert np.max(np.abs(pw - a.astype(np.float64) ** (1.0 / BETA)) / pw) < 1e-6
    assert np.isnan(oracle.edm_math_probe(2, np.array([-1.0], np.float32), np.array([0.3], np.float32))[0])
    u = np.linspace(-0.99999, 0.99999, 100001).astype(np.float32)
    ei = oracle.edm_math_probe(3, u).astype(np.float64)
    assert np.max(np.abs(ei - erfinv(u.astype(np.float64))) / np.maximum(np.abs(erfinv(u.astype(np.float64))), 1e-3)) < 2e-6


def test_coupling_table_and_seed_indices():
    p = oracle.edm_default_params()
    w = oracle.edm_coupling(p)
    N, L = 1024, 3.0
    x = -L + (2 * L / N) * np.arange(N)
    w0 = (11 * np.exp(-5 * np.abs(x)) - 7 * np.exp(-3.5 * np.abs(x))) * 2 * L / N
    assert np.allclose(w, np.roll(w0, -N // 2), rtol=3e-6, atol=1e-9)       # circshift by N/2, :826-841
    assert w[0] == np.float32(4.0 * 2 * 3 / 1024)                           # distance 0: (a1-a2)*2L/N, exact
    ind = oracle.edm_seed_indices(p, Z_DRIVER)
    assert ind[0] == 512
    for m in (1, 2):                                                         # first grid point left of -c*Z_m
        target = -Z_DRIVER[0] * Z_DRIVER[m]
        xi = -3 + (2 * ind[m] * 3.0) / 1024
        assert xi < target <= -3 + (2 * (ind[m] + 1) * 3.0) / 1024
    # [D5] no qualifying point: the previous entry is kept
    stale = oracle.edm_seed_indices(p, [0.3310, 50.0, 1.3557], prev=[7, 9, 11])
    assert stale[1] == 9


def test_lift_profile_shape():
    p = oracle.edm_default_params()
    U = np.array([Z_DRIVER[0], 0.0, Z_DRIVER[1], Z_DRIVER[2]], np.float32)
    v, s = oracle.edm_lift(p, U)
    ok = ~np.isnan(v)
    assert ok.sum() > 800 and np.all(v[ok] < 1.0) and np.all(v[ok] >= 0.0)   # sub-threshold, clamped (:538)
    # 0*inf poisoning of the unselected branch (boolean-multiply structure, :522-528): far ahead of the wave
    # (exp overflow at x < c*U_m - 88.7*c/beta): v is NaN from index 857 on, s from 820 on
    assert np.isnan(v[-1]) and np.all(np.isnan(v[857:])) and not np.isnan(v[856])
    assert np.all(np.isfinite(s[:820])) and np.all(np.isnan(s[820:]))


def test_residual_is_small_at_the_drivers_initial_guess():
    """The reference's Newton solve starts from Z_DRIVER (Driver.cu:24) because it is near a root: a faithful
    restatement of lift/evolve/restrict must give |F| << 1 there, and every realisation must be accepted."""
    p = oracle.edm_default_params(n_real=2, mean_quirk=0)       # the true mean: the physics check
    f, d = oracle.edm_compute_f(p, Z_DRIVER, nthreads=2)
    assert np.all(d["accept"] == 1)
    assert np.linalg.norm(f) < 2e-2
    # bumps sit near c*T - c*U_m after time T: 3 ordered positions, one grid cell apart for last/crossed
    assert np.all(d["i1"].astype(int) - d["i0"].astype(int) == 1)
    assert np.all(d["t0"] <= 5.0) and np.all(d["t1"] > 5.0)
    x = d["restricted"].reshape(3, 2)[:, 0]
    assert x[0] > x[1] > x[2] and abs(x[0] - Z_DRIVER[0] * 5.0) < 0.02
    # sigma = 0: realisations are identical
    assert np.all(d["restricted"].reshape(3, 2)[:, 0] == d["restricted"].reshape(3, 2)[:, 1])
    # f = -c*U - mean + c*T in fp64 (:239)
    mean = d["sums"][:3] / d["sums"][3]
    c = Z_DRIVER[0]
    fexp = -c * np.array([0.0, Z_DRIVER[1], Z_DRIVER[2]]) - mean.astype(np.float32).astype(np.float64) + c * 5.0
    assert np.allclose(f, fexp, rtol=0, atol=1e-7)
    assert np.all(d["sums"][4:] == 0.0)                          # no realisation-0 term without the reference rule


def test_reference_averaging_is_the_default_and_leaves_realisation_0_out():
    """EventDrivenMap.cu:800-802 overwrites accept[0] with the count, :817 then tests accept[index]==1 and :822 divides
    by accept[0]: with R accepted, identical realisations the reference returns x*(R-1)/R.  That is the default
    (mean_quirk = 1); the partial block carries realisation 0 separately so that shards can be added."""
    assert oracle.edm_default_params().mean_quirk == 1
    R = 4
    pq = oracle.edm_default_params(n_real=R)
    pt = oracle.edm_default_params(n_real=R, mean_quirk=0)
    fq, dq = oracle.edm_compute_f(pq, Z_DRIVER, nthreads=2)
    ft, dt = oracle.edm_compute_f(pt, Z_DRIVER, nthreads=2)
    x = dq["restricted"].reshape(3, R)
    assert np.array_equal(dq["restricted"], dt["restricted"]) and np.all(dq["accept"] == 1)
    assert dq["sums"][3] == R and np.array_equal(dq["sums"][4:], x[:, 0].astype(np.float64))
    assert np.array_equal(dq["sums"][:3], x[:, 1:].astype(np.float64).sum(axis=1))
    mean_q = (dq["sums"][:3] / R).astype(np.float32).astype(np.float64)
    c = Z_DRIVER[0]
    assert np.array_equal(fq, (-c * np.array([0.0, Z_DRIVER[1], Z_DRIVER[2]]) - mean_q) + c * 5.0)
    assert np.allclose(ft - fq, -x[:, 0] / R, rtol=0, atol=1e-6)      # the reference's mean is short by x_0 / R
    assert np.array_equal(oracle.edm_residual_from_sums(pq, Z_DRIVER, dq["sums"]), fq)
    # count == 1: realisation 0 is summed whatever its flag (accept[0] then holds 1)
    p1 = oracle.edm_default_params(n_real=1)
    f1, d1 = oracle.edm_compute_f(p1, Z_DRIVER)
    assert np.array_equal(f1, oracle.edm_compute_f(oracle.edm_default_params(n_real=1, mean_quirk=0), Z_DRIVER)[0])
    assert np.array_equal(oracle.edm_residual_from_sums(p1, Z_DRIVER, d1["sums"]), f1)
    # shards: blocks add; only the shard holding realisation 0 applies the rule
    ph = oracle.edm_default_params(n_real=5, beta_stddev=0.3, seed=99)
    fw, dw = oracle.edm_compute_f(ph, Z_DRIVER, nthreads=2)
    tot = np.zeros(7)
    for lo, hi in ((0, 2), (2, 5)):
        tot += oracle.edm_compute_f(oracle.edm_default_params(n_real=hi - lo, real_offset=lo, beta_stddev=0.3, seed=99), Z_DRIVER)[1]["sums"]
    assert np.allclose(tot, dw["sums"], rtol=1e-15, atol=0)
    assert np.allclose(oracle.edm_residual_from_sums(ph, Z_DRIVER, tot), fw, rtol=0, atol=1e-7)


def test_heterogeneous_beta_is_deterministic_and_centred():
    p = oracle.edm_default_params(beta_stddev=0.5, seed=123)
    b = np.array([oracle.edm_beta(p, r, i) for r in range(4) for i in range(0, 1024, 7)])
    assert abs(b.mean() - BETA) < 0.08 and 0.4 < b.std() < 0.6
    assert oracle.edm_beta(p, 3, 5) == oracle.edm_beta(p, 3, 5)
    p0 = oracle.edm_default_params()
    assert oracle.edm_beta(p0, 3, 5) == np.float32(BETA)
