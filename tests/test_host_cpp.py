"""The C++ host layer (armadillocudalinearinterpolation_amd/host + include/mi355_arma*.hpp).

CPU part: it compiles with plain g++ (Armadillo when installed, otherwise the stand-in) and its Newton loop /
linear algebra self-test passes.  GPU part: the Armadillo-facing wrappers and the Driver.cu problem run on
the MI355X and agree with the oracle (wrappers: bit-exact; Newton root: within the solver tolerance)."""
import json
import os
import subprocess

import numpy as np
import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "armadillocudalinearinterpolation_amd", "host")


def _build():
    from armadillocudalinearinterpolation_amd import _build as b
    b.build_lib()
    subprocess.check_call(["make", "-s", "-C", HOST, "all"])


def test_host_layer_builds_and_selftest_passes():
    _build()
    out = subprocess.run([os.path.join(HOST, "host_selftest")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all passed" in out.stdout


def _lcg_queries(n, seed, scale, shift):
    s, out = seed, np.empty(n)
    for i in range(n):
        s = (s * 6364136223846793005 + 1442695040888963407) & (2**64 - 1)
        out[i] = (s >> 11) * 2.0 ** -53 * scale + shift
    return out, s


@pytest.mark.gpu
def test_arma_interp_bench_runs_and_its_checksum_matches_the_oracle():
    """The C++ end-to-end timing program of the Armadillo-shaped calls (host/arma_interp_bench.cpp) at a small size:
    the checksum it prints over every 9973rd result must equal the oracle's on the same LCG queries."""
    _build()
    nq, ng = 20_000_000, 50_000                          # > 16 M queries: the chunked, pinned host path
    out = subprocess.run([os.path.join(HOST, "arma_interp_bench"), str(nq), str(ng)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    chk = float([ln for ln in out.stdout.splitlines() if ln.startswith("checksum")][0].split()[1])
    X = np.arange(ng) / (ng - 1)
    Y = np.sin(6.283185307179586 * X) + 0.5 * X
    idx = np.arange(0, nq, 9973, dtype=np.uint64)
    # LCG state after k steps by composing the affine map x -> a x + c with itself (repeated squaring)
    a, c, s0, mask = 6364136223846793005, 1442695040888963407, 0x5EED0003, (1 << 64) - 1

    def state_after(k):
        A, C, pa, pc = 1, 0, a, c                          # (A, C): accumulated map, (pa, pc): a^(2^j) map
        while k:
            if k & 1:
                A, C = (pa * A) & mask, (pa * C + pc) & mask
            pa, pc = (pa * pa) & mask, (pa * pc + pc) & mask
            k >>= 1
        return (A * s0 + C) & mask

    xs = np.array([(state_after(int(i) + 1) >> 11) * 2.0 ** -53 for i in idx])
    ref = oracle.interp1_bracket(X, Y, xs)
    acc = 0.0
    for v in ref:                                          # same summation order as the C++ loop
        acc += v
    assert chk == acc


@pytest.mark.gpu
def test_arma_wrappers_match_oracle(tmp_path):
    _build()
    out = subprocess.run([os.path.join(HOST, "arma_wrappers_test"), str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    ng, nq = 10000, 100000
    X = np.arange(ng) / (ng - 1)
    Y = np.sin(6.283185307179586 * X) + 0.5 * X
    XI, s = _lcg_queries(nq, 42, 1.1, -0.05)
    rd = lambda f: np.fromfile(os.path.join(tmp_path, f), dtype=np.float64)  # noqa: E731
    ref = oracle.interp1_arma(X, Y, XI)
    assert np.array_equal(rd("w_interp1.bin"), ref, equal_nan=True)
    assert np.array_equal(rd("w_table_nan.bin"), ref, equal_nan=True)
    assert np.array_equal(rd("w_table_extrap.bin"), oracle.interp1_arma(X, Y, XI, extrap=-1.0))
    assert np.array_equal(rd("w_group_table.bin"), ref, equal_nan=True)            # query shards over a device group
    assert np.array_equal(rd("w_group1_table_extrap.bin"), oracle.interp1_arma(X, Y, XI, extrap=-1.0))
    xr = np.fromfile(os.path.join(tmp_path, "w_restrict.bin"), dtype=np.float32)
    assert np.array_equal(xr, oracle.restrict_f32([4, 4.5, 1, 4.99], [512, 100, 1023, 0], [6, 5.25, 9, 5.01],
                                                  [514, 101, 1023, 1], 5.0, 3.0, 1024))
    nx, ny, n2 = 40, 25, 20000
    xg = np.array([0.1 * j * (1.0 + 0.01 * j) for j in range(nx)])
    yg = np.array([-1.0 + 0.2 * i for i in range(ny)])
    Z = np.sin(xg)[None, :] * np.cos(yg)[:, None] + 0.1 * xg[None, :] * yg[:, None]
    xq, yq = np.empty(n2), np.empty(n2)
    for k in range(n2):
        s = (s * 6364136223846793005 + 1442695040888963407) & (2**64 - 1)
        xq[k] = (s >> 11) * 2.0 ** -53 * (xg[-1] + 0.2) - 0.1
        s = (s * 6364136223846793005 + 1442695040888963407) & (2**64 - 1)
        yq[k] = (s >> 11) * 2.0 ** -53 * 5.2 - 1.1
    # libm sin/cos of the C++ program and numpy agree on this platform (same glibc); compare to 1e-15 to be safe
    got = rd("w_interp2.bin")
    ref2 = oracle.interp2_bilinear(xg, yg, Z, xq, yq)
    assert np.array_equal(np.isnan(got), np.isnan(ref2))
    assert np.nanmax(np.abs(got - ref2)) < 1e-14
    assert np.array_equal(rd("w_group_interp2.bin"), got, equal_nan=True)       # the same pairs sharded over a device group


def _oracle_f(p, u, seed_ind=None, n_real_reference_mean=0):
    """Oracle residual.  n_real_reference_mean = R > 0: the residual of R IDENTICAL realisations (sigma = 0) averaged
    as the reference does (realisation 0 left out of the sum, R in the divisor, EventDrivenMap.cu:800-802,:817,:822),
    formed from ONE oracle realisation: the device adds R-1 copies of the fp32 position x in fp64 -- exact, x has 24
    significant bits and R < 2^29 -- so the mean is fl32((R-1) x / R) (the shortcut is checked against the full oracle
    at R = 5 in test_reference_mean_shortcut_equals_full_oracle)."""
    if not n_real_reference_mean:
        f, d = oracle.edm_compute_f(p, u, seed_ind=seed_ind, nthreads=8, debug=True)
        return f, d["seed_ind"]
    R = int(n_real_reference_mean)
    assert p.beta_stddev == 0.0 and p.n_real == 1
    _, d = oracle.edm_compute_f(p, u, seed_ind=seed_ind, debug=True)
    S = p.n_spikes
    x = d["restricted"].reshape(S, 1)[:, 0].astype(np.float64)
    if int(d["accept"][0]) != 1:
        mean = np.full(S, np.nan)                                   # 0 / 0, as on the device
    elif R == 1:
        mean = x
    else:
        mean = ((R - 1) * x / R).astype(np.float32).astype(np.float64)
    U0 = np.concatenate([[u[0], 0.0], np.asarray(u, dtype=np.float64)[1:]])
    f = (-U0[0] * U0[1:S + 1] - mean) + U0[0] * float(p.time_horizon)
    return f, d["seed_ind"]


def test_reference_mean_shortcut_equals_full_oracle():
    Z = [0.3310, 0.6914, 1.3557]
    full, _ = oracle.edm_compute_f(oracle.edm_default_params(n_grid=512, n_real=5), Z, nthreads=5)
    short, _ = _oracle_f(oracle.edm_default_params(n_grid=512, n_real=1), np.array(Z), n_real_reference_mean=5)
    assert np.array_equal(full, short)


def _newton_on_oracle(p, Z0, tol, max_it, eps, damping=1.0, n_real_reference_mean=0):
    """tests-only restatement of NewtonSolver.cpp:40-197 driving the ORACLE residual"""
    u = np.array(Z0, dtype=np.float64)
    kw = dict(n_real_reference_mean=n_real_reference_mean)
    f, seed = _oracle_f(p, u, None, **kw)
    hist, it = [np.linalg.norm(f)], 0
    while it < max_it and not hist[-1] <= tol:
        J = np.empty((3, 3))
        for i in range(3):
            du = u.copy()
            du[i] += eps
            df, seed = _oracle_f(p, du, seed, **kw)
            J[:, i] = (df - f) * eps ** -1
        u = u + damping * np.linalg.solve(J, -f)
        it += 1
        f, seed = _oracle_f(p, u, seed, **kw)
        hist.append(np.linalg.norm(f))
    return u, hist, it


@pytest.mark.gpu
def test_driver_newton_solve_matches_oracle_newton(tmp_path):
    """BASELINE config 5 at 1 GPU: the Driver.cu problem (tol 1e-4, maxIt 10, FD eps 1e-2, damping 1).

    With sigma = 0 the residual is a piecewise-smooth function of Z (event indices are discrete), and the
    oracle-driven Newton iteration shows that Driver.cu's live setting (512 grid points) does NOT reach 1e-4 in
    10 iterations while 1024 grid points converges in 8.  The GPU path must reproduce exactly that behaviour:
    same iterates at 1024 points (root within the Newton tolerance), same early history and the same
    not-converged exit flag at 512 points."""
    _build()
    Z0 = [float(np.float32(0.3310)), float(np.float32(0.6914)), float(np.float32(1.3557))]
    # ---- 1024 grid points, the reference's averaging (the driver's default; EventDrivenMap.cu:800-824): realisation 0
    # is left out of the sum and 1000 stays in the divisor, which moves every f_m by x_m / 1000 ~ 1e-3 -- ten times the
    # Newton tolerance.  The GPU path must follow the oracle's Newton iteration on THAT residual.
    js = os.path.join(tmp_path, "driver1024ref.json")
    out = subprocess.run([os.path.join(HOST, "driver"), "--real", "1000", "--threads", "1024", "--json", js],
                         capture_output=True, text=True)
    rq = json.load(open(js))
    assert rq["mean"] == "reference"
    p1 = oracle.edm_default_params(n_grid=1024, n_real=1)
    try:
        uq, histq, itq = _newton_on_oracle(p1, Z0, 1e-4, 10, 1e-2, n_real_reference_mean=1000)
        okq = bool(histq[-1] <= 1e-4)
    except np.linalg.LinAlgError:
        uq, histq, itq, okq = None, [], 0, False
    assert rq["converged"] == okq and (out.returncode == 0) == okq
    fq0, _ = _oracle_f(p1, np.array(Z0), n_real_reference_mean=1000)
    assert np.allclose(rq["f0_1024"], fq0, rtol=0, atol=2e-7)
    ft0, _ = oracle.edm_compute_f(oracle.edm_default_params(n_grid=1024, n_real=2, mean_quirk=0), Z0)
    assert np.all(np.abs(np.array(rq["f0_1024"]) - ft0) > 5e-4)              # not the true mean's residual
    if okq:
        assert rq["iterations"] == itq
        assert np.allclose(rq["solution"], uq, rtol=0, atol=1e-4) and np.allclose(rq["history"], histq, rtol=0, atol=5e-6)
    elif rq["history"] and histq:
        assert np.allclose(rq["history"][:3], histq[:3], rtol=0, atol=5e-6)
    # ---- 1024 grid points, the true mean: converges
    js = os.path.join(tmp_path, "driver1024.json")
    out = subprocess.run([os.path.join(HOST, "driver"), "--real", "1000", "--threads", "1024", "--json", js,
                          "--stability", "--true-mean"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "The method converged after" in out.stdout and "Newton Method" in out.stdout
    r = json.load(open(js))
    assert r["converged"] and r["iterations"] <= 10
    assert r["residual_evaluations"] == 1 + 4 * r["iterations"]
    # sigma = 0: the residual does not depend on the number of realisations, so 2 oracle realisations suffice
    p = oracle.edm_default_params(n_grid=1024, n_real=2, mean_quirk=0)
    u, hist, it = _newton_on_oracle(p, Z0, 1e-4, 10, 1e-2)
    assert it == r["iterations"] and r["mean"] == "true"
    assert np.allclose(r["solution"], u, rtol=0, atol=1e-4)                 # same root within the Newton tolerance
    assert np.allclose(r["history"], hist, rtol=0, atol=5e-6)
    assert r["history"][-1] <= 1e-4
    f1024, _ = oracle.edm_compute_f(p, Z0)
    assert np.allclose(r["f0_1024"], f1024, rtol=0, atol=2e-7)
    # Stability (Stability.cpp:52-111): eigenvalues of I + FD Jacobian at the solution vs numpy on the oracle's
    us = np.array(r["solution"])
    f0, d = oracle.edm_compute_f(p, us)
    J = np.empty((3, 3))
    for i in range(3):
        du = us.copy()
        du[i] += 1e-2
        df, d = oracle.edm_compute_f(p, du, seed_ind=d["seed_ind"])
        J[:, i] = (df - f0) * (1e-2) ** -1
    ev_ref = np.sort_complex(np.linalg.eigvals(J + np.eye(3)))
    ev = np.sort_complex(np.array([complex(a, b) for a, b in r["eigenvalues"]]))
    assert np.allclose(ev, ev_ref, rtol=0, atol=1e-3)
    assert r["n_unstable"] == int(np.sum(np.abs(ev_ref) > 1.0))
    # ---- 512 grid points (Driver.cu:69): not converged after 10 iterations, like the oracle iteration
    js = os.path.join(tmp_path, "driver512.json")
    dbg = os.path.join(tmp_path, "dumps")
    os.makedirs(dbg)
    out = subprocess.run([os.path.join(HOST, "driver"), "--real", "1000", "--json", js, "--debug", dbg, "--true-mean"],
                         capture_output=True, text=True)
    assert out.returncode == 1 and "The method failed" in out.stdout          # not converged, or a NaN/singular Jacobian
    r = json.load(open(js))
    p = oracle.edm_default_params(n_grid=512, n_real=2, mean_quirk=0)
    try:
        u, hist, it = _newton_on_oracle(p, Z0, 1e-4, 10, 1e-2)
        oracle_ok = bool(hist[-1] <= 1e-4)
    except np.linalg.LinAlgError:
        oracle_ok = False
    assert not oracle_ok and not r["converged"]
    if r["history"]:
        assert np.allclose(r["history"][:3], hist[:3], rtol=0, atol=5e-6)
    # debug taps (the reference's Save* dumps): one %f per line, sizes S*R / R / N / S
    n = lambda f: sum(1 for _ in open(os.path.join(dbg, f)))  # noqa: E731
    assert n("testAverages.dat") == 3000 and n("testAcceptFlag.dat") == 1000 and n("testLift.dat") == 512
    assert n("testAveraged.dat") == 3 and n("testLastSpikeTime.dat") == 3000 and n("test.dat") == 512


def _run_driver(tmp_path, name, *args):
    js = os.path.join(tmp_path, name + ".json")
    out = subprocess.run([os.path.join(HOST, "driver"), "--json", js, "--quiet", *args], capture_output=True, text=True)
    return out, json.load(open(js))


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_config5_per_gpu_share_newton(tmp_path):
    """BASELINE configs[4] at one GPU's share: the whole Newton solve of the Driver.cu problem (Driver.cu:15-37,59,
    69-71; NewtonSolver.cpp:40-197; tol 1e-4, maxIt 10, FD eps 1e-2, damping 1) with R = 125 000 realisations per
    residual evaluation, EXACT math, the reference's averaging (the driver's defaults).

    * 1024 grid points: same exit flag and iteration count as the oracle-driven Newton iteration on the residual of
      125 000 identical realisations, history within 5e-6, root within the Newton tolerance;
    * --dedup (one realisation evolved, replicated) must give the same JSON, field for field, apart from the wall time;
    * 512 grid points (Driver.cu:69 as shipped): the oracle iteration's exit flag and its first iterates."""
    _build()
    R = 125_000
    Z0 = [float(np.float32(0.3310)), float(np.float32(0.6914)), float(np.float32(1.3557))]
    out, r = _run_driver(tmp_path, "cfg5_1024", "--real", str(R), "--threads", "1024")
    assert r["n_real"] == R and r["n_grid"] == 1024 and r["math"] == "exact" and r["mean"] == "reference"
    p1 = oracle.edm_default_params(n_grid=1024, n_real=1)
    try:
        u, hist, it = _newton_on_oracle(p1, Z0, 1e-4, 10, 1e-2, n_real_reference_mean=R)
        ok = bool(hist[-1] <= 1e-4)
    except np.linalg.LinAlgError:
        u, hist, it, ok = None, [], 0, False
    assert r["converged"] == ok and (out.returncode == 0) == ok, (r, hist)
    f0, _ = _oracle_f(p1, np.array(Z0), n_real_reference_mean=R)
    assert np.allclose(r["f0_1024"], f0, rtol=0, atol=2e-7)
    if hist:
        assert r["iterations"] == it
        assert r["residual_evaluations"] == 1 + 4 * it
        assert np.allclose(r["history"], hist, rtol=0, atol=5e-6)
        assert np.allclose(r["solution"], u, rtol=0, atol=1e-4)
    # the opt-in shortcut for sigma = 0 must not change a single digit of the solve
    out_d, rd = _run_driver(tmp_path, "cfg5_1024_dedup", "--real", str(R), "--threads", "1024", "--dedup")
    assert out_d.returncode == out.returncode
    strip = lambda d: {k: v for k, v in d.items() if k != "solve_seconds"}  # noqa: E731
    assert strip(rd) == strip(r)
    # Driver.cu:69 as shipped: 512 grid points for the solve
    out5, r5 = _run_driver(tmp_path, "cfg5_512", "--real", str(R))
    assert r5["n_grid"] == 512 and r5["n_real"] == R
    p5 = oracle.edm_default_params(n_grid=512, n_real=1)
    try:
        u5, hist5, it5 = _newton_on_oracle(p5, Z0, 1e-4, 10, 1e-2, n_real_reference_mean=R)
        ok5 = bool(hist5[-1] <= 1e-4)
    except np.linalg.LinAlgError:
        u5, hist5, it5, ok5 = None, [], 0, False
    assert r5["converged"] == ok5 and (out5.returncode == 0) == ok5, (r5, hist5)
    if hist5 and r5["history"]:
        n = min(len(hist5), len(r5["history"]))
        assert np.allclose(r5["history"][:min(n, 3)], hist5[:min(n, 3)], rtol=0, atol=5e-6)
        if ok5:
            assert r5["iterations"] == it5 and np.allclose(r5["solution"], u5, rtol=0, atol=1e-4)
    print("config 5 at the 125k share: %d iterations, |F| = %.3g, %.2f s (dedup %.3f s); 512 points: converged=%s, %.2f s"
          % (r["iterations"], r["history"][-1] if r["history"] else float("nan"), r["solve_seconds"],
             rd["solve_seconds"], r5["converged"], r5["solve_seconds"]))
