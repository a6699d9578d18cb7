"""Generate tests/golden/*.npz -- small golden vectors for the interpolation path.

The reference (kyle-wedgwood/ArmadilloCUDALinearInterpolation) holds no tests,
fixtures or golden outputs, cannot be built here and has
no Python to import, so nothing is generated FROM the reference.  The vectors
below are produced by the CPU oracle (oracle/) and, before being written,
cross-checked against independent implementations available offline:
  * numpy.interp                      (fp64, different blend formula: <= 4 ulp),
  * scipy RegularGridInterpolator     (bilinear),
  * mpmath at 50 digits               (exact value of the two/four-point blend).
These cross-checks are NOT the reference; they guard the oracle against
transcription errors.  Each fixture stores inputs (or the closed-form recipe +
SplitMix64 seed that regenerates them), expected outputs, and a SHA-256 of the
full expected output where only a sample is stored.

Run from the repo root:  python scripts/make_golden.py
"""
import hashlib
import os
import sys

import mpmath
import numpy as np
from scipy.interpolate import RegularGridInterpolator

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
mpmath.mp.dps = 50


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def mp_interp1(X, Y, q):
    """Exact linear interpolant at q (mpmath), bracket = largest l with X[l] <= q."""
    l = int(np.searchsorted(X, q, side="right") - 1)
    r = min(l + 1, len(X) - 1)
    if r == l:
        return mpmath.mpf(float(Y[l])), l, r
    xa, xb, ya, yb, qq = (mpmath.mpf(float(v)) for v in (X[l], X[r], Y[l], Y[r], q))
    w = (qq - xa) / (xb - xa)
    return (1 - w) * ya + w * yb, l, r


def check_mp(X, Y, q, got, what):
    worst = 0.0
    for k in range(len(q)):
        if not (X[0] <= q[k] <= X[-1]):
            continue
        ex, l, r = mp_interp1(X, Y, q[k])
        scale = max(abs(float(ex)), abs(Y[l]), abs(Y[r]), 1e-300)
        worst = max(worst, abs(float(mpmath.mpf(float(got[k])) - ex)) / scale)
    assert worst <= 1e-15, (what, worst)
    return worst


def config1():
    """BASELINE config 1: NG=1e4 uniform grid on [0,1], Y=sin(2 pi X)+0.5 X; NQ=1e5."""
    ng, nq = 10_000, 100_000
    X = np.arange(ng) / (ng - 1)
    Y = np.sin(2 * np.pi * X) + 0.5 * X
    for tag, xi in (("uniform", np.arange(nq) / (nq - 1)), ("random", oracle.splitmix_uniform(0x5EED0001, nq))):
        ref = oracle.interp1_arma(X, Y, xi)
        assert np.array_equal(ref, oracle.interp1_bracket(X, Y, xi))
        npy = np.interp(xi, X, Y)
        assert np.max(np.abs(ref - npy) / np.maximum(np.abs(npy), 1.0)) < 1e-15
        idx = np.linspace(0, nq - 1, 4096).astype(np.int64)
        w = check_mp(X, Y, xi[idx[::16]], ref[idx[::16]], "config1-" + tag)
        np.savez_compressed(os.path.join(OUT, "interp1_config1_%s.npz" % tag), ng=ng, nq=nq,
                            seed=np.uint64(0x5EED0001), sample_idx=idx, sample_xi=xi[idx], sample_yi=ref[idx],
                            sha256_full=np.array(sha(ref)))
        print("config1", tag, "max rel err vs mpmath %.2e" % w, sha(ref)[:12])


def nonuniform():
    """512-node sorted non-uniform grid, 4096 random queries incl. out-of-range, nodes and NaN."""
    ng, nq = 512, 4096
    u = oracle.splitmix_uniform(0x5EED0002, ng)
    X = (np.arange(ng) + 0.5 * u) / ng
    Y = np.cos(7 * X) * np.exp(-X) + X * X
    xi = oracle.splitmix_uniform(0x5EED0012, nq) * 1.1 - 0.05
    xi[:6] = [X[0], X[-1], X[17], np.nan, X[0] - 1e-12, X[-1] + 1e-12]
    ref = oracle.interp1_arma(X, Y, xi)
    assert np.array_equal(ref, oracle.interp1_bracket(X, Y, xi), equal_nan=True)
    inside = (xi >= X[0]) & (xi <= X[-1])
    npy = np.interp(xi[inside], X, Y)
    assert np.max(np.abs(ref[inside] - npy)) < 1e-15
    check_mp(X, Y, xi[::8], ref[::8], "nonuniform")
    # unsorted / duplicated X exercises the sanitising front end
    perm = np.argsort(oracle.splitmix_uniform(99, ng))
    Xs = np.concatenate([X[perm], X[perm][:5]])
    Ys = np.concatenate([Y[perm], Y[perm][:5]])
    assert np.array_equal(oracle.interp1_arma(Xs, Ys, xi), ref, equal_nan=True)
    np.savez_compressed(os.path.join(OUT, "interp1_nonuniform.npz"), X=X, Y=Y, XI=xi, YI=ref,
                        X_shuffled_dup=Xs, Y_shuffled_dup=Ys)
    print("nonuniform ok")


def wild():
    """Strongly clustered grid (bucket-index mode of the HIP table)."""
    X = np.unique(np.sort(oracle.splitmix_uniform(77, 3000) ** 6))
    Y = np.cos(5 * X)
    xi = oracle.splitmix_uniform(78, 4096) * (X[-1] - X[0]) * 1.02 + X[0] - 0.01 * (X[-1] - X[0])
    ref = oracle.interp1_arma(X, Y, xi)
    assert np.array_equal(ref, oracle.interp1_bracket(X, Y, xi), equal_nan=True)
    check_mp(X, Y, xi[::8], ref[::8], "wild")
    np.savez_compressed(os.path.join(OUT, "interp1_clustered.npz"), X=X, Y=Y, XI=xi, YI=ref)
    print("clustered ok")


def bilinear():
    """64x48 bilinear case, scattered queries (BASELINE config 3 formula)."""
    nx, ny, nq = 64, 48, 4096
    xg = np.arange(nx) / (nx - 1)
    yg = np.arange(ny) / (ny - 1)
    Z = np.sin(2 * np.pi * yg)[:, None] * np.cos(2 * np.pi * xg)[None, :] + xg[None, :] * yg[:, None]
    q = oracle.splitmix_uniform(0x5EED0004, 2 * nq)
    xq, yq = q[:nq] * 1.04 - 0.02, q[nq:] * 1.04 - 0.02
    xq[:4] = [0.0, 1.0, xg[5], np.nan]
    yq[:4] = [1.0, 1.0, yg[7], 0.5]
    ref = oracle.interp2_bilinear(xg, yg, Z, xq, yq)
    inside = (xq >= 0) & (xq <= 1) & (yq >= 0) & (yq <= 1)
    rgi = RegularGridInterpolator((yg, xg), Z, method="linear")
    sc = rgi(np.stack([yq[inside], xq[inside]], axis=1))
    assert np.max(np.abs(sc - ref[inside])) < 5e-15, np.max(np.abs(sc - ref[inside]))
    assert np.all(np.isnan(ref[~inside]))
    np.savez_compressed(os.path.join(OUT, "interp2_bilinear.npz"), xg=xg, yg=yg, Z=Z, XQ=xq, YQ=yq, ZQ=ref)
    print("bilinear ok, max |oracle - scipy| = %.2e" % np.max(np.abs(sc - ref[inside])))


def restrict_case():
    """3 x 1000 Restrict + masked mean case (Driver.cu sizes: S=3, R=1000, N=1024, L=3, T=5)."""
    S, R, N = 3, 1000, 1024
    rng = np.random.default_rng(20260105)
    t0 = (rng.random(S * R) * 5).astype(np.float32)
    t1 = (5 + rng.random(S * R)).astype(np.float32)
    i0 = rng.integers(0, N, S * R).astype(np.uint16)
    i1 = np.minimum(i0 + rng.integers(0, 3, S * R), N - 1).astype(np.uint16)
    acc = (rng.random(R) < 0.9).astype(np.uint32)
    out = oracle.restrict_f32(t0, i0, t1, i1, 5.0, 3.0, N)
    # independent fp64 evaluation of the same formula: fp32 result within 4 ulp of it
    h = 6.0 / N
    x0, x1 = -3 + h * i0.astype(np.float64), -3 + h * i1.astype(np.float64)
    ex = x0 + (5.0 - t0.astype(np.float64)) * (x1 - x0) / (t1.astype(np.float64) - t0.astype(np.float64))
    assert np.max(np.abs(out - ex) / np.maximum(np.abs(ex), 1e-3)) < 1e-5
    mean, cnt = oracle.masked_mean_f32(out, acc, S)
    mean_q, _ = oracle.masked_mean_f32(out, acc, S, quirk=True)
    ex_mean = np.array([ex[m * R:(m + 1) * R][acc == 1].mean() for m in range(S)])
    assert np.max(np.abs(mean - ex_mean)) < 1e-5
    np.savez_compressed(os.path.join(OUT, "restrict_3x1000.npz"), t0=t0, i0=i0, t1=t1, i1=i1, accept=acc,
                        T=np.float32(5), L=np.float32(3), N=np.uint32(N), out=out, mean=mean, mean_quirk=mean_q,
                        count=np.uint32(cnt))
    print("restrict ok; KAT:", oracle.restrict_f32([4], [512], [6], [514], 5.0, 3.0, 1024))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    config1()
    nonuniform()
    wild()
    bilinear()
    restrict_case()
    print("fixtures written to", OUT)
