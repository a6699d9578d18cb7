"""Property-based GPU parity: random grids (clustered, huge/tiny magnitudes, negative, near-degenerate spacings)
and random queries; libmi355interp must equal the oracle bit for bit in every table mode, and must never index
out of bounds whatever the guess arithmetic does (a faulting kernel can take the whole GPU down)."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import oracle

pytestmark = pytest.mark.gpu


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


grid_kind = st.sampled_from(["uniform", "linspace", "jitter", "cluster", "geometric", "huge", "tiny", "two", "plateau"])


def make_grid(kind, n, seed):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        X = -3.0 + 0.125 * np.arange(n)
    elif kind == "linspace":
        X = np.linspace(rng.uniform(-5, 0), rng.uniform(0.1, 7), n)
    elif kind == "jitter":
        X = (np.arange(n) + 0.9 * rng.random(n)) * rng.uniform(1e-3, 10)
    elif kind == "cluster":
        X = np.unique(np.sort(rng.random(n) ** 8 * rng.choice([1.0, 1e6, 1e-6])))
    elif kind == "geometric":
        X = np.unique(1e-12 * 1.05 ** np.arange(n))
    elif kind == "huge":
        X = np.unique(np.sort(rng.uniform(-1e307, 1e307, n)))
    elif kind == "tiny":
        X = np.unique(np.sort(rng.uniform(-1e-300, 1e-300, n)))
    elif kind == "two":
        X = np.array([rng.uniform(-1, 0), rng.uniform(0.5, 1)])
    else:   # nodes a few ulp apart
        base = rng.uniform(1, 2)
        X = np.unique(base + np.arange(n) * np.spacing(base) * rng.integers(1, 4))
    return X


@settings(max_examples=60, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(kind=grid_kind, n=st.integers(2, 3000), nq=st.integers(0, 5000), seed=st.integers(0, 2**31 - 1),
       extrap=st.sampled_from([float("nan"), -1.5, 0.0]))
def test_interp1_matches_oracle_on_arbitrary_grids(mi_ctx, kind, n, nq, seed, extrap):
    import armadillocudalinearinterpolation_amd as mi
    X = make_grid(kind, n, seed)
    if X.size < 2:
        return
    rng = np.random.default_rng(seed + 1)
    Y = rng.standard_normal(X.size) * rng.choice([1.0, 1e200, 1e-200])
    span = X[-1] - X[0]
    q = X[0] + rng.random(nq) * span if np.isfinite(span) else rng.choice(X, nq)
    if nq >= 8:
        q[:8] = [X[0], X[-1], np.nan, np.nextafter(X[0], -np.inf), np.nextafter(X[-1], np.inf), X[X.size // 2],
                 np.inf, -np.inf]
    if nq >= 40:
        q[8:40] = rng.choice(X, 32)                     # exactly on nodes
    ref = oracle.interp1_bracket(X, Y, q, extrap=extrap)
    grid = mi.Grid1.from_nodes(mi_ctx, X, Y, sanitise=False)
    got = grid.interp(_t(q), extrap=extrap).cpu().numpy()
    assert np.array_equal(got, ref, equal_nan=True), (kind, n, nq, seed, grid.info())


@settings(max_examples=25, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(nx=st.integers(2, 70), ny=st.integers(2, 70), nq=st.integers(0, 3000), seed=st.integers(0, 2**31 - 1),
       compact=st.booleans(), kx=st.sampled_from(["uniform", "jitter", "cluster"]), ky=st.sampled_from(["uniform", "jitter", "cluster"]))
def test_interp2_matches_oracle(mi_ctx, nx, ny, nq, seed, compact, kx, ky):
    import armadillocudalinearinterpolation_amd as mi
    xg, yg = make_grid(kx, nx, seed), make_grid(ky, ny, seed + 7)
    if xg.size < 2 or yg.size < 2:
        return
    rng = np.random.default_rng(seed + 3)
    Z = rng.standard_normal((yg.size, xg.size))
    xq = xg[0] + (rng.random(nq) * 1.1 - 0.05) * (xg[-1] - xg[0])
    yq = yg[0] + (rng.random(nq) * 1.1 - 0.05) * (yg[-1] - yg[0])
    if nq >= 4:
        xq[:4] = [xg[0], xg[-1], xg[-1], np.nan]
        yq[:4] = [yg[-1], yg[-1], yg[0], yg[0]]
    ref = oracle.interp2_bilinear(xg, yg, Z, xq, yq)
    got = mi.Grid2.from_axes(mi_ctx, xg, yg, Z, compact=compact).interp(_t(xq), _t(yq)).cpu().numpy()
    assert np.array_equal(got, ref, equal_nan=True), (nx, ny, nq, seed, compact, kx, ky)
