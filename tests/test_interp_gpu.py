"""GPU parity suite for the interpolation hot path: libmi355interp.so (through the C ABI) against the CPU
oracle on the same inputs, against the committed golden vectors, and -- at BASELINE.json's full size --
through size-independent properties.  Bar: fp64 results BIT-EXACT (same formula, no FMA contraction on
either side), which is stricter than the 1e-12 relative tolerance north_star states."""
import hashlib
import os

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _t(a):
    import torch
    if a.dtype == np.uint16:
        a = a.view(np.int16)
    elif a.dtype == np.uint32:
        a = a.view(np.int32)
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _eq(a, b):
    return np.array_equal(a, b, equal_nan=True)


# ---------------------------------------------------------------- golden vectors
@pytest.mark.parametrize("tag", ["uniform", "random"])
def test_config1_golden(mi_ctx, golden_dir, tag):
    import armadillocudalinearinterpolation_amd as mi
    g = _load(golden_dir, "interp1_config1_%s.npz" % tag)
    ng, nq = int(g["ng"]), int(g["nq"])
    X = np.arange(ng) / (ng - 1)
    Y = np.sin(2 * np.pi * X) + 0.5 * X
    xi = np.arange(nq) / (nq - 1) if tag == "uniform" else oracle.splitmix_uniform(int(g["seed"]), nq)
    grid = mi.Grid1.from_nodes(mi_ctx, X, Y)
    out = grid.interp(_t(xi)).cpu().numpy()
    assert _eq(out[g["sample_idx"]], g["sample_yi"])
    assert hashlib.sha256(out.tobytes()).hexdigest() == str(g["sha256_full"])
    # host convenience entry points (what the arma::vec wrapper calls)
    assert _eq(grid.interp_host(xi), out)
    assert _eq(mi.interp1(mi_ctx, X, Y, xi), out)


@pytest.mark.parametrize("name,mode", [("interp1_nonuniform.npz", 3), ("interp1_clustered.npz", 2)])
def test_general_grid_golden(mi_ctx, golden_dir, name, mode):
    import armadillocudalinearinterpolation_amd as mi
    g = _load(golden_dir, name)
    grid = mi.Grid1.from_nodes(mi_ctx, g["X"], g["Y"])
    assert grid.info()["mode"] == mode
    assert _eq(grid.interp(_t(g["XI"])).cpu().numpy(), g["YI"])
    assert _eq(grid.interp(_t(g["XI"]), extrap=-7.5).cpu().numpy(),
               oracle.interp1_bracket(g["X"], g["Y"], g["XI"], extrap=-7.5))
    if "X_shuffled_dup" in g.files:
        assert _eq(mi.interp1(mi_ctx, g["X_shuffled_dup"], g["Y_shuffled_dup"], g["XI"]), g["YI"])


def test_bilinear_golden(mi_ctx, golden_dir):
    import armadillocudalinearinterpolation_amd as mi
    g = _load(golden_dir, "interp2_bilinear.npz")
    grid = mi.Grid2.from_axes(mi_ctx, g["xg"], g["yg"], g["Z"])
    assert _eq(grid.interp(_t(g["XQ"]), _t(g["YQ"])).cpu().numpy(), g["ZQ"])
    compact = mi.Grid2.from_axes(mi_ctx, g["xg"], g["yg"], g["Z"], compact=True)      # column-pair layout
    assert _eq(compact.interp(_t(g["XQ"]), _t(g["YQ"])).cpu().numpy(), g["ZQ"])
    assert _eq(grid.interp_host(g["XQ"], g["YQ"]), g["ZQ"])
    nx, ny = g["xg"].size, g["yg"].size
    gu = mi.Grid2.uniform(mi_ctx, 0.0, 1.0 / (nx - 1), nx, 0.0, 1.0 / (ny - 1), ny, g["Z"])
    ref = oracle.interp2_bilinear_uniform(0.0, 1.0 / (nx - 1), nx, 0.0, 1.0 / (ny - 1), ny, g["Z"], g["XQ"], g["YQ"])
    assert _eq(gu.interp(_t(g["XQ"]), _t(g["YQ"])).cpu().numpy(), ref)


def test_restrict_and_mean_golden(mi_ctx, golden_dir):
    import armadillocudalinearinterpolation_amd as mi
    g = _load(golden_dir, "restrict_3x1000.npz")
    T, L, N = float(g["T"]), float(g["L"]), int(g["N"])
    t0, i0, t1, i1, acc = _t(g["t0"]), _t(g["i0"]), _t(g["t1"]), _t(g["i1"]), _t(g["accept"])
    out = mi.restrict(mi_ctx, t0, i0, t1, i1, T, L, N)
    assert np.array_equal(out.cpu().numpy(), g["out"])                       # bit-exact fp32
    for quirk, key in ((False, "mean"), (True, "mean_quirk")):
        m, c = mi.masked_mean(mi_ctx, out, acc, 3, quirk=quirk)
        f = mi.restrict_mean(mi_ctx, t0, i0, t1, i1, acc, T, L, N, 3, quirk=quirk, want_restricted=True, want_sums=True)
        for got in (m.cpu().numpy(), f["mean"].cpu().numpy()):
            # fp64 partial sums are added in a different order than the oracle's index order:
            # tolerance 1 ulp of fp32 (see oracle/interp_oracle.c orc_masked_mean_f32)
            assert np.all(np.abs(got - g[key]) <= np.spacing(np.abs(g[key]).astype(np.float32)))
        assert int(c.item()) == int(g["count"]) == int(f["count"].item())
        assert np.array_equal(f["restricted"].cpu().numpy(), g["out"])
        blk = f["sums"].cpu().numpy()                                        # partial block [sums | count | x0]
        assert blk.shape == (7,) and blk[3] == int(g["count"])
        x0 = blk[4:] if quirk else 0.0                                       # realisation 0 travels separately
        tot = blk[:3] + (x0 if (quirk and int(g["count"]) == 1) else 0.0)
        assert np.allclose(tot / int(g["count"]), g[key], rtol=1e-6)
        assert np.array_equal(blk[4:], g["out"].reshape(3, -1)[:, 0].astype(np.float64) if quirk else np.zeros(3))
    # in-place form of the reference (out aliases lastSpikeTime, EventDrivenMap.cu:783)
    t0c = t0.clone()
    mi.restrict(mi_ctx, t0c, i0, t1, i1, T, L, N, out=t0c)
    assert np.array_equal(t0c.cpu().numpy(), g["out"])
    # SURVEY 8c known answer, exact in fp32
    kat = mi.restrict(mi_ctx, _t(np.float32([4])), _t(np.uint16([512])), _t(np.float32([6])), _t(np.uint16([514])),
                      5.0, 3.0, 1024)
    assert kat.cpu().numpy()[0] == np.float32(0.005859375)


# ---------------------------------------------------------------- seeded parity vs the oracle
@pytest.mark.parametrize("nq", [0, 1, 2, 3, 255, 256, 257, 1023, 100003])
def test_ragged_sizes_all_modes(mi_ctx, nq):
    import armadillocudalinearinterpolation_amd as mi
    import torch
    ng = 4097
    u = oracle.splitmix_uniform(1234, ng)
    grids = {
        "uniformX": (np.arange(ng) / (ng - 1), 0),            # closed form detected -> implicit table
        "jitterX": ((np.arange(ng) + 0.5 * u) / ng, 3),
        "clustered": (np.unique(np.sort(u ** 5)), 2),
    }
    q = oracle.splitmix_uniform(4321 + nq, nq) * 1.1 - 0.05
    if nq > 8:
        q[:4] = [np.nan, -1.0, 2.0, 0.0]
    for name, (X, mode) in grids.items():
        Y = np.sin(9 * X) + X
        grid = mi.Grid1.from_nodes(mi_ctx, X, Y)
        assert grid.info()["mode"] == mode, name
        ref = oracle.interp1_bracket(X, Y, q)
        assert _eq(grid.interp(_t(q)).cpu().numpy(), ref), name
        if nq > 3:   # misaligned (8-byte but not 16-byte aligned) query / result pointers -> scalar kernel
            buf = _t(np.concatenate([[0.0], q]))
            out = torch.zeros(nq + 1, dtype=torch.float64, device="cuda:0")
            grid.interp(buf[1:], out=out[1:])
            assert _eq(out[1:].cpu().numpy(), ref), name
    # implicit uniform table
    ng2, x0, dx = 3000, -1.25, 1.0 / 2999
    Y = np.cos(np.arange(ng2) * 0.01)
    gu = mi.Grid1.uniform(mi_ctx, x0, dx, Y)
    qu = q * 1.2 + x0
    assert _eq(gu.interp(_t(qu)).cpu().numpy(), oracle.interp1_uniform(x0, dx, Y, qu))


def test_closed_form_detection_is_bit_exact(mi_ctx):
    """Explicit grids that a closed form reproduces bit for bit are stored as Y only (mode 0); the result must
    still equal the oracle evaluated on the EXPLICIT abscissae, everywhere including one ulp around every node."""
    import armadillocudalinearinterpolation_amd as mi
    n = 20001
    grids = {
        "i/(n-1)": np.arange(n) / (n - 1),
        "linspace": np.linspace(-2.5, 7.25, n),                 # last node pinned to the stop value
        "x0+i*dx": 0.1 + 0.3 * np.arange(n),
        "fma-like": -1.0 + 2.0 ** -7 * np.arange(n),
        "span*(i/den)": 3.0 + 5.0 * (np.arange(n) / (n - 1)),
    }
    for name, X in grids.items():
        Y = np.sin(X) + 0.01 * X * X
        g = mi.Grid1.from_nodes(mi_ctx, X, Y)
        assert g.info()["mode"] == 0, name
        assert g.info()["table_bytes"] == 8 * (n + 1)
        q = np.concatenate([X, np.nextafter(X, np.inf), np.nextafter(X, -np.inf), 0.5 * (X[1:] + X[:-1]),
                            oracle.splitmix_uniform(5, 50000) * (X[-1] - X[0]) * 1.01 + X[0]])
        assert _eq(g.interp(_t(q)).cpu().numpy(), oracle.interp1_bracket(X, Y, q)), name
    # one perturbed node breaks the closed form -> explicit table (mode 1), still exact
    X = np.arange(n) / (n - 1)
    X[777] = np.nextafter(X[777], 1.0)
    g = mi.Grid1.from_nodes(mi_ctx, X, np.cos(X))
    assert g.info()["mode"] == 3
    q = oracle.splitmix_uniform(6, 50000)
    assert _eq(g.interp(_t(q)).cpu().numpy(), oracle.interp1_bracket(X, np.cos(X), q))


def test_explicit_grids_with_analytic_guess_are_bit_exact_around_every_node(mi_ctx):
    """Explicit {x,y} tables with an analytic guess come in two forms: mode 3, the centred guess (grid within one cell
    of a straight line: bracket = G-1 or G, no walk) and mode 1, the generic guess + bounded walk.  Both must equal the oracle on every node,
    one ulp either side of every node, cell midpoints and random queries -- the places where an off-by-one bracket
    would show."""
    import armadillocudalinearinterpolation_amd as mi
    n = 30011
    i = np.arange(n, dtype=np.float64)
    u = oracle.splitmix_uniform(99, n)
    grids = {
        "jitter_half_cell": ((i + 0.5 * u) / n, 3),                      # BASELINE's non-uniform variant
        "jitter_0.9_cell": ((i + 0.9 * u) / n, 3),                       # still within one cell of a line
        "one_perturbed_node": (np.where(i == 777, np.nextafter(i / (n - 1), 1.0), i / (n - 1)), 3),
        "stretched": ((i / (n - 1)) ** 1.00002 * 3.0 - 1.0, 3),          # slow drift of about a quarter cell
        "jitter_1.5_cells": (np.unique(np.sort((i + 1.5 * u) / n)), 1),  # more than a cell: generic guess + walk
        "offset_scale": (1e6 + 1e-3 * (i + 0.3 * np.sin(i)), 3),         # large offset, small spacing
    }
    for name, (X, want_mode) in grids.items():
        X = np.ascontiguousarray(X)
        Y = np.sin(5 * (X - X[0]) / (X[-1] - X[0])) + 0.25 * X
        g = mi.Grid1.from_nodes(mi_ctx, X, Y, sanitise=False)
        assert g.info()["mode"] == want_mode, name
        q = np.concatenate([X, np.nextafter(X, np.inf), np.nextafter(X, -np.inf), 0.5 * (X[1:] + X[:-1]),
                            oracle.splitmix_uniform(7, 200000) * (X[-1] - X[0]) * 1.01 + X[0] - 0.005 * (X[-1] - X[0]),
                            [np.nan, -np.inf, np.inf, X[0], X[-1]]])
        assert _eq(g.interp(_t(q)).cpu().numpy(), oracle.interp1_bracket(X, Y, q)), name


@pytest.mark.parametrize("n,kind", [(3, "closed"), (1000, "closed"), (4096, "closed"), (10_000, "closed"),
                                    (16_383, "closed"), (16_384, "closed"), (2049, "jitter"), (5000, "jitter"),
                                    (8191, "jitter"), (8192, "jitter")])
def test_small_table_lds_path_equals_streaming_path(mi_ctx, n, kind):
    """Tables of at most 16 K nodes (closed-form abscissae) are copied into LDS by each workgroup when the query
    set is large and not declared ordered; the result must be bit-identical to the streaming kernel (ORDERED hint)
    and to the oracle, ragged/odd sizes, NaN and out-of-range queries included.  16 384 nodes + the padding node is
    one entry more than LDS holds and must fall back to the streaming kernel.  The same for {x,y} tables with a
    centred guess (mode 3, at most 8191 nodes)."""
    import armadillocudalinearinterpolation_amd as mi
    import torch
    if kind == "closed":
        X = np.linspace(-1.5, 2.25, n)
    else:
        X = -1.5 + 3.75 * (np.arange(n) + 0.5 * oracle.splitmix_uniform(n, n)) / n
    Y = np.cos(3 * X) + 0.1 * X
    grid = mi.Grid1.from_nodes(mi_ctx, X, Y, sanitise=False)
    assert grid.info()["mode"] == (0 if kind == "closed" else 3)
    for nq in (8 * (1 << 17), 9 * (1 << 17) + 4097, 8 * (1 << 17) + 1):
        g = torch.Generator(device="cuda:0").manual_seed(n + nq)
        xq = torch.rand(nq, dtype=torch.float64, device="cuda:0", generator=g) * 3.9 - 1.6
        xq[:5] = torch.tensor([float("nan"), -7.0, 7.0, float(X[0]), float(X[-1])], dtype=torch.float64)
        xq[-2:] = torch.tensor([float(X[-1]), float("nan")], dtype=torch.float64)
        try:
            mi_ctx.set_query_order(1)                                  # unordered: LDS-table kernel where it applies
            a = grid.interp(xq, extrap=4.5)
            mi_ctx.set_query_order(2)                                  # streaming kernel
            b = grid.interp(xq, extrap=4.5)
            mi_ctx.set_query_order(0)                                  # auto, twice: first call, then predicted
            c = grid.interp(xq, extrap=4.5)
            mi_ctx.synchronize()
            d = grid.interp(xq, extrap=4.5)
        finally:
            mi_ctx.set_query_order(0)
        for other in (b, c, d):
            assert torch.equal(a.view(torch.int64), other.view(torch.int64))
        idx = torch.cat([torch.arange(0, 3000, device="cuda:0"), torch.arange(0, nq, 1009, device="cuda:0"),
                         torch.arange(nq - 3000, nq, device="cuda:0")])
        ref = oracle.interp1_bracket(X, Y, xq[idx].cpu().numpy(), extrap=4.5)
        assert np.array_equal(a[idx].cpu().numpy(), ref, equal_nan=True)


def test_chunked_host_paths_equal_device_paths(mi_ctx):
    """Host convenience entry points (what the arma::vec / arma::mat wrappers call) pipeline large calls in chunks of
    8 M queries with the caller's arrays pinned for the duration of the call; results must equal the device entry
    points, for a ragged last chunk, for a reused and an in-place result array."""
    import armadillocudalinearinterpolation_amd as mi
    import torch
    nq = 2 * (8 << 20) + 12345
    rng = np.random.default_rng(3)
    X = np.linspace(0.0, 1.0, 50_001)
    g1 = mi.Grid1.from_nodes(mi_ctx, X, np.sin(9 * X))
    q = rng.random(nq) * 1.02 - 0.01
    q[:3] = [np.nan, 0.0, 1.0]
    dev = g1.interp(_t(q)).cpu().numpy()
    assert _eq(g1.interp_host(q), dev)
    buf = np.empty_like(q)
    assert g1.interp_host(q, out=buf) is buf and _eq(buf, dev)
    inplace = q.copy()
    g1.interp_host(inplace, out=inplace)
    assert _eq(inplace, dev)
    nx, ny = 300, 200
    Z = rng.standard_normal((ny, nx))
    g2 = mi.Grid2.uniform(mi_ctx, 0.0, 1.0 / (nx - 1), nx, 0.0, 1.0 / (ny - 1), ny, Z)
    yq = rng.random(nq)
    dev2 = g2.interp(_t(q), _t(yq)).cpu().numpy()
    assert _eq(g2.interp_host(q, yq), dev2)
    assert _eq(g2.interp_host(q, q), g2.interp(_t(q), _t(q)).cpu().numpy())     # both coordinate arrays are one array


def test_queries_on_nodes_and_cell_midpoints(mi_ctx):
    import armadillocudalinearinterpolation_amd as mi
    rng = np.random.default_rng(8)
    X = np.cumsum(rng.random(5000) + 1e-6)
    Y = rng.standard_normal(5000)
    grid = mi.Grid1.from_nodes(mi_ctx, X, Y)
    assert np.array_equal(grid.interp(_t(X.copy())).cpu().numpy(), Y)            # exact at every node
    mid = 0.5 * (X[:-1] + X[1:])
    assert _eq(grid.interp(_t(mid)).cpu().numpy(), oracle.interp1_bracket(X, Y, mid))
    nxt = np.nextafter(X, np.inf)[:-1]                                            # one ulp right of each node
    prv = np.nextafter(X, -np.inf)[1:]
    for q in (nxt, prv):
        assert _eq(grid.interp(_t(q)).cpu().numpy(), oracle.interp1_bracket(X, Y, q))


def test_device_resident_table_and_hipgraph_capture(mi_ctx):
    """Table built from device pointers; the device entry point allocates nothing and never synchronises, so it
    can be captured into a HIP graph and replayed (mi355_interp.h conventions)."""
    import armadillocudalinearinterpolation_amd as mi
    import torch
    rng = np.random.default_rng(21)
    X = np.cumsum(rng.random(30000) + 0.01)
    Y = np.sin(X)
    grid = mi.Grid1.from_device_nodes(mi_ctx, _t(X), _t(Y))
    q = rng.random(200001) * (X[-1] - X[0]) + X[0]
    ref = oracle.interp1_bracket(X, Y, q)
    assert _eq(grid.interp(_t(q)).cpu().numpy(), ref)
    with pytest.raises(mi.MiError):
        mi.Grid1.from_device_nodes(mi_ctx, _t(X[::-1].copy()), _t(Y))          # not increasing
    # capture on a side stream, replay twice with fresh inputs in the same buffers
    xq = _t(q)
    out = torch.zeros_like(xq)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        mi_ctx.use_torch_stream()
        grid.interp(xq, out=out)                                               # warm-up outside capture
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            mi_ctx.use_torch_stream()
            grid.interp(xq, out=out)
    torch.cuda.current_stream().wait_stream(side)
    mi_ctx.use_torch_stream()
    for seed in (1, 2):
        q2 = np.random.default_rng(seed).random(q.size) * (X[-1] - X[0]) + X[0]
        xq.copy_(_t(q2))
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert _eq(out.cpu().numpy(), oracle.interp1_bracket(X, Y, q2))


def test_grid_validation_errors(mi_ctx):
    import armadillocudalinearinterpolation_amd as mi
    with pytest.raises(mi.MiError) as e:
        mi.Grid1.from_nodes(mi_ctx, [0.0, 2.0, 1.0], [0.0, 1.0, 2.0], sanitise=False)
    assert e.value.code == 2
    with pytest.raises(mi.MiError):
        mi.Grid1.from_nodes(mi_ctx, [1.0, 1.0], [0.0, 1.0])          # < 2 unique nodes
    with pytest.raises(mi.MiError):
        mi.Grid1.from_nodes(mi_ctx, [0.0, np.nan, 1.0], [0.0, 1.0, 2.0])
    with pytest.raises(mi.MiError):
        mi.Grid1.uniform(mi_ctx, 0.0, -1.0, [0.0, 1.0])
    with pytest.raises(ValueError):
        mi.Grid1.from_nodes(mi_ctx, [0.0, 1.0], [0.0, 1.0, 2.0])


def test_scattered_bilinear_seeded(mi_ctx):
    import armadillocudalinearinterpolation_amd as mi
    nx, ny, nq = 257, 129, 200001
    xg = np.cumsum(oracle.splitmix_uniform(1, nx) + 0.01)
    yg = np.cumsum(oracle.splitmix_uniform(2, ny) ** 3 + 1e-4)          # y axis: binary-search path
    Z = np.sin(xg)[None, :] * np.cos(3 * yg)[:, None]
    q = oracle.splitmix_uniform(3, 2 * nq)
    xq = q[:nq] * (xg[-1] - xg[0]) * 1.02 + xg[0] - 0.01
    yq = q[nq:] * (yg[-1] - yg[0]) * 1.02 + yg[0] - 0.001
    xq[:5] = [xg[0], xg[-1], xg[10], xg[-1], np.nan]
    yq[:5] = [yg[0], yg[-1], yg[-1], yg[3], yg[1]]
    grid = mi.Grid2.from_axes(mi_ctx, xg, yg, Z)
    assert _eq(grid.interp(_t(xq), _t(yq)).cpu().numpy(), oracle.interp2_bilinear(xg, yg, Z, xq, yq, nthreads=4))


@pytest.mark.parametrize("compact", [False, True])
@pytest.mark.parametrize("shape", [(257, 129), (64, 1500), (3000, 2), (2, 2)])
def test_bilinear_shapes_layouts_and_special_queries(mi_ctx, shape, compact):
    """Table shapes from 2 x 2 to 3000 x 2, both resident layouts (quad cells / column pairs), explicit axes within a third of
    a cell of a line (guess + walk) and uniform axes, with out-of-range / NaN / infinite / node-exact queries and an odd
    query count: every result bit-equal to the oracle.  (Inputs of the round-3 tests of the call-wide cell ordering, which
    left the product in round 4 -- scripts/exp_interp2_ordered.hpp.)"""
    import armadillocudalinearinterpolation_amd as mi
    ctx = mi_ctx
    nx, ny = shape
    nq = 5 * 4096 + 1237                                   # five tiles and a ragged tail
    rng = np.random.default_rng(nx * 1000 + ny)
    jx, jy = rng.random(nx) * 0.3, rng.random(ny) * 0.3
    xg = (np.arange(nx) + jx) / nx * 2.0 - 0.5             # explicit, within a third of a cell of a line: guess + walk
    yg = (np.arange(ny) + jy) / ny * 3.0 + 1.0
    Z = np.sin(3 * xg)[None, :] * np.cos(2 * yg)[:, None] + 0.1 * rng.random((ny, nx))
    xq = rng.random(nq) * (xg[-1] - xg[0]) * 1.04 + xg[0] - 0.02 * (xg[-1] - xg[0])
    yq = rng.random(nq) * (yg[-1] - yg[0]) * 1.04 + yg[0] - 0.02 * (yg[-1] - yg[0])
    xq[:6] = [xg[0], xg[-1], xg[nx // 2], xg[-1], np.nan, xg[1]]
    yq[:6] = [yg[0], yg[-1], yg[-1], yg[ny // 2], yg[1], np.inf]
    xq[4096:4096 + nx] = xg                                 # a pile of node-exact abscissae in the second tile
    ref = oracle.interp2_bilinear(xg, yg, Z, xq, yq, nthreads=4)
    grid = mi.Grid2.from_axes(ctx, xg, yg, Z, compact=compact)
    assert grid.info()["table_bytes"] == nx * ny * (16 if compact else 32)
    assert _eq(grid.interp(_t(xq), _t(yq)).cpu().numpy(), ref)
    assert _eq(grid.interp(_t(xq[1:]), _t(yq[1:])).cpu().numpy(), ref[1:])        # 8-byte aligned only: the scalar kernel
    grid.close()
    # uniform axes (implicit nodes)
    gu = mi.Grid2.uniform(ctx, -0.5, 2.0 / nx, nx, 1.0, 3.0 / ny, ny, Z, compact=compact)
    want = oracle.interp2_bilinear_uniform(-0.5, 2.0 / nx, nx, 1.0, 3.0 / ny, ny, Z, xq, yq, nthreads=4)
    assert _eq(gu.interp(_t(xq), _t(yq)).cpu().numpy(), want)
    gu.close()


def test_bilinear_skewed_query_sets(mi_ctx):
    """A 2048 x 1536 table (96 MiB of quad cells): queries piled into one corner, onto one grid line, and spread evenly
    must all come out as the oracle's."""
    import armadillocudalinearinterpolation_amd as mi
    ctx = mi_ctx
    nx, ny, nq = 1536, 2048, 40 * 4096
    rng = np.random.default_rng(7)
    Z = rng.random((ny, nx))
    g = mi.Grid2.uniform(ctx, 0.0, 1.0 / (nx - 1), nx, 0.0, 1.0 / (ny - 1), ny, Z)
    for kind in ("even", "corner", "line"):
        xq, yq = rng.random(nq), rng.random(nq)
        if kind == "corner":
            xq, yq = xq * 0.01, 1.0 - yq * 0.01
        elif kind == "line":
            xq[:] = 700.0 / (nx - 1)
        ref = oracle.interp2_bilinear_uniform(0.0, 1.0 / (nx - 1), nx, 0.0, 1.0 / (ny - 1), ny, Z, xq, yq, nthreads=8)
        assert _eq(g.interp(_t(xq), _t(yq)).cpu().numpy(), ref), kind
    g.close()


def test_bilinear_binary_search_axes(mi_ctx):
    """strongly non-uniform axes (no usable linear guess: binary search on both)"""
    import armadillocudalinearinterpolation_amd as mi
    nx, ny, nq = 257, 129, 3 * 4096
    xg = np.cumsum(oracle.splitmix_uniform(1, nx) + 0.01)
    yg = np.cumsum(oracle.splitmix_uniform(2, ny) ** 3 + 1e-4)
    Z = np.sin(xg)[None, :] * np.cos(3 * yg)[:, None]
    q = oracle.splitmix_uniform(3, 2 * nq)
    xq, yq = q[:nq] * (xg[-1] - xg[0]) + xg[0], q[nq:] * (yg[-1] - yg[0]) + yg[0]
    grid = mi.Grid2.from_axes(mi_ctx, xg, yg, Z)
    assert _eq(grid.interp(_t(xq), _t(yq)).cpu().numpy(), oracle.interp2_bilinear(xg, yg, Z, xq, yq, nthreads=4))
    grid.close()


# ---------------------------------------------------------------- BASELINE-size property tests
def test_config2_full_size_properties(mi_ctx):
    """1e8 random queries over a 1e6-node table (BASELINE.json configs[1]), the very query set bench.py times
    (SplitMix64 seed 0x5EED0003, generated on the device): properties + sampled parity."""
    import armadillocudalinearinterpolation_amd as mi
    from armadillocudalinearinterpolation_amd import synth
    import torch
    NG, NQ = 10**6, 10**8
    X = np.arange(NG) / (NG - 1)
    Y = np.sin(2 * np.pi * X) + 0.5 * X
    grid = mi.Grid1.from_nodes(mi_ctx, X, Y, sanitise=False)
    xq = synth.splitmix_uniform(0x5EED0003, NQ, torch.device("cuda", 0))
    assert np.array_equal(xq[:1000].cpu().numpy(), oracle.splitmix_uniform(0x5EED0003, 1000))   # the oracle's stream
    out = grid.interp(xq)
    # sampled bit parity with the oracle (2e5 queries spread over the vector)
    idx = torch.arange(0, NQ, NQ // 200000, device="cuda:0")
    ref = oracle.interp1_bracket(X, Y, xq[idx].cpu().numpy(), nthreads=8)
    assert np.array_equal(out[idx].cpu().numpy(), ref)
    # shard-and-concatenate equals whole (two unequal shards, second one starting at an odd offset)
    cut = 33_333_333
    out2 = torch.empty_like(out)
    grid.interp(xq[:cut].contiguous(), out=out2[:cut])
    grid.interp(xq[cut:].contiguous(), out=out2[cut:])
    assert torch.equal(out, out2)
    # monotone table on [0, 0.2] (Y' > 0 there): sorted queries give sorted results
    sub = torch.sort(xq[:10_000_000] * 0.2).values
    r = grid.interp(sub)
    assert bool((r[1:] >= r[:-1]).all())
    # affine table: reproduced to a few ulp at every one of the 1e8 points
    ga = mi.Grid1.from_nodes(mi_ctx, X, 3.0 * X - 1.0, sanitise=False)
    ra = ga.interp(xq)
    assert float((ra - (3.0 * xq - 1.0)).abs().max()) <= 1e-15
    # implicit-uniform table on a power-of-two step grid is bitwise equal to the explicit table
    NP = 2**20 + 1
    Xp = np.arange(NP) * 2.0 ** -20
    Yp = np.cos(5 * Xp)
    a = mi.Grid1.from_nodes(mi_ctx, Xp, Yp, sanitise=False).interp(xq)
    b = mi.Grid1.uniform(mi_ctx, 0.0, 2.0 ** -20, Yp).interp(xq)
    assert torch.equal(a, b)


@pytest.mark.parametrize("kind", ["closed_form", "jitter", "clustered"])
def test_region_sweep_path_equals_streaming_path(mi_ctx, kind):
    """Large unordered query sets over a table bigger than L2 take the region-sweep kernel (LDS ordering by table
    region); its output must be bit-identical to the streaming kernel's and to the oracle, in the caller's order,
    for every table mode, with a ragged tail, NaN / out-of-range queries, and whatever the order hint says."""
    import armadillocudalinearinterpolation_amd as mi
    import torch
    n = 1_000_000
    u = oracle.splitmix_uniform(0x5EED0002, n)
    X = {"closed_form": np.arange(n) / (n - 1), "jitter": (np.arange(n) + 0.5 * u) / n,
         "clustered": np.unique(np.sort(u ** 3))}[kind]
    Y = np.sin(7 * X) + X
    grid = mi.Grid1.from_nodes(mi_ctx, X, Y, sanitise=False)
    assert grid.info()["mode"] == {"closed_form": 0, "jitter": 3, "clustered": 2}[kind]
    nq = 256 * 8 * 8192 + 12345                                   # >= 8 tiles per CU, plus a ragged tail
    g = torch.Generator(device="cuda:0").manual_seed(77)
    xq = torch.rand(nq, dtype=torch.float64, device="cuda:0", generator=g) * 1.02 - 0.01
    xq[:4] = torch.tensor([float("nan"), -5.0, 5.0, float(X[0])], dtype=torch.float64)
    xq[-3:] = torch.tensor([float(X[-1]), float("nan"), 0.5], dtype=torch.float64)
    outs = {}
    try:
        for name, hint in (("auto", 0), ("random", 1), ("ordered", 2)):
            mi_ctx.set_query_order(hint)
            outs[name] = grid.interp(xq, extrap=-3.25)
    finally:
        mi_ctx.set_query_order(0)
    assert torch.equal(outs["random"].view(torch.int64), outs["ordered"].view(torch.int64))
    assert torch.equal(outs["auto"].view(torch.int64), outs["ordered"].view(torch.int64))
    # AUTO again: the first call had no verdict yet (both kernels launched, gated on the device flag); by now the
    # probe's verdict has reached the host and predicts the kernel (region sweep only)
    mi_ctx.synchronize()
    again = grid.interp(xq, extrap=-3.25)
    assert torch.equal(again.view(torch.int64), outs["ordered"].view(torch.int64))
    idx = torch.cat([torch.arange(0, 4096, device="cuda:0"), torch.arange(0, nq, 997, device="cuda:0"),
                     torch.arange(nq - 20000, nq, device="cuda:0")])
    ref = oracle.interp1_bracket(X, Y, xq[idx].cpu().numpy(), extrap=-3.25, nthreads=8)
    assert np.array_equal(outs["auto"][idx].cpu().numpy(), ref, equal_nan=True)
    # ordered input through AUTO: the first call is mispredicted from the random set above (region sweep on ordered
    # data), the second follows the new verdict (streaming kernel); both must equal the hinted paths
    xs = torch.sort(xq[4:-3]).values.contiguous()
    a1 = grid.interp(xs)
    mi_ctx.synchronize()
    a2 = grid.interp(xs)
    try:
        mi_ctx.set_query_order(1)                                   # force the region sort on ordered data
        b = grid.interp(xs)
        mi_ctx.set_query_order(2)                                   # streaming kernel
        c = grid.interp(xs)
        mi_ctx.set_query_order(0)                                   # hint change forgets the verdict: gated launch
        a3 = grid.interp(xs)
    finally:
        mi_ctx.set_query_order(0)
    for other in (a2, a3, b, c):
        assert torch.equal(a1.view(torch.int64), other.view(torch.int64))
    # hipGraph: capture the AUTO call (probe + whatever launch plan is in force), replay it on query sets of the
    # other kind in the same buffers -- the plan is frozen in the graph, the results must not care
    buf = xs.clone()
    out = torch.zeros_like(buf)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    try:
        with torch.cuda.stream(side):
            mi_ctx.use_torch_stream()
            grid.interp(buf, out=out)                                   # warm-up outside capture
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                mi_ctx.use_torch_stream()
                grid.interp(buf, out=out)
        torch.cuda.current_stream().wait_stream(side)
    finally:
        mi_ctx.use_torch_stream()
    for src, want in ((xq[4:-3], outs["ordered"][4:-3]), (xs, a1)):
        buf.copy_(src)
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        # same queries, default extrap (NaN) here vs -3.25 above: compare where the reference value is in range
        if want is a1:
            assert torch.equal(out.view(torch.int64), want.view(torch.int64))
        else:
            inside = (src >= float(X[0])) & (src <= float(X[-1]))
            assert torch.equal(out[inside].view(torch.int64), want[inside].view(torch.int64))
            assert bool(torch.isnan(out[~inside]).all())


def test_config3_full_grid_sampled(mi_ctx):
    """4096x4096 table (BASELINE.json configs[2]) at its full size: the 1e8 scattered queries bench.py times (SplitMix64
    seed 0x5EED0004: first half x, second half y), sampled parity (2e5 queries spread over the vector) + bilinear
    exactness at every one of the 1e8 points + both resident layouts bit-equal."""
    import armadillocudalinearinterpolation_amd as mi
    from armadillocudalinearinterpolation_amd import synth
    import torch
    n = 4096
    ax = np.arange(n) / (n - 1)
    Z = np.sin(2 * np.pi * ax)[:, None] * np.cos(2 * np.pi * ax)[None, :] + ax[None, :] * ax[:, None]
    grid = mi.Grid2.uniform(mi_ctx, 0.0, 1.0 / (n - 1), n, 0.0, 1.0 / (n - 1), n, Z)
    NQ = 100_000_000
    q2 = synth.splitmix_uniform(0x5EED0004, 2 * NQ, torch.device("cuda", 0))
    xq, yq = q2[:NQ], q2[NQ:]
    assert grid.info()["table_bytes"] == 32 * n * n
    out = grid.interp(xq, yq)
    compact = mi.Grid2.uniform(mi_ctx, 0.0, 1.0 / (n - 1), n, 0.0, 1.0 / (n - 1), n, Z, compact=True)
    assert torch.equal(compact.interp(xq, yq), out)
    compact.close()
    idx = torch.arange(0, NQ, 500, device="cuda:0")
    ref = oracle.interp2_bilinear_uniform(0.0, 1.0 / (n - 1), n, 0.0, 1.0 / (n - 1), n, Z,
                                          xq[idx].cpu().numpy(), yq[idx].cpu().numpy(), nthreads=8)
    assert np.array_equal(out[idx].cpu().numpy(), ref)
    # a bilinear function is reproduced to rounding everywhere
    Zb = 2.0 * ax[None, :] - ax[:, None] + 0.5 * ax[None, :] * ax[:, None] + 1.0
    gb = mi.Grid2.uniform(mi_ctx, 0.0, 1.0 / (n - 1), n, 0.0, 1.0 / (n - 1), n, Zb)
    rb = gb.interp(xq, yq)
    assert float((rb - (2.0 * xq - yq + 0.5 * xq * yq + 1.0)).abs().max()) < 1e-14


def test_restrict_mean_cfg4_size(mi_ctx):
    """S=3 x R=1e6 (BASELINE config 4's interpolation step): fused kernel == separate kernels == oracle."""
    import armadillocudalinearinterpolation_amd as mi
    S, R, N = 3, 1_000_000, 1024
    rng = np.random.default_rng(4)
    t0 = (rng.random(S * R) * 5).astype(np.float32)
    t1 = (5 + rng.random(S * R)).astype(np.float32)
    i0 = rng.integers(0, N - 2, S * R).astype(np.uint16)
    i1 = (i0 + 1).astype(np.uint16)
    acc = (rng.random(R) < 0.97).astype(np.uint32)
    ref = oracle.restrict_f32(t0, i0, t1, i1, 5.0, 3.0, N)
    mref, cref = oracle.masked_mean_f32(ref, acc, S)
    d = [_t(a) for a in (t0, i0, t1, i1, acc)]
    out = mi.restrict(mi_ctx, d[0], d[1], d[2], d[3], 5.0, 3.0, N)
    assert np.array_equal(out.cpu().numpy(), ref)
    f = mi.restrict_mean(mi_ctx, *d, 5.0, 3.0, N, S, want_restricted=True)
    assert np.array_equal(f["restricted"].cpu().numpy(), ref)
    assert int(f["count"].item()) == cref
    assert np.all(np.abs(f["mean"].cpu().numpy() - mref) <= np.spacing(np.abs(mref)))
    # run-to-run reproducibility (no float atomics)
    f2 = mi.restrict_mean(mi_ctx, *d, 5.0, 3.0, N, S)
    assert np.array_equal(f["mean"].cpu().numpy(), f2["mean"].cpu().numpy())


def test_host_path_error_leaves_nothing_pinned(mi_ctx, monkeypatch):
    """a failing HIP call inside the chunked, pinned host path must fall through to the stream drain and
    the un-pinning (MI_TEST_FAIL_HOST_CHUNK injects the failure); afterwards no caller range stays registered, the
    same arrays work again, and two calls sharing one query array keep it pinned until the last one is done."""
    import armadillocudalinearinterpolation_amd as mi
    from armadillocudalinearinterpolation_amd import _lib
    L = _lib.load()
    ng, nq = 1000, (17 << 20) + 5                          # > 2 chunks of 8 M: the pinned, pipelined path
    X = np.arange(ng) / (ng - 1)
    Y = np.cos(3 * X)
    grid = mi.Grid1.from_nodes(mi_ctx, X, Y)
    xi = oracle.splitmix_uniform(77, nq)
    assert L.mi_debug_pinned_ranges() == 0
    monkeypatch.setenv("MI_TEST_FAIL_HOST_CHUNK", "1")
    with pytest.raises(mi.MiError) as e:
        grid.interp_host(xi)
    assert "MI_TEST_FAIL_HOST_CHUNK" in str(e.value) and L.mi_debug_pinned_ranges() == 0
    g2 = mi.Grid2.uniform(mi_ctx, 0.0, 0.5, 3, 0.0, 0.5, 3, np.arange(9.0).reshape(3, 3))
    with pytest.raises(mi.MiError):
        g2.interp_host(xi, xi)
    assert L.mi_debug_pinned_ranges() == 0
    monkeypatch.delenv("MI_TEST_FAIL_HOST_CHUNK")
    got = grid.interp_host(xi)
    assert L.mi_debug_pinned_ranges() == 0
    idx = np.arange(0, nq, 4099)
    assert np.array_equal(got[idx], oracle.interp1_bracket(X, Y, xi[idx]))


def test_both_sweep_kernel_forms_are_bit_identical(tmp_path):
    """The region sweep has two forms (MI_SWEEP_VARIANT, read once per process): the pipelined two-group kernel (default)
    and the one-phase-after-the-other kernel.  Each runs in a process of its own on the same seeded inputs (closed-form
    and {x,y} tables, ragged tail) and must reproduce the streaming kernel -- hence each other -- bit for bit."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = r"""
import sys, numpy as np, torch, hashlib
sys.path.insert(0, %r)
import armadillocudalinearinterpolation_amd as mi
from armadillocudalinearinterpolation_amd import synth
ctx = mi.Context(0)
out = []
for kind in ("closed", "jitter"):
    ng = 1_000_000
    X = np.arange(ng) / (ng - 1)
    if kind == "jitter":
        X = (np.arange(ng) + 0.5 * synth.splitmix_uniform(7, ng, torch.device("cpu")).numpy()) / ng
    Y = np.sin(2 * np.pi * X) + 0.5 * X
    grid = mi.Grid1.from_nodes(ctx, X, Y, sanitise=False)
    xq = synth.splitmix_uniform(11, 20_000_000 + 12_345, torch.device("cuda", 0)) * 1.02 - 0.01
    ctx.set_query_order(1)                      # unordered: region sweep
    a = grid.interp(xq)
    ctx.set_query_order(2)                      # ordered: streaming kernel
    b = grid.interp(xq)
    assert torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)), kind
    out.append(hashlib.sha256(a.cpu().numpy().tobytes()).hexdigest())
print("DIGEST", " ".join(out))
""" % root
    digests = []
    for variant in ("1", "2"):
        env = dict(os.environ, MI_SWEEP_VARIANT=variant)
        r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        digests.append([ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][-1])
    assert digests[0] == digests[1]
