"""N > 1 path on CPU: world_size-2 gloo processes shard realisations / queries exactly as the GPU ranks do;
the oracle stands in for the per-rank device computation (no GPU here), the host logic under test is the
product's: shard_bounds, the all-reduce of partial sums, mi_edm_residual_from_sums (a host-only C-ABI call)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from armadillocudalinearinterpolation_amd import sharding

Z = [0.3310, 0.6914, 1.3557]
KW = dict(n_grid=512, n_real=6, beta_stddev=0.3, seed=424242, mean_quirk=0)      # the true mean: Newton stays finite at R = 6
KWQ = dict(KW, mean_quirk=1)                                                        # the reference's averaging (the default)


def test_shard_bounds_cover_exactly():
    for n in (0, 1, 7, 8, 100, 10**8 + 3):
        for world in (1, 2, 3, 8):
            b = [sharding.shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _finish(Zv, sc, kw=KW):
    import ctypes as C
    from armadillocudalinearinterpolation_amd import _lib, api
    L = _lib.load()
    p = api.default_edm_params(**kw)
    f = np.empty(3)
    Zv, sc = np.ascontiguousarray(Zv, dtype=np.float64), np.ascontiguousarray(sc, dtype=np.float64)
    _lib.check(L.mi_edm_residual_from_sums(C.byref(p), C.c_void_p(Zv.ctypes.data), C.c_void_p(sc.ctypes.data),
                                           C.c_void_p(f.ctypes.data)))
    return f


def _local(Zv, lo, hi, kw=KW):
    p = oracle.edm_default_params(**dict(kw, n_real=hi - lo, real_offset=lo))
    _, d = oracle.edm_compute_f(p, Zv)
    return d["sums"]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sr = sharding.ShardedResidual(KW["n_real"], _local, _finish)
        f = sr.ComputeF(Z)
        # the same with the reference's averaging: only the rank that holds realisation 0 leaves it out of its sums
        srq = sharding.ShardedResidual(KW["n_real"], lambda Zv, lo, hi: _local(Zv, lo, hi, KWQ), lambda Zv, sc: _finish(Zv, sc, KWQ))
        fq = srq.ComputeF(Z)
        # the replicated Newton loop over the sharded residual (config 5's structure), 2 iterations
        from armadillocudalinearinterpolation_amd import newton
        pars = newton.ParameterList(tolerance=1e-12, maxIterations=2, printOutput=False, finiteDifferenceEpsilon=1e-2)
        u, hist, conv, it = newton.NewtonSolver(sr, Z, pars).Solve()
        # query sharding + all-gather reassembly of an interp1 result
        X = np.linspace(0, 1, 257)
        Y = np.sin(5 * X)
        xi = oracle.splitmix_uniform(9, 1001)
        lo, hi = sharding.shard_bounds(xi.size, rank, world)
        local = torch.from_numpy(oracle.interp1_bracket(X, Y, xi[lo:hi]))
        sizes = [b - a for a, b in (sharding.shard_bounds(xi.size, r, world) for r in range(world))]
        full = sharding.all_gather_results(local, sizes).numpy()
        q.put((rank, f, full, u, hist, fq))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world2_gloo_residual_and_allgather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=90) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process reference over all realisations
    pf = oracle.edm_default_params(**KW)
    f_ref, d = oracle.edm_compute_f(pf, Z)
    X = np.linspace(0, 1, 257)
    whole = oracle.interp1_bracket(X, np.sin(5 * X), oracle.splitmix_uniform(9, 1001))
    # single-process Newton on the unsharded oracle residual
    from armadillocudalinearinterpolation_amd import newton

    class Whole:
        def ComputeF(self, Zv):
            return oracle.edm_compute_f(pf, Zv)[0]
    pars = newton.ParameterList(tolerance=1e-12, maxIterations=2, printOutput=False, finiteDifferenceEpsilon=1e-2)
    u_ref, hist_ref, _, _ = newton.NewtonSolver(Whole(), Z, pars).Solve()
    fq_ref, dq = oracle.edm_compute_f(oracle.edm_default_params(**KWQ), Z)
    assert np.all(dq["accept"] == 1) and not np.allclose(fq_ref, f_ref, rtol=0, atol=1e-3)   # x_0 / R is missing from the mean
    for rank, f, full, u, hist, fq in out:
        assert np.allclose(f, f_ref, rtol=0, atol=2e-7), (rank, f, f_ref)
        assert np.allclose(fq, fq_ref, rtol=0, atol=2e-7), (rank, fq, fq_ref)
        assert np.array_equal(full, whole)
        assert np.allclose(u, u_ref, rtol=0, atol=1e-5) and np.allclose(hist, hist_ref, rtol=0, atol=1e-5)
    assert np.array_equal(out[0][1], out[1][1])          # every rank holds the same residual
