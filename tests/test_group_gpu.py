"""Multi-GPU behind the C ABI (mi_group_*) on the one-GPU box: a group of ONE device exercises the RCCL binding
(ncclCommInitAll over 1 rank, all-gather), groups that name GPU 0 two or three times exercise the shard arithmetic, the
concurrent per-shard pipelines, the device-to-device gather and the summation of the partial blocks.  Every result must
equal the single-context path (bit for bit where the arithmetic is exact) and the oracle."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

Z = [0.3310, 0.6914, 1.3557]


def _table():
    ng = 50_001
    X = np.arange(ng) / (ng - 1)
    return X, np.sin(2 * np.pi * X) + 0.5 * X


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_group_interp1_host_and_device_shards(mi_ctx, devices):
    import torch
    import armadillocudalinearinterpolation_amd as mi
    X, Y = _table()
    nq = 300_007                                           # ragged: shards differ in size
    xi = oracle.splitmix_uniform(5, nq) * 1.1 - 0.05       # some out of range
    ref = oracle.interp1_arma(X, Y, xi)
    grp = mi.Group(devices)
    assert len(grp) == len(devices)
    tab = grp.grid1(X, Y)
    assert np.array_equal(tab.interp_host(xi), ref, equal_nan=True)
    assert np.array_equal(tab.interp_host(xi, extrap=-2.0), oracle.interp1_arma(X, Y, xi, extrap=-2.0))
    # device-resident equal shards + gather of the whole result on every member
    P = len(devices)
    n = 65_536 + 2
    shards = [torch.from_numpy(xi[r * n:(r + 1) * n].copy()).cuda() for r in range(P)]
    outs, full = tab.interp_dev(shards, gather=True)
    for r in range(P):
        assert np.array_equal(outs[r].cpu().numpy(), ref[r * n:(r + 1) * n], equal_nan=True)
        assert np.array_equal(full[r].cpu().numpy(), ref[:P * n], equal_nan=True)       # every member holds every shard
    tab.close()
    grp.close()


@pytest.mark.parametrize("chunks", [2, 4, 7])
@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
def test_chunked_gather_equals_the_unchunked_call(mi_ctx, devices, chunks):
    """mi_group_set_gather_chunks: the all-gather hidden behind the kernels (chunk k exchanged on a second stream -- grouped
    ncclBroadcast on the one-rank communicator of devices [0], device-to-device copies for the repeated device -- while the
    kernel of chunk k+1 runs) must leave every member with exactly the bytes of the unchunked call, for shard sizes that
    do and do not divide by the chunk count, in place and out of place, call after call on the same buffers."""
    import torch
    import armadillocudalinearinterpolation_amd as mi
    X, Y = _table()
    P = len(devices)
    grp = mi.Group(devices)
    tab = grp.grid1(X, Y)
    for n in (65_536 + 2, 1_000_001 if P == 1 else 300_001, 4 * chunks):
        xi = oracle.splitmix_uniform(11 + n, P * n) * 1.1 - 0.05
        ref = oracle.interp1_arma(X, Y, xi)
        shards = [torch.from_numpy(xi[r * n:(r + 1) * n].copy()).cuda() for r in range(P)]
        grp.set_gather_chunks(1)
        outs0, full0 = tab.interp_dev(shards, gather=True)
        grp.set_gather_chunks(chunks)
        for rep in range(2):                               # twice: the second call reuses streams, events and buffers
            outs, full = tab.interp_dev(shards, gather=True)
            for r in range(P):
                assert torch.equal(outs[r].view(torch.int64), outs0[r].view(torch.int64))
                assert torch.equal(full[r].view(torch.int64), full0[r].view(torch.int64))
                assert np.array_equal(full[r].cpu().numpy(), ref, equal_nan=True)
        # in place: every member's shard already lives inside its gathered vector
        fulls = [torch.full((P * n,), -7.0, dtype=torch.float64, device="cuda") for _ in range(P)]
        inplace = [fulls[r][r * n:(r + 1) * n] for r in range(P)]
        tab.interp_dev(shards, out=inplace, gather=True, gathered=fulls)
        for r in range(P):
            assert np.array_equal(fulls[r].cpu().numpy(), ref, equal_nan=True)
    tab.close()
    grp.close()


@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
def test_group_interp2_host_and_device_shards(mi_ctx, devices):
    """BASELINE config 3's shape over a group: scattered bilinear queries, table replicated, query shards."""
    import torch
    import armadillocudalinearinterpolation_amd as mi
    nx, ny = 193, 157
    x = np.linspace(-1.0, 2.0, nx)
    y = np.cumsum(0.5 + oracle.splitmix_uniform(3, ny))             # non-uniform axis
    zz = np.sin(3 * y)[:, None] * np.cos(2 * x)[None, :] + 0.1 * x[None, :] * y[:, None]
    nq = 200_003
    xq = oracle.splitmix_uniform(8, nq) * 3.2 - 1.1                  # some out of range
    yq = y[0] + oracle.splitmix_uniform(9, nq) * (y[-1] - y[0]) * 1.05 - 0.02
    ref = oracle.interp2_bilinear(x, y, zz, xq, yq)
    grp = mi.Group(devices)
    tab = grp.grid2(x, y, zz)
    assert np.array_equal(tab.interp_host(xq, yq), ref, equal_nan=True)
    assert np.array_equal(tab.interp_host(xq, yq, extrap=7.0), oracle.interp2_bilinear(x, y, zz, xq, yq, extrap=7.0))
    P, n = len(devices), 50_001
    xs = [torch.from_numpy(xq[r * n:(r + 1) * n].copy()).cuda() for r in range(P)]
    ys = [torch.from_numpy(yq[r * n:(r + 1) * n].copy()).cuda() for r in range(P)]
    outs, full = tab.interp_dev(xs, ys, gather=True)
    for r in range(P):
        assert np.array_equal(outs[r].cpu().numpy(), ref[r * n:(r + 1) * n], equal_nan=True)
        assert np.array_equal(full[r].cpu().numpy(), ref[:P * n], equal_nan=True)
    tab.close()
    grp.close()


def test_shard_bounds_match_python(mi_ctx):
    import armadillocudalinearinterpolation_amd as mi
    from armadillocudalinearinterpolation_amd import sharding
    for n in (0, 1, 7, 8, 100, 10**8 + 3, 2**33 + 5):
        for world in (1, 2, 3, 8):
            assert [mi.shard_bounds(n, r, world) for r in range(world)] == [sharding.shard_bounds(n, r, world) for r in range(world)]


@pytest.mark.parametrize("mean_quirk", [1, 0])
@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_group_compute_f_equals_single_device(mi_ctx, devices, mean_quirk):
    import armadillocudalinearinterpolation_amd as mi
    grp = mi.Group(devices)
    # sigma = 0: the partial sums are exact, the sharded residual is BIT-identical to the unsharded one
    R = 1501
    one = mi.EventDrivenMap(mi_ctx, [13.0589], R, n_grid=512, mean_quirk=mean_quirk)
    f1, p1 = one.ComputeF(Z, want_partial=True)
    ge = grp.edm([13.0589], R, n_grid=512, mean_quirk=mean_quirk)
    fg, pg = ge.ComputeF(Z, want_partial=True)
    assert np.array_equal(fg, f1) and np.array_equal(pg, p1)
    assert [ge.shard_bounds(r) for r in range(len(devices))] == [mi.shard_bounds(R, r, len(devices)) for r in range(len(devices))]
    # sigma > 0: realisations differ (the draw is keyed by the GLOBAL realisation index), sums add up in another order
    one.SetParameterStdDev(0.3)
    f1, p1 = one.ComputeF(Z, want_partial=True)
    ge.params.beta_stddev = 0.3
    ge._push()
    fg, pg = ge.ComputeF(Z, want_partial=True)
    assert pg[3] == p1[3] and np.allclose(pg[:3], p1[:3], rtol=1e-12, atol=0) and np.array_equal(pg[4:], p1[4:])
    assert np.allclose(fg, f1, rtol=0, atol=2e-7)
    po = oracle.edm_default_params(n_grid=512, n_real=64, beta_stddev=0.3, mean_quirk=mean_quirk)
    ge.params.n_real = 64                                   # setter path: new shard layout
    ge._push()
    fo, _ = oracle.edm_compute_f(po, Z, nthreads=8)
    assert np.allclose(ge.ComputeF(Z), fo, rtol=0, atol=2e-7)
    one.close()
    ge.close()
    grp.close()


def test_group_rccl_reduce_with_one_rank(mi_ctx):
    """MI_GROUP_REDUCE_RCCL binds librccl at run time and forms the communicator; with one rank the all-reduce is the
    identity, so the residual must equal the host-reduce one."""
    import armadillocudalinearinterpolation_amd as mi
    grp = mi.Group([0])
    ge = grp.edm([13.0589], 300, n_grid=512)
    f_host = ge.ComputeF(Z)
    grp.set_reduce(mi.Group.REDUCE_RCCL)
    assert np.array_equal(ge.ComputeF(Z), f_host)
    with pytest.raises(mi.MiError):                        # a rehearsal group (repeated device) cannot use RCCL
        mi.Group([0, 0]).set_reduce(mi.Group.REDUCE_RCCL)
    ge.close()
    grp.close()


def test_driver_with_sharded_realisations_matches_single_device(tmp_path):
    """host/driver --devices 0,0: the Driver.cu Newton solve with the realisations in two shards (one ComputeF per
    residual, AbstractNonlinearProblem.hpp:11) gives the same iterates as the single-device run."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drv = os.path.join(root, "armadillocudalinearinterpolation_amd", "host", "driver")
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(drv), "all"])
    res = []
    for extra in ([], ["--devices", "0,0"], ["--gpus", "1"]):
        js = os.path.join(tmp_path, "d%d.json" % len(res))
        subprocess.run([drv, "--real", "1001", "--threads", "1024", "--quiet", "--json", js] + extra, capture_output=True, text=True)
        res.append(json.load(open(js)))
    assert res[1]["shards"] == 2 and res[2]["shards"] == 1
    for r in res[1:]:
        assert r["solution"] == res[0]["solution"] and r["history"] == res[0]["history"] and r["converged"] == res[0]["converged"]
