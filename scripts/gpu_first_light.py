"""First-light check on a real MI355X: interp1 (3 table modes), interp2, restrict, masked mean
against the CPU oracle, plus a quick timing of the headline config.  Not a test: see tests/."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from armadillocudalinearinterpolation_amd import _lib  # noqa: E402

_lib.load(strict=False)
import armadillocudalinearinterpolation_amd as mi  # noqa: E402

ctx = mi.Context(0)
print(ctx.device_info())
dev = torch.device("cuda:0")
res = {}

# ---- interp1 parity
ng, nq = 10000, 100001
X = np.arange(ng) / (ng - 1)
Y = np.sin(2 * np.pi * X) + 0.5 * X
xi = oracle.splitmix_uniform(0x5EED0001, nq)
xi[:8] = [0.0, 1.0, X[5], X[ng - 1], -1e-9, 1 + 1e-9, np.nan, 0.5 * (X[3] + X[4])]
ref = oracle.interp1_bracket(X, Y, xi)
g = mi.Grid1.from_nodes(ctx, X, Y)
print("grid1 general:", g.info())
got = g.interp(torch.from_numpy(xi).to(dev)).cpu().numpy()
res["interp1_mode1_bitexact"] = bool(np.array_equal(got, ref, equal_nan=True))
# non-uniform jittered grid
u = oracle.splitmix_uniform(0x5EED0002, ng)
Xn = (np.arange(ng) + 0.5 * u) / ng
refn = oracle.interp1_bracket(Xn, Y, xi)
gn = mi.Grid1.from_nodes(ctx, Xn, Y)
print("grid1 jittered:", gn.info())
gotn = gn.interp(torch.from_numpy(xi).to(dev)).cpu().numpy()
res["interp1_jitter_bitexact"] = bool(np.array_equal(gotn, refn, equal_nan=True))
# wildly non-uniform grid -> bucket mode
Xw = np.sort(oracle.splitmix_uniform(77, ng) ** 6)
Xw = np.unique(Xw)
Yw = np.cos(5 * Xw)
xiw = oracle.splitmix_uniform(78, nq) * (Xw[-1] - Xw[0]) * 1.02 + Xw[0] - 0.01 * (Xw[-1] - Xw[0])
refw = oracle.interp1_bracket(Xw, Yw, xiw)
gw = mi.Grid1.from_nodes(ctx, Xw, Yw)
print("grid1 wild:", gw.info())
gotw = gw.interp(torch.from_numpy(xiw).to(dev)).cpu().numpy()
res["interp1_mode2_bitexact"] = bool(np.array_equal(gotw, refw, equal_nan=True))
# uniform implicit
gu = mi.Grid1.uniform(ctx, 0.0, 1.0 / (ng - 1), Y)
refu = oracle.interp1_uniform(0.0, 1.0 / (ng - 1), Y, xi)
gotu = gu.interp(torch.from_numpy(xi).to(dev)).cpu().numpy()
res["interp1_mode0_bitexact"] = bool(np.array_equal(gotu, refu, equal_nan=True))
# host one-shot
goth = mi.interp1(ctx, X, Y, xi)
res["interp1_host_bitexact"] = bool(np.array_equal(goth, oracle.interp1_arma(X, Y, xi), equal_nan=True))

# ---- interp2 parity
nx, ny, n2 = 64, 48, 50001
xg = np.arange(nx) / (nx - 1)
yg = np.arange(ny) / (ny - 1)
Z = np.sin(2 * np.pi * yg)[:, None] * np.cos(2 * np.pi * xg)[None, :] + xg[None, :] * yg[:, None]
q = oracle.splitmix_uniform(0x5EED0004, 2 * n2)
xq, yq = q[:n2].copy() * 1.02 - 0.01, q[n2:].copy() * 1.02 - 0.01
xq[:3] = [0.0, 1.0, np.nan]
yq[:3] = [1.0, 1.0, 0.5]
g2 = mi.Grid2.from_axes(ctx, xg, yg, Z)
r2 = oracle.interp2_bilinear(xg, yg, Z, xq, yq)
got2 = g2.interp(torch.from_numpy(xq).to(dev), torch.from_numpy(yq).to(dev)).cpu().numpy()
res["interp2_explicit_bitexact"] = bool(np.array_equal(got2, r2, equal_nan=True))
g2u = mi.Grid2.uniform(ctx, 0.0, 1.0 / (nx - 1), nx, 0.0, 1.0 / (ny - 1), ny, Z)
r2u = oracle.interp2_bilinear_uniform(0.0, 1.0 / (nx - 1), nx, 0.0, 1.0 / (ny - 1), ny, Z, xq, yq)
got2u = g2u.interp(torch.from_numpy(xq).to(dev), torch.from_numpy(yq).to(dev)).cpu().numpy()
res["interp2_uniform_bitexact"] = bool(np.array_equal(got2u, r2u, equal_nan=True))

# ---- restrict + masked mean
S, R, N = 3, 1000, 1024
rng = np.random.default_rng(5)
t0 = (rng.random(S * R) * 5).astype(np.float32)
t1 = (5 + rng.random(S * R)).astype(np.float32)
i0 = rng.integers(0, N, S * R).astype(np.uint16)
i1 = rng.integers(0, N, S * R).astype(np.uint16)
acc = (rng.random(R) < 0.9).astype(np.uint32)
rr = oracle.restrict_f32(t0, i0, t1, i1, 5.0, 3.0, N)
tt = lambda a: torch.from_numpy(a.view(np.int16) if a.dtype == np.uint16 else (a.view(np.int32) if a.dtype == np.uint32 else a)).to(dev)
gr = mi.restrict(ctx, tt(t0), tt(i0), tt(t1), tt(i1), 5.0, 3.0, N).cpu().numpy()
res["restrict_bitexact"] = bool(np.array_equal(gr, rr))
for quirk in (False, True):
    m_ref, c_ref = oracle.masked_mean_f32(rr, acc, S, quirk=quirk)
    m, c = mi.masked_mean(ctx, tt(rr), tt(acc), S, quirk=quirk)
    f = mi.restrict_mean(ctx, tt(t0), tt(i0), tt(t1), tt(i1), tt(acc), 5.0, 3.0, N, S, quirk=quirk, want_restricted=True)
    res["mean_quirk%d" % quirk] = [m.cpu().numpy().tolist(), m_ref.tolist(), int(c.item()), c_ref,
                                    f["mean"].cpu().numpy().tolist(), bool(np.array_equal(f["restricted"].cpu().numpy(), rr))]

# ---- headline timing: 1e8 random queries, 1e6-node table
NG, NQ = 10**6, 10**8
X6 = np.arange(NG) / (NG - 1)
Y6 = np.sin(2 * np.pi * X6) + 0.5 * X6
xq6 = torch.rand(NQ, dtype=torch.float64, device=dev)
out = torch.empty_like(xq6)
tm = ctx.timer()
for name, grid in (("general(mode1)", mi.Grid1.from_nodes(ctx, X6, Y6, sanitise=False)),
                   ("uniform(mode0)", mi.Grid1.uniform(ctx, 0.0, 1.0 / (NG - 1), Y6))):
    for qname, qq in (("random", xq6), ("sorted", torch.sort(xq6).values)):
        grid.interp(qq, out=out)
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            tm.start(); grid.interp(qq, out=out); tm.stop()
            ts.append(tm.elapsed_ms())
        ms = float(np.median(ts))
        res["time_%s_%s" % (name, qname)] = {"ms": ms, "pts_per_s": NQ / ms * 1e3, "GBps_alg": 16 * NQ / ms / 1e6,
                                              "frac_of_8TBps": 16 * NQ / ms / 1e6 / 8000}
        print(name, qname, res["time_%s_%s" % (name, qname)], flush=True)
# spot-check the big run against the oracle on a sample
samp = xq6[:200000].cpu().numpy()
gbig = mi.Grid1.from_nodes(ctx, X6, Y6, sanitise=False)
res["big_sample_bitexact"] = bool(np.array_equal(gbig.interp(xq6[:200000].clone()).cpu().numpy(),
                                                 oracle.interp1_bracket(X6, Y6, samp, nthreads=8)))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/first_light.json", "w"), indent=1)
print(json.dumps(res, indent=1))
