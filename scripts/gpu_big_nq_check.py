import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import oracle
import armadillocudalinearinterpolation_amd as mi
ctx = mi.Context(0)
n = 1_000_000
X = np.arange(n) / (n - 1); Y = np.sin(7 * X) + X
g = mi.Grid1.from_nodes(ctx, X, Y)
nq = (1 << 31) + 12345
xq = torch.empty(nq, dtype=torch.float64, device="cuda:0")
gen = torch.Generator(device="cuda:0").manual_seed(5)
step = 1 << 28
for off in range(0, nq, step):
    m = min(step, nq - off)
    xq[off:off + m] = torch.rand(m, dtype=torch.float64, device="cuda:0", generator=gen) * 1.02 - 0.01
out = torch.empty_like(xq)
idx = torch.cat([torch.arange(0, 5000, device="cuda:0"), torch.randint(0, nq, (200000,), device="cuda:0"), torch.arange(nq - 5000, nq, device="cuda:0"),
                 torch.arange((1 << 31) - 3000, (1 << 31) + 3000, device="cuda:0"), torch.arange((1 << 32) // 8 * 4 - 2000, (1 << 32) // 8 * 4 + 2000, device="cuda:0") % nq])
ref = oracle.interp1_bracket(X, Y, xq[idx].cpu().numpy(), extrap=-2.5, nthreads=8)
for hint in (1, 2, 0):
    ctx.set_query_order(hint)
    out.zero_()
    g.interp(xq, out=out, extrap=-2.5)
    ctx.synchronize()
    ok = np.array_equal(out[idx].cpu().numpy(), ref, equal_nan=True)
    print("nq = 2^31 + 12345, hint", hint, "sample parity:", ok, flush=True)
    assert ok
ctx.set_query_order(0)
