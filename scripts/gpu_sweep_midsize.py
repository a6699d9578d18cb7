"""Sweep vs streaming kernel for mid-size tables (between LDS and L2 capacity), random queries.  MI_SWEEP_MIN_BYTES=0
lets the sweep run on any table size (tuning hook)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
os.environ["MI_SWEEP_MIN_BYTES"] = "0"
import armadillocudalinearinterpolation_amd as mi
from armadillocudalinearinterpolation_amd import synth
ctx = mi.Context(0)
nq = 100_000_000
u = synth.splitmix_uniform(0x5EED0003, nq, "cuda:0")
out = torch.empty_like(u)
sizes = [int(a) for a in sys.argv[1].split(',')] if len(sys.argv) > 1 else (20_000, 32_000, 64_000, 128_000, 200_000, 300_000, 500_000)
kinds = sys.argv[2].split(',') if len(sys.argv) > 2 else ('closed', 'jitter')
for ng in sizes:
    for kind in kinds:
        X = np.arange(ng) / (ng - 1) if kind == "closed" else (np.arange(ng) + 0.5 * np.random.default_rng(ng).random(ng)) / ng
        g = mi.Grid1.from_nodes(ctx, X, np.sin(X), sanitise=False)
        res = {}
        for name, hint in (("sweep", 1), ("stream", 2)):
            ctx.set_query_order(hint)
            for _ in range(2): g.interp(u, out=out)
            t = mi.Timer(ctx); t.start()
            for _ in range(5): g.interp(u, out=out)
            t.stop(); ctx.synchronize()
            res[name] = t.elapsed_ms() / 5
        ctx.set_query_order(0)
        print("ng %7d %-6s table %6.2f MB mode %d : sweep %.4f  stream %.4f  ratio %.2f" % (ng, kind, g.info()["table_bytes"] / 1e6, g.info()["mode"], res["sweep"], res["stream"], res["stream"] / res["sweep"]), flush=True)
