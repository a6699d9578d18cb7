"""Evolve time of ComputeF for few realisations (the latency form of the kernel) and for dedup_identical (not a test)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import armadillocudalinearinterpolation_amd as mi  # noqa: E402

ctx = mi.Context(0)
Z = [0.3310, 0.6914, 1.3557]
for N, R, kw in ((1024, 1000, {}), (512, 1000, {}), (1024, 256, {}), (512, 256, {}), (1024, 125000, {"dedup_identical": 1})):
    edm = mi.EventDrivenMap(ctx, [13.0589], R, n_grid=N, **kw)
    edm.ComputeF(Z)
    best = 1e30
    for _ in range(5):
        f = edm.ComputeF(Z)
        best = min(best, edm.last_timings()["evolve_ms"])
    print("N=%d R=%d %s evolve %.3f ms f0 %.9g" % (N, R, "dedup" if kw else "", best, f[0]), flush=True)
    edm.close()
