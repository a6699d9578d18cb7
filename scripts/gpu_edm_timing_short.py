"""Evolve time of ComputeF at the per-GPU share of BASELINE configs[3] and two smaller cases (not a test)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import armadillocudalinearinterpolation_amd as mi  # noqa: E402

ctx = mi.Context(0)
Z = [0.3310, 0.6914, 1.3557]
for mode, name in ((mi.MATH_EXACT, "exact"), (mi.MATH_FAST, "fast")):
    for N, R, sigma in ((1024, 125000, 0.0), (512, 125000, 0.0), (1024, 16384, 0.3)):
        edm = mi.EventDrivenMap(ctx, [13.0589], R, n_grid=N, math_mode=mode, beta_stddev=sigma)
        edm.ComputeF(Z)
        best = 1e30
        for _ in range(2):
            f = edm.ComputeF(Z)
            best = min(best, edm.last_timings()["evolve_ms"])
        print("%s N=%d R=%d sigma=%g evolve %.2f ms f0 %.9g" % (name, N, R, sigma, best, f[0]), flush=True)
        edm.close()
