"""Evolve time of the four headline ComputeF shapes (not a test).  Usage: gpu_edm_timing_short.py [repeats]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import armadillocudalinearinterpolation_amd as mi  # noqa: E402

rep = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctx = mi.Context(0)
Z = [0.3310, 0.6914, 1.3557]
for mode, name in ((mi.MATH_EXACT, "exact"), (mi.MATH_FAST, "fast")):
    for N, R, sigma in ((1024, 125000, 0.0), (512, 125000, 0.0), (1024, 16384, 0.3), (1024, 4000, 0.0), (512, 4000, 0.0)):
        edm = mi.EventDrivenMap(ctx, [13.0589], R, n_grid=N, math_mode=mode, beta_stddev=sigma)
        edm.ComputeF(Z)
        best = min(edm.last_timings()["evolve_ms"] for _ in range(rep) if edm.ComputeF(Z) is not None)
        print("%s N=%d R=%d sigma=%g evolve %.2f ms" % (name, N, R, sigma, best), flush=True)
        edm.close()
