"""Evolve time at small and middling realisation counts (not a test).  Usage: gpu_edm_timing_small.py [repeats]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import armadillocudalinearinterpolation_amd as mi  # noqa: E402

rep = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ctx = mi.Context(0)
Z = [0.3310, 0.6914, 1.3557]
for N in (1024, 512):
    for R in (600, 1000, 2000, 3000, 4000, 8000):
        edm = mi.EventDrivenMap(ctx, [13.0589], R, n_grid=N)
        edm.ComputeF(Z)
        best = min(edm.last_timings()["evolve_ms"] for _ in range(rep) if edm.ComputeF(Z) is not None)
        print("exact N=%d R=%d evolve %.3f ms" % (N, R, best), flush=True)
        edm.close()
