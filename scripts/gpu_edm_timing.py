"""Time EventDrivenMap::ComputeF on the GPU (not a test).  Usage: gpu_edm_timing.py [tag]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import armadillocudalinearinterpolation_amd as mi  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "edm"
ctx = mi.Context(0)
Z = [0.3310, 0.6914, 1.3557]
res = {}
for mode, name in ((mi.MATH_EXACT, "exact"), (mi.MATH_FAST, "fast")):
    for N in (1024, 512):
        for R, sigma in ((1000, 0.0), (16384, 0.0), (125000, 0.0), (16384, 0.3)):
            edm = mi.EventDrivenMap(ctx, [13.0589], R, n_grid=N, math_mode=mode, beta_stddev=sigma)
            edm.ComputeF(Z)
            t = time.perf_counter()
            f = edm.ComputeF(Z)
            wall = time.perf_counter() - t
            tm = edm.last_timings()
            key = "%s_N%d_R%d_sigma%g" % (name, N, R, sigma)
            res[key] = {"wall_ms": wall * 1e3, **tm, "f": f.tolist(), "real_per_s": R / (tm["evolve_ms"] * 1e-3)}
            print(key, "wall %.2f ms evolve %.2f ms  %.0f realisations/s" % (wall * 1e3, tm["evolve_ms"], res[key]["real_per_s"]), flush=True)
            edm.close()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/edm_timing_%s.json" % tag, "w"), indent=1)
