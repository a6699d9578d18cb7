#!/usr/bin/env python3
"""Bounded first check of the Evolve kernels on the GPU box (not a test): a few small ComputeF calls with the event cap
lowered so that a wrong kernel cannot run for long, every stage tap compared with the oracle.
Run as: timeout -k 10 300 python3 scripts/gpu_edm_quickcheck.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import armadillocudalinearinterpolation_amd as mi  # noqa: E402
import oracle  # noqa: E402

Z = [0.3310, 0.6914, 1.3557]
ctx = mi.Context(0)
for form in ("4", "1"):
    os.environ["MI_EDM_WAVES_PER_REALISATION"] = form
    for kw in (dict(n_grid=64, n_real=4), dict(n_grid=512, n_real=5), dict(n_grid=1024, n_real=8), dict(n_grid=1000, n_real=3),
               dict(n_grid=1024, n_real=5, beta_stddev=0.3, seed=7), dict(n_grid=512, n_real=3500)):
        kw = dict(kw, max_events=3000)
        R = kw.pop("n_real")
        print("start form", form, kw, "R", R, flush=True)
        t = time.perf_counter()
        edm = mi.EventDrivenMap(ctx, [13.0589], R, **kw)
        f = edm.ComputeF(Z)
        dbg = edm.debug_read()
        wall = time.perf_counter() - t
        fo, d = oracle.edm_compute_f(oracle.edm_default_params(n_real=R, **kw), Z, nthreads=8)
        same = all(np.array_equal(dbg[k], d[k], equal_nan=True) for k in ("v", "s", "w", "t0", "i0", "t1", "i1", "accept"))
        extra = edm.debug_counters() if form == "1" else {}
        print("form", form, kw, "R", R, "wall %.1f ms" % (wall * 1e3), "evolve %.2f ms" % edm.last_timings()["evolve_ms"],
              "taps equal:", same, "f", f, extra, flush=True)
        edm.close()
