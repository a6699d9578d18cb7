#!/usr/bin/env python3
"""Where the wave-per-realisation Evolve kernel spends its time (library built with -DMI_EVOLVE_TIMING=1): wave-time
shares of the Newton rounds, the arg-min + uniform exponentials, the state pass and the bookkeeping, and events per
realisation.  ComputeF at the reference's parameters, R realisations of N neurons."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import armadillocudalinearinterpolation_amd as mi  # noqa: E402
from armadillocudalinearinterpolation_amd import _lib  # noqa: E402

L = _lib.load()
hook = L.mi_debug_evolve_timing
hook.argtypes = [C.c_void_p]
hook.restype = C.c_int
ctx = mi.Context(0)
Z = [0.3310, 0.6914, 1.3557]
out = (C.c_ulonglong * 8)()
for mode, name in ((mi.MATH_EXACT, "exact"), (mi.MATH_FAST, "fast")):
    for N, R, sigma in ((1024, 32768, 0.0), (512, 32768, 0.0), (1024, 32768, 0.3)):
        edm = mi.EventDrivenMap(ctx, [13.0589], R, n_grid=N, math_mode=mode, beta_stddev=sigma)
        edm.ComputeF(Z)
        assert hook(out) == 0
        edm.ComputeF(Z)
        tm = edm.last_timings()
        assert hook(out) == 0
        t = [float(out[j]) for j in range(4)]
        tot = sum(t)
        print("%s N=%d R=%d sigma=%g: evolve %.2f ms | events/realisation %.1f | wave time: Newton rounds %.1f %%, arg-min + "
              "uniform exps %.1f %%, state pass %.1f %%, bookkeeping %.1f %% | ticks per event %.0f" %
              (name, N, R, sigma, tm["evolve_ms"], out[4] / R, 100 * t[0] / tot, 100 * t[1] / tot, 100 * t[2] / tot,
               100 * t[3] / tot, tot / max(1, out[4])), flush=True)
        print("    slices per event that reach will_fire's division %.2f, its log/exp %.2f (of %d)" %
              (out[5] / max(1, out[4]), out[6] / max(1, out[4]), (N + 63) // 64), flush=True)
        edm.close()
